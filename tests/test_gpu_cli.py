"""End to end through the preserved entry point: jpeg directory + upstream-format checkpoint -> label files
(reference README.md:77 invocation; output consumed by reference src/process_yolo/geocode_results.py:123-172)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = [0, 1, 2, 3, 19, 20, 21, 22, 23, 24]


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    from aquaculture_amd import checkpoint, tiles
    d = tmp_path_factory.mktemp("cli")
    tiles.write_synthetic_jpegs(str(d / "jpegs"), TILES, size=640)
    checkpoint.write_synthetic_checkpoint(str(d / "multilabel_farms_synth.pt"), "yolov5m", 5)
    return d


def _run(workdir, name, extra=(), launcher=(), env=None):
    cmd = [sys.executable, *launcher, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"),
           "--source", str(workdir / "jpegs"), "--nosave", "--save-txt", "--save-conf", "--project", str(workdir / "runs"), "--name", name,
           "--batch-size", "4", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout, workdir / "runs" / name / "labels"


def test_cli_writes_labels_the_consumer_can_parse(workdir, lib):
    from aquaculture_amd import checkpoint, dataloader, tiles
    from oracle import yolov5_oracle as O
    out, labels = _run(workdir, "exp")
    assert "labels saved to" in out and "Speed:" in out
    files = sorted(os.listdir(labels))
    stems = {tiles.tile_name(i)[:-5] for i in TILES}
    assert files and {f[:-4] for f in files} <= stems
    # oracle on the same decoded jpegs (fp32 is the CLI default = detect.py without --half)
    model = O.model_from_checkpoint(checkpoint.load_checkpoint(str(workdir / "multilabel_farms_synth.pt")))
    same = total = 0
    for i in TILES:
        stem = tiles.tile_name(i)[:-5]
        im = dataloader.read_rgb(str(workdir / "jpegs" / (stem + ".jpeg")))
        want = O.label_lines(O.detect_tiles(model, im[None])[0], (640, 640), (640, 640))
        path = labels / (stem + ".txt")
        if not want:
            assert not path.exists()          # no detections => no file
            continue
        arr = np.loadtxt(path)
        arr = arr[None] if arr.ndim == 1 else arr
        assert arr.shape == (len(want), 6) and set(arr[:, 0]) <= {0, 1, 2, 3, 4}
        assert np.all(np.diff(arr[:, 5]) >= -1e-6) and arr[:, 1:5].min() >= 0 and arr[:, 1:5].max() <= 1
        got = open(path).read().splitlines()
        pool = {}
        for l in got:
            f = l.split()
            pool.setdefault(" ".join(f[:5]), []).append(float(f[5]))
        for l in want:
            f = l.split()
            hit = [c for c in pool.get(" ".join(f[:5]), []) if abs(c - float(f[5])) <= 1e-4]
            same += bool(hit)
            total += 1
    assert total > 500 and same >= 0.995 * total


def test_cli_classes_and_agnostic_nms(workdir, lib):
    """--classes keeps exactly the listed classes' lines (per-class suppression does not see the other classes); --agnostic-nms lets boxes of
    different classes suppress each other: fewer lines on the synthetic head, which stacks classes on the same spots (the kernel's agreement
    with the oracle, bit for bit, is tests/test_gpu_nms.py).  [UPSTREAM detect.py: non_max_suppression(pred, conf, iou, classes, agnostic_nms, max_det)]"""
    _, ref = _run(workdir, "nmsopt_ref", extra=("--half",))
    _, flt = _run(workdir, "nmsopt_cls", extra=("--half", "--classes", "1", "3"))
    _, agn = _run(workdir, "nmsopt_agn", extra=("--half", "--agnostic-nms"))
    n_ref = n_agn = 0
    assert os.listdir(ref)
    for f in sorted(os.listdir(ref)):
        lines = open(ref / f).read().splitlines()
        want = [l for l in lines if l.split()[0] in ("1", "3")]
        got = open(flt / f).read().splitlines() if os.path.exists(flt / f) else []
        assert got == want, f
        a = open(agn / f).read().splitlines() if os.path.exists(agn / f) else []
        assert all(len(l.split()) == 6 for l in a), f
        n_ref += len(lines)
        n_agn += len(a)
    assert 0 < n_agn < n_ref                                   # the synthetic head stacks classes on the same spots


def test_cli_two_ranks_write_the_same_label_set(workdir, lib):
    """torchrun with 2 ranks (both on the one GPU, gloo as the rehearsal backend): strided shard + final gather;
    the union of the ranks' label files equals the single-process run byte for byte (tiles are independent)."""
    _, ref = _run(workdir, "exp_ref", extra=("--half",))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out, lab = _run(workdir, "exp_ddp", extra=("--half",), env={"AQ_DIST_BACKEND": "gloo"},
                    launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port)))
    assert "on 2 GPU(s)" in out
    assert sorted(os.listdir(ref)) == sorted(os.listdir(lab))
    for f in os.listdir(ref):
        assert open(ref / f).read() == open(lab / f).read(), f


def test_cli_on_real_size_1024_tiles(tmp_path, lib):
    """Real tiles are 1024x1024 (reference src/load_data/tile_tifs.py:13, src/utils.py:17-19): decoded on the host,
    letterboxed 1024 -> 640 on the device, boxes written normalised by the ORIGINAL size, as the consumer assumes
    (reference src/process_yolo/geocode_results.py:160-163 multiplies by IM_WIDTH = 1024)."""
    from aquaculture_amd import checkpoint, dataloader, tiles
    from oracle import yolov5_oracle as O
    idx = [19, 3]
    tiles.write_synthetic_jpegs(str(tmp_path / "jpegs"), idx, size=1024)
    checkpoint.write_synthetic_checkpoint(str(tmp_path / "multilabel_farms_synth.pt"), "yolov5m", 5)
    out, labels = _run(tmp_path, "exp1024")
    assert "at shape (1, 3, 640, 640)" in out
    model = O.model_from_checkpoint(checkpoint.synthetic_checkpoint("yolov5m", 5))
    for i in idx:
        stem = tiles.tile_name(i)[:-5]
        im0 = dataloader.read_rgb(str(tmp_path / "jpegs" / (stem + ".jpeg")))
        assert im0.shape == (1024, 1024, 3)
        want = O.label_lines(O.detect_tiles(model, dataloader.letterbox(im0)[None])[0], (640, 640), (1024, 1024))
        got = open(labels / (stem + ".txt")).read().splitlines()
        assert len(got) == len(want) and len(want) > 20
        pool = {}
        for l in got:
            f = l.split()
            pool.setdefault(" ".join(f[:5]), []).append(float(f[5]))
        same = sum(any(abs(c - float(l.split()[5])) <= 1e-4 for c in pool.get(" ".join(l.split()[:5]), [])) for l in want)
        assert same >= 0.99 * len(want)
        arr = np.loadtxt(labels / (stem + ".txt"))
        assert int(1024 * (arr[:, 1] + arr[:, 3] / 2).max()) <= 1024


def test_cli_geocodes_its_own_labels(workdir, lib, tmp_path):
    """--geocode-bboxes: the consumer's arithmetic (reference src/process_yolo/geocode_results.py:106-197) run as a batch op on the
    label files the sweep just wrote; checked against the scalar oracle on those same files."""
    import json
    from aquaculture_amd import tiles
    from oracle import geocode_oracle as GO
    inds = sorted({int(tiles.tile_name(i)[:-5].split("_")[1]) for i in TILES})
    csv_path = tmp_path / "wanted_bboxes.csv"
    with open(csv_path, "w") as f:                          # 1200 m squares like the reference's table
        f.write(",geometry\n")
        for k in inds:
            x0, y0 = 300000.0 + 1200.0 * k, 5200000.0 + 1200.0 * (k % 7)
            f.write(f'{k},"POLYGON (({x0 + 1200} {y0}, {x0 + 1200} {y0 + 1200}, {x0} {y0 + 1200}, {x0} {y0}, {x0 + 1200} {y0}))"\n')
    out, labels = _run(workdir, "geo", extra=("--geocode-bboxes", str(csv_path)))
    gj = labels.parent / "detections.geojson"
    assert "detections geocoded to" in out and gj.exists()
    d = json.load(open(gj))
    bboxes = {k: (300000.0 + 1200.0 * k, 5200000.0 + 1200.0 * (k % 7), 300000.0 + 1200.0 * k + 1200, 5200000.0 + 1200.0 * (k % 7) + 1200) for k in inds}
    want = []
    for fn in sorted(os.listdir(labels)):
        rows = np.loadtxt(labels / fn, ndmin=2)
        want += GO.geocode_label_rows(fn, rows, bboxes)
    assert len(d["features"]) == len(want) > 0
    for ft, w in zip(d["features"], want):
        p = ft["properties"]
        assert (p["image"], p["xmin"], p["xmax"], p["ymin"], p["ymax"], p["type"], p["year"]) == \
               (w["image"], w["xmin"], w["xmax"], w["ymin"], w["ymax"], w["type"], w["year"])
        assert p["det_conf"] == w["det_conf"] and abs(p["e_min_3035"] - w["e_min_3035"]) < 1e-6
        lon, lat = GO.mercator_to_lonlat(w["geometry"][2], w["geometry"][1])      # ring starts at (maxx, miny)
        assert abs(ft["geometry"]["coordinates"][0][0][0] - lon) < 1e-12 and abs(ft["geometry"]["coordinates"][0][0][1] - lat) < 1e-12


def test_decode_processes_and_threads_give_identical_labels(workdir, lib):
    """The jpeg decode worker processes (shared-memory pinned ring) and the in-process decode threads feed the engine the same pixels."""
    _, la = _run(workdir, "dec_procs", extra=("--half", "--workers", "3"))
    _, lb = _run(workdir, "dec_threads", extra=("--half", "--workers", "3", "--decode-threads"))
    fa, fb = sorted(os.listdir(la)), sorted(os.listdir(lb))
    assert fa == fb and fa
    for f in fa:
        assert open(la / f).read() == open(lb / f).read()


def test_scene_mode_equals_the_sweep_over_lossless_tiles(tmp_path, lib):
    """--tile-scenes: a scene raster cut into tiles on the device (order, sizes and names of reference src/load_data/tile_tifs.py:33-47,
    edge tiles included) gives byte-identical label files to the ordinary sweep over the same tiles stored losslessly."""
    from PIL import Image
    from aquaculture_amd import checkpoint, scenes, tiles
    checkpoint.write_synthetic_checkpoint(str(tmp_path / "multilabel_farms_synth.pt"), "yolov5m", 5)
    ids = {(0, 0): 19, (1024, 0): 3, (0, 1024): 20, (1024, 1024): 19}
    scene = np.zeros((1500, 2048, 3), np.uint8)
    for (x0, y0), i in ids.items():
        t = tiles.synthetic_tile(i, 1024)
        scene[y0:y0 + 1024, x0:x0 + 1024] = t[: min(1024, 1500 - y0)]
    (tmp_path / "scenes").mkdir()
    (tmp_path / "jpegs").mkdir()
    spath = tmp_path / "scenes" / "ORTHOIMAGERY.ORTHOPHOTOS2015_7.tif"
    Image.fromarray(scene).save(spath)
    grid = scenes.tile_grid(2048, 1500)
    assert [g[2:] for g in grid] == [(1024, 1024), (1024, 476), (1024, 1024), (1024, 476)]
    for x0, y0, w, h in grid:
        Image.fromarray(scene[y0:y0 + h, x0:x0 + w]).save(tmp_path / "jpegs" / (scenes.tile_stem(str(spath), x0, y0) + ".png"))
    _, ref = _run(tmp_path, "tiles")
    cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(tmp_path / "multilabel_farms_synth.pt"), "--source",
           str(tmp_path / "scenes"), "--tile-scenes", "--nosave", "--save-txt", "--save-conf", "--project", str(tmp_path / "runs"), "--name",
           "scene", "--batch-size", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "image 4/4" in r.stdout and "4 images" in r.stdout
    lab = tmp_path / "runs" / "scene" / "labels"
    fr, fs = sorted(os.listdir(ref)), sorted(os.listdir(lab))
    assert fr == fs and len(fr) >= 3 and all(f.count("_") == 3 for f in fs)
    for f in fr:
        assert open(ref / f).read() == open(lab / f).read(), f


def test_scene_mode_two_ranks(tmp_path, lib):
    """Scenes are sharded i % world; the union of the ranks' label files equals the single-process scene sweep and the final gather
    accounts for every detection (gloo rehearsal of the RCCL path, both ranks on the one GPU)."""
    from PIL import Image
    from aquaculture_amd import checkpoint, tiles
    checkpoint.write_synthetic_checkpoint(str(tmp_path / "multilabel_farms_synth.pt"), "yolov5m", 5)
    (tmp_path / "scenes").mkdir()
    t19, t20 = tiles.synthetic_tile(19, 1024), tiles.synthetic_tile(20, 1024)
    Image.fromarray(t19).save(tmp_path / "scenes" / "ORTHOIMAGERY.ORTHOPHOTOS2015_1.tif")
    Image.fromarray(np.concatenate([t20, t19], 0)).save(tmp_path / "scenes" / "ORTHOIMAGERY.ORTHOPHOTOS2015_2.tif", compression="tiff_lzw")
    base = [os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(tmp_path / "multilabel_farms_synth.pt"), "--source",
            str(tmp_path / "scenes"), "--tile-scenes", "--nosave", "--save-txt", "--save-conf", "--half", "--project", str(tmp_path / "runs")]
    r1 = subprocess.run([sys.executable, *base, "--name", "one"], capture_output=True, text=True, timeout=420)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), *base, "--name", "two"], capture_output=True, text=True, timeout=420,
                        env=dict(os.environ, AQ_DIST_BACKEND="gloo"))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    assert "3 images" in r1.stdout and "3 images" in r2.stdout and "on 2 GPU(s)" in r2.stdout
    a, b = tmp_path / "runs" / "one" / "labels", tmp_path / "runs" / "two" / "labels"
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) and len(os.listdir(a)) == 3
    for f in os.listdir(a):
        assert open(a / f).read() == open(b / f).read(), f


def test_cli_autotune_keeps_the_labels(workdir, lib, tmp_path):
    """--autotune on: the first batch is used to time every tile configuration per conv layer; the table is cached and the second
    run installs it without timing.  fp32 results may move by accumulation order only (halo vs implicit-GEMM K order): same boxes
    within 1e-4 (a handful of box edges one pixel off after rounding), same counts."""
    env = {"AQ_TUNE_CACHE": str(tmp_path / "tune.json")}
    _, off = _run(workdir, "tune_off", extra=("--autotune", "off"))
    out1, on1 = _run(workdir, "tune_on", extra=("--autotune", "on"), env=env)
    assert "autotuned" in out1 and os.path.exists(env["AQ_TUNE_CACHE"])
    out2, on2 = _run(workdir, "tune_on2", extra=("--autotune", "on"), env=env)
    assert "autotuned" in out2
    fa = sorted(os.listdir(off))
    assert fa == sorted(os.listdir(on1)) == sorted(os.listdir(on2)) and fa
    for f in fa:
        a, b = np.loadtxt(off / f, ndmin=2), np.loadtxt(on1 / f, ndmin=2)
        assert a.shape == b.shape and np.array_equal(np.sort(a[:, 0]), np.sort(b[:, 0]))      # near-ties in conf may swap lines
        # nearest row of the same class: a 1e-6 change before `.round()` can move a box edge by one pixel (1/640), nothing more
        d = np.abs(a[:, None, 1:] - b[None, :, 1:]).max(-1) + 10.0 * (a[:, None, 0] != b[None, :, 0])
        near = d.min(1)
        assert near.max() <= 2.0 / 640 + 1e-4 and (near <= 1e-4).mean() >= 0.99, (f, near.max(), (near <= 1e-4).mean())
        assert open(on1 / f).read() == open(on2 / f).read()          # cached table = the same kernels again
    # the tuned table serves the ragged last batch too (10 tiles, batches of 4 -> 4, 4, 2): with another batch size every tile
    # sits in a differently sized batch, and bf16 -- where a different kernel would round differently -- gives the same bytes
    envh = {"AQ_TUNE_CACHE": str(tmp_path / "tune_half.json")}
    _, h4 = _run(workdir, "tune_half4", extra=("--autotune", "on", "--half"), env=envh)
    cmd3 = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"), "--source",
            str(workdir / "jpegs"), "--nosave", "--save-txt", "--save-conf", "--project", str(workdir / "runs"), "--name", "tune_half3",
            "--batch-size", "3", "--autotune", "off", "--half"]
    import json
    table = json.load(open(envh["AQ_TUNE_CACHE"]))
    (key, cfgs), = table.items()
    table[key.replace(":4x640x640:", ":3x640x640:")] = cfgs          # the same kernels for batches of 3 (installed from the cache)
    json.dump(table, open(envh["AQ_TUNE_CACHE"], "w"))
    cmd3[cmd3.index("off")] = "on"
    r = subprocess.run(cmd3, capture_output=True, text=True, timeout=420, env=dict(os.environ, **envh))
    assert r.returncode == 0, r.stderr[-2000:]
    h3 = workdir / "runs" / "tune_half3" / "labels"
    assert sorted(os.listdir(h3)) == sorted(os.listdir(h4))
    for f in os.listdir(h4):
        assert open(h3 / f, "rb").read() == open(h4 / f, "rb").read(), f


def test_cli_resume_skips_recorded_tiles(workdir, lib):
    """`--resume` (missing item 4): a sweep whose manifests record part of the tiles -- here: a first run over a
    directory holding only some of them, then a crash-cut record -- processes exactly the rest, and the label directory ends up
    byte-identical to an uninterrupted sweep's.  Tiles without detections are skipped too (no label file says so; the manifest does)."""
    import shutil
    full_out, full = _run(workdir, "resume_full", extra=("--quiet",))
    part = workdir / "jpegs_part"
    part.mkdir(exist_ok=True)
    names = sorted(os.listdir(workdir / "jpegs"))
    for n in names[:6]:
        shutil.copy(workdir / "jpegs" / n, part / n)
    cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"),
           "--source", str(part), "--nosave", "--save-txt", "--save-conf", "--project", str(workdir / "runs"), "--name", "resume_run", "--batch-size", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    run = workdir / "runs" / "resume_run"
    recorded = open(run / "done.rank0.txt").read().split()
    assert sorted(recorded) == sorted(n[:-5] for n in names[:6])
    with open(run / "done.rank0.txt", "ab") as f:
        f.write(names[7][:-5].encode()[:-3])                 # a record cut short by a crash: must not count
    out, labels = _run(workdir, "resume_run", extra=("--resume",))
    assert f"skips 6 of its share" in out
    processed = [l for l in out.splitlines() if l.startswith("image ")]
    assert len(processed) == len(names) - 6 and not any(n[:-5] in l for l in processed for n in names[:6])
    assert sorted(os.listdir(labels)) == sorted(os.listdir(full))
    for f in os.listdir(full):
        assert open(labels / f, "rb").read() == open(full / f, "rb").read(), f
    assert sorted(open(run / "done.rank0.txt").read().split()) == sorted(n[:-5] for n in names)
    # a resume with other settings is refused before any tile is touched: the labels of two configurations never mix
    import json
    assert json.load(open(run / "run_params.json"))["conf_thres"] == 0.25
    before = {f: open(labels / f, "rb").read() for f in os.listdir(labels)}
    r = subprocess.run(cmd[:-4] + ["--name", "resume_run", "--batch-size", "4", "--resume", "--conf-thres", "0.6"], capture_output=True, text=True, timeout=420)
    assert r.returncode != 0 and "RunParamsMismatch" in r.stderr and "conf_thres" in r.stderr, r.stderr[-2000:]
    assert {f: open(labels / f, "rb").read() for f in os.listdir(labels)} == before


def test_cli_mixed_tile_sizes_do_not_end_the_sweep(tmp_path, lib, workdir):
    """More than 17 tiles of one size plus one smaller edge tile (the reference's tiler cuts those for scenes that are not a multiple of
    1024, reference src/load_data/tile_tifs.py:35-36), placed where a sampled size check does not look: every tile gets processed, the
    odd one through the generic loader (the sweep used to abort midway with label files half written)."""
    from PIL import Image
    from aquaculture_amd import tiles
    d = tmp_path / "mixed"
    d.mkdir()
    idx = list(range(40, 60))
    tiles.write_synthetic_jpegs(str(d), idx, size=256)
    names = sorted(os.listdir(d))
    victim = names[11]                                        # not a multiple of len // 16: a sampling check misses it
    im = Image.open(d / victim)
    im.crop((0, 0, 160, 224)).save(d / victim, quality=90)
    cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"), "--source", str(d),
           "--nosave", "--save-txt", "--save-conf", "--project", str(tmp_path / "runs"), "--name", "mixed", "--batch-size", "8", "--imgsz", "256"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert "1 of 20 images differ in size" in r.stdout and "20 images," in r.stdout
    processed = [l for l in r.stdout.splitlines() if l.startswith("image ")]
    assert len(processed) == 20 and any(victim in l and "256x192" in l for l in processed)      # 224 x 160 letterboxed for --imgsz 256
    assert sorted(open(tmp_path / "runs" / "mixed" / "done.rank0.txt").read().split()) == sorted(n[:-5] for n in names)


def test_split_jpeg_decode_gives_identical_labels(workdir, lib):
    """--jpeg-decode split (Huffman in the worker processes, IDCT / chroma upsampling / colour conversion on the GPU) against --jpeg-decode
    host (Pillow's libjpeg-turbo in the workers): the decoded pixels are byte-identical (tests/test_jpeg.py), so the label files are too;
    `auto` picks the split path for a directory of baseline 4:2:0 JPEGs and says so."""
    out_h, lab_h = _run(workdir, "jpeg_host", extra=("--quiet", "--half", "--jpeg-decode", "host"))
    out_s, lab_s = _run(workdir, "jpeg_split", extra=("--quiet", "--half", "--jpeg-decode", "split"))
    out_a, lab_a = _run(workdir, "jpeg_auto", extra=("--quiet", "--half"))
    assert "jpeg decode: split" in out_s and "jpeg decode: split" in out_a and "jpeg decode: split" not in out_h
    names = sorted(os.listdir(lab_h))
    assert names and names == sorted(os.listdir(lab_s)) == sorted(os.listdir(lab_a))
    for n in names:
        ref = open(lab_h / n, "rb").read()
        assert open(lab_s / n, "rb").read() == ref and open(lab_a / n, "rb").read() == ref, n


def test_cli_with_the_collectives_on_rccl_writes_the_same_labels(workdir, lib):
    """AQ_DIST_FORCE=1: save-directory broadcast, barrier, bounded detection gather and the counter reductions run through RCCL ("nccl") in
    a world of one rank -- what a one-GPU box can execute of SURVEY.md 8e on the real backend; label bytes as the plain run."""
    _, ref = _run(workdir, "exp_plain_bf16", extra=("--half",))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out, lab = _run(workdir, "exp_rccl1", extra=("--half",), env={"AQ_DIST_FORCE": "1", "AQ_DIST_BACKEND": "nccl", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                                                                 "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    a = {f: open(ref / f, "rb").read() for f in sorted(os.listdir(ref))}
    b = {f: open(lab / f, "rb").read() for f in sorted(os.listdir(lab))}
    assert a and a == b


def test_cli_names_the_file_a_decode_worker_rejects(workdir, lib, tmp_path):
    """A file whose frame header passes the directory scan but whose scan header names a Huffman table that was never defined: the worker's
    error reaches the parent at the batch's flush, the sweep stops with a non-zero exit code and the message names the file -- in both decode
    modes (libjpeg refuses such a file as well; upstream: `assert im0 is not None, 'Image Not Found'`)."""
    import shutil
    src = tmp_path / "jpegs"
    shutil.copytree(workdir / "jpegs", src)
    victim = sorted(os.listdir(src))[3]
    data = bytearray((src / victim).read_bytes())
    sos = data.index(b"\xff\xda")
    assert data[sos + 4] == 3                                     # three components: (id, tables) pairs follow
    data[sos + 6] = 0x33                                          # component 1: DC table 3 / AC table 3 -- neither exists
    (src / victim).write_bytes(bytes(data))
    for mode in ("split", "host"):
        cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"), "--source", str(src),
               "--nosave", "--save-txt", "--save-conf", "--project", str(tmp_path / "runs"), "--name", "bad_" + mode, "--batch-size", "4", "--half", "--jpeg-decode", mode]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
        assert r.returncode != 0 and victim in (r.stderr + r.stdout), (mode, r.stdout[-1500:], r.stderr[-1500:])


def test_cli_fp8_calibrates_on_the_sweeps_tiles_and_records_the_scales(workdir, lib):
    """`--precision fp8` used to quantise every sweep with scales calibrated on eight synthetic 640-px tiles.  The CLI
    now calibrates on tiles sampled from --source before the first batch, records the scales beside run_params.json, and a --resume
    installs the recorded ones (so the rest of an interrupted sweep is quantised as its beginning was): the label bytes of a resumed fp8
    sweep equal the uninterrupted one's."""
    import json
    import shutil
    out, full = _run(workdir, "fp8_full", extra=("--quiet", "--precision", "fp8"))
    assert "fp8: 14 activation scales calibrated on this sweep's tiles" in out
    rec = json.load(open(workdir / "runs" / "fp8_full" / "fp8_scales.json"))
    assert len(rec) == 14 and all(v > 0 for v in rec.values()) and all(k.endswith(".cv2") for k in rec)
    assert json.load(open(workdir / "runs" / "fp8_full" / "run_params.json"))["precision"] == "fp8"
    # an interrupted sweep: first six tiles, then --resume over the whole directory with the recorded scales
    part = workdir / "jpegs_fp8_part"
    part.mkdir(exist_ok=True)
    names = sorted(os.listdir(workdir / "jpegs"))
    for n in names[:6]:
        shutil.copy(workdir / "jpegs" / n, part / n)
    cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"), "--source", str(part),
           "--nosave", "--save-txt", "--save-conf", "--project", str(workdir / "runs"), "--name", "fp8_resume", "--batch-size", "4", "--precision", "fp8", "--quiet"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    run = workdir / "runs" / "fp8_resume"
    shutil.copy(workdir / "runs" / "fp8_full" / "fp8_scales.json", run / "fp8_scales.json")      # (the part directory calibrated on other tiles)
    out, labels = _run(workdir, "fp8_resume", extra=("--resume", "--quiet", "--precision", "fp8"))
    assert json.load(open(run / "fp8_scales.json")) == rec
    for f in sorted(os.listdir(full)):
        stem = f[:-4]
        if any(n.startswith(stem) for n in names[:6]):
            continue                                               # written by the first part with ITS calibration
        assert open(labels / f, "rb").read() == open(full / f, "rb").read(), f


def test_cli_refuses_a_truncated_tile_in_both_decode_modes(workdir, lib, tmp_path):
    """a half-written tile used to decode to grey blocks in split mode, get no labels and be recorded as done.  Both
    modes now stop with the file's name, and the done-manifest does not list it."""
    import shutil
    src = tmp_path / "jpegs"
    shutil.copytree(workdir / "jpegs", src)
    victim = sorted(os.listdir(src))[5]
    data = (src / victim).read_bytes()
    (src / victim).write_bytes(data[: len(data) * 2 // 3])
    for mode in ("split", "host"):
        cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"), "--source", str(src),
               "--nosave", "--save-txt", "--save-conf", "--project", str(tmp_path / "runs"), "--name", "trunc_" + mode, "--batch-size", "4", "--half", "--jpeg-decode", mode]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
        assert r.returncode != 0 and victim in r.stderr, (mode, r.stderr[-1500:])
        done = tmp_path / "runs" / ("trunc_" + mode) / "done.rank0.txt"
        assert not done.exists() or victim[:-5] not in open(done).read().split()


def test_gpu_jpeg_decode_gives_identical_labels(workdir, lib, tmp_path):
    """--jpeg-decode gpu (round 4: the Huffman stage on the GPU too, one lane per image, super-batches) against --jpeg-decode host: the
    coefficient blocks equal the host decoder's (tests/test_jpeg.py), so the label files are byte-identical; with a super-batch smaller
    than the directory the rotation of the three buffers is exercised; a tile cut short is refused with its name."""
    import shutil
    out_h, lab_h = _run(workdir, "gjpeg_host", extra=("--quiet", "--half", "--jpeg-decode", "host"))
    out_g, lab_g = _run(workdir, "gjpeg_gpu", extra=("--quiet", "--half", "--jpeg-decode", "gpu"), env={"AQ_JPEG_GPU_SUPERBATCH": "4"})
    # `auto` takes the GPU decoder once this rank's share reaches AQ_JPEG_GPU_AUTO_MIN images (32,768 by default: the split path below that)
    out_a, lab_a = _run(workdir, "gjpeg_auto", extra=("--quiet", "--half"), env={"AQ_JPEG_GPU_AUTO_MIN": "1"})
    assert "jpeg decode: gpu" in out_g and "jpeg decode: gpu" in out_a
    names = sorted(os.listdir(lab_h))
    assert names and names == sorted(os.listdir(lab_g)) == sorted(os.listdir(lab_a))
    for n in names:
        assert open(lab_g / n, "rb").read() == open(lab_h / n, "rb").read() == open(lab_a / n, "rb").read(), n
    src = tmp_path / "jpegs"
    shutil.copytree(workdir / "jpegs", src)
    victim = sorted(os.listdir(src))[5]
    data = (src / victim).read_bytes()
    (src / victim).write_bytes(data[: len(data) * 2 // 3])
    cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", str(workdir / "multilabel_farms_synth.pt"), "--source", str(src),
           "--nosave", "--save-txt", "--save-conf", "--project", str(tmp_path / "runs"), "--name", "gjpeg_trunc", "--batch-size", "4", "--half", "--jpeg-decode", "gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
    assert r.returncode != 0 and victim in r.stderr, r.stderr[-1500:]
