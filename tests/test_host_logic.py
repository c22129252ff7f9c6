"""Host side of the path: label writer (byte-identical to the oracle's), letterbox, tile names, CLI, C ABI surface."""
import os
import re

import numpy as np
import pytest

from aquaculture_amd import dataloader, detect, postprocess, tiles
from oracle import yolov5_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_dets(rng, n, size):
    d = np.zeros((n, 6), np.float32)
    xy = rng.uniform(-20, size + 20, (n, 2))
    wh = rng.uniform(1, 120, (n, 2))
    d[:, 0:2] = xy - wh / 2
    d[:, 2:4] = xy + wh / 2
    d[:, 4] = np.sort(rng.uniform(0.25, 1.0, n))[::-1]
    d[:, 5] = rng.integers(0, 5, n)
    return d


@pytest.mark.parametrize("img1,img0", [((640, 640), (640, 640)), ((640, 640), (1024, 1024)), ((480, 640), (500, 700)), ((640, 384), (1000, 600))])
def test_label_text_is_byte_identical_to_oracle(img1, img0):
    rng = np.random.default_rng(hash((img1, img0)) % 2 ** 31)
    det = _random_dets(rng, 300, max(img1))
    want = O.label_lines(det, img1, img0)
    got = postprocess.format_rows(postprocess.detections_to_rows(det, img1, img0)).splitlines()
    assert got == want
    assert postprocess.detections_to_rows(np.zeros((0, 6), np.float32), img1, img0).shape == (0, 6)


def test_label_file_contract_for_the_reference_consumer(tmp_path):
    """reference src/process_yolo/geocode_results.py:140-172: np.loadtxt rows cls xc yc w h conf; ndim == 1 for one line;
    no file when there are no detections (it only downloads images that have a label file, :46-55)."""
    det = _random_dets(np.random.default_rng(1), 5, 640)
    stem = tiles.tile_name(7).replace(".jpeg", "")
    assert postprocess.write_label_file(str(tmp_path), stem, postprocess.detections_to_rows(det, (640, 640), (1024, 1024)))
    arr = np.loadtxt(tmp_path / (stem + ".txt"))
    assert arr.shape == (5, 6) and set(arr[:, 0]) <= {0, 1, 2, 3, 4} and np.all(np.diff(arr[:, 5]) >= 0)
    assert int(1024 * (arr[0, 1] - arr[0, 3] / 2)) >= 0
    assert not postprocess.write_label_file(str(tmp_path), "empty", np.zeros((0, 6), np.float32))
    assert not (tmp_path / "empty.txt").exists()
    one = postprocess.detections_to_rows(det[:1], (640, 640), (640, 640))
    postprocess.write_label_file(str(tmp_path), "one", one)
    assert np.loadtxt(tmp_path / "one.txt").ndim == 1
    postprocess.write_label_file(str(tmp_path), "one", one)      # upstream opens in append mode
    assert np.loadtxt(tmp_path / "one.txt").shape == (2, 6)


def test_tile_name_codec_follows_reference_data():
    """reference src/utils.py:372-389; all names in reference output/cf_images.csv have exactly four '_' fields."""
    for i, year in ((0, 2015), (37, 2004), (123456, 2021)):
        n = tiles.tile_name(i, year)
        assert n.endswith(".jpeg") and len(n[:-5].split("_")) == 4
        spec = tiles.parse_tile_name(n)
        assert spec["year"] == str(year) and int(spec["x_offset"]) % 1024 == 0 and int(spec["y_offset"]) % 1024 == 0
        assert int(os.path.basename(n).split("_")[0][-4:]) == year          # geocode_results.py:143
    assert tiles.tile_name(5, 2021).startswith("ORTHOIMAGERY.ORTHOPHOTOS.ORTHO-EXPRESS.2021_")
    with pytest.raises(ValueError):
        tiles.parse_tile_name("a_b_c.jpeg")
    a, b = tiles.synthetic_tile(3), tiles.synthetic_tile(3)
    assert a.dtype == np.uint8 and a.shape == (640, 640, 3) and a.flags["C_CONTIGUOUS"] and np.array_equal(a, b)
    assert not np.array_equal(a, tiles.synthetic_tile(4))


def test_letterbox_geometry_and_resize():
    assert dataloader.letterbox_geometry((640, 640)) == ((640, 640), (0, 0, 0, 0))
    assert dataloader.letterbox_geometry((1024, 1024)) == ((640, 640), (0, 0, 0, 0))     # real tiles: scale 0.625, no pad
    (nw, nh), pads = dataloader.letterbox_geometry((500, 700))
    assert (nw, nh) == (640, 457) and pads == (11, 12, 0, 0) and (nh + 23) % 32 == 0
    im = np.full((1024, 1024, 3), 77, np.uint8)
    out = dataloader.letterbox(im)
    assert out.shape == (640, 640, 3) and np.all(out == 77)                              # constants survive fixed-point bilinear
    ramp = np.tile(np.arange(1024, dtype=np.float32)[None, :, None] / 4, (1024, 1, 3)).astype(np.uint8)
    r = dataloader.resize_linear_u8(ramp, 640, 640).astype(np.int32)
    assert np.all(np.diff(r[0, :, 0]) >= 0) and abs(int(r[0, 320, 0]) - int(ramp[0, 512, 0])) <= 1
    out = dataloader.letterbox(np.zeros((500, 700, 3), np.uint8))
    assert out.shape == (480, 640, 3) and np.all(out[:11] == 114) and np.all(out[-12:] == 114) and np.all(out[11:-12] == 0)
    assert dataloader.check_img_size([640, 650]) == [640, 672]


def test_load_images_sorted_sharded_and_batched(tmp_path):
    paths = tiles.write_synthetic_jpegs(str(tmp_path), range(7), size=64)
    (tmp_path / "notes.txt").write_text("not an image")
    files = dataloader.list_images(str(tmp_path))
    assert files == sorted(paths)
    ds0 = dataloader.LoadImages(str(tmp_path), 64, shard=(0, 2), workers=2)
    ds1 = dataloader.LoadImages(str(tmp_path), 64, shard=(1, 2), workers=1)
    assert sorted(ds0.indices + ds1.indices) == list(range(7)) and len(ds0) == 4 and ds0.total == 7
    batches = list(ds0.batches(3))
    assert [b[1].shape for b in batches] == [(3, 64, 64, 3), (1, 64, 64, 3)] and batches[0][1].dtype == np.uint8
    assert batches[0][2][0] == (64, 64)
    with pytest.raises(FileNotFoundError):
        dataloader.list_images(str(tmp_path / "missing"))


def test_cli_flags_and_paths(tmp_path):
    opt = detect.parse_opt(["--weights", "w.pt", "--source", "data/jpegs", "--nosave", "--save-txt", "--save-conf"])   # reference README.md:77
    assert opt.imgsz == [640, 640] and opt.conf_thres == 0.25 and opt.iou_thres == 0.45 and opt.max_det == 1000
    assert opt.save_txt and opt.save_conf and opt.nosave and not opt.half and opt.name == "exp"
    assert detect.parse_opt(["--img", "1280"]).imgsz == [1280, 1280]
    p = detect.increment_path(tmp_path / "exp")
    assert p.name == "exp"
    p.mkdir()
    assert detect.increment_path(tmp_path / "exp").name == "exp2"
    (tmp_path / "exp2").mkdir()
    assert detect.increment_path(tmp_path / "exp").name == "exp3"
    assert detect.increment_path(tmp_path / "exp", exist_ok=True).name == "exp"


def test_library_exports_every_declared_symbol(lib):
    """-m 'not gpu': the C-ABI library loads and exports every function include/aq_engine.h declares."""
    from aquaculture_amd import engine
    hdr = open(os.path.join(ROOT, "include", "aq_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(aq_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 24
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.aq_version() >= 1 and lib.aq_conv_num_configs() >= 10


def test_no_cpu_fallback(synth_ck):
    """The product path must fail loudly without a GPU; it never routes through the oracle or PyTorch eager."""
    import torch
    from aquaculture_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.Engine(synth_ck, "bf16")
    with pytest.raises(RuntimeError):
        engine.nms(torch.zeros(1, 10, 10), 5)
    src = "".join(open(os.path.join(ROOT, "aquaculture_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "aquaculture_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_bf16_byte_normalisation_by_reciprocal_is_exact():
    """The bf16 stem kernel converts uint8 pixels with v * (1/255) instead of v / 255 ([UPSTREAM detect.py: im.float() / 255]): after
    round-to-nearest-even to bf16 the two agree for every byte value (in fp32 they differ for 126 of them, so fp32 mode keeps v / 255)."""
    import numpy as np
    v = np.arange(256, dtype=np.float32)
    a = (v / np.float32(255.0)).astype(np.float32)
    b = (v * np.float32(1.0 / 255.0)).astype(np.float32)

    def bf16(x):
        u = x.view(np.uint32).astype(np.uint64)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    assert np.array_equal(bf16(a), bf16(b)) and int((a != b).sum()) > 0


def test_shipped_tuned_tables_match_the_plans():
    """aquaculture_amd/data/tuned_tables.json (round 3: the tile configurations of BASELINE.json's geometries, timed on MI355X and shipped so
    that two runs of one build launch the same kernels): every entry has one config per plan op of its model / precision, conv ops only."""
    import json
    from aquaculture_amd import spec
    tab = json.load(open(os.path.join(ROOT, "aquaculture_amd", "data", "tuned_tables.json")))
    assert any(k.startswith("yolov5m:nc5:pbf16:64x640x640:") for k in tab), "the headline geometry must ship"
    for key, cfgs in tab.items():
        variant, nc, prec = key.split(":")[0], int(key.split(":")[1][2:]), key.split(":")[2][1:]
        plan = spec.build_plan(variant, nc, fused_bottleneck=prec in ("bf16", "fp8", "fp8w"))
        assert len(cfgs) == len(plan.ops) == int(key.rsplit("ops", 1)[1]), key
        for c, o in zip(cfgs, plan.ops):
            assert (c == -1) == (o.kind != spec.OP_CONV) or c >= -1, (key, o.name, c)
            assert o.kind == spec.OP_CONV or c == -1, (key, o.name, c)


def test_batch_postprocess_equals_the_per_tile_one_bit_for_bit():
    """postprocess.batch_rows (one numpy pass per batch, what the CLI's writer threads use) against detections_to_rows per tile: empty tiles,
    full tiles, boxes outside the image, three original sizes."""
    import numpy as np
    from aquaculture_amd import postprocess as P
    rng = np.random.default_rng(0)
    B, M = 9, 50
    cnt = rng.integers(0, M + 1, B)
    cnt[3], cnt[5] = 0, M
    det = np.zeros((B, M, 6), np.float32)
    det[..., :2] = rng.uniform(-20, 700, (B, M, 2))
    det[..., 2:4] = det[..., :2] + rng.uniform(0, 300, (B, M, 2))
    det[..., 4] = rng.uniform(0, 1, (B, M))
    det[..., 5] = rng.integers(0, 5, (B, M))
    for shp in ((1024, 1024), (640, 640), (500, 700)):
        rows, offs = P.batch_rows(det, cnt, (640, 640), shp)
        assert offs[-1] == cnt.sum() == rows.shape[0]
        for b in range(B):
            want = P.detections_to_rows(det[b, :cnt[b]], (640, 640), shp)
            got = rows[offs[b]:offs[b + 1]]
            assert want.shape == got.shape and np.array_equal(want.view(np.uint32), got.view(np.uint32)), (shp, b)
    rows, offs = P.batch_rows(det, np.zeros(B, np.int64), (640, 640), (640, 640))
    assert rows.shape == (0, 6) and not offs.any()


def test_batch_label_writer_writes_the_bytes_of_the_per_tile_formatter(tmp_path):
    """aq_write_label_files (one C call per batch, round 4) against the per-tile path it replaces: the same files with the same bytes, no
    file for a tile without rows, an existing file truncated; an unwritable directory is an error that names the tile."""
    import numpy as np
    import pytest
    from aquaculture_amd import engine, postprocess
    rng = np.random.default_rng(3)
    counts = np.array([5, 0, 1, 340, 0, 17])
    offs = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    rows = np.empty((int(offs[-1]), 6), np.float32)
    rows[:, 0] = rng.integers(0, 5, rows.shape[0])
    rows[:, 1:5] = rng.integers(0, 1025, (rows.shape[0], 4)) / np.float32(1024)
    rows[:, 5] = rng.uniform(0.25, 1, rows.shape[0])
    rows[3, 1:5] = [0, 1e-5, 1, 0.5]                                  # exponent form, integers
    stems = [f"ORTHOIMAGERY.ORTHOPHOTOS2015_{i}_0_1024" for i in range(len(counts))]
    (tmp_path / f"{stems[2]}.txt").write_bytes(b"stale content that is longer than the new file\n" * 20)
    for save_conf in (True, False):
        n = engine.write_label_files(str(tmp_path), stems, rows, offs, save_conf)
        assert n == int((counts > 0).sum())
        for t, stem in enumerate(stems):
            f = tmp_path / f"{stem}.txt"
            if counts[t] == 0:
                assert not f.exists()
                continue
            want = engine.format_label_rows(rows[offs[t]:offs[t + 1]], save_conf)
            assert f.read_bytes() == want == postprocess.format_rows(rows[offs[t]:offs[t + 1]], save_conf).encode()
    with pytest.raises(OSError, match=stems[0]):
        engine.write_label_files(str(tmp_path / "missing_dir"), stems, rows, offs, True)
