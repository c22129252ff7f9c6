"""Resume of an interrupted sweep (missing item 4): the done-manifest survives a SIGKILL, a restarted sweep skips
exactly the tiles that were recorded -- also those without a label file -- and records the rest.  CPU only: the sweep here is a
stand-in loop over fake tiles using the same DoneManifest / LoadImages(skip_stems=...) code the CLI uses (tests/test_gpu_cli.py
runs the real `yolov5/detect.py --resume`).  Reference idiom this replaces: skip-if-exists, reference src/load_data/tile_tifs.py:40-41."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
from PIL import Image

from aquaculture_amd.dataloader import LoadImages
from aquaculture_amd.manifest import DoneManifest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEP = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
from aquaculture_amd.dataloader import LoadImages
from aquaculture_amd.manifest import DoneManifest
src, run, rank, world, delay = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
done = DoneManifest.load(run)
ds = LoadImages(src, shard=(rank, world), workers=1, raw=True, skip_stems=done)
m = DoneManifest(run, rank); m.open()
log = open(os.path.join(run, f"processed.rank{rank}.log"), "a")
for s in range(0, len(ds.files), 4):                       # batches of 4 tiles
    stems = [os.path.splitext(os.path.basename(p))[0] for p in ds.files[s:s + 4]]
    for st in stems:
        if int(st.split("_")[1]) % 3 == 0:                 # only every third tile has detections -> a label file
            with open(os.path.join(run, "labels", st + ".txt"), "wb") as f:
                f.write(b"0 0.5 0.5 0.1 0.1 0.9\n")
        log.write(st + "\n"); log.flush()
    m.add(stems)
    time.sleep(delay)
m.close()
print("finished", flush=True)
'''


def _tiles(d, n):
    os.makedirs(d, exist_ok=True)
    for i in range(n):
        Image.fromarray(np.full((8, 8, 3), i % 251, np.uint8)).save(os.path.join(d, f"TILE_{i:04d}.png"))


def test_manifest_add_load_and_partial_last_line(tmp_path):
    run = str(tmp_path / "run")
    m = DoneManifest(run, 0)
    m.open()
    m.add(["a_1", "a_2"])
    m.add([])
    m.close()
    with open(os.path.join(run, "done.rank1.txt"), "wb") as f:      # another rank; its last record was cut by a crash
        f.write(b"b_1\nb_2\nb_3_partia")
    assert DoneManifest.load(run) == {"a_1", "a_2", "b_1", "b_2"}
    m = DoneManifest(run, 1)
    m.open()                                                          # re-opening after the crash starts on a fresh line
    m.add(["b_3"])
    m.close()
    assert DoneManifest.load(run) == {"a_1", "a_2", "b_1", "b_2", "b_3"}      # the cut record was dropped, not completed
    assert DoneManifest.load(str(tmp_path / "nothing")) == set()


def test_killed_sweep_resumes_without_repeating_or_losing_tiles(tmp_path):
    src, run = str(tmp_path / "tiles"), str(tmp_path / "run")
    n = 60
    _tiles(src, n)
    os.makedirs(os.path.join(run, "labels"))
    script = str(tmp_path / "sweep.py")
    with open(script, "w") as f:
        f.write(SWEEP)
    # first attempt: 2 ranks, slow; kill both hard after a few batches
    procs = [subprocess.Popen([sys.executable, script, ROOT, src, run, str(r), "2", "0.15"], stdout=subprocess.PIPE, text=True) for r in range(2)]
    deadline = time.time() + 60
    while time.time() < deadline and len(DoneManifest.load(run)) < 12:
        time.sleep(0.05)
    for p in procs:
        p.send_signal(signal.SIGKILL)
    for p in procs:
        p.wait()
    done1 = DoneManifest.load(run)
    assert 12 <= len(done1) < n, len(done1)
    # restart with a DIFFERENT world size (3 ranks): the union of the old manifests is honoured
    procs = [subprocess.Popen([sys.executable, script, ROOT, src, run, str(r), "3", "0"], stdout=subprocess.PIPE, text=True) for r in range(3)]
    for p in procs:
        assert "finished" in p.communicate(timeout=120)[0]
    all_stems = {f"TILE_{i:04d}" for i in range(n)}
    assert DoneManifest.load(run) == all_stems
    processed = []
    for name in os.listdir(run):
        if name.startswith("processed."):
            processed += open(os.path.join(run, name)).read().split()
    assert set(processed) == all_stems                           # nothing lost
    # a tile recorded before the kill is never processed again; tiles in flight at the kill may be (their label bytes are the same)
    from collections import Counter
    again = [s for s, c in Counter(processed).items() if c > 1]
    assert not (set(again) & done1), sorted(set(again) & done1)
    assert len(again) <= 8                                       # at most the two batches that were in flight
    labels = {os.path.splitext(x)[0] for x in os.listdir(os.path.join(run, "labels"))}
    assert labels == {s for s in all_stems if int(s.split("_")[1]) % 3 == 0}
    for x in os.listdir(os.path.join(run, "labels")):
        assert open(os.path.join(run, "labels", x), "rb").read() == b"0 0.5 0.5 0.1 0.1 0.9\n"      # never appended twice


def test_loadimages_skip_keeps_global_numbering_and_subset(tmp_path):
    src = str(tmp_path / "tiles")
    _tiles(src, 10)
    Image.fromarray(np.zeros((5, 9, 3), np.uint8)).save(os.path.join(src, "TILE_0010.png"))      # an edge tile of another size
    ds = LoadImages(src, shard=(1, 2), workers=2, raw=True, skip_stems={"TILE_0003", "TILE_0004"})
    assert ds.total == 11 and ds.indices == [1, 5, 7, 9] and ds.skipped == 1
    ds = LoadImages(src, shard=(0, 1), workers=2, raw=True)
    sizes = ds.scan_sizes()
    assert sizes.count((8, 8)) == 10 and sizes[10] == (9, 5)
    sub = ds.subset([10, 2])
    assert sub.indices == [10, 2] and [os.path.basename(f) for f in sub.files] == ["TILE_0010.png", "TILE_0002.png"] and len(ds) == 11


def test_run_params_record_refuses_a_resume_with_other_settings(tmp_path):
    """labels written with different weights / thresholds / image size must not be mixed in one directory."""
    import json
    import pytest
    from aquaculture_amd.manifest import RunParamsMismatch, check_run_params, file_digest
    run = str(tmp_path / "run")
    os.makedirs(run)
    w = str(tmp_path / "w.pt")
    open(w, "wb").write(b"weights v1")
    params = {"weights_sha256": file_digest(w), "conf_thres": 0.25, "iou_thres": 0.45, "max_det": 1000, "imgsz": [640, 640], "precision": "bf16", "save_conf": True}
    check_run_params(run, params, resume=False)
    assert json.load(open(os.path.join(run, "run_params.json"))) == params
    check_run_params(run, dict(params), resume=True)                         # same settings: accepted
    with pytest.raises(RunParamsMismatch, match="conf_thres"):
        check_run_params(run, dict(params, conf_thres=0.5), resume=True)
    open(w, "wb").write(b"weights v2")
    with pytest.raises(RunParamsMismatch, match="weights_sha256"):
        check_run_params(run, dict(params, weights_sha256=file_digest(w)), resume=True)
    assert json.load(open(os.path.join(run, "run_params.json"))) == params   # a refused resume leaves the record alone
    check_run_params(run, dict(params, conf_thres=0.5), resume=False)        # a fresh run (--exist-ok) overwrites it
    assert json.load(open(os.path.join(run, "run_params.json")))["conf_thres"] == 0.5
    old = str(tmp_path / "old_run")                                          # a directory from before the record existed: accepted, recorded
    os.makedirs(old)
    check_run_params(old, params, resume=True)
    assert os.path.exists(os.path.join(old, "run_params.json"))
