"""S2 alone: the HIP NMS (aq_nms through the C ABI) must reproduce the oracle's non_max_suppression BIT FOR BIT
on hand-built pred tensors that hit the edge cases: empty input, confidence ties, the strict '>'
IoU boundary, cross-class overlap (class-offset trick), n > max_det, the > 2048-candidate path, the 30,000 cap."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NC = 5


def _pred(boxes_xywh, obj, cls_conf):
    n = len(boxes_xywh)
    p = np.zeros((1, n, 5 + NC), np.float32)
    p[0, :, :4] = boxes_xywh
    p[0, :, 4] = obj
    p[0, :, 5:] = cls_conf
    return p


def _run(pred, conf=0.25, iou=0.45, max_det=1000, agnostic=False, classes=None):
    from aquaculture_amd import engine
    from oracle import yolov5_oracle as O
    ref = O.non_max_suppression(pred, conf, iou, max_det, agnostic=agnostic, classes=classes)
    dets, counts = engine.nms(torch.from_numpy(pred).cuda().contiguous(), NC, conf, iou, max_det, agnostic=agnostic, classes=classes)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    for b, r in enumerate(ref):
        assert counts[b] == r.shape[0], (counts[b], r.shape[0])
        assert np.array_equal(dets[b, :counts[b]], r)
    return ref


def _random_pred(rng, n, spread=600.0, B=1):
    p = np.zeros((B, n, 5 + NC), np.float32)
    p[..., 0:2] = rng.uniform(20, spread, (B, n, 2))
    p[..., 2:4] = rng.uniform(4, 80, (B, n, 2))
    p[..., 4] = rng.uniform(0, 1, (B, n))
    p[..., 5:] = rng.uniform(0, 1, (B, n, NC))
    return p.astype(np.float32)


def test_empty_and_below_threshold(lib):
    rng = np.random.default_rng(0)
    p = _random_pred(rng, 500)
    p[..., 4] = 0.2                      # nothing passes obj > 0.25
    assert _run(p)[0].shape[0] == 0
    p[..., 4] = 0.3
    p[..., 5:] = 0.5                     # obj*cls = 0.15: passes the first threshold, fails the second
    assert _run(p)[0].shape[0] == 0


def test_confidence_ties_are_ordered_by_index(lib):
    boxes = [[100 + 50 * i, 100, 20, 20] for i in range(8)]
    p = _pred(boxes, 0.9, np.tile([[0.1, 0.8, 0.8, 0.2, 0.1]], (8, 1)))   # every row: conf 0.72, first max class = 1
    r = _run(p)[0]
    assert r.shape[0] == 8 and np.all(r[:, 5] == 1) and np.all(np.diff(r[:, 0]) > 0)


def test_strict_iou_boundary(lib):
    # IoU of the second box with the first is exactly 0.5: kept at thr 0.5 (strict '>'), suppressed at 0.45
    boxes = [[100, 100, 40, 40], [100, 110, 40, 20]]
    p = _pred(boxes, [0.9, 0.8], np.tile([[0.9, 0, 0, 0, 0]], (2, 1)))
    assert _run(p, iou=0.5)[0].shape[0] == 2
    assert _run(p, iou=0.45)[0].shape[0] == 1


def test_cross_class_overlap_is_not_suppressed(lib):
    boxes = [[200, 200, 50, 50], [200, 200, 50, 50], [201, 200, 50, 50]]
    cls = np.array([[0.9, 0, 0, 0, 0], [0, 0.9, 0, 0, 0], [0.8, 0, 0, 0, 0]], np.float32)
    r = _run(_pred(boxes, 0.9, cls))[0]
    assert r.shape[0] == 2 and set(r[:, 5]) == {0.0, 1.0}


def test_agnostic_and_class_filter(lib):
    """detect.py --agnostic-nms / --classes [UPSTREAM non_max_suppression(classes, agnostic)]: with agnostic the class-0 and class-1 boxes on
    the same spot suppress each other; a class filter removes the other classes' candidates before the suppression."""
    boxes = [[200, 200, 50, 50], [200, 200, 50, 50], [201, 200, 50, 50], [400, 400, 30, 30]]
    cls = np.array([[0.9, 0, 0, 0, 0], [0, 0.95, 0, 0, 0], [0.8, 0, 0, 0, 0], [0, 0, 0, 0.7, 0]], np.float32)
    p = _pred(boxes, 0.9, cls)
    r = _run(p, agnostic=True)[0]
    assert r.shape[0] == 2 and list(r[:, 5]) == [1.0, 3.0]            # the class-1 box wins the spot, the far class-3 box stays
    r = _run(p, classes=[0, 3])[0]
    assert r.shape[0] == 2 and set(r[:, 5]) == {0.0, 3.0}
    assert _run(p, classes=[4])[0].shape[0] == 0
    r = _run(p, agnostic=True, classes=[0, 3])[0]
    assert r.shape[0] == 2 and set(r[:, 5]) == {0.0, 3.0}
    rng = np.random.default_rng(11)
    for n in (700, 3000):                                             # both the bit-matrix path and the greedy fallback
        q = _random_pred(rng, n, B=2)
        q[..., 4] = rng.uniform(0.5, 1.0, q.shape[:2])
        q[..., 5:] = rng.uniform(0.6, 1.0, q.shape[:2] + (NC,))
        assert all(x.shape[0] > 5 for x in _run(q, agnostic=True))
        assert all(x.shape[0] > 5 and set(x[:, 5]) <= {1.0, 2.0} for x in _run(q, classes=[1, 2]))


def test_max_det_truncation(lib):
    rng = np.random.default_rng(1)
    p = _random_pred(rng, 1500, spread=3000.0)
    p[..., 2:4] = 3.0                    # tiny boxes: almost nothing suppressed
    p[..., 4] = rng.uniform(0.6, 1.0, p.shape[:2])
    p[..., 5] = 0.99
    assert _run(p, max_det=300)[0].shape[0] == 300
    assert _run(p, max_det=1000)[0].shape[0] == 1000


@pytest.mark.parametrize("n,seed", [(300, 2), (2048, 3), (2049, 4), (6000, 5)])
def test_random_dense_matches_oracle(lib, n, seed):
    """Dense overlapping boxes across both the bit-matrix path (n <= 2048) and the greedy fallback."""
    rng = np.random.default_rng(seed)
    p = _random_pred(rng, n, B=2)
    p[..., 4] = rng.uniform(0.5, 1.0, p.shape[:2])
    p[..., 5:] = rng.uniform(0.6, 1.0, p.shape[:2] + (NC,))
    r = _run(p)
    assert all(x.shape[0] > 10 for x in r)


def test_max_nms_cap_30000(lib):
    """More than 30,000 rows pass both thresholds: only the 30,000 most confident enter NMS [UPSTREAM max_nms]."""
    rng = np.random.default_rng(6)
    n = 33000
    p = np.zeros((1, n, 5 + NC), np.float32)
    p[0, :, 0] = (np.arange(n) % 200) * 10 + 5
    p[0, :, 1] = (np.arange(n) // 200) * 10 + 5
    p[0, :, 2:4] = 6.0
    p[0, :, 4] = rng.uniform(0.5, 1.0, n)
    p[0, :, 5] = 0.99
    r = _run(p, max_det=40000)[0]
    assert r.shape[0] == 30000


def test_engine_decode_plus_nms_equals_nms_on_raw_pred(lib, synth_ck, monkeypatch):
    """The compact-candidate path (decode writes only rows with obj > conf_thres) and S1 -> S2 through the full pred tensor agree bit for
    bit.  (With the Detect heads fused with their decode -- csrc/head_decode.hip, the default of `infer` -- the head convs sum K in a
    different order, so that comparison is one of tolerances: tests/test_gpu_head_decode.py.)"""
    from aquaculture_amd import engine, tiles
    monkeypatch.setenv("AQ_DISABLE_HEAD_FUSION", "1")
    eng = engine.Engine(synth_ck, "bf16")
    t = torch.from_numpy(tiles.synthetic_batch([3, 4], 640)).cuda()
    d0, c0 = eng.infer(t)
    d1, c1 = engine.nms(eng.forward_raw(t), synth_ck.nc)
    assert torch.equal(c0, c1)
    for b in range(2):
        assert torch.equal(d0[b, :c0[b]], d1[b, :c1[b]])


@pytest.mark.parametrize("n_cand", [1500, 2049, 3500, 4096, 4097, 6000])
def test_large_tiles_take_the_4096_candidate_bit_matrix(lib, n_cand):
    """1280-px tiles (100,800 rows; BASELINE.json configs[4]) are launched on nms_kernel<4096> (round 3): candidate counts on both sides
    of the old 2,048 and the new 4,096 limit (beyond it: the bitonic path), all bit for bit the oracle's."""
    rng = np.random.default_rng(n_cand)
    N = 100800
    p = _random_pred(rng, N, spread=1240.0)
    p[..., 4] = 0.1                                           # nothing passes ...
    idx = rng.choice(N, n_cand, replace=False)
    p[0, idx, 4] = rng.uniform(0.5, 1.0, n_cand)              # ... but n_cand rows
    p[0, idx, 5:] = rng.uniform(0.6, 1.0, (n_cand, NC))
    r = _run(p, max_det=1000)[0]
    assert r.shape[0] > 100
