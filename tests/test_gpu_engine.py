"""End-to-end parity of the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

fp32 engine  vs fp32 oracle           : the parity gate named by BASELINE.json's north_star
                                        (boxes/confidences within 1e-4, identical post-NMS box counts).
bf16 engine  vs bf16-emulating oracle : same rounding points (bf16 storage, fp32 accumulate), so only
                                        summation order differs; tolerance documents what bf16 costs.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiles_small():
    from aquaculture_amd import tiles
    return tiles.synthetic_batch([0, 19], 128)


@pytest.fixture(scope="module")
def tiles_640():
    from aquaculture_amd import tiles
    return tiles.synthetic_batch([0, 1, 2], 640)


def _engine(ck, precision):
    from aquaculture_amd import engine
    return engine.Engine(ck, precision)


def test_intermediate_tensors_fp32(lib, synth_ck, tiles_small):
    """Per-module outputs at 128x128 (backbone, neck, raw heads) against the oracle's taps."""
    from oracle import yolov5_oracle as O
    eng = _engine(synth_ck, "fp32")
    m = O.model_from_checkpoint(synth_ck)
    m.taps = {}
    pred_ref = m.forward(O.preprocess(tiles_small))
    B = tiles_small.shape[0]
    pred = eng.forward_raw(torch.from_numpy(tiles_small).cuda())
    torch.cuda.synchronize()
    names = {"out0": "model.0", "out1": "model.1", "out2": "model.2", "out3": "model.3", "out5": "model.5",
             "out7": "model.7", "out8": "model.8", "out9": "model.9", "out13": "model.13", "out17": "model.17",
             "out20": "model.20", "out23": "model.23"}
    for tname, key in names.items():
        got = eng.tensor_by_name(tname, B).float().cpu().permute(0, 3, 1, 2)
        ref = m.taps[key]
        scale = ref.abs().max().item()
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * max(scale, 1.0), f"{key}: max err {err} (scale {scale})"
    for lvl in range(3):
        got = eng.tensor_by_name(f"head{lvl}", B).float().cpu()[..., :3 * (synth_ck.nc + 5)].permute(0, 3, 1, 2)
        ref = m.taps[f"model.24.m.{lvl}"]
        assert (got - ref).abs().max().item() <= 1e-3, f"head {lvl}"
    torch.testing.assert_close(pred.cpu(), pred_ref, rtol=1e-4, atol=1e-3)


def test_forward_raw_fp32_640(lib, synth_ck, tiles_640):
    from oracle import yolov5_oracle as O
    eng = _engine(synth_ck, "fp32")
    m = O.model_from_checkpoint(synth_ck)
    ref = m.forward(O.preprocess(tiles_640))
    pred = eng.forward_raw(torch.from_numpy(tiles_640).cuda()).cpu()
    assert pred.shape == ref.shape == (3, 25200, 10)
    # confidences (sigmoid outputs) within 1e-4 absolute; boxes within 1e-4 of the image size (0.064 px)
    assert (pred[..., 4:] - ref[..., 4:]).abs().max().item() <= 1e-4
    assert (pred[..., :4] - ref[..., :4]).abs().max().item() <= 640 * 1e-4


def _match(dets, counts, ref_list, box_tol, conf_tol):
    """Same number of boxes per tile, and a one-to-one pairing (same class) within tolerance.  Rows are compared
    as a set, not by position: two confidences that differ by 1e-7 may legitimately sort in either order."""
    for b, ref in enumerate(ref_list):
        n = int(counts[b])
        assert n == ref.shape[0], f"tile {b}: {n} boxes vs oracle {ref.shape[0]}"
        got = dets[b, :n]
        assert np.all(np.diff(got[:, 4]) <= 0), f"tile {b}: confidences not descending"
        used = np.zeros(n, bool)
        for r in ref:
            d = np.abs(got[:, :4] - r[:4]).max(1) / box_tol + np.abs(got[:, 4] - r[4]) / conf_tol
            d[used | (got[:, 5] != r[5])] = np.inf
            j = int(np.argmin(d))
            assert np.abs(got[j, :4] - r[:4]).max() <= box_tol and abs(got[j, 4] - r[4]) <= conf_tol, \
                f"tile {b}: oracle box {r} has no engine match (closest {got[j]})"
            used[j] = True


def test_infer_fp32_matches_oracle_detections(lib, synth_ck, tiles_640):
    """The north-star gate: identical post-NMS box counts, boxes/conf within 1e-4 (boxes normalised by 640)."""
    from oracle import yolov5_oracle as O
    eng = _engine(synth_ck, "fp32")
    m = O.model_from_checkpoint(synth_ck)
    ref = O.detect_tiles(m, tiles_640)
    dets, counts = eng.infer(torch.from_numpy(tiles_640).cuda())
    _match(dets.cpu().numpy(), counts.cpu().numpy(), ref, box_tol=640 * 1e-4, conf_tol=1e-4)
    assert sum(r.shape[0] for r in ref) > 100   # the case actually exercises NMS


def test_infer_f16x3_matches_oracle_detections(lib, synth_ck, tiles_640):
    """The same north-star gate on the FAST parity mode (--precision f16x3: fp32 activations, conv products as three fp16 MFMAs on
    hi / lo halves): identical post-NMS box counts, boxes / conf within 1e-4 of the fp32 oracle."""
    from oracle import yolov5_oracle as O
    eng = _engine(synth_ck, "f16x3")
    m = O.model_from_checkpoint(synth_ck)
    ref = O.detect_tiles(m, tiles_640)
    dets, counts = eng.infer(torch.from_numpy(tiles_640).cuda())
    _match(dets.cpu().numpy(), counts.cpu().numpy(), ref, box_tol=640 * 1e-4, conf_tol=1e-4)
    pred = eng.forward_raw(torch.from_numpy(tiles_640).cuda()).cpu()
    refp = m.forward(O.preprocess(tiles_640))
    assert (pred[..., 4:] - refp[..., 4:]).abs().max().item() <= 1e-4
    assert (pred[..., :4] - refp[..., :4]).abs().max().item() <= 640 * 1e-4


def test_nms_kernel_bitexact_on_oracle_pred(lib, synth_ck, tiles_640):
    """S2 alone: feed the ORACLE's pred to the HIP NMS; output must equal the oracle's NMS bit for bit."""
    from aquaculture_amd import engine
    from oracle import yolov5_oracle as O
    m = O.model_from_checkpoint(synth_ck)
    pred = m.forward(O.preprocess(tiles_640))
    ref = O.non_max_suppression(pred.numpy())
    dets, counts = engine.nms(pred.cuda().contiguous(), synth_ck.nc)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    for b, r in enumerate(ref):
        assert counts[b] == r.shape[0]
        assert np.array_equal(dets[b, :counts[b]], r)


def test_infer_bf16_close_to_emulated_oracle(lib, synth_ck, tiles_640):
    """bf16 engine vs an oracle that rounds weights/activations to bf16 at the same points.

    Measured on MI355X (rounds 1 and 2): |d conf| mean 6.8e-3, p99.9 4.9e-2, max 8.8e-2; |d box| mean 0.68 px;
    post-NMS counts 312/363/348 vs 313/357/346.  tests/test_gpu_bf16_deviation.py shows where it comes from: module by
    module, for every kernel selection, the engine deviates from an fp64-accumulation oracle by 0.6-1.04x what
    fp32-vs-fp64 accumulation alone causes inside the oracle (profiles/r02_bf16_deviation.json) -- accumulation-order
    noise re-rounded to bf16 layer after layer and amplified by the synthetic head, not a kernel's own error.  The
    bounds below are 1.5x the measured values; the fp32 mode is the 1e-4 parity gate."""
    from oracle import yolov5_oracle as O
    eng = _engine(synth_ck, "bf16")
    m = O.model_from_checkpoint(synth_ck, O.q_bf16)
    ref = m.forward(O.preprocess(tiles_640))
    t = torch.from_numpy(tiles_640).cuda()
    pred = eng.forward_raw(t).cpu()
    d_conf = (pred[..., 4:] - ref[..., 4:]).abs().flatten()
    assert d_conf.mean().item() <= 1.04e-2
    assert d_conf.kthvalue(int(0.999 * d_conf.numel()))[0].item() <= 7.3e-2
    assert (pred[..., :4] - ref[..., :4]).abs().mean().item() <= 1.02
    ref_counts = [r.shape[0] for r in O.non_max_suppression(ref.numpy())]
    _, counts = eng.infer(t)
    for got, want in zip(counts.cpu().tolist(), ref_counts):
        assert abs(got - want) <= 8            # measured: at most 5 of ~340 boxes per tile over the 16 golden tiles


def test_infer_deterministic(lib, synth_ck, tiles_640):
    """Atomics are used for candidate compaction; results must not depend on their order."""
    eng = _engine(synth_ck, "bf16")
    t = torch.from_numpy(tiles_640).cuda()
    d0, c0 = eng.infer(t)
    d0, c0 = d0.clone(), c0.clone()
    for _ in range(3):
        d1, c1 = eng.infer(t)
        assert torch.equal(c0, c1)
        for b in range(t.shape[0]):
            assert torch.equal(d0[b, :c0[b]], d1[b, :c1[b]])


MAX_DIFFERING_LINES = 4      # of 5,581 golden label lines (round 3's 0.5 % bound allowed 27); measured on MI355X in round 4: 2, both named in the test's output


def test_fp32_engine_reproduces_golden_detections_and_label_text(lib, synth_ck):
    """All 16 config-1 tiles against the committed fixtures (tests/golden/g3_*): same box count per tile, boxes/conf
    within 1e-4, and the label text the reference's consumer parses is the same (a digit may differ only where a
    coordinate sits within 1e-4 px of a rounding boundary)."""
    import json
    import os
    from aquaculture_amd import postprocess, tiles
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, "g3_detections_640.npz"))
    with open(os.path.join(gold, "g3_labels_640.json")) as f:
        labels = json.load(f)
    eng = _engine(synth_ck, "fp32")
    x = tiles.synthetic_batch(range(16), 640)
    dets, counts = eng.infer(torch.from_numpy(x).cuda())
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    _match(dets, counts, [g[f"det_{i}"] for i in range(16)], box_tol=640 * 1e-4, conf_tol=1e-4)
    total = same = 0
    differing = []                                        # (tile, golden line) without a partner in the engine's file
    for i in range(16):
        got = postprocess.format_rows(postprocess.detections_to_rows(dets[i, :counts[i]], (640, 640), (640, 640))).splitlines()
        want = labels[tiles.tile_name(i)].split("\n")
        assert len(got) == len(want)
        # `cls xc yc w h` come from integer-rounded pixels: identical TEXT; conf is printed to 6 significant digits, so
        # an fp32-rounding-level difference (1e-6) may change its last digit: compare it as a number.
        pool = {}
        for l in got:
            f = l.split()
            pool.setdefault(" ".join(f[:5]), []).append(float(f[5]))
        for l in want:
            f = l.split()
            cands = pool.get(" ".join(f[:5]), [])
            hit = [c for c in cands if abs(c - float(f[5])) <= 1e-4]
            if hit:
                cands.remove(hit[0])
                same += 1
            else:
                differing.append((i, l))
        total += len(want)
    # The consumer is digit-sensitive, so the bound is a COUNT of named lines, not a percentage: a box whose fp32 coordinate
    # sits within the engine-vs-oracle difference (<= 1e-4 x 640 px, asserted above) of a .5 rounding boundary prints a neighbouring integer
    # pixel.  Measured on MI355X (round 4): see MAX_DIFFERING_LINES below; every such line must be one of those boundary cases.
    print(f"golden labels: {same}/{total} lines identical as text (conf compared as a number); differing: {differing}")
    assert len(differing) <= MAX_DIFFERING_LINES, f"{len(differing)} of {total} label lines differ: {differing}"


def test_full_batch_is_batch_invariant(lib, synth_ck):
    """Size-independent property at BASELINE.json's batch (64 tiles, 640x640, bf16): every tile's detections are bit-identical
    to running that tile in a batch of 4 -- tiles are independent work units (reference src/load_data/tile_tifs.py:33-47)."""
    from aquaculture_amd import tiles
    eng = _engine(synth_ck, "bf16")
    x = torch.from_numpy(np.concatenate([tiles.synthetic_batch(range(8), 640)] * 8, 0)).cuda()     # 64 tiles
    d64, c64 = eng.infer(x)
    d64, c64 = d64.clone(), c64.clone()
    assert c64.shape[0] == 64 and int(c64.min()) > 0
    for rep in range(1, 8):                                   # the 8 copies of each tile agree with each other
        assert torch.equal(c64[:8], c64[8 * rep:8 * rep + 8])
    d4, c4 = eng.infer(x[:4].contiguous())
    assert torch.equal(c4, c64[:4])
    for b in range(4):
        assert torch.equal(d4[b, :c4[b]], d64[b, :c64[b]])


def _oracle_letterbox(im):
    """oracle/ref_kernels.c ref_letterbox_u8: the scalar C restatement of OpenCV's 8-bit INTER_LINEAR resize + border 114."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle")], check=True)
    cl = C.CDLL(os.path.join(root, "oracle", "_build", "libref_kernels.so"))
    im = np.ascontiguousarray(im)
    hw = (C.c_int * 2)()
    u8 = C.POINTER(C.c_uint8)
    cl.ref_letterbox_u8(im.ctypes.data_as(u8), im.shape[0], im.shape[1], 640, 640, 1, 1, 32, None, hw)
    out = np.empty((hw[0], hw[1], 3), np.uint8)
    cl.ref_letterbox_u8(im.ctypes.data_as(u8), im.shape[0], im.shape[1], 640, 640, 1, 1, 32, out.ctypes.data_as(u8), hw)
    return out


@pytest.mark.parametrize("shape", [(1024, 1024), (640, 640), (500, 700), (1000, 600), (320, 480)])
def test_device_letterbox_is_bit_exact_with_the_oracle(lib, shape):
    """aq_letterbox_u8 vs the C oracle (oracle/ref_kernels.c ref_letterbox_u8, OpenCV's fixed-point INTER_LINEAR restated as scalar
    loops + border 114): identical bytes; the product's host restatement (dataloader.letterbox) is held to the same oracle.
    (The oracle is unpinned against OpenCV itself: none is installed; tests/golden/g9_letterbox.json pins it against regressions.)"""
    from aquaculture_amd import dataloader, engine
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    ims = rng.integers(0, 256, (3,) + shape + (3,), dtype=np.uint8)
    want = np.stack([_oracle_letterbox(im) for im in ims], 0)
    got = engine.letterbox_device(torch.from_numpy(ims).cuda()).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(dataloader.letterbox(ims[0]), want[0])


@pytest.mark.parametrize("shape,batch", [((384, 640), 1), ((640, 384), 3), ((32, 32), 2), ((96, 160), 5)])
def test_fp32_engine_on_ragged_shapes(lib, synth_ck, shape, batch):
    """Letterbox with auto=True yields any multiple-of-32 rectangle (e.g. 480x640 for a 500x700 image); batches can be
    of any size (the last batch of a shard).  Same engine object, changing geometry between calls."""
    from aquaculture_amd import tiles
    from oracle import yolov5_oracle as O
    H, W = shape
    eng = _engine(synth_ck, "fp32")
    m = O.model_from_checkpoint(synth_ck)
    big = tiles.synthetic_batch(range(batch), max(640, H, W))
    x = np.ascontiguousarray(big[:, :H, :W, :])
    ref_pred = m.forward(O.preprocess(x))
    t = torch.from_numpy(x).cuda()
    pred = eng.forward_raw(t).cpu()
    assert pred.shape == ref_pred.shape == (batch, 3 * (H // 8 * (W // 8) + H // 16 * (W // 16) + H // 32 * (W // 32)), 10)
    assert (pred[..., 4:] - ref_pred[..., 4:]).abs().max().item() <= 1e-4
    assert (pred[..., :4] - ref_pred[..., :4]).abs().max().item() <= 640 * 1e-4   # boxes scale with the anchors (up to 373 px), not the tile
    ref = O.non_max_suppression(ref_pred.numpy())
    dets, counts = eng.infer(t)
    assert counts.cpu().tolist() == [r.shape[0] for r in ref]
    # a second geometry on the same engine (workspace and layout are re-planned per call)
    x2 = np.ascontiguousarray(big[:1, :64, :96, :])
    p2 = eng.forward_raw(torch.from_numpy(x2).cuda()).cpu()
    r2 = m.forward(O.preprocess(x2))
    assert (p2[..., 4:] - r2[..., 4:]).abs().max().item() <= 1e-4


def test_engine_rejects_bad_input(lib, synth_ck):
    """Error behaviour at the boundary: wrong dtype / layout / size raise, nothing is launched."""
    eng = _engine(synth_ck, "bf16")
    with pytest.raises(ValueError):
        eng.infer(torch.zeros(1, 640, 640, 3, dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        eng.infer(torch.zeros(1, 3, 640, 640, dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError, match="multiple of the max stride"):
        eng.infer(torch.zeros(1, 100, 100, 3, dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError, match="bad thresholds"):
        eng.infer(torch.zeros(1, 64, 64, 3, dtype=torch.uint8, device="cuda"), conf_thres=1.5)


@pytest.mark.parametrize("scene_hw,tile_hw", [((2048, 3072), (1024, 1024)), ((1500, 2500), (476, 1024)), ((1500, 2500), (1024, 452)),
                                              ((700, 900), (640, 640))])
def test_scene_tiles_letterbox_equals_letterbox_of_explicit_crops(lib, scene_hw, tile_hw):
    """aq_letterbox_tiles_u8 (tiles cut out of one scene raster by the letterbox kernel, reference src/load_data/tile_tifs.py:33-47)
    == aq_letterbox_u8 on the same tiles copied out first == the host restatement: identical bytes."""
    from aquaculture_amd import dataloader, engine, scenes
    rng = np.random.default_rng(scene_hw[0] + tile_hw[1])
    scene = rng.integers(0, 256, scene_hw + (3,), dtype=np.uint8)
    h, w = tile_hw
    origins = [(x0, y0) for x0, y0, tw, th in scenes.tile_grid(scene_hw[1], scene_hw[0], max(h, w) if (h, w) != (640, 640) else 640)
               if (th, tw) == (h, w)]
    if (h, w) == (640, 640):
        origins = [(0, 0), (260, 60), (17, 33)]                 # any in-raster origin works, not only grid points
    assert origins
    crops = np.stack([scene[y0:y0 + h, x0:x0 + w] for x0, y0 in origins], 0)
    dev = torch.from_numpy(scene).cuda()
    got = engine.letterbox_scene_tiles(dev, origins, (h, w)).cpu().numpy()
    ref = engine.letterbox_device(torch.from_numpy(np.ascontiguousarray(crops)).cuda()).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert np.array_equal(got[0], dataloader.letterbox(crops[0]))


def test_scene_tiles_outside_the_raster_are_refused(lib):
    from aquaculture_amd import engine
    dev = torch.zeros((1100, 1100, 3), dtype=torch.uint8, device="cuda")
    for origin in [(100, 0), (0, 77), (-1, 0)]:
        with pytest.raises(RuntimeError, match="leaves the raster"):
            engine.letterbox_scene_tiles(dev, [(0, 0), origin], (1024, 1024))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(2, 20, 20, 384), (3, 13, 17, 16), (1, 5, 3, 8), (2, 40, 40, 64), (1, 80, 80, 32)])
def test_sppf_pools_match_chained_max_pool2d(lib, precision, shape):
    """aq_sppf_pool: y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2) written into slices 1..3 of the [x|y1|y2|y3] buffer ([UPSTREAM SPPF.forward]:
    three chained MaxPool2d(5, 1, 2)); bit-exact against torch (max has no rounding).  Small planes run the fused in-LDS kernel
    (separable passes), large ones the three-launch path; the buffer has a channel offset and extra channels on both sides."""
    import ctypes as C
    import torch.nn.functional as F
    from aquaculture_amd import engine
    B, H, W, c = shape
    dt = torch.bfloat16 if precision == "bf16" else torch.float32
    g = torch.Generator().manual_seed(H * 31 + c)
    ld, off = 4 * c + 16, 8
    buf = torch.full((B, H, W, ld), -7.0, dtype=dt)
    x = torch.randn(B, H, W, c, generator=g).to(dt)
    buf[..., off:off + c] = x
    dev = buf.cuda()
    engine._check(lib.aq_sppf_pool(dev.data_ptr(), ld, off, c, B, H, W, 0 if precision == "bf16" else 1, torch.cuda.current_stream().cuda_stream))
    got = dev.cpu()
    y = x.float().permute(0, 3, 1, 2)
    for s in range(1, 4):
        y = F.max_pool2d(y, 5, 1, 2)
        assert torch.equal(got[..., off + s * c: off + (s + 1) * c].float(), y.permute(0, 2, 3, 1)), s
    assert torch.equal(got[..., off:off + c], x) and (got[..., :off] == -7.0).all() and (got[..., off + 4 * c:] == -7.0).all()


def test_benchmark_configuration_nms_invariants(lib, synth_ck):
    """The exact bench.py configuration -- 64 tiles, 640x640, bf16, AUTOTUNED kernels (direct / fused / one-workgroup-per-tile forms
    included) -- checked through properties that need no oracle at this size: per tile the detections are sorted by confidence, above
    the threshold, at most max_det, finite, and no two kept boxes of one class overlap by more than the IoU threshold (what
    [UPSTREAM non_max_suppression] + torchvision nms guarantee).  The tuned engine also agrees with the untuned one to bf16 noise."""
    from aquaculture_amd import tiles
    conf_thres, iou_thres, max_det = 0.25, 0.45, 1000
    x = torch.from_numpy(tiles.synthetic_batch(range(64), 640)).cuda()
    base = _engine(synth_ck, "bf16")
    d0, c0 = base.infer(x, conf_thres, iou_thres, max_det)
    d0, c0 = d0.cpu(), c0.cpu()
    eng = _engine(synth_ck, "bf16")
    cfgs = eng.autotune(x)
    assert any(c >= 4096 for c in cfgs) or any(c in (1000, 1001, 1002) for c in cfgs)       # the tuner did pick non-default forms
    dets, counts = eng.infer(x, conf_thres, iou_thres, max_det)
    dets, counts = dets.cpu(), counts.cpu()
    assert counts.shape == (64,) and int(counts.max()) <= max_det and int(counts.sum()) > 1000
    for b in range(64):
        n = int(counts[b])
        d = dets[b, :n]
        assert torch.isfinite(d).all()
        conf, cls = d[:, 4], d[:, 5]
        assert (conf >= conf_thres).all() and (conf[:-1] >= conf[1:]).all() and ((cls >= 0) & (cls < 5) & (cls == cls.round())).all()
        assert (d[:, 2] >= d[:, 0]).all() and (d[:, 3] >= d[:, 1]).all()     # w = (2 sigmoid)^2 * anchor may underflow to a zero-width box
        if n > 1:
            # IoU as the suppression itself sees it: on the class-offset boxes (box + cls * max_wh, max_wh = 7680 [UPSTREAM
            # non_max_suppression]) in fp32.  At coordinates of up to 3e4 one fp32 ulp is 0.002-0.004 px, so the IoU of the raw boxes
            # can sit 1e-4 away from the one that was thresholded; on the offset boxes the two agree to rounding.
            bx = d[:, :4] + (cls * 7680.0)[:, None]
            x1 = torch.maximum(bx[:, None, 0], bx[None, :, 0]); y1 = torch.maximum(bx[:, None, 1], bx[None, :, 1])
            x2 = torch.minimum(bx[:, None, 2], bx[None, :, 2]); y2 = torch.minimum(bx[:, None, 3], bx[None, :, 3])
            inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
            area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
            iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=1e-9)
            same = (cls[:, None] == cls[None, :]) & ~torch.eye(n, dtype=torch.bool)
            assert float(iou[same].max()) <= iou_thres + 1e-5 if same.any() else True, (b, float(iou[same].max()))
    # tuned vs untuned kernels: same detections up to bf16 accumulation-order noise
    assert (counts - c0).abs().max() <= max(4, int(0.05 * int(c0.max())))
    assert abs(int(counts.sum()) - int(c0.sum())) <= 0.01 * int(c0.sum())
