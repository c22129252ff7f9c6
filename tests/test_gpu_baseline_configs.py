"""BASELINE.json configs[3] and configs[4] AT THEIR OWN SIZES: yolov5m fp8 at batch 128 / 640x640 and yolov5x bf16
at batch 16 / 1280x1280 through the C ABI, with the shipped tile-configuration tables the bench uses.  Tile counts decide which planar
tiling or fallback a layer gets, so the small-batch parity tests (tests/test_gpu_fp8.py, tests/test_gpu_yolov5x.py) say nothing about
what runs here.  No oracle finishes in seconds at these sizes; the checks are the size-independent properties of the path:
  * what [UPSTREAM non_max_suppression] + torchvision nms guarantee per tile (sorted by confidence, above the threshold, at most max_det,
    finite, valid classes, no two kept boxes of a class above the IoU threshold on the class-offset boxes);
  * batch invariance: a tile's detections are bit-identical whether it runs in the full batch or in a batch of 4 (tiles are independent
    work units, reference src/load_data/tile_tifs.py:33-47), and identical copies of a tile inside one batch agree;
  * which kernel family every 3x3 layer ACTUALLY launched (aq_engine_last_launch): the fp8 pairs must not fall back to bf16 at batch 128.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def nms_invariants(dets, counts, conf_thres, iou_thres, max_det, nc):
    dets, counts = dets.cpu(), counts.cpu()
    assert int(counts.max()) <= max_det and int(counts.min()) >= 0
    for b in range(counts.shape[0]):
        n = int(counts[b])
        d = dets[b, :n]
        assert torch.isfinite(d).all(), b
        conf, cls = d[:, 4], d[:, 5]
        assert (conf >= conf_thres).all() and (conf[:-1] >= conf[1:]).all() and ((cls >= 0) & (cls < nc) & (cls == cls.round())).all(), b
        assert (d[:, 2] >= d[:, 0]).all() and (d[:, 3] >= d[:, 1]).all(), b
        if n > 1:
            # IoU as the suppression sees it: class-offset boxes (box + cls * 7680 [UPSTREAM non_max_suppression max_wh]) in fp32
            bx = d[:, :4] + (cls * 7680.0)[:, None]
            x1 = torch.maximum(bx[:, None, 0], bx[None, :, 0]); y1 = torch.maximum(bx[:, None, 1], bx[None, :, 1])
            x2 = torch.minimum(bx[:, None, 2], bx[None, :, 2]); y2 = torch.minimum(bx[:, None, 3], bx[None, :, 3])
            inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
            area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
            iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=1e-9)
            same = (cls[:, None] == cls[None, :]) & ~torch.eye(n, dtype=torch.bool)
            if same.any():
                assert float(iou[same].max()) <= iou_thres + 1e-5, (b, float(iou[same].max()))


def conv3x3_ops(eng):
    from aquaculture_amd import spec
    return [i for i, o in enumerate(eng.plan.ops) if o.kind == spec.OP_CONV and o.k == 3]


def test_configs3_fp8_engine_at_batch_128(lib, synth_ck):
    """yolov5m, fp8 (e4m3 on both MFMA operands of the 14 wide Bottleneck 3x3 layers), batch 128, 640x640 = BASELINE.json configs[3]."""
    from aquaculture_amd import engine, tiles
    base = tiles.synthetic_batch(range(16), 640)
    x = torch.from_numpy(np.concatenate([base] * 8, 0)).cuda()                  # 128 tiles: 8 copies of 16
    eng = engine.Engine(synth_ck, "fp8", fp8_calibration=x[:16])
    eng.autotune(x)
    assert eng.tuned_from == "shipped table"                                    # the bench's kernels, not a fresh timing
    d128, c128 = eng.infer(x, 0.25, 0.45, 1000)
    d128, c128 = d128.clone(), c128.clone()
    fam128 = eng.last_launches()
    pairs = eng.fp8_pairs()
    assert len(pairs) == 14
    for prod, cons in pairs:                                                    # no silent fall-back to bf16 at this batch size
        assert fam128[cons][0] == "pl3x3_f8", (eng.plan.ops[cons].name, fam128[cons])
        assert fam128[prod][0] == "direct1x1_f8out", (eng.plan.ops[prod].name, fam128[prod])
    others = [i for i in conv3x3_ops(eng) if i not in {c for _, c in pairs}]
    assert {fam128[i][0] for i in others} <= {"pl3x3s2", "direct3x3s2", "igemm_or_halo"}, [(eng.plan.ops[i].name, fam128[i]) for i in others]
    assert sum(fam128[i][0] == "pl3x3s2" for i in others) == 4                  # model.5 / 7 / 18 / 21 on the planar stride-2 family
    assert sum(f == "bottleneck" for f, _ in fam128) == 8 and sum(f == "downblock" for f, _ in fam128) == 1 and sum(f == "head_decode" for f, _ in fam128) == 3
    # the eight 1x1 layers with K >= 768 (model.8 / 23 cv1|cv2 and cv3, SPPF cv1 / cv2, model.10, model.13 cv1|cv2) on the assembly 1x1 of round 4
    wide = [i for i, o in enumerate(eng.plan.ops) if o.kind == 1 and o.k == 1 and o.level < 0 and o.src.channels >= 768]
    assert len(wide) == 8 and all(fam128[i][0] == "asm1x1" for i in wide), [(eng.plan.ops[i].name, fam128[i]) for i in wide]
    assert int(c128.sum()) > 2000
    nms_invariants(d128, c128, 0.25, 0.45, 1000, 5)
    for rep in range(1, 8):                                                     # the 8 copies of each tile agree, box for box
        assert torch.equal(c128[:16], c128[16 * rep:16 * rep + 16]), rep
        for b in range(16):
            assert torch.equal(d128[b, :c128[b]], d128[16 * rep + b, :c128[b]]), (rep, b)
    d4, c4 = eng.infer(x[:4].contiguous(), 0.25, 0.45, 1000)
    fam4 = eng.last_launches()
    assert [fam4[c][0] for _, c in pairs] == ["pl3x3_f8"] * 14                  # same kernels at batch 4, hence bit-identical tiles
    assert torch.equal(c4, c128[:4])
    for b in range(4):
        assert torch.equal(d4[b, :c4[b]], d128[b, :c128[b]]), b
    eng.close()


def test_configs4_yolov5x_bf16_at_batch_16_1280px(lib):
    """yolov5x, bf16, batch 16, 1280x1280 tiles (100,800 candidates per tile) = one GPU's share of BASELINE.json configs[4]."""
    from aquaculture_amd import checkpoint, engine, tiles
    ck = checkpoint.synthetic_checkpoint("yolov5x", 5)
    base = tiles.synthetic_batch([0, 1, 19, 40], 1280)
    x = torch.from_numpy(np.concatenate([base] * 4, 0)).cuda()                  # 16 tiles: 4 copies of 4
    eng = engine.Engine(ck, "bf16")
    assert eng.plan.variant == "yolov5x" and eng.num_candidates(1280, 1280) == 100800
    eng.autotune(x)
    assert eng.tuned_from == "shipped table"
    d16, c16 = eng.infer(x, 0.25, 0.45, 1000)
    d16, c16 = d16.clone(), c16.clone()
    fam16 = eng.last_launches()
    ops3 = conv3x3_ops(eng)
    assert len(ops3) >= 49 and all(fam16[i][0] != "none" for i in ops3)       # every 3x3 layer of yolov5x ran, on SOME family ...
    fams = {}
    for i in ops3:
        fams.setdefault(fam16[i][0], []).append(eng.plan.ops[i].name)
    print("yolov5x 3x3 families at batch 16 / 1280:", {k: len(v) for k, v in fams.items()})
    # ... and which: the planar families cover Cout = k x 192 only (yolov5m's widths), so yolov5x's 160 / 320 / 640 / 1280-channel layers run
    # on the implicit-GEMM / halo kernels -- stated here so that a new family that starts to apply shows up as a test change, not silently
    assert set(fams) <= {"igemm_or_halo", "pl3x3", "pl3x3s2", "direct3x3s2"}
    assert int(c16.min()) > 0
    nms_invariants(d16, c16, 0.25, 0.45, 1000, 5)
    for rep in range(1, 4):
        assert torch.equal(c16[:4], c16[4 * rep:4 * rep + 4]), rep
        for b in range(4):
            assert torch.equal(d16[b, :c16[b]], d16[4 * rep + b, :c16[b]]), (rep, b)
    d4, c4 = eng.infer(x[:4].contiguous(), 0.25, 0.45, 1000)
    fam4 = eng.last_launches()
    same_kernels = all(fam4[i] == fam16[i] for i in range(len(fam16)))
    assert same_kernels, [(eng.plan.ops[i].name, fam4[i], fam16[i]) for i in range(len(fam16)) if fam4[i] != fam16[i]]
    assert torch.equal(c4, c16[:4])
    for b in range(4):
        assert torch.equal(d4[b, :c4[b]], d16[b, :c16[b]]), b
    eng.close()
