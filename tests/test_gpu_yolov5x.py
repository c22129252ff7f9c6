"""BASELINE.json configs[4]: the same engine on yolov5x (depth 1.33, width 1.25: 80/160/320/640/1280 channels, 10 groups per
tap, C3 depths 4/8/12/4) -- nothing in the kernels is specific to yolov5m."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ck_x():
    from aquaculture_amd import checkpoint
    return checkpoint.synthetic_checkpoint("yolov5x", 5)


def test_yolov5x_fp32_matches_oracle(lib, ck_x):
    from aquaculture_amd import engine, tiles
    from oracle import yolov5_oracle as O
    x = tiles.synthetic_batch([0, 19], 128)
    m = O.model_from_checkpoint(ck_x)
    ref_pred = m.forward(O.preprocess(x))
    eng = engine.Engine(ck_x, "fp32")
    assert eng.plan.variant == "yolov5x" and eng.plan.channels == (80, 160, 320, 640, 1280)
    pred = eng.forward_raw(torch.from_numpy(x).cuda()).cpu()
    assert (pred[..., 4:] - ref_pred[..., 4:]).abs().max().item() <= 1e-4
    dbox = (pred[..., :4] - ref_pred[..., :4]).abs().max().item()
    print(f"yolov5x fp32 vs oracle at 128 px: max |d box| {dbox:.3e} px, max |d conf| {(pred[..., 4:] - ref_pred[..., 4:]).abs().max().item():.3e}")
    # north_star's 1e-4 of the tile size, with no slack on top (round 3 added 1e-3 px): measured on MI355X (round 4) 9.1e-3 px on boxes
    # whose extent the anchors set (up to 373 px x 4, whatever the tile size), and 2.5e-5 on confidences
    assert dbox <= 128 * 1e-4
    ref = O.non_max_suppression(ref_pred.numpy())
    dets, counts = eng.infer(torch.from_numpy(x).cuda())
    assert counts.cpu().tolist() == [r.shape[0] for r in ref]


def test_yolov5x_bf16_1280px_tiles(lib, ck_x):
    """1280x1280 tiles (100,800 candidates per tile): runs, is deterministic and batch invariant."""
    from aquaculture_amd import engine, tiles
    eng = engine.Engine(ck_x, "bf16")
    x = torch.from_numpy(tiles.synthetic_batch([0, 1], 1280)).cuda()
    assert eng.num_candidates(1280, 1280) == 100800
    d2, c2 = eng.infer(x)
    d2, c2 = d2.clone(), c2.clone()
    d1, c1 = eng.infer(x[:1].contiguous())
    assert int(c2.min()) > 0 and c1[0] == c2[0] and torch.equal(d1[0, :c1[0]], d2[0, :c2[0]])
    assert float(d2[0, :c2[0], 4].min()) > 0.25 and float(d2[0, :c2[0], 5].max()) <= 4
