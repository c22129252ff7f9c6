"""model.0 + model.1 + model.2.cv1|cv2 in one launch (aq_stemdown, csrc/downblock.hip STEM form; [UPSTREAM detect.py `im / 255`,
models/common.py Conv.forward_fuse x 3], reached through reference README.md:77): bit-identical to aq_stem_conv followed by aq_downblock,
which have their own parity tests against F.conv2d (tests/test_gpu_conv.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [
    # B, Hi, Wi: the benchmark tile, a ragged one (output 37 x 23: partial tiles on both edges), one tile smaller than a workgroup tile,
    # and more tiles than CUs at a size whose rows are not a multiple of the patch width
    (2, 640, 640), (3, 148, 92), (1, 16, 24), (5, 256, 384),
])
def test_stemdown_is_bit_identical_to_stem_then_downblock(lib, case):
    from aquaculture_amd import engine
    B, Hi, Wi = case
    g = torch.Generator().manual_seed(Hi + Wi)
    tiles = torch.randint(0, 256, (B, Hi, Wi, 3), generator=g, dtype=torch.uint8).cuda()
    ws = torch.randn(48, 3, 6, 6, generator=g) * 0.25
    bs = torch.randn(48, generator=g) * 0.3
    wa = torch.randn(96, 48, 3, 3, generator=g) * (2.0 / (9 * 48)) ** 0.5
    ba = torch.randn(96, generator=g) * 0.2
    wb = torch.randn(96, 96, 1, 1, generator=g) * (2.0 / 96) ** 0.5
    bb = torch.randn(96, generator=g) * 0.2
    x = engine.stem_conv_nhwc(tiles, ws, bs, act=True, precision="bf16")
    ref = engine.downblock_nhwc(x, wa, ba, wb, bb)
    outw = torch.full((B, Hi // 4, Wi // 4, 96 + 16), 5.0, dtype=torch.bfloat16, device="cuda")
    got = engine.stemdown_nhwc(tiles, ws, bs, wa, ba, wb, bb, out=outw[..., 8:104])
    assert torch.equal(got.cpu().view(torch.int16), ref.cpu().view(torch.int16))
    assert (outw[..., :8] == 5.0).all() and (outw[..., 104:] == 5.0).all(), "wrote outside its channel slice"
    assert float(ref.float().abs().mean()) > 0.05


def test_engine_with_the_stem_fused_into_the_downblock(lib, synth_ck, monkeypatch):
    """AQ_STEMDOWN=1: the same detections, bit for bit (the fused launch is bit-identical to its two halves)."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([2, 9], 256)).cuda()
    outs = []
    for on in ("0", "1"):
        monkeypatch.setenv("AQ_STEMDOWN", on)
        eng = engine.Engine(synth_ck, "bf16")
        dets, counts = eng.infer(x)
        outs.append((dets.cpu().clone(), counts.cpu().clone()))
        del eng
    assert torch.equal(outs[0][1], outs[1][1]) and int(outs[0][1].sum()) > 10
    for b in range(2):
        n = int(outs[0][1][b])
        assert torch.equal(outs[0][0][b, :n], outs[1][0][b, :n])
