"""Detect head fused with its decode (csrc/head_decode.hip; [UPSTREAM models/yolo.py Detect.forward] + the candidate filter of
[UPSTREAM utils/general.py non_max_suppression], reached through reference README.md:77): the kernel against a plain fp32 restatement
on the same bf16-rounded operands, and the engine's `infer` with and without the fusion (AQ_DISABLE_HEAD_FUSION=1 = conv + decode)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ANCHORS = [(10.0, 13.0), (16.0, 30.0), (33.0, 23.0)]


@pytest.mark.parametrize("case", [
    # B, ny, nx, cin: the three yolov5m levels at 640 px (batch cut down), a ragged one (pixel count not a multiple of 16), yolov5s widths
    (3, 80, 80, 192), (5, 40, 40, 384), (7, 20, 20, 768), (2, 13, 9, 256), (1, 5, 7, 128), (2, 12, 12, 1024),
])
def test_head_decode_matches_reference(lib, case):
    from aquaculture_amd import engine
    B, ny, nx, cin = case
    nc, na, no, stride, off, thr = 5, 3, 10, 16.0, 1000, 0.25
    g = torch.Generator().manual_seed(cin + ny)
    xw = (torch.randn(B, ny, nx, cin + 16, generator=g) * 0.7).bfloat16()
    x = xw[..., 8:8 + cin]
    w = torch.randn(na * no, cin, generator=g) * (1.5 / cin ** 0.5)
    b = torch.randn(na * no, generator=g) * 0.5
    b[4::no] -= 0.5                                          # roughly a third of the candidates pass
    cap = na * ny * nx
    counts, cand, rows = engine.head_decode_level(x.cuda() if False else xw.cuda()[..., 8:8 + cin], w, b, off, stride, ANCHORS, nc, thr, cap)
    raw = (x.float().reshape(-1, cin) @ w.bfloat16().float().t() + b).reshape(B, ny, nx, na, no)
    sig = torch.sigmoid(raw.double()).float()
    yy, xx = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
    ref = torch.empty_like(sig)
    ref[..., 0] = (sig[..., 0] * 2 + (xx - 0.5)[None, :, :, None]) * stride
    ref[..., 1] = (sig[..., 1] * 2 + (yy - 0.5)[None, :, :, None]) * stride
    anch = torch.tensor(ANCHORS)
    ref[..., 2:4] = (sig[..., 2:4] * 2) ** 2 * anch[None, None, None]
    ref[..., 4:] = sig[..., 4:]
    idx = off + (torch.arange(na)[None, None, :] * (ny * nx) + (torch.arange(ny)[:, None, None] * nx + torch.arange(nx)[None, :, None]))
    counts, cand, rows = counts.cpu(), cand.cpu(), rows.cpu()
    total = 0
    for bi in range(B):
        obj = ref[bi, ..., 4]
        sure = (obj - thr).abs() > 2e-4                      # away from the threshold the pass / fail decision must agree
        want = set(idx[sure & (obj > thr)].tolist())
        maybe = set(idx[~sure].tolist())
        n = int(counts[bi])
        got = cand[bi, :n].tolist()
        assert len(set(got)) == n and want <= set(got) <= want | maybe, (bi, n, len(want))
        lut = {int(i): k for k, i in enumerate(got)}
        flat_ref = ref[bi].permute(2, 0, 1, 3).reshape(-1, no)       # candidate order: a, y, x
        sel = torch.tensor([i - off for i in got])
        torch.testing.assert_close(rows[bi, :n], flat_ref[sel], rtol=2e-4, atol=2e-3)
        assert (cand[bi, n:] == -1).all() and lut
        total += n
    assert total > 0.1 * B * cap


def test_engine_infer_with_and_without_head_fusion(lib, synth_ck, monkeypatch):
    """Same detections either way, up to the summation order of the head convs (fp32 accumulate, K split differently)."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([1, 6, 12], 256)).cuda()
    outs = []
    for off in ("1", "0"):
        monkeypatch.setenv("AQ_DISABLE_HEAD_FUSION", off)
        eng = engine.Engine(synth_ck, "bf16")
        dets, counts = eng.infer(x)
        outs.append((dets.cpu().clone(), counts.cpu().clone()))
        del eng
    (d0, c0), (d1, c1) = outs
    assert int(c0.sum()) > 20
    assert (c0 - c1).abs().max() <= 1                                 # a box within 1e-6 of a threshold may flip
    for bi in range(x.shape[0]):
        if c0[bi] != c1[bi]:
            continue
        n = int(c0[bi])
        # the lists are sorted by confidence: two detections whose confidences differ by less than the two paths' rounding may swap places, so
        # match every row to its nearest row of the other list (one to one) instead of comparing position by position
        a, b = d0[bi, :n], d1[bi, :n]
        dist = (a[:, None, :4] - b[None, :, :4]).abs().amax(-1)
        nearest = dist.argmin(1)
        assert sorted(nearest.tolist()) == list(range(n)), "not a one-to-one match"
        b = b[nearest]
        assert (nearest - torch.arange(n)).abs().max() <= 2          # ... and only neighbours swap
        torch.testing.assert_close(a[:, :4], b[:, :4], rtol=0, atol=2e-2)
        torch.testing.assert_close(a[:, 4], b[:, 4], rtol=0, atol=1e-4)
        assert torch.equal(a[:, 5], b[:, 5])
