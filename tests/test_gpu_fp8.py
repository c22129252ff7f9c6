"""fp8 on both MFMA operands (BASELINE.json configs[3] "fp8 weights (CDNA4 fp8 MFMA)"): the planar 3x3 kernel's fp8 family
(gen_conv3x3_pl_asm.py f8nb13, v_mfma_f32_16x16x128_f8f6f4) against F.conv2d on the SAME quantised operands -- the products of two e4m3
values are exact in fp32, so the only differences are fp32 summation order and the bf16 rounding of the output."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def test_host_e4m3_rounding_is_torchs(lib):
    """aq_f32_to_e4m3 (the packer's weight quantiser) == torch.float8_e4m3fn's round-to-nearest-even on a dense sweep of the range,
    every code's exact value, every midpoint between neighbouring codes, subnormals and the saturation edge."""
    codes = torch.arange(0, 127, dtype=torch.uint8)                        # 0 .. 0x7e: the finite non-negative codes
    vals = codes.view(torch.float8_e4m3fn).float()
    mids = (vals[:-1] + vals[1:]) / 2
    g = torch.Generator().manual_seed(7)
    sweep = torch.cat([vals, mids, torch.nextafter(mids, torch.tensor(0.0)), torch.nextafter(mids, torch.tensor(1e9)),
                       torch.rand(20000, generator=g) * 460.0, torch.rand(20000, generator=g) * 0.05, torch.tensor([447.9, 448.0])])
    sweep = torch.cat([sweep, -sweep])
    sweep = sweep[sweep.abs() <= 448.0]
    want = sweep.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([lib.aq_f32_to_e4m3(float(v)) for v in sweep.tolist()], dtype=np.uint8)
    bad = np.nonzero((got != want) & ~((want & 0x7f) == 0) | ((got & 0x7f) != (want & 0x7f)))[0]     # (the sign of a zero may differ)
    assert bad.size == 0, [(float(sweep[i]), hex(got[i]), hex(want[i])) for i in bad[:8]]


F8_CASES = [
    # B, H, W, Cin, Cout, residual mode (None / "sep" / "inplace"), act
    (3, 40, 40, 192, 192, "inplace", True),    # yolov5m model.6 Bottleneck.cv2: three 64-channel chunks, in-place shortcut
    (2, 20, 20, 384, 384, "sep", True),        # model.8: two M tiles, six chunks
    (1, 9, 7, 192, 192, "sep", True),          # sub-tile ragged image: every border case inside one tile
    (2, 13, 24, 128, 192, None, True),         # two chunks only (the ring wraps into the next tile at once)
    (5, 20, 20, 192, 192, None, False),        # no activation, no shortcut
    (40, 40, 40, 192, 192, "inplace", True),   # more tiles than CUs: persistent workgroups walk several tiles
    (70, 20, 20, 384, 384, "sep", True),       # the same with two M tiles
    (3, 10, 10, 768, 768, "sep", True),        # four M tiles, twelve chunks
    (2, 6, 6, 64, 192, None, True),            # ONE chunk: every tile starts and ends the ring
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", F8_CASES)
def test_planar_conv3x3_f8_matches_reference(lib, case):
    from aquaculture_amd import engine
    B, H, W, cin, c, resmode, act = case
    assert lib.aq_conv3x3_pl_f8_supported(cin, c, B, H, W)
    g = torch.Generator().manual_seed(c * 3 + H * 5 + cin)
    x = torch.randn(B, H, W, cin + 32, generator=g).abs() * 0.9 - 0.25          # SiLU-like range: mostly positive, a negative tail
    act_scale = float(x.abs().max()) / 448.0
    xq = (x / act_scale).to(torch.float8_e4m3fn)                                 # the codes a producing kernel would have written
    xw = xq.cuda()
    xs = xw[..., 16:16 + cin]
    w = torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    w = w * (10.0 ** torch.linspace(-1, 1, c)).view(-1, 1, 1, 1)                 # per-channel scales two orders of magnitude apart
    b = torch.randn(c, generator=g) * 0.2
    ws = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0
    wq = (w / ws).to(torch.float8_e4m3fn)
    outw = torch.full((B, H, W, c + 24), 7.0, dtype=torch.bfloat16, device="cuda")
    out = outw[..., 16:16 + c]
    res = None
    if resmode == "sep":
        resw = (torch.randn(B, H, W, c + 8, generator=g)).bfloat16().cuda()
        res = resw[..., 8:]
    elif resmode == "inplace":
        out.copy_((torch.randn(B, H, W, c, generator=g)).bfloat16())
        res = out
    res_host = res.float().cpu().clone() if res is not None else None
    engine.conv3x3_pl_f8_nhwc(xs, act_scale, w, b, act, residual=res, out=out)
    ref = F.conv2d(xq[..., 16:16 + cin].float().double().permute(0, 3, 1, 2), wq.float().double(), None, padding=1)
    ref = ref * (act_scale * ws.view(1, -1, 1, 1).double()) + b.double().view(1, -1, 1, 1)
    ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
    if res_host is not None:
        ref = ref + res_host.double()
    got = out.float().cpu()
    assert (outw[..., :16] == 7.0).all() and (outw[..., 16 + c:] == 7.0).all(), "wrote outside its channel slice"
    scale = ref.abs().amax(dim=(0, 1, 2)).clamp_min(1e-6).float()
    err = ((got - ref.float()).abs() / scale).max().item()
    assert err <= 2 ** -7, err                                                   # bf16 output rounding (+ fp32 summation order)
    torch.testing.assert_close(got, ref.float().bfloat16().float(), rtol=2 ** -6, atol=2e-2 * float(scale.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("c,H,W", [(192, 40, 40), (384, 20, 20), (192, 7, 9)])
def test_direct_conv1x1_writes_the_e4m3_codes_of_its_output(lib, c, H, W):
    """The producer half of an fp8 pair: aq_conv1x1_direct_f8out == float8_e4m3fn(SiLU(conv1x1) / scale) -- code for code, but for
    values whose fp32 sum lands on another side of a rounding boundary in another summation order (a fraction of a percent, one code apart)."""
    from aquaculture_amd import engine
    B = 3
    g = torch.Generator().manual_seed(c + H)
    xw = (torch.randn(B, H, W, c + 16, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:8 + c]
    w = torch.randn(c, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    b = torch.randn(c, generator=g) * 0.2
    ref = F.silu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float(), b)).permute(0, 2, 3, 1)
    scale = float(ref.abs().max()) / 448.0 * 0.9                                # 10 % of the range saturates: the clamp is exercised
    want = (ref / scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    wk = np.ascontiguousarray(w.permute(0, 2, 3, 1).float().numpy())
    n = C.c_size_t()
    wp = wk.ctypes.data_as(C.POINTER(C.c_float))
    engine._check(lib.aq_pack_conv1x1_direct(wp, c, c, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device="cuda")
    engine._check(lib.aq_pack_conv1x1_direct(wp, c, c, wbuf.data_ptr(), C.byref(n), engine._stream_ptr()))
    bbuf = b.float().cuda()
    pitch = 2 * c + 32                                                            # codes live in the first bytes of a wider (bf16-sized) row
    out = torch.full((B, H, W, pitch), 0xAA, dtype=torch.uint8, device="cuda")
    engine._check(lib.aq_conv1x1_direct_f8out(x.data_ptr(), c + 16, 0, out.data_ptr(), pitch, 16, c, c, wbuf.data_ptr(), bbuf.data_ptr(),
                                              B * H * W, 1, scale, engine._stream_ptr()))
    torch.cuda.synchronize()
    got = out[..., 16:16 + c].cpu()
    assert (out[..., :16] == 0xAA).all() and (out[..., 16 + c:] == 0xAA).all()
    gv, wv = got.view(torch.float8_e4m3fn).float(), want.float()
    assert not torch.isnan(gv).any() and float(gv.max()) == 448.0               # saturated, never NaN
    same = (got == want.view(torch.uint8)).float().mean().item()
    assert same > 0.99, same
    assert ((gv - wv).abs() <= 0.126 * wv.abs().clamp_min(2.0 ** -9)).all()    # the others: the neighbouring code


@pytest.mark.gpu
def test_fp8_engine_matches_the_fp8_oracle(lib, synth_ck):
    """Engine precision "fp8" (14 Bottleneck pairs on the fp8 MFMA, everything else bf16) against the oracle emulating the same
    quantisers with the engine's calibrated scales, and against the plain bf16 engine: the fp8 layers add quantisation error of their own
    (3 mantissa bits), which must stay of the order the oracle's own fp8 emulation shows against its bf16 emulation."""
    from aquaculture_amd import engine, tiles
    from oracle import yolov5_oracle as O
    x = tiles.synthetic_batch([0, 5, 19], 256)
    eng = engine.Engine(synth_ck, "fp8", fp8_calibration=x)
    assert len(eng.fp8_scales) == 14 and all(s > 0 for s in eng.fp8_scales.values())
    pred = eng.forward_raw(torch.from_numpy(x).cuda()).float().cpu()
    m8 = O.model_from_checkpoint(synth_ck, O.q_bf16, f8_scales=eng.fp8_scales)
    m16 = O.model_from_checkpoint(synth_ck, O.q_bf16)
    ref8, ref16 = m8.forward(O.preprocess(x)), m16.forward(O.preprocess(x))
    bf = engine.Engine(synth_ck, "bf16")
    p16 = bf.forward_raw(torch.from_numpy(x).cuda()).float().cpu()
    d_eng = (pred[..., 4] - ref8[..., 4]).abs()               # engine fp8 vs oracle fp8: accumulation-order noise only
    d_q = (ref8[..., 4] - ref16[..., 4]).abs()                # what the quantisation itself does (oracle vs oracle)
    d_bf = (p16[..., 4] - ref16[..., 4]).abs()                # the bf16 engine's own distance from its oracle
    print(f"objectness |d|: fp8 engine vs fp8 oracle mean {d_eng.mean():.2e} max {d_eng.max():.2e}; fp8 vs bf16 oracle mean {d_q.mean():.2e} "
          f"max {d_q.max():.2e}; bf16 engine vs bf16 oracle mean {d_bf.mean():.2e} max {d_bf.max():.2e}")
    # The tail is bounded on the 99.9th percentile and against the quantisation's own effect, not on a ratio of two maxima: each maximum is one
    # logit of 12 k and moves by a third between kernel builds -- round 4's assembly Bottleneck made BOTH engines closer to their oracles
    # (max 0.163 -> 0.137 and 0.056 -> 0.042) and failed "max <= 3 x max" for it.
    q999 = lambda t: float(torch.quantile(t.flatten(), 0.999))
    assert float(d_eng.mean()) <= 3.0 * float(d_bf.mean()) + 1e-4 and q999(d_eng) <= 3.0 * q999(d_bf) + 1e-3 and float(d_eng.max()) <= float(d_q.max())
    assert float((pred[..., 4] - p16[..., 4]).abs().mean()) > 0.0                  # the fp8 layers did run
    dets, counts = eng.infer(torch.from_numpy(x).cuda())
    assert int(counts.sum()) > 0


FP8_BOUNDS = {"dconf_mean": 0.0372, "dbox_mean_px": 4.34, "count_diff_frac": 0.0586}   # 1.5 x measured (profiles/r03_fp8_accuracy.json): 0.0248, 2.89 px, 218 of 5581 boxes


@pytest.mark.gpu
def test_fp8_accuracy_report_on_the_golden_tiles(lib, synth_ck):
    """fp8 engine (14 Bottleneck pairs on the fp8 MFMA, scales calibrated on tiles 16-23 -- NOT the tiles it is scored on) vs the fp32
    oracle and beside the bf16 engine, 16 golden tiles at 640 px; written to gpurun_out/r03_fp8_accuracy.json (copied to profiles/).  The
    seeded synthetic head amplifies feature noise ~60x: a worst case for any reduced-precision mode."""
    import json
    import os
    from aquaculture_amd import engine, tiles
    from oracle import yolov5_oracle as O
    x = tiles.synthetic_batch(range(16), 640)
    xt = torch.from_numpy(x).cuda()
    res, preds = {}, {}
    for mode in ("fp8", "bf16"):
        eng = engine.Engine(synth_ck, mode, fp8_calibration=tiles.synthetic_batch(range(16, 24), 640) if mode == "fp8" else None)
        preds[mode] = eng.forward_raw(xt).cpu()
        _, c = eng.infer(xt)
        res[mode + "_counts"] = c.cpu().tolist()
        if mode == "fp8":
            res["act_scales"] = eng.fp8_scales
        eng.close()
    m = O.model_from_checkpoint(synth_ck)
    ref = torch.cat([m.forward(O.preprocess(x[i:i + 1])) for i in range(16)], 0)
    ref_counts = [r.shape[0] for r in O.non_max_suppression(ref.numpy())]
    res["fp32_oracle_counts"] = ref_counts
    for mode in ("fp8", "bf16"):
        dc = (preds[mode][..., 4:] - ref[..., 4:]).abs().flatten()
        db = (preds[mode][..., :4] - ref[..., :4]).abs().flatten()
        res[mode + " vs fp32 oracle"] = {"dconf_mean": float(dc.mean()), "dconf_p999": float(dc.kthvalue(int(0.999 * dc.numel()))[0]), "dconf_max": float(dc.max()),
                                         "dbox_mean_px": float(db.mean()), "dbox_p999_px": float(db.kthvalue(int(0.999 * db.numel()))[0]),
                                         "count_diff_max": max(abs(p - q) for p, q in zip(res[mode + "_counts"], ref_counts)),
                                         "count_diff_sum": sum(abs(p - q) for p, q in zip(res[mode + "_counts"], ref_counts)), "boxes_total": sum(ref_counts)}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r03_fp8_accuracy.json"), "w") as f:
            json.dump(res, f, indent=1)
    a8 = res["fp8 vs fp32 oracle"]
    print(json.dumps({k: v for k, v in res.items() if "vs" in k}, indent=1))
    assert a8["dconf_mean"] <= FP8_BOUNDS["dconf_mean"] and a8["dbox_mean_px"] <= FP8_BOUNDS["dbox_mean_px"], a8
    assert a8["count_diff_sum"] <= FP8_BOUNDS["count_diff_frac"] * a8["boxes_total"], a8
