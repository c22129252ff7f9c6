"""Parity of the implicit-GEMM conv kernel (aq_conv2d through the C ABI) against torch CPU conv2d.

fp32 mode: exact-fp32 MFMA vs F.conv2d fp32 -> tolerance 2e-5 relative to the output scale (summation order only).
bf16 mode: the oracle is F.conv2d on bf16-ROUNDED inputs/weights in fp32 (the kernel's own semantics);
           outputs may differ by one bf16 ulp where the fp32 sums straddle a rounding boundary.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nhwc, w, b, stride, pad, act, res, quant):
    x = x_nhwc.permute(0, 3, 1, 2).float()
    if quant:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    y = F.conv2d(x, w, b, stride=stride, padding=pad)
    if act:
        y = F.silu(y)
    if res is not None:
        y = y + res.permute(0, 3, 1, 2).float()
    return y.permute(0, 2, 3, 1).contiguous()


CASES = [
    # B, H, W, cin, cout, k, stride, act, residual
    (2, 20, 20, 64, 64, 1, 1, True, False),
    (1, 16, 24, 48, 96, 3, 2, True, False),
    (2, 20, 20, 192, 192, 3, 1, True, True),
    (1, 40, 40, 96, 48, 1, 1, True, False),
    (3, 8, 8, 384, 32, 1, 1, False, False),
    (1, 32, 32, 16, 48, 3, 1, True, False),
    (2, 12, 20, 768, 384, 1, 1, True, False),
    (1, 9, 7, 32, 64, 3, 1, True, True),      # ragged spatial size: tile tail + every border case
    (3, 40, 40, 192, 192, 3, 1, True, True),  # a real layer shape (model.6 bottleneck), several tiles, batch seams inside tiles
    (2, 20, 20, 48, 48, 3, 1, True, False),   # 48 channels: partial K chunk (6 of 8 groups)
    (1, 24, 16, 96, 96, 3, 1, True, True),    # 96 channels: one full + one half chunk
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_conv_matches_reference(lib, case, precision):
    from aquaculture_amd import engine
    B, H, W, cin, cout, k, stride, act, use_res = case
    g = torch.Generator().manual_seed(1234 + cin + cout)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g) if use_res else None
    dt = torch.float32 if precision == "fp32" else torch.bfloat16
    xd = x.to(dt).cuda()
    rd = res.to(dt).cuda() if res is not None else None
    ref = _ref(xd.cpu(), w, b, stride, pad, act, rd.cpu() if rd is not None else None, precision == "bf16")
    ran = 0
    for cfg in range(lib.aq_conv_num_configs()):
        try:
            out = engine.conv2d_nhwc(xd, w, b, stride=stride, act=act, residual=rd, precision=precision, cfg=cfg).cpu().float()
        except RuntimeError as err:          # tile shape not applicable to this layer (e.g. halo kernel on a 1x1 conv)
            assert "halo conv" in str(err), err
            continue
        ran += 1
        assert out.shape == ref.shape
        if precision == "fp32":
            torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)
        else:
            # within one bf16 ulp (2^-8 relative) of the exactly-rounded reference
            torch.testing.assert_close(out, ref.bfloat16().float(), rtol=2 ** -7, atol=1e-3)
        # the same tile shape launched as one workgroup per tile: scheduling only, so the bytes are the same
        out1 = engine.conv2d_nhwc(xd, w, b, stride=stride, act=act, residual=rd, precision=precision,
                                  cfg=cfg | engine.CONV_CFG_ONE_TILE_PER_WG).cpu().float()
        assert torch.equal(out1, out), cfg
    assert ran >= 10


@pytest.mark.parametrize("case", [c for c in CASES if c[3] % 8 == 0])
def test_conv_split_mode_is_fp32_grade(lib, case):
    """AQ_F16X3 (fp32 activations, every product as three fp16 MFMAs on hi / lo halves, fp32 accumulate) against an fp64 reference:
    the error must be of fp32's own order -- within 4x the exact-fp32 kernel's error on the same layer, and inside fp32 mode's
    tolerance -- on every tile shape; weights spanning four orders of magnitude across output channels exercise the per-channel scale."""
    from aquaculture_amd import engine
    B, H, W, cin, cout, k, stride, act, use_res = case
    g = torch.Generator().manual_seed(4321 + cin + cout)
    x = torch.randn(B, H, W, cin, generator=g) * 2.0
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    w = w * (10.0 ** torch.linspace(-3, 1, cout)).view(-1, 1, 1, 1)            # rows from 1e-3 to 10 times the usual scale
    b = torch.randn(cout, generator=g) * 0.1
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g) if use_res else None
    xd = x.cuda()
    rd = res.cuda() if res is not None else None
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=pad)
    if act:
        y = F.silu(y)
    if res is not None:
        y = y + res.double().permute(0, 3, 1, 2)
    ref = y.permute(0, 2, 3, 1)
    scale = ref.abs().amax(dim=(0, 1, 2), keepdim=True).clamp_min(1e-30)           # per output channel: the rows differ by 1e4
    ran = 0
    for cfg in range(lib.aq_conv_num_configs()):
        try:
            out = engine.conv2d_nhwc(xd, w, b, stride=stride, act=act, residual=rd, precision="f16x3", cfg=cfg).cpu().double()
        except RuntimeError as err:
            assert "halo conv" in str(err), err
            continue
        ran += 1
        e3 = ((out - ref).abs() / scale).max().item()
        if ran == 1:
            o32 = engine.conv2d_nhwc(xd, w, b, stride=stride, act=act, residual=rd, precision="fp32", cfg=cfg).cpu().double()
            e32 = ((o32 - ref).abs() / scale).max().item()
        assert e3 <= max(4.0 * e32, 2e-6), (cfg, e3, e32)
        torch.testing.assert_close(out.float(), ref.float(), rtol=2e-5, atol=2e-5 * float(scale.max()))
    assert ran >= 10


def test_conv_split_mode_refuses_what_it_cannot_split(lib):
    from aquaculture_amd import engine
    x = torch.randn(1, 8, 8, 12).cuda()
    with pytest.raises(RuntimeError, match="multiple of 8"):
        engine.conv2d_nhwc(x, torch.randn(16, 12, 3, 3), torch.zeros(16), precision="f16x3")


def test_conv_f32_out_head(lib):
    """Detect-head form: bf16 inputs, fp32 output, no activation, cout padded to 32."""
    from aquaculture_amd import engine
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 20, 20, 192, generator=g).bfloat16()
    w = torch.zeros(32, 192, 1, 1)
    w[:30] = torch.randn(30, 192, 1, 1, generator=g) * 0.1
    b = torch.zeros(32)
    b[:30] = torch.randn(30, generator=g)
    out = engine.conv2d_nhwc(x.cuda(), w, b, act=False, precision="bf16", out_f32=True).cpu()
    ref = _ref(x, w, b, 1, 0, False, None, True)
    assert out.dtype == torch.float32
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 128, 96), (3, 32, 160), (1, 640, 640)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_stem_matches_reference(lib, shape, precision):
    """aq_stem_conv: uint8 tiles -> x/255 -> Conv(3, 48, 6, 2, 2) + bias + SiLU, against F.conv2d on the same values
    (bf16 mode: inputs and weights rounded to bf16 first, fp32 accumulate).  Widths 96/160 exercise partial 64-column tiles."""
    from aquaculture_amd import engine
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    w = torch.randn(48, 3, 6, 6, generator=g) * 0.2
    b = torch.randn(48, generator=g) * 0.1
    xin = (x.permute(0, 3, 1, 2).float() / 255)
    wq = w
    if precision == "bf16":
        xin, wq = xin.bfloat16().float(), w.bfloat16().float()
    ref = F.silu(F.conv2d(xin, wq, b, stride=2, padding=2)).permute(0, 2, 3, 1).contiguous()
    out = engine.stem_conv_nhwc(x.cuda(), w, b, precision=precision).cpu().float()
    assert out.shape == ref.shape == (B, H // 2, W // 2, 48)
    if precision == "fp32":
        torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)
    else:
        torch.testing.assert_close(out, ref.bfloat16().float(), rtol=2 ** -7, atol=1e-3)


def test_fused_and_two_kernel_stems_agree(lib, synth_ck):
    """The fused stem and the preprocess + space-to-depth conv path are two routes to the same layer."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([0, 19], 128)).cuda()
    outs = []
    for fused in (True, False):
        eng = engine.Engine(synth_ck, "fp32", fused_stem=fused)
        assert (eng.plan.ops[0].kind == 6) == fused
        eng.forward_raw(x)
        torch.cuda.synchronize()
        outs.append(eng.tensor_by_name("out0", 2).float().cpu().clone())
    torch.testing.assert_close(outs[0], outs[1], rtol=2e-5, atol=2e-5)


def _bottleneck_reference(x, w1, b1, w2, b2, shortcut=True):
    """bf16 two-kernel semantics: weights and t rounded to bf16, fp32 accumulation, bf16 output."""
    xf = x.float().permute(0, 3, 1, 2)
    t = F.silu(F.conv2d(xf, w1.bfloat16().float(), b1)).bfloat16().float()
    y = F.silu(F.conv2d(t, w2.bfloat16().float(), b2, padding=1))
    if shortcut:
        y = y + xf
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("c", [16, 32, 48, 64, 96])
@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 40, 56), (3, 160, 160), (1, 7, 5)])
def test_fused_bottleneck_matches_reference(lib, c, shape):
    """aq_bottleneck: x + SiLU(cv2_3x3(SiLU(cv1_1x1(x)))) in one launch; input and output are channel slices of wider
    NHWC tensors (as in the C3 concat buffer).  Shapes cover whole tiles, ragged edges, many tiles and a sub-tile image."""
    from aquaculture_amd import engine
    B, H, W = shape
    g = torch.Generator().manual_seed(c * 1000 + H)
    xw = (torch.randn(B, H, W, 2 * c, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:8 + c]
    w1 = torch.randn(c, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    w2 = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
    b1, b2 = torch.randn(c, generator=g) * 0.2, torch.randn(c, generator=g) * 0.2
    outw = torch.full((B, H, W, c + 16), 7.0, dtype=torch.bfloat16, device="cuda")
    for shortcut in (True, False):
        engine.bottleneck_nhwc(x, w1, b1, w2, b2, shortcut, out=outw[..., 16:])
        ref = _bottleneck_reference(x.cpu(), w1, b1, w2, b2, shortcut)
        got = outw[..., 16:].float().cpu()
        assert (outw[..., :16] == 7.0).all(), "wrote outside its channel slice"
        err = (got - ref).abs()
        # one bf16 ulp of the result plus the effect of rare 1-ulp flips of t (accumulation order differs from torch's)
        assert (err <= 2 ** -7 * ref.abs() + 2e-2).all(), float(err.max())
        assert float(err.mean()) < 3e-3


def test_fused_bottleneck_refuses_in_place(lib):
    from aquaculture_amd import engine
    x = torch.zeros(1, 16, 32, 48, dtype=torch.bfloat16, device="cuda")
    w1, w2, b = torch.zeros(48, 48, 1, 1), torch.zeros(48, 48, 3, 3), torch.zeros(48)
    with pytest.raises(RuntimeError, match="overlaps"):
        engine.bottleneck_nhwc(x, w1, b, w2, b, True, out=x)


def test_fused_and_two_kernel_bottlenecks_agree(lib, synth_ck):
    """bf16 engines with and without the fused Bottleneck op: same C3 output up to bf16 rounding noise."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([0, 19], 128)).cuda()
    outs = []
    for fused in (True, False):
        eng = engine.Engine(synth_ck, "bf16", fused_bottleneck=fused)
        assert any(o.kind == 7 for o in eng.plan.ops) == fused
        eng.forward_raw(x)
        torch.cuda.synchronize()
        outs.append(eng.tensor_by_name("out2", 2).float().cpu().clone())
    err = (outs[0] - outs[1]).abs()
    assert float(err.mean()) < 5e-3 and float(err.max()) < 0.25, (float(err.mean()), float(err.max()))


@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 36, 44), (2, 320, 320), (1, 6, 10)])
def test_fused_downblock_matches_reference(lib, shape):
    """aq_downblock: SiLU(conv1x1(SiLU(conv3x3/s2(x)))), 48 -> 96 -> 96, in one launch; the input is a channel slice of a wider
    tensor, the output a slice too.  Shapes: one exact tile, ragged tiles in both directions, many tiles, a sub-tile image."""
    from aquaculture_amd import engine
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    xw = (torch.randn(B, H, W, 64, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:56]
    wa = torch.randn(96, 48, 3, 3, generator=g) * (2.0 / (9 * 48)) ** 0.5
    wb = torch.randn(96, 96, 1, 1, generator=g) * (2.0 / 96) ** 0.5
    ba, bb = torch.randn(96, generator=g) * 0.2, torch.randn(96, generator=g) * 0.2
    outw = torch.full((B, H // 2, W // 2, 112), 7.0, dtype=torch.bfloat16, device="cuda")
    engine.downblock_nhwc(x, wa, ba, wb, bb, out=outw[..., 16:])
    xf = x.float().cpu().permute(0, 3, 1, 2)
    t = F.silu(F.conv2d(xf, wa.bfloat16().float(), ba, stride=2, padding=1)).bfloat16().float()
    ref = F.silu(F.conv2d(t, wb.bfloat16().float(), bb)).permute(0, 2, 3, 1).contiguous()
    got = outw[..., 16:].float().cpu()
    assert (outw[..., :16] == 7.0).all(), "wrote outside its channel slice"
    err = (got - ref).abs()
    assert (err <= 2 ** -7 * ref.abs() + 2e-2).all(), float(err.max())
    assert float(err.mean()) < 3e-3


def test_fused_downblock_in_engine(lib, synth_ck):
    """bf16 engines with the fused ops (default) and without them agree on the first C3's output up to bf16 rounding noise."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([3, 11], 128)).cuda()
    outs = []
    for fused in (True, False):
        eng = engine.Engine(synth_ck, "bf16", fused_bottleneck=fused)
        assert any(o.kind == 8 for o in eng.plan.ops) == fused
        eng.forward_raw(x)
        torch.cuda.synchronize()
        outs.append(eng.tensor_by_name("out2", 2).float().cpu().clone())
    err = (outs[0] - outs[1]).abs()
    assert float(err.mean()) < 5e-3 and float(err.max()) < 0.25, (float(err.mean()), float(err.max()))


@pytest.mark.parametrize("cin,cout", [(96, 96), (192, 192), (384, 192), (384, 384)])
@pytest.mark.parametrize("npix_shape", [(1, 16, 16), (2, 37, 41), (3, 160, 160)])
def test_direct_conv1x1_matches_reference(lib, cin, cout, npix_shape):
    """aq_conv1x1_direct vs F.conv2d on bf16-rounded operands; input and output are channel slices of wider tensors; pixel counts
    cover one partial tile, ragged multi-tile and many tiles."""
    from aquaculture_amd import engine
    B, H, W = npix_shape
    g = torch.Generator().manual_seed(cin + H)
    xw = (torch.randn(B, H, W, cin + 16, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:8 + cin]
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.2
    outw = torch.full((B, H, W, cout + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    for act in (True, False):
        engine.conv1x1_direct_nhwc(x, w, b, act, out=outw[..., 8:])
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float(), b)
        ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
        got = outw[..., 8:].float().cpu()
        assert (outw[..., :8] == 7.0).all()
        torch.testing.assert_close(got, ref.bfloat16().float(), rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("cin,cout", [(768, 768), (768, 384), (1536, 768), (96, 384), (384, 1536)])
@pytest.mark.parametrize("npix_shape", [(1, 5, 7), (1, 13, 16), (2, 37, 41), (16, 40, 40)])
@pytest.mark.parametrize("family", [13, 7])
def test_asm_conv1x1_matches_reference(lib, monkeypatch, cin, cout, npix_shape, family):
    """aq_conv1x1_asm (generated assembly; both tile heights: 208 and 112 pixels x 384 channels) vs F.conv2d on bf16-rounded operands: K = 96
    (one chunk) .. 1536, one / two / four channel tiles; input and output are channel slices of wider tensors; pixel counts: less than one
    tile, exactly one 208-pixel tile, ragged multi-tile, more tiles than CUs (several tiles per workgroup: the loads run ahead across tile
    boundaries)."""
    from aquaculture_amd import engine
    monkeypatch.setenv("AQ_C1_ASM_NB", str(family))
    B, H, W = npix_shape
    g = torch.Generator().manual_seed(cin + H)
    xw = (torch.randn(B, H, W, cin + 16, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:8 + cin]
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.2
    outw = torch.full((B, H, W, cout + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    engine.conv1x1_asm_nhwc(x, w, b, out=outw[..., 4:4 + cout])
    ref = F.silu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float(), b)).permute(0, 2, 3, 1)
    got = outw[..., 4:4 + cout].float().cpu()
    assert (outw[..., :4] == 7.0).all() and (outw[..., 4 + cout:] == 7.0).all()
    torch.testing.assert_close(got, ref.bfloat16().float(), rtol=2 ** -7, atol=2e-3)


def test_direct_conv1x1_in_engine(lib, synth_ck):
    """Forcing the direct kernel on every 1x1 layer it supports leaves the head outputs within bf16 noise of the default engine."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([5], 128)).cuda()
    ref_eng = engine.Engine(synth_ck, "bf16")
    ref = ref_eng.forward_raw(x).float().cpu()
    eng = engine.Engine(synth_ck, "bf16")
    forced = 0
    for i, o in enumerate(eng.plan.ops):
        if o.kind == 1 and o.k == 1 and o.res is None and eng.lib.aq_conv1x1_direct_supported(o.src.channels, o.dst.channels) and o.level < 0:
            eng.set_conv_config(i, engine.CONV_CFG_DIRECT1X1)
            forced += 1
        if o.kind == 1 and o.k == 3 and o.stride == 2 and eng.lib.aq_conv3x3s2_direct_supported(o.src.channels, o.dst.channels):
            eng.set_conv_config(i, engine.CONV_CFG_DIRECT3X3S2)
            forced += 1
    assert forced >= 5
    got = eng.forward_raw(x).float().cpu()
    err = (got - ref).abs()
    assert float(err[..., 4].max()) < 0.1 and float(err[..., :4].mean()) < 1.0, (float(err[..., 4].max()), float(err[..., :4].mean()))


@pytest.mark.parametrize("shape", [(2, 8, 32), (1, 36, 44), (2, 160, 160), (1, 6, 10)])
def test_direct_conv3x3s2_matches_reference(lib, shape):
    """aq_conv3x3s2_direct (96 -> 192, the plain form of the down-block kernel) vs F.conv2d on bf16-rounded operands; slices in and out;
    one exact tile, ragged tiles, many tiles, a sub-tile image; with and without SiLU."""
    from aquaculture_amd import engine
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 5 + W)
    xw = (torch.randn(B, H, W, 112, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:104]
    w = torch.randn(192, 96, 3, 3, generator=g) * (2.0 / (9 * 96)) ** 0.5
    b = torch.randn(192, generator=g) * 0.2
    outw = torch.full((B, H // 2, W // 2, 200), 7.0, dtype=torch.bfloat16, device="cuda")
    for act in (True, False):
        engine.conv3x3s2_direct_nhwc(x, w, b, act, out=outw[..., 8:])
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float(), b, stride=2, padding=1)
        ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
        got = outw[..., 8:].float().cpu()
        assert (outw[..., :8] == 7.0).all()
        torch.testing.assert_close(got, ref.bfloat16().float(), rtol=2 ** -7, atol=4e-3)


PL_S2_CASES = [
    # B, H, W (input), Cin, Cout, act
    (3, 80, 80, 192, 384, True),      # yolov5m model.5: several tiles per image, batch seams inside tiles, two M tiles, six 32-channel chunks
    (2, 40, 40, 384, 768, True),      # model.7: four M tiles, twelve chunks
    (5, 80, 80, 192, 192, True),      # model.18
    (3, 40, 40, 384, 384, False),     # model.21, no activation
    (1, 18, 14, 64, 192, True),       # sub-tile ragged image: every border case inside one tile; two chunks (the ring wraps at once)
    (2, 26, 48, 96, 192, True),       # three chunks, odd output sizes (13 x 24)
    (40, 80, 80, 192, 192, True),     # more tiles than CUs: persistent workgroups walk several tiles (prefetch across the tile seam)
    (70, 40, 40, 128, 384, True),     # the same with two M tiles
    (2, 4, 4, 64, 192, False),        # 2 x 2 outputs per image: every tap of every pixel touches padding or another parity plane's border
]


@pytest.mark.parametrize("case", PL_S2_CASES)
def test_planar_conv3x3_s2_matches_reference(lib, case):
    """aq_conv3x3_pl_s2 (generated assembly, family s2nb13: parity planes of the input in padded output coordinates) vs F.conv2d with
    stride 2 on bf16-rounded operands; input and output are channel slices of wider tensors."""
    from aquaculture_amd import engine
    B, H, W, cin, c, act = case
    assert engine.load_library().aq_conv3x3_pl_s2_supported(cin, c, B, H, W)
    g = torch.Generator().manual_seed(c * 5 + H * 3 + cin)
    xw = (torch.randn(B, H, W, cin + 16, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:8 + cin]
    w = torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(c, generator=g) * 0.2
    outw = torch.full((B, H // 2, W // 2, c + 24), 7.0, dtype=torch.bfloat16, device="cuda")
    out = outw[..., 16:16 + c]
    engine.conv3x3_pl_s2_nhwc(x, w, b, act, out=out)
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float(), b, stride=2, padding=1)
    ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
    got = out.float().cpu()
    assert (outw[..., :16] == 7.0).all() and (outw[..., 16 + c:] == 7.0).all(), "wrote outside its channel slice"
    torch.testing.assert_close(got, ref.bfloat16().float(), rtol=2 ** -7, atol=4e-3)


def test_planar_conv3x3_s2_rejects_what_it_cannot_tile(lib):
    """Odd image sizes, channel counts off the 32 / 192 grids and output rows too long for the tile's region are refused (the engine
    then keeps the implicit-GEMM kernels), not mis-computed."""
    from aquaculture_amd import engine
    L = engine.load_library()
    assert L.aq_conv3x3_pl_s2_supported(192, 384, 64, 80, 80) and L.aq_conv3x3_pl_s2_supported(384, 768, 64, 40, 40)
    assert not L.aq_conv3x3_pl_s2_supported(192, 384, 2, 81, 80) and not L.aq_conv3x3_pl_s2_supported(192, 384, 2, 80, 82 + 1)
    assert not L.aq_conv3x3_pl_s2_supported(48, 192, 2, 80, 80) and not L.aq_conv3x3_pl_s2_supported(192, 200, 2, 80, 80)
    assert not L.aq_conv3x3_pl_s2_supported(96, 192, 64, 160, 160)        # model.3: an 80-pixel output row does not fit the 304-row region
    x = torch.zeros(2, 160, 160, 96, dtype=torch.bfloat16, device="cuda")       # (a single image has no seam and would still fit)
    with pytest.raises(RuntimeError, match="unsupported"):
        engine.conv3x3_pl_s2_nhwc(x, torch.zeros(192, 96, 3, 3), torch.zeros(192))


PL_CASES = [
    # B, H, W, Cin, Cout, residual mode (None / "sep" / "inplace"), act
    (3, 40, 40, 192, 192, "inplace", True),    # yolov5m model.6 Bottleneck.cv2: several tiles, batch seams inside tiles, in-place shortcut
    (2, 20, 20, 384, 384, "sep", True),        # model.8: two M tiles, six chunks
    (1, 9, 7, 192, 192, "sep", True),          # sub-tile ragged image: every border case inside one tile
    (2, 13, 24, 128, 192, None, True),         # two chunks only (the ring wraps into the next tile at once)
    (5, 20, 20, 192, 192, None, False),        # no activation, no shortcut
    (2, 36, 44, 256, 384, "inplace", True),    # four chunks, two M tiles, odd sizes
    (40, 40, 40, 192, 192, "inplace", True),   # more tiles than CUs: persistent workgroups walk several tiles (prefetch across the tile seam)
    (70, 20, 20, 384, 384, "sep", True),       # the same with two M tiles
    (3, 10, 10, 768, 768, "sep", True),        # model.9-sized: four M tiles, twelve chunks
    (4, 12, 12, 128, 576, None, True),         # three M tiles (not a power of two: the HIP-source kernel for every NB)
]


@pytest.mark.parametrize("case", PL_CASES)
@pytest.mark.parametrize("nb", ["13asm", "13slot-asm", "13pm-asm", "7asm", "8asm", 13, 10, 7])
def test_planar_conv3x3_matches_reference(lib, case, nb, monkeypatch):
    """aq_conv3x3_pl vs F.conv2d on bf16-rounded operands; input, output and shortcut are channel slices of wider tensors; every
    pixel-block count of the kernel, tiles that end inside images, at image seams and past the end of the batch.  "13asm", "7asm", "8asm"
    are the families of the hand-scheduled assembly build (gen_conv3x3_pl_asm.py; 7 and 8: two workgroups per CU, two ring buffers,
    residual fetched in the epilogue), the plain numbers the HIP-source kernel."""
    from aquaculture_amd import engine
    B, H, W, cin, c, resmode, act = case
    monkeypatch.setenv("AQ_PL_ASM", "1" if isinstance(nb, str) else "0")
    # "13asm": what the engine runs -- the pixel-major builds for 20- and 40-wide images (pm13w20 / pm13w40), the slot-major build (nb13)
    # elsewhere; "13slot-asm": nb13 at every width; "13pm-asm": the any-width pixel-major build (pm13) at every width
    if nb in ("13slot-asm", "13pm-asm"):
        monkeypatch.setenv("AQ_PL_PM", "0" if nb == "13slot-asm" else "2")
        nb = "13asm"
    if isinstance(nb, str) and not lib.aq_conv3x3_pl_asm_family(int(nb[:-3])):
        pytest.skip("experimental assembly family: built only with AQ_GEN_EXPERIMENTAL=1")
    nb = int(nb[:-3]) if isinstance(nb, str) else nb
    monkeypatch.setenv("AQ_PL_NB", str(nb))
    g = torch.Generator().manual_seed(c * 7 + H * 3 + nb)
    xw = (torch.randn(B, H, W, cin + 16, generator=g) * 0.8).bfloat16().cuda()
    x = xw[..., 8:8 + cin]
    w = torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(c, generator=g) * 0.2
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
    if nb == 8 and (W > 40 or (c // 192) & (c // 192 - 1)):
        with pytest.raises(RuntimeError, match="no tile of this kernel fits"):     # NB = 8 exists as assembly only: 256 region rows, 2^k M tiles
            engine.conv3x3_pl_nhwc(x, w, b, act, residual=res, out=out)
        return
    engine.conv3x3_pl_nhwc(x, w, b, act, residual=res, out=out)
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float(), b, padding=1)
    ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
    if res_host is not None:
        ref = ref + res_host
    got = out.float().cpu()
    assert (outw[..., :16] == 7.0).all() and (outw[..., 16 + c:] == 7.0).all(), "wrote outside its channel slice"
    torch.testing.assert_close(got, ref.bfloat16().float(), rtol=2 ** -7, atol=4e-3)


@pytest.mark.parametrize("case", [(5, 40, 40, 192, 192, True), (9, 20, 20, 384, 384, True), (3, 40, 40, 256, 192, False), (2, 13, 24, 128, 192, True)])
def test_planar_builds_are_bit_identical(lib, case, monkeypatch):
    """The pixel-major builds (pm13w40 / pm13w20; pm13 at other widths under AQ_PL_PM=2), the slot-major assembly build (nb13) and the
    HIP-source kernel are the same algorithm on the same tiles with the same MFMA order: their outputs are equal BIT FOR BIT, so which one
    the launcher picks for an image width never changes a label byte."""
    from aquaculture_amd import engine
    B, H, W, cin, c, with_res = case
    g = torch.Generator().manual_seed(H * 31 + cin)
    x = (torch.randn(B, H, W, cin, generator=g) * 0.8).bfloat16().cuda()
    w = torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(c, generator=g) * 0.2
    res = torch.randn(B, H, W, c, generator=g).bfloat16().cuda() if with_res else None
    outs = {}
    for name, env in (("default", {}), ("slot-major", {"AQ_PL_PM": "0"}), ("pixel-major any width", {"AQ_PL_PM": "2"}), ("HIP source", {"AQ_PL_ASM": "0", "AQ_PL_NB": "13"})):
        for k in ("AQ_PL_PM", "AQ_PL_ASM", "AQ_PL_NB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = torch.zeros(B, H, W, c, dtype=torch.bfloat16, device="cuda")
        engine.conv3x3_pl_nhwc(x, w, b, True, residual=res, out=out)
        outs[name] = out.view(torch.int16).cpu()
    for name, o in outs.items():
        assert torch.equal(o, outs["default"]), f"{name} differs from the default build in {int((o != outs['default']).sum())} values"


def test_planar_conv3x3_in_engine(lib, synth_ck):
    """Forcing the planar kernel on every 3x3/s1 layer it supports leaves the head outputs within bf16 noise of the default engine."""
    from aquaculture_amd import engine, tiles
    x = torch.from_numpy(tiles.synthetic_batch([5, 7], 128)).cuda()
    ref = engine.Engine(synth_ck, "bf16").forward_raw(x).float().cpu()
    eng = engine.Engine(synth_ck, "bf16")
    forced = 0
    for i, o in enumerate(eng.plan.ops):
        if o.kind == 1 and o.k == 3 and o.stride == 1 and eng.lib.aq_conv3x3_pl_supported(o.src.channels, o.dst.channels):
            eng.set_conv_config(i, engine.CONV_CFG_PL3X3)
            forced += 1
    assert forced >= 14
    got = eng.forward_raw(x).float().cpu()
    err = (got - ref).abs()
    assert float(err[..., 4].max()) < 0.1 and float(err[..., :4].mean()) < 1.0, (float(err[..., 4].max()), float(err[..., :4].mean()))
