"""fp8w mode, host side (CPU): the numpy e4m3fn quantiser of the product (aquaculture_amd/quant.py) against torch's own float8_e4m3fn
conversion (the restatement the oracle uses), the exactness claim the mode rests on (every dequantised weight is a bf16 value), and
that engine-side packing and the oracle quantise a whole checkpoint to the same numbers."""
import numpy as np
import torch

from aquaculture_amd import checkpoint, quant, spec
from oracle import yolov5_oracle as O


def test_e4m3_round_and_codes_match_torch_float8():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 1, 200_000) * rng.choice([1e-4, 1e-3, 0.02, 1, 30, 200], 200_000),
                        [0, -0.0, 448, -448, 2 ** -9, 2 ** -10, 1.5 * 2 ** -9, 2.5 * 2 ** -9, 2 ** -6, 2 ** -6 * 1.0625, 240, 232, 248, 440, 447.9,
                         0.0625 * 1.0625, 0.0625 * 1.1875, 17, 18, 19, 21, 23, 25, 27]]).astype(np.float32)
    # ties: exact midpoints between neighbouring e4m3 values in several binades
    mids = np.array([(8 + m + 0.5) * 2.0 ** (e - 3) for e in range(-6, 9) for m in range(8)], np.float32)
    x = np.clip(np.concatenate([x, mids, -mids]), -448, 448)
    mine = quant.e4m3_round(x)
    t8 = torch.from_numpy(x).to(torch.float8_e4m3fn)
    assert np.array_equal(mine, t8.float().numpy())
    assert np.array_equal(quant.e4m3_encode(mine), t8.view(torch.uint8).numpy())
    assert quant.is_bf16_exact(mine)


def test_rows_get_power_of_two_scales_and_stay_bf16_exact():
    rng = np.random.default_rng(2)
    w = (rng.normal(0, 1, (48, 3, 3, 64)) * rng.lognormal(0, 2, (48, 1, 1, 1))).astype(np.float32)
    w[5] = 0
    deq, codes, e = quant.quantize_rows(w)
    assert deq.shape == w.shape and codes.dtype == np.uint8 and quant.is_bf16_exact(deq)
    s = np.ldexp(1.0, e)
    amax = np.abs(w.reshape(48, -1)).max(1)
    assert np.all(amax / s <= 448) and np.all((amax / s > 224) | (amax == 0))          # the smallest power of two that fits
    assert np.all(deq[5] == 0)
    rel = np.abs(deq - w).reshape(48, -1).max(1) / np.maximum(amax, 1e-30)
    assert rel.max() <= 16.0 / 224.0 + 1e-6                # half a step (16) of the top binade over the smallest scaled maximum (224)
    assert np.array_equal(deq, O.wq_fp8_e4m3(torch.from_numpy(w)).numpy())


def test_engine_packing_and_oracle_quantise_the_checkpoint_alike(synth_ck):
    plan = spec.build_plan("yolov5m", 5, fused_bottleneck=True)
    packed = checkpoint.pack_plan_weights(synth_ck, plan, "fp8")
    native = checkpoint.pack_plan_weights(synth_ck, plan)
    m = O.model_from_checkpoint(synth_ck, O.q_bf16, O.wq_fp8_e4m3)
    seen = 0
    for op, pk, nat in zip(plan.conv_ops(), packed, native):
        if op.name.startswith("model.24"):
            assert np.array_equal(pk.weight, nat.weight)            # the Detect head keeps its weights
            continue
        assert quant.is_bf16_exact(pk.weight), op.name
        assert np.array_equal(pk.bias, nat.bias)
        if op.kind == spec.OP_CONV and len(op.weight_keys) == 1 and not op.meta.get("stem_s2d"):
            w_or, _ = m._wb(op.weight_keys[0])
            assert np.array_equal(pk.weight, w_or.permute(0, 2, 3, 1).numpy()), op.name
            seen += 1
    assert seen >= 30


def test_fp8_weight_stream_image_is_half_the_bf16_one():
    """aq_pack_conv3x3_pl_w8 size query (no GPU): one byte per weight, against two in aq_pack_conv3x3_pl's fragment stream."""
    import ctypes as C
    from aquaculture_amd import engine
    lib = engine.load_library()
    n8, n16 = C.c_size_t(), C.c_size_t()
    w = np.zeros((192, 3, 3, 128), np.float32)
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.aq_pack_conv3x3_pl_w8(wp, None, 128, 192, None, C.byref(n8), None, None) == 0
    assert lib.aq_pack_conv3x3_pl(wp, 128, 192, None, C.byref(n16), None) == 0
    assert n8.value * 2 == n16.value == 192 * 9 * 128 * 2
    assert lib.aq_conv3x3_pl_w8_supported(192, 192, 64, 40, 40) == 1 and lib.aq_conv3x3_pl_w8_supported(192, 576, 64, 40, 40) == 0
