"""Where does the bf16 engine's deviation come from?  (VERDICT round 1, weak point 1b.)

Three implementations of the SAME bf16 pipeline (bf16 weights and activations rounded at the same points, wide accumulation in between):

    ref64   the oracle with fp64 accumulation between two roundings          (oracle/yolov5_oracle.py q_bf16_f64)
    ref32   the oracle with fp32 accumulation in PyTorch's summation order   (q_bf16)
    engine  the HIP engine: fp32 accumulation in MFMA order, four kernel selections (fused ops, two-kernel Bottlenecks, direct
            1x1 / 3x3-s2 kernels forced, planar 3x3 kernel forced)

|ref32 - ref64| per module is the noise that accumulation order alone causes; it grows through the network because every module
re-rounds its input differences to bf16.  If |engine - ref64| tracks that floor module by module, for every kernel selection, no
kernel contributes an error of its own (a wrong rounding point, a low-precision SiLU, a missing k-step would show up as a jump at the
module where that kernel first runs).  The test asserts that, and writes the table (gpurun_out/r02_bf16_deviation.json, copied to
profiles/) together with the detection-level numbers on the 16 golden tiles that tests/test_gpu_engine.py bounds.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAPS = {"out0": "model.0", "out1": "model.1", "out2": "model.2", "out3": "model.3", "out5": "model.5", "out7": "model.7", "out8": "model.8",
        "out9": "model.9", "out13": "model.13", "out17": "model.17", "out20": "model.20", "out23": "model.23"}
ORDER = ["model.0", "model.1", "model.2", "model.3", "model.5", "model.7", "model.8", "model.9", "model.13", "model.17", "model.20", "model.23",
         "model.24.m.0", "model.24.m.1", "model.24.m.2"]


def _variants(ck):
    from aquaculture_amd import engine
    def force(eng, pred):
        n = 0
        for i, o in enumerate(eng.plan.ops):
            c = pred(eng, o)
            if c is not None:
                eng.set_conv_config(i, c)
                n += 1
        return n
    def direct(eng, o):
        if o.kind != 1:
            return None
        if o.k == 1 and o.res is None and o.level < 0 and eng.lib.aq_conv1x1_direct_supported(o.src.channels, o.dst.channels):
            return engine.CONV_CFG_DIRECT1X1
        if o.k == 3 and o.stride == 2 and eng.lib.aq_conv3x3s2_direct_supported(o.src.channels, o.dst.channels):
            return engine.CONV_CFG_DIRECT3X3S2
        return None
    def planar(eng, o):
        ok = o.kind == 1 and o.k == 3 and o.stride == 1 and eng.lib.aq_conv3x3_pl_supported(o.src.channels, o.dst.channels)
        return engine.CONV_CFG_PL3X3 if ok else None
    v = {}
    v["fused ops, heuristic tiles"] = engine.Engine(ck, "bf16")
    v["two-kernel Bottlenecks / down-block"] = engine.Engine(ck, "bf16", fused_bottleneck=False)
    e = engine.Engine(ck, "bf16")
    assert force(e, direct) >= 5
    v["direct 1x1 and 3x3/s2 kernels forced"] = e
    e = engine.Engine(ck, "bf16")
    assert force(e, planar) >= 14
    v["planar 3x3 kernel forced"] = e
    return v


def _stats(x, ref):
    """x, ref: float64 tensors of one module.  Errors in units of the bf16 ulp of the reference element, over elements that are not
    tiny (|ref| >= rms / 16: an ulp of a near-zero element says nothing), and relative to the module's rms."""
    d = (x - ref).abs()
    rms = float(ref.pow(2).mean().sqrt())
    big = ref.abs() >= rms / 16
    ulp = torch.pow(2.0, torch.floor(torch.log2(ref.abs().clamp(min=1e-30))) - 7)
    u = (d / ulp)[big]
    q = torch.quantile(u[:: max(1, u.numel() // 2_000_000)], torch.tensor([0.5, 0.99], dtype=u.dtype))
    return {"rms": rms, "mean_abs_over_rms": float(d.mean()) / rms, "ulp_p50": float(q[0]), "ulp_p99": float(q[1]), "ulp_max": float(u.max()),
            "frac_over_1ulp": float((u > 1).double().mean())}


def test_bf16_deviation_tracks_the_accumulation_order_floor(lib, synth_ck):
    from aquaculture_amd import tiles
    from oracle import yolov5_oracle as O
    x = tiles.synthetic_batch([0, 19], 128)
    B = x.shape[0]
    m64 = O.model_from_checkpoint(synth_ck, O.q_bf16_f64); m64.taps = {}
    m32 = O.model_from_checkpoint(synth_ck, O.q_bf16); m32.taps = {}
    m64.forward(O.preprocess(x).double())
    m32.forward(O.preprocess(x))
    report = {"tiles": "synthetic tiles 0 and 19 at 128 px", "unit": "bf16 ulps of the reference element (|ref| >= rms/16); mean |d| / rms",
              "floor (fp32 vs fp64 accumulation in the oracle)": {k: _stats(m32.taps[k].double(), m64.taps[k]) for k in ORDER}, "engine": {}}
    floor = report["floor (fp32 vs fp64 accumulation in the oracle)"]
    xt = torch.from_numpy(x).cuda()
    for name, eng in _variants(synth_ck).items():
        eng.forward_raw(xt)
        torch.cuda.synchronize()
        got = {key: eng.tensor_by_name(t, B).double().cpu().permute(0, 3, 1, 2) for t, key in TAPS.items()}
        for lvl in range(3):
            got[f"model.24.m.{lvl}"] = eng.tensor_by_name(f"head{lvl}", B).double().cpu()[..., :3 * (synth_ck.nc + 5)].permute(0, 3, 1, 2)
        # the fused down-block never writes model.1's own output (it goes straight into model.2's stacked 1x1)
        order = [k for k in ORDER if not (k == "model.1" and any(o.kind == 8 for o in eng.plan.ops))]
        rows = {k: _stats(got[k], m64.taps[k]) for k in order}
        report["engine"][name] = rows
        eng.close()
        for k in order:
            e, f = rows[k], floor[k]
            # tracks the floor: no module's error exceeds what accumulation order alone produces in the oracle by more than 2.5x
            # (+ 0.1 % of the rms for the first modules, where the floor is nearly zero: the stem sums 108 products)
            assert e["mean_abs_over_rms"] <= 2.5 * f["mean_abs_over_rms"] + 1e-3, (name, k, e, f)
            assert e["frac_over_1ulp"] <= 2.5 * f["frac_over_1ulp"] + 0.02, (name, k, e, f)   # and no more elements off by 2+ ulps than there
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r02_bf16_deviation_modules.json"), "w") as f:
            json.dump(report, f, indent=1)


def test_bf16_detections_on_the_golden_tiles(lib, synth_ck):
    """The 16 golden tiles at 640 px: bf16 engine against the bf16-emulating oracle (same rounding points) and against the fp32 oracle
    (what a user of detect.py without --half would get).  Bounds = 1.5 x the values measured on MI355X (profiles/r02_bf16_deviation.json)."""
    from aquaculture_amd import engine, tiles
    from oracle import yolov5_oracle as O
    x = tiles.synthetic_batch(range(16), 640)
    eng = engine.Engine(synth_ck, "bf16")
    xt = torch.from_numpy(x).cuda()
    pred = eng.forward_raw(xt).cpu()
    dets, counts = eng.infer(xt)
    counts = counts.cpu().tolist()
    res = {}
    for label, quant in (("vs bf16-emulating oracle", O.q_bf16), ("vs fp32 oracle", O.q_fp32)):
        m = O.model_from_checkpoint(synth_ck, quant)
        ref = torch.cat([m.forward(O.preprocess(x[i:i + 1])) for i in range(16)], 0)
        ref_counts = [r.shape[0] for r in O.non_max_suppression(ref.numpy())]
        dc = (pred[..., 4:] - ref[..., 4:]).abs()
        db = (pred[..., :4] - ref[..., :4]).abs()
        per_tile = [{"tile": i, "count": counts[i], "oracle_count": ref_counts[i], "dconf_mean": float(dc[i].mean()), "dconf_max": float(dc[i].max()),
                     "dbox_mean_px": float(db[i].mean()), "dbox_max_px": float(db[i].max())} for i in range(16)]
        flat_c, flat_b = dc.flatten(), db.flatten()
        res[label] = {"dconf_mean": float(flat_c.mean()), "dconf_p999": float(flat_c.kthvalue(int(0.999 * flat_c.numel()))[0]), "dconf_max": float(flat_c.max()),
                      "dbox_mean_px": float(flat_b.mean()), "dbox_p999_px": float(flat_b.kthvalue(int(0.999 * flat_b.numel()))[0]), "dbox_max_px": float(flat_b.max()),
                      "count_diff_max": max(abs(a - b) for a, b in zip(counts, ref_counts)), "count_diff_sum": sum(abs(a - b) for a, b in zip(counts, ref_counts)),
                      "boxes_total": sum(ref_counts), "per_tile": per_tile}
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r02_bf16_deviation_detections.json"), "w") as f:
            json.dump(res, f, indent=1)
    a = res["vs bf16-emulating oracle"]
    assert a["dconf_mean"] <= BOUNDS["dconf_mean"] and a["dconf_p999"] <= BOUNDS["dconf_p999"] and a["dbox_mean_px"] <= BOUNDS["dbox_mean_px"], a
    assert a["count_diff_max"] <= BOUNDS["count_diff_max"], a


# 1.5 x the values of profiles/r02_bf16_deviation.json (MI355X, this round's kernels, heuristic tile shapes)
BOUNDS = {"dconf_mean": 1.04e-2, "dconf_p999": 7.3e-2, "dbox_mean_px": 1.02, "count_diff_max": 8}   # measured: 6.9e-3, 4.84e-2, 0.678 px, 5 of ~340 boxes
