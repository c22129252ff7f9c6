#!/usr/bin/env python3
"""A/B timing of the 3x3 / stride-2 kernels on the yolov5m layer shapes at batch 64 (model.5 / 7 / 18 / 21): the planar stride-2 kernel
(aq_conv3x3_pl_s2, generated assembly) against the implicit-GEMM tile shapes the tuner used to pick for them (aq_conv2d with
AQ_CONV_CFG).  Operands rotate through several buffers so nothing but the weights is cache resident.
Usage: python tools/time_conv3x3_s2.py [--reps 20] [--stamp] [--abl 1,2,3,4,7,8]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aquaculture_amd import engine as E

LAYERS = (("model.5", 80, 80, 192, 384), ("model.7", 40, 40, 384, 768), ("model.18", 80, 80, 192, 192), ("model.21", 40, 40, 384, 384))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--old", default="6,10", help="aq_conv2d tile configs to time beside the planar kernel")
    ap.add_argument("--stamp", action="store_true", help="run the stamped build once and print per-phase cycle sums per wave")
    ap.add_argument("--abl", default="", help="timing-only ablations of the stamped build (1 no weight loads, 2 no LDS-DMA, 4 no B reads, 8 no MFMAs; sums)")
    ap.add_argument("--layers", default="", help="comma list of layer names (default: all four)")
    a = ap.parse_args()
    lib = E.load_library()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for name, H, W, cin, cout in LAYERS:
        if a.layers and name not in a.layers.split(","):
            continue
        B = a.batch
        g = torch.Generator().manual_seed(cin + cout)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        nbuf = 4
        xs = [(torch.randn(B, H, W, cin, generator=g) * 0.8).bfloat16().to(dev) for _ in range(nbuf)]
        outs = [torch.empty(B, H // 2, W // 2, cout, dtype=torch.bfloat16, device=dev) for _ in range(nbuf)]
        wk = np.ascontiguousarray(w.permute(0, 2, 3, 1).float().numpy())
        wp = wk.ctypes.data_as(C.POINTER(C.c_float))
        n = C.c_size_t()
        E._check(lib.aq_pack_conv3x3_pl_s2(wp, cin, cout, None, C.byref(n), None))
        wpl = torch.empty(n.value, dtype=torch.uint8, device=dev)
        E._check(lib.aq_pack_conv3x3_pl_s2(wp, cin, cout, wpl.data_ptr(), C.byref(n), st))
        wold = E.pack_conv_weights(w, "bf16", dev)
        bb = torch.zeros(cout + 1024, dtype=torch.float32, device=dev)
        bb[:cout] = torch.randn(cout, generator=g) * 0.2
        zero = E._zero_page(dev)
        flops = 2.0 * B * (H // 2) * (W // 2) * cin * cout * 9

        def run_pl(i):
            E._check(lib.aq_conv3x3_pl_s2(xs[i % nbuf].data_ptr(), cin, 0, cin, outs[i % nbuf].data_ptr(), cout, 0, cout,
                                          wpl.data_ptr(), bb.data_ptr(), B, H, W, 1, st))

        def run_old(i):
            E._check(lib.aq_conv2d(xs[i % nbuf].data_ptr(), cin, 0, cin, outs[i % nbuf].data_ptr(), cout, 0, cout, None, 0, 0,
                                   wold.data_ptr(), bb.data_ptr(), B, H, W, 3, 2, 1, 1, 0, 0, zero.data_ptr(), st))

        def timeit(fn):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.reps * 1e3

        # interleaved rounds in one process (rule 24): planar, old, planar, old ...
        res = {}
        for rnd in range(3):
            res.setdefault("planar s2 asm", []).append(timeit(run_pl))
            for cfg in [int(v) for v in a.old.split(",") if v]:
                for flag in (0, E.CONV_CFG_ONE_TILE_PER_WG):
                    os.environ["AQ_CONV_CFG"] = str(cfg | flag)
                    try:
                        res.setdefault(f"igemm cfg {cfg | flag}", []).append(timeit(run_old))
                    except RuntimeError as err:
                        res[f"igemm cfg {cfg | flag}"] = [float("nan")]
            os.environ.pop("AQ_CONV_CFG", None)
        for k, v in res.items():
            us = float(np.median(v))
            print(f"{name} {cin}->{cout} {H}x{W} B{B}  {k:18s}: median {us:8.1f} us (min {min(v):.1f})  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
        # parity spot check of the timed configuration against the implicit-GEMM kernel (bit-level differences only from accumulation order)
        run_pl(0)
        torch.cuda.synchronize()
        got = outs[0].float().clone()
        os.environ["AQ_CONV_CFG"] = "6"
        run_old(0)
        torch.cuda.synchronize()
        os.environ.pop("AQ_CONV_CFG", None)
        d = (got - outs[0].float()).abs()
        print(f"{name}: |planar - igemm| max {float(d.max()):.4f} mean {float(d.mean()):.6f}", flush=True)
        if not a.stamp:
            continue
        for abl_ in [""] + [v for v in a.abl.split(",") if v]:
            buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
            E._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
            os.environ["AQ_PL_ASM"] = "2"
            if abl_:
                os.environ["AQ_PL_ASM_ABL"] = abl_
            for i in range(3):
                run_pl(i)
            torch.cuda.synchronize()
            buf.zero_()
            run_pl(3)
            torch.cuda.synchronize()
            lib.aq_debug_conv_stamp(None, 0)
            os.environ.pop("AQ_PL_ASM", None)
            os.environ.pop("AQ_PL_ASM_ABL", None)
            t = buf.cpu().view(-1, 8).double()
            t = t[t[:, 6] > 0]
            if t.shape[0] == 0:
                print(f"{name} stamped run wrote no rows: skipped", flush=True)
                continue
            names = ["prologue", "chunk-barrier", "stream", "tile-setup", "epilogue", "chunk-top"]
            life, ticks = t[:, 6], t[:, 7]
            print(f"{name} stamped ABL={abl_ or 0}: waves {t.shape[0]}  lifetime {life.mean():.0f} cycles (min {life.min():.0f} max {life.max():.0f}) = "
                  f"{ticks.mean() * 10:.0f} ns -> clock {(life / ticks).median() * 100:.0f} MHz | " + " ".join(f"{n}={t[:, k].mean():.0f}" for k, n in enumerate(names)), flush=True)


if __name__ == "__main__":
    main()
