#!/usr/bin/env python3
"""Time aq_bottleneck alone on yolov5m's fused-Bottleneck shapes (batch 64 at 640 px), inputs and outputs rotated through > 600 MB
so that nothing is cache-resident, and (--stamp) print the stamped assembly build's per-wave phase cycle sums.

The kernel selection is a process-wide switch (AQ_BTL_ASM=0: HIP-source kernel; default: the assembly kernel where it applies), so an
A/B is two runs of this tool on the same box:

    python tools/time_bottleneck.py ; AQ_BTL_ASM=0 python tools/time_bottleneck.py
"""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aquaculture_amd import engine  # noqa: E402

SHAPES = [(48, 64, 160, 160, "model.2.m"), (96, 64, 80, 80, "model.4.m / model.17.m")]
ASM_PHASES = ["prologue", "C-vmcnt-wait", "phase-B", "C-setup+mfma", "C-epilogue", "barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--only", type=int, default=0, help="hidden width to time (48 or 96; 0 = both)")
    ap.add_argument("--out-ld", type=int, default=0, help="channels per output row (default 2 C: the slice is half of a C3 concat row; C = dense)")
    ap.add_argument("--in-ld", type=int, default=0, help="channels per input row (default 2 C)")
    a = ap.parse_args()
    lib = engine.load_library()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    which = "HIP-source kernel (AQ_BTL_ASM=0)" if os.environ.get("AQ_BTL_ASM") == "0" else "assembly kernel where it applies (C = 48)"
    print(f"# {which}")
    for c, B, H, W, name in SHAPES:
        if a.only and a.only != c:
            continue
        npix = B * H * W
        ld = a.in_ld or 2 * c                                 # as in the C3 concat buffer: the slice is half of the row
        old = a.out_ld or 2 * c
        per = npix * c * 2 * 2
        nbuf = max(2, int(640e6 // (npix * ld * 2 * 2)) + 1)
        base = (torch.randn(npix, ld, generator=g) * 0.5).bfloat16().cuda()
        xs = [base.clone() for _ in range(nbuf)]
        ys = [torch.empty(npix, old, dtype=torch.bfloat16, device="cuda") for _ in range(nbuf)]
        w1 = np.ascontiguousarray((torch.randn(c, 1, 1, c, generator=g) * (2.0 / c) ** 0.5).numpy())
        w2 = np.ascontiguousarray((torch.randn(c, 3, 3, c, generator=g) * (2.0 / (9 * c)) ** 0.5).numpy())
        n = C.c_size_t()
        p1, p2 = w1.ctypes.data_as(C.POINTER(C.c_float)), w2.ctypes.data_as(C.POINTER(C.c_float))
        engine._check(lib.aq_pack_bottleneck_weights(p1, p2, c, None, C.byref(n), None))
        wbuf = torch.empty(n.value, dtype=torch.uint8, device="cuda")
        engine._check(lib.aq_pack_bottleneck_weights(p1, p2, c, wbuf.data_ptr(), C.byref(n), st))
        bias = (torch.randn(2 * c, generator=g) * 0.1).cuda()

        def run(i, shortcut=1):
            engine._check(lib.aq_bottleneck(xs[i % nbuf].data_ptr(), ld, 0, ys[i % nbuf].data_ptr(), old, old - c, c, wbuf.data_ptr(), bias.data_ptr(),
                                            B, H, W, shortcut, st))
        for sc in (1, 0):
            for i in range(5):
                run(i, sc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters):
                run(i, sc)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            flops = 2.0 * npix * c * c * 10
            print(f"C={c:3d} {B}x{H}x{W} shortcut={sc}: {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {per / us / 1e6:5.2f} TB/s (in + out once)  {name}", flush=True)
        if a.stamp and c == 96 and os.environ.get("AQ_BTL_ASM") != "0" and os.environ.get("AQ_BTL96_ASM") != "0":
            buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
            engine._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
            try:
                run(0)
                torch.cuda.synchronize()
            finally:
                lib.aq_debug_conv_stamp(None, 0)
            t = buf.cpu().view(-1, 8)[:, :4]
            t = t[t.sum(1) > 0].double()
            print(f"  stamped build: {t.shape[0]} waves, {t.sum(1).mean():.0f} cycles per wave | " +
                  " ".join(f"{nm}={v:.0f}" for nm, v in zip(["prologue+phase-B", "barrier+C-setup", "C-mfma", "C-epilogue"], t.mean(0).tolist())))
        if a.stamp and c == 48 and os.environ.get("AQ_BTL_ASM") != "0":
            buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
            engine._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
            try:
                run(0)
                torch.cuda.synchronize()
            finally:
                lib.aq_debug_conv_stamp(None, 0)
            t = buf.cpu().view(-1, 8)[:, :6]
            t = t[t.sum(1) > 0].double()
            tot = t.sum(1)
            print(f"  stamped build: {t.shape[0]} waves, {tot.mean():.0f} cycles per wave | " +
                  " ".join(f"{nm}={v:.0f}" for nm, v in zip(ASM_PHASES, t.mean(0).tolist())))
            for grp in (0, 1):
                sel = t.view(-1, 8, 6)[:, 4 * grp:4 * grp + 4].reshape(-1, 6)
                print(f"    waves {4 * grp}-{4 * grp + 3}: " + " ".join(f"{nm}={v:.0f}" for nm, v in zip(ASM_PHASES, sel.mean(0).tolist())))


if __name__ == "__main__":
    main()
