#!/usr/bin/env python3
"""Time aq_conv1x1_asm (generated assembly, csrc/gen_conv1x1_asm.py) alone on yolov5m's K >= 768 1x1 shapes (batch 64 at 640 px), beside the
implicit-GEMM kernel's numbers in profiles/r04_per_op_pmc.txt -- inputs and outputs rotated through > 600 MB so nothing is cache-resident.
--stamp: per-wave phase cycle sums of the stamped build.  AQ_C1_ASM_KERNEL=<name> times an ablation kernel of an AQ_GEN_EXPERIMENTAL=1 build.

    python tools/time_conv1x1_asm.py [--iters 30] [--stamp]
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

SHAPES = [  # (cin, cout, pixels at batch 64, layer)
    (768, 768, 64 * 20 * 20, "model.8.cv1|cv2 / cv3, model.23.cv1|cv2 / cv3"),
    (768, 384, 64 * 20 * 20, "model.9.cv1, model.10"),
    (1536, 768, 64 * 20 * 20, "model.9.cv2"),
    (768, 384, 64 * 40 * 40, "model.13.cv1|cv2"),
    (384, 384, 64 * 40 * 40, "model.6.cv1|cv2 / cv3, 13.cv3, 20.*  (direct kernel's layers)"),
    (384, 384, 64 * 20 * 20, "model.8.m.*.cv1, model.23.m.*.cv1  (direct kernel's layers)"),
]
PHASES = ["prologue", "stream", "barrier", "epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--only", type=int, default=-1, help="index of the one shape to time")
    ap.add_argument("--hot", action="store_true", help="one buffer set instead of > 600 MB of them: operands stay in L2 / MALL between launches")
    a = ap.parse_args()
    lib = engine.load_library()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    print(f"# kernel: {os.environ.get('AQ_C1_ASM_KERNEL') or 'conv1x1_asm_nb13 / nb7 (AQ_C1_ASM_NB=' + os.environ.get('AQ_C1_ASM_NB', 'auto') + ')'}")
    for si, (cin, cout, npix, name) in enumerate(SHAPES):
        if a.only >= 0 and a.only != si:
            continue
        per = npix * (cin + cout) * 2
        nbuf = 1 if a.hot else max(2, int(640e6 // per) + 1)
        base = (torch.randn(npix, cin, generator=g) * 0.5).bfloat16().cuda()
        xs = [base.clone() for _ in range(nbuf)]
        ys = [torch.empty(npix, cout, dtype=torch.bfloat16, device="cuda") for _ in range(nbuf)]
        w = np.ascontiguousarray((torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5).numpy())
        n = C.c_size_t()
        wp = w.ctypes.data_as(C.POINTER(C.c_float))
        engine._check(lib.aq_pack_conv1x1_asm(wp, cin, cout, None, C.byref(n), None))
        wbuf = torch.empty(n.value, dtype=torch.uint8, device="cuda")
        engine._check(lib.aq_pack_conv1x1_asm(wp, cin, cout, wbuf.data_ptr(), C.byref(n), st))
        bias = (torch.randn(cout, generator=g) * 0.1).cuda()

        def run(i):
            engine._check(lib.aq_conv1x1_asm(xs[i % nbuf].data_ptr(), cin, 0, ys[i % nbuf].data_ptr(), cout, 0, cin, cout, wbuf.data_ptr(),
                                             bias.data_ptr(), npix, 1, st))
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        fl = 2.0 * npix * cin * cout
        ref = torch.nn.functional.silu(xs[0][:4096].float() @ torch.from_numpy(w).cuda().bfloat16().float().t() + bias)
        run(0)
        torch.cuda.synchronize()
        err = float((ys[0][:4096].float() - ref).abs().max())
        print(f"{cin:4d}->{cout:4d} {npix:7d} px  {us:7.1f} us  {fl / us * 1e-6:6.0f} TFLOP/s  {per / us / 1e6:5.2f} TB/s  maxerr {err:.3f}  {name}", flush=True)
        if a.stamp:
            buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
            engine._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
            try:
                run(0)
                torch.cuda.synchronize()
            finally:
                lib.aq_debug_conv_stamp(None, 0)
            t = buf.cpu().view(-1, 8)[:, :len(PHASES)]
            t = t[t.sum(1) > 0].double()
            print(f"  stamped build: {t.shape[0]} waves, {t.sum(1).mean():.0f} cycles per wave | " +
                  " ".join(f"{nm}={v:.0f}" for nm, v in zip(PHASES, t.mean(0).tolist())))


if __name__ == "__main__":
    main()
