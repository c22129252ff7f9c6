#!/usr/bin/env python3
"""Time aq_conv1x1_direct alone on the yolov5m 1x1 shapes it serves (batch 64 at 640 px) against the HBM floor of each layer.

Inputs and outputs rotate through enough buffers (> 600 MB in total) that no launch finds its operands in L2 / MALL.
Usage (GPU box): python tools/time_conv1x1.py [--iters 40]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aquaculture_amd import engine  # noqa: E402

SHAPES = [  # (cin, cout, pixels at batch 64, layer)
    (96, 96, 64 * 160 * 160, "model.2.cv3"),
    (192, 192, 64 * 80 * 80, "model.4.cv1|cv2 / cv3, 17.cv3"),
    (192, 192, 64 * 40 * 40, "model.6/13/20 m.cv1"),
    (384, 384, 64 * 40 * 40, "model.6.cv1|cv2 / cv3, 13.cv3, 20.*"),
    (384, 384, 64 * 20 * 20, "model.8/23 m.cv1"),
    (384, 192, 64 * 40 * 40, "model.14"),
    (384, 192, 64 * 80 * 80, "model.17.cv1|cv2"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--lib", default=None, help="alternative libaqengine.so (A/B runs)")
    a = ap.parse_args()
    lib = engine.load_library(a.lib) if a.lib else engine.load_library()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    for cin, cout, npix, name in SHAPES:
        per = npix * (cin + cout) * 2
        nbuf = max(2, int(640e6 // per) + 1)
        xs = [(torch.randn(npix, cin, generator=g) * 0.5).bfloat16().cuda() for _ in range(min(nbuf, 2))]
        xs = (xs * nbuf)[:nbuf] if nbuf <= 2 else [xs[i % 2].clone() for i in range(nbuf)]
        ys = [torch.empty(npix, cout, dtype=torch.bfloat16, device="cuda") for _ in range(nbuf)]
        w = np.ascontiguousarray((torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5).numpy())
        n = C.c_size_t()
        wp = w.ctypes.data_as(C.POINTER(C.c_float))
        engine._check(lib.aq_pack_conv1x1_direct(wp, cin, cout, None, C.byref(n), None))
        wbuf = torch.empty(n.value, dtype=torch.uint8, device="cuda")
        engine._check(lib.aq_pack_conv1x1_direct(wp, cin, cout, wbuf.data_ptr(), C.byref(n), st))
        bias = torch.zeros(cout, device="cuda")

        def run(i):
            engine._check(lib.aq_conv1x1_direct(xs[i % nbuf].data_ptr(), cin, 0, ys[i % nbuf].data_ptr(), cout, 0, cin, cout, wbuf.data_ptr(),
                                                bias.data_ptr(), npix, 1, st))
        for i in range(5):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        floor = per / 8e12 * 1e6
        ref = torch.nn.functional.silu(xs[0][:4096].float() @ torch.from_numpy(w).cuda().bfloat16().float().t())
        run(0)
        torch.cuda.synchronize()
        err = float((ys[0][:4096].float() - ref).abs().max())
        print(f"{cin:4d}->{cout:4d} {npix:8d} px  {us:7.1f} us  {per / us / 1e6:5.2f} TB/s  floor {floor:5.1f} us ({100 * floor / us:4.1f} %)  maxerr {err:.3f}  {name}", flush=True)


if __name__ == "__main__":
    main()
