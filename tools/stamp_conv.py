#!/usr/bin/env python3
"""Where do the conv kernels spend their cycles?  Runs STAMPED diagnostic builds of a few tile shapes on real layer
shapes (through aq_conv2d) and prints each phase's share of the wave lifetime (mean over waves).

Phases: 0 prologue | 1 stage setup (cursor, tile decode) | 2 compute (ds_read + MFMA + interleaved LDS-DMA issue)
        3 vmcnt wait | 4 barrier | 5 barrier before the epilogue | 6 epilogue | 7 wait + barrier after the epilogue
Read SHARES, not lengths: stamps fence the instruction stream (cdna_hip_programming.md section 7).
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaculture_amd import engine  # noqa: E402

NAMES = ["prologue", "stage-setup", "compute", "vmcnt-wait", "barrier", "pre-epi-bar", "epilogue", "post-epi"]
# (name, B, H, W, cin, cout, k, stride, residual, [configs])    igemm ids: 1, 6, 7, 10 ; halo id: n_igemm + 0
LAYERS = [
    ("model.23.m 3x3 384@20", 64, 20, 20, 384, 384, 3, 1, False, [1, 6, "h0"]),
    ("model.6.m  3x3 192@40", 64, 40, 40, 192, 192, 3, 1, True, [1, 6, "h0"]),
    ("model.4.m  3x3  96@80", 64, 80, 80, 96, 96, 3, 1, True, [7, 6]),
    ("model.2.m  3x3  48@160", 64, 160, 160, 48, 48, 3, 1, True, [10, 7]),
    ("model.2.m  1x1  48@160", 64, 160, 160, 48, 48, 1, 1, False, [10]),
    ("model.6.m  1x1 192@40", 64, 40, 40, 192, 192, 1, 1, False, [6, 10]),
    ("model.0 stem 16->48@320", 64, 320, 320, 16, 48, 3, 1, False, [10]),
]


BTL_NAMES = ["prologue", "vmcnt-wait", "tile-barrier", "dma-issue", "phase-B", "mid-barrier", "C-mfma", "C-epilogue"]


def bottlenecks(lib, dev):
    """Stamped builds of the fused Bottleneck kernel (csrc/bottleneck.hip) on yolov5m's two shapes."""
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    g = torch.Generator().manual_seed(0)
    for name, B, H, W, c in (("model.2.m fused 48@160", 64, 160, 160, 48), ("model.4.m fused 96@80", 64, 80, 80, 96)):
        x = torch.randn(B, H, W, c, generator=g).bfloat16().to(dev)
        w1 = torch.randn(c, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
        w2 = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
        b = torch.randn(c, generator=g) * 0.1
        buf.zero_()
        engine._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
        try:
            engine.bottleneck_nhwc(x, w1, b, w2, b, True)
        finally:
            lib.aq_debug_conv_stamp(None, 0)
        t = buf.cpu().view(-1, 8)
        t = t[t.sum(1) > 0].double()
        tot = t.sum(1)
        share = (t / tot[:, None]).mean(0) * 100
        print(f"{name:26s} waves {t.shape[0]:5d} ticks/wave {tot.mean():9.0f} | " + " ".join(f"{n}={v:4.1f}%" for n, v in zip(BTL_NAMES, share.tolist())))


def main():
    lib = engine.load_library()
    if "--bottleneck" in sys.argv:
        return bottlenecks(lib, torch.device("cuda", 0))
    n_igemm = 21
    dev = torch.device("cuda", 0)
    buf = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    g = torch.Generator().manual_seed(0)
    for name, B, H, W, cin, cout, k, s, res, cfgs in LAYERS:
        x = torch.randn(B, H, W, cin, generator=g).bfloat16().to(dev)
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        r = torch.randn(B, H // s, W // s, cout, generator=g).bfloat16().to(dev) if res else None
        for cfg in cfgs:
            cid = n_igemm + int(cfg[1:]) if isinstance(cfg, str) else cfg
            buf.zero_()
            engine._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
            try:
                for _ in range(2):
                    engine.conv2d_nhwc(x, w, b, stride=s, act=True, residual=r, precision="bf16", cfg=cid)
            finally:
                lib.aq_debug_conv_stamp(None, 0)
            t = buf.cpu().view(-1, 8)
            t = t[t.sum(1) > 0].double()
            if t.shape[0] == 0:
                print(f"{name:26s} cfg {cfg}: no stamped build")
                continue
            tot = t.sum(1)
            share = (t / tot[:, None]).mean(0) * 100
            print(f"{name:26s} cfg {str(cfg):3s} waves {t.shape[0]:5d} cycles/wave {tot.mean():9.0f} | " +
                  " ".join(f"{n}={v:4.1f}%" for n, v in zip(NAMES, share.tolist())))


if __name__ == "__main__":
    main()
