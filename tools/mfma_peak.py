#!/usr/bin/env python3
"""Sustained bf16 MFMA rate of this chip (register-only loop, no memory traffic): the practical ceiling under DVFS."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaculture_amd import engine  # noqa: E402

lib = engine.load_library()
out = torch.zeros(16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for blocks_per_cu, sign in ((1, 1), (2, 1), (1, -1), (2, -1)):      # sign -1: v_mfma_f32_16x16x32_bf16 (same FLOPs per iteration)
    blocks, iters = 256 * blocks_per_cu, 20000 * sign
    engine._check(lib.aq_debug_mfma_peak(blocks, 2000 * sign, out.data_ptr(), st))
    torch.cuda.synchronize()
    best = 0.0
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine._check(lib.aq_debug_mfma_peak(blocks, iters, out.data_ptr(), st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = blocks * 4 * abs(iters) * 8 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12
        best = max(best, tf)
        print(f"{'32x32x16' if sign > 0 else '16x16x32'} {blocks_per_cu} block(s)/CU x 4 waves: {ms:8.3f} ms  {tf:7.1f} TFLOP/s  -> {tf / 256 / 4 / (2 * 32 * 32 * 16 / 32) * 1e12 / 1e9:5.2f} GHz-equivalent")
