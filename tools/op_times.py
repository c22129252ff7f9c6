#!/usr/bin/env python3
"""Per-op HIP-event times of one engine step (single stream), autotuned: python tools/op_times.py [--batch 64] [--grep model.9]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aquaculture_amd import checkpoint, engine, tiles  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--grep", default="")
ap.add_argument("--lib", default=None)
ap.add_argument("--variant", default="yolov5m")
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--force-asm1x1", action="store_true", help="the generated-assembly wide 1x1 on every layer it takes (otherwise: the tuned table's pick)")
a = ap.parse_args()
if a.lib:
    engine.load_library(a.lib)
eng = engine.Engine(checkpoint.synthetic_checkpoint(a.variant, 5), "bf16")
x = torch.from_numpy(tiles.synthetic_batch(range(a.batch), a.size)).cuda()
eng.autotune(x)
if a.force_asm1x1:
    for i, o in enumerate(eng.plan.ops):
        if o.kind == 1 and o.k == 1 and o.res is None and o.level < 0 and eng.lib.aq_conv1x1_asm_supported(o.src.channels, o.dst.channels) and \
                not eng.lib.aq_conv1x1_direct_supported(o.src.channels, o.dst.channels):
            eng.set_conv_config(i, engine.CONV_CFG_ASM1X1)
eng.infer(x)
eng.profile(True, a.reps)
for _ in range(a.reps):
    eng.infer(x)
torch.cuda.synchronize()
ms, calls = eng.op_times_ms()
for i, o in enumerate(eng.plan.ops):
    if a.grep in o.name:
        tf = o.flops_per_tile * a.batch / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else 0.0
        fam = eng.last_launches()[i][0]
        print(f"{i:3d} {o.name:28s} k{o.k} s{o.stride} {o.src.channels if o.src else 0:5d}->{o.dst.channels if o.dst else 0:5d} {ms[i] * 1e3:8.1f} us {tf:8.1f} TF/s  {fam}")
print(f"total {ms.sum():.3f} ms over {calls} calls")
