#!/usr/bin/env python3
"""Per tile-configuration time of chosen conv layers inside the running engine (what the autotuner sees, as a table).

Usage (GPU box): python tools/sweep_conv_cfg.py --ops model.6.m.0.cv2 model.8.m.0.cv2 [--batch 64] [--reps 20]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aquaculture_amd import checkpoint, engine, tiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", nargs="+", default=["model.6.m.0.cv2", "model.8.m.0.cv2", "model.7", "model.18"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None, help="alternative libaqengine.so (A/B runs)")
    ap.add_argument("--top", type=int, default=12)
    a = ap.parse_args()
    if a.lib:
        engine.load_library(a.lib)
    ck = checkpoint.synthetic_checkpoint("yolov5m", 5)
    eng = engine.Engine(ck, "bf16")
    x = torch.from_numpy(tiles.synthetic_batch(range(a.batch), 640)).cuda()
    eng.autotune(x)
    lib = eng.lib
    ncfg = lib.aq_conv_num_configs()
    names = [o.name for o in eng.plan.ops]
    for name in a.ops:
        oi = names.index(name)
        tuned = lib.aq_engine_get_conv_config(eng.handle, oi)
        rows = []
        for cfg in list(range(ncfg)) + [c | engine.CONV_CFG_ONE_TILE_PER_WG for c in range(ncfg)] + [engine.CONV_CFG_DIRECT1X1, engine.CONV_CFG_DIRECT3X3S2]:
            try:
                eng.set_conv_config(oi, cfg)
            except RuntimeError:
                continue
            try:
                eng.forward_raw(x)
            except RuntimeError:
                continue
            eng.profile(True, a.reps)
            for _ in range(a.reps):
                eng.forward_raw(x)
            torch.cuda.synchronize()
            ms, calls = eng.op_times_ms()
            eng.profile(False)
            bm, bn = C.c_int(), C.c_int()
            if (cfg & 4095) < ncfg:
                lib.aq_conv_config_tiles(cfg, C.byref(bm), C.byref(bn))
            rows.append((float(ms[oi]) * 1e3, cfg, bm.value, bn.value))
        eng.set_conv_config(oi, -1)
        rows.sort()
        print(f"{name} (op {oi}, tuned cfg {tuned}):")
        for us, cfg, bm, bn in rows[:a.top]:
            print(f"   cfg {cfg & 4095:4d}{'+1/wg' if cfg >= 4096 else '     '}  {bm:3d}x{bn:3d}  {us:8.1f} us")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
