#!/usr/bin/env python3
"""A/B timing at batch 64, 640-px tiles: aq_stem_conv + aq_downblock (two launches) against aq_stemdown (one).  python tools/time_stemdown.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aquaculture_amd import engine as E

lib = E.load_library()
dev = torch.device("cuda", 0)
B, Hi, Wi = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 640, 640
g = torch.Generator().manual_seed(0)
fp = C.POINTER(C.c_float)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr())
nbuf = 4
tiles = [torch.randint(0, 256, (B, Hi, Wi, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(nbuf)]
ws = np.ascontiguousarray((torch.randn(48, 6, 6, 3, generator=g) * 0.25).numpy())
wa = np.ascontiguousarray((torch.randn(96, 3, 3, 48, generator=g) * 0.07).numpy())
wb = np.ascontiguousarray((torch.randn(96, 1, 1, 96, generator=g) * 0.14).numpy())
n = C.c_size_t()
E._check(lib.aq_pack_stem_weights(ws.ctypes.data_as(fp), 48, 0, None, C.byref(n), None))
wsb = torch.empty(n.value, dtype=torch.uint8, device=dev)
E._check(lib.aq_pack_stem_weights(ws.ctypes.data_as(fp), 48, 0, vp(wsb), C.byref(n), st()))
E._check(lib.aq_pack_downblock_weights(wa.ctypes.data_as(fp), wb.ctypes.data_as(fp), None, C.byref(n), None))
wdb = torch.empty(n.value, dtype=torch.uint8, device=dev)
E._check(lib.aq_pack_downblock_weights(wa.ctypes.data_as(fp), wb.ctypes.data_as(fp), vp(wdb), C.byref(n), st()))
bs = torch.zeros(64, device=dev)
bd = torch.zeros(192, device=dev)
x = torch.empty((B, Hi // 2, Wi // 2, 48), dtype=torch.bfloat16, device=dev)
outs = [torch.empty((B, Hi // 4, Wi // 4, 96), dtype=torch.bfloat16, device=dev) for _ in range(nbuf)]


def two(i):
    E._check(lib.aq_stem_conv(vp(tiles[i % nbuf]), vp(x), 48, 0, 48, vp(wsb), vp(bs), B, Hi, Wi, 1, 0, st()))
    E._check(lib.aq_downblock(vp(x), 48, 0, vp(outs[i % nbuf]), 96, 0, vp(wdb), vp(bd), B, Hi // 2, Wi // 2, st()))


def one(i):
    E._check(lib.aq_stemdown(vp(tiles[i % nbuf]), vp(outs[i % nbuf]), 96, 0, vp(wsb), vp(bs), vp(wdb), vp(bd), B, Hi, Wi, st()))


def timeit(fn, reps=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in (("stem + downblock", two), ("stemdown (fused)", one), ("stem + downblock", two), ("stemdown (fused)", one)):
    print(f"B={B}  {name:18s} {timeit(fn):8.1f} us", flush=True)
