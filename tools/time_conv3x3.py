#!/usr/bin/env python3
"""A/B timing of the 3x3/s1 kernels on the real yolov5m layer shapes at batch 64: the planar kernel (aq_conv3x3_pl, every pixel-block
count) against the implicit-GEMM / halo tile shapes (aq_conv2d with AQ_CONV_CFG).  Operands rotate through several buffers so
nothing but the weights is cache resident.  Usage: python tools/time_conv3x3.py [--reps 20]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aquaculture_amd import engine as E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--old", default="21,30", help="aq_conv2d configs to time beside the planar kernel (21 = halo 192x256)")
    ap.add_argument("--stamp", action="store_true", help="run the stamped build once and print per-phase cycle sums per wave")
    ap.add_argument("--abl", default="", help="timing-only ablation builds of the planar kernel to time (comma list of AQ_PL_ABL values)")
    ap.add_argument("--nbs", default="13,10,7", help="pixel-block counts of the planar kernel to time ('' = automatic choice only)")
    ap.add_argument("--lib", default=None, help="another build of libaqengine.so to time (experiments)")
    a = ap.parse_args()
    lib = E.load_library(a.lib) if a.lib else E.load_library()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for (H, W, c) in ((40, 40, 192), (20, 20, 384)):
        B = a.batch
        g = torch.Generator().manual_seed(c)
        w = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
        b = torch.randn(c, generator=g) * 0.2
        nbuf = 6
        xs = [(torch.randn(B, H, W, c, generator=g) * 0.8).bfloat16().to(dev) for _ in range(nbuf)]
        outs = [torch.randn(B, H, W, 2 * c, generator=g).bfloat16().to(dev) for _ in range(nbuf)]   # shortcut in place, slice of a concat buffer
        wk = np.ascontiguousarray(w.permute(0, 2, 3, 1).float().numpy())
        wp = wk.ctypes.data_as(C.POINTER(C.c_float))
        n = C.c_size_t()
        E._check(lib.aq_pack_conv3x3_pl(wp, c, c, None, C.byref(n), None))
        wpl = torch.empty(n.value, dtype=torch.uint8, device=dev)
        E._check(lib.aq_pack_conv3x3_pl(wp, c, c, wpl.data_ptr(), C.byref(n), st))
        wold = E.pack_conv_weights(w, "bf16", dev)
        bb = torch.zeros(c + 512, dtype=torch.float32, device=dev)
        bb[:c] = b
        zero = E._zero_page(dev)
        flops = 2.0 * B * H * W * c * c * 9

        def run_pl(i):
            o = outs[i % nbuf]
            E._check(lib.aq_conv3x3_pl(xs[i % nbuf].data_ptr(), c * 2, 16, c, o.data_ptr(), 2 * c, 0, c, o.data_ptr(), 2 * c, 0,
                                       wpl.data_ptr(), bb.data_ptr(), B, H, W, 1, st))

        from aquaculture_amd import quant
        wq = np.ascontiguousarray(quant.quantize_rows(wk.reshape(c, -1))[0].reshape(wk.shape))
        wqp = wq.ctypes.data_as(C.POINTER(C.c_float))
        bh = np.ascontiguousarray(b.numpy())
        E._check(lib.aq_pack_conv3x3_pl_w8(wqp, bh.ctypes.data_as(C.POINTER(C.c_float)), c, c, None, C.byref(n), None, None))
        w8 = torch.empty(n.value, dtype=torch.uint8, device=dev)
        sb8 = torch.empty(2048, dtype=torch.float32, device=dev)
        E._check(lib.aq_pack_conv3x3_pl_w8(wqp, bh.ctypes.data_as(C.POINTER(C.c_float)), c, c, w8.data_ptr(), C.byref(n), sb8.data_ptr(), st))

        def run_w8(i):
            o = outs[i % nbuf]
            E._check(lib.aq_conv3x3_pl_w8(xs[i % nbuf].data_ptr(), c * 2, 16, c, o.data_ptr(), 2 * c, 0, c, o.data_ptr(), 2 * c, 0,
                                          w8.data_ptr(), sb8.data_ptr(), B, H, W, 1, st))

        def run_old(i):
            o = outs[i % nbuf]
            E._check(lib.aq_conv2d(xs[i % nbuf].data_ptr(), c, 0, c, o.data_ptr(), 2 * c, 0, c, o.data_ptr(), 2 * c, 0,
                                   wold.data_ptr(), bb.data_ptr(), B, H, W, 3, 1, 1, 1, 0, 0, zero.data_ptr(), st))

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

        for nb in [int(v) for v in a.nbs.split(',') if v]:
            os.environ["AQ_PL_NB"] = str(nb)
            for asm in {13: (1, 0), 7: (1, 0), 8: (1,)}.get(nb, (0,)):
                os.environ["AQ_PL_ASM"] = str(asm)
                try:
                    us = timeit(run_pl)
                    print(f"{c}ch {H}x{W} B{B}  planar NB={nb:2d} {'asm' if asm else 'hip'}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
                except RuntimeError as err:
                    print(f"{c}ch planar NB={nb}: {err}")
        os.environ.pop("AQ_PL_NB", None)
        os.environ.pop("AQ_PL_ASM", None)
        fams = [nb for nb in (13, 7, 8) if lib.aq_conv3x3_pl_asm_family(nb) == 1]       # (7, 8: only in AQ_GEN_EXPERIMENTAL=1 builds)
        asm_abl = [f"asm{nb}:{v}" for nb in (7, 13) if nb in fams for v in a.abl.split(",") if v]
        for sab in ([f"asm{nb}" for nb in fams] + asm_abl + [16] + [16 + int(v) for v in a.abl.split(",") if v and int(v) < 8] if a.stamp else []):
            buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
            E._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
            if isinstance(sab, str):
                nb_, _, abl_ = sab[3:].partition(":")
                os.environ["AQ_PL_ASM"], os.environ["AQ_PL_NB"], sab = "2", nb_, 16 + int(abl_ or 0)
                if abl_:
                    os.environ["AQ_PL_ASM_ABL"] = abl_
            else:
                os.environ["AQ_PL_ABL"] = str(sab)
            for i in range(3):
                run_pl(i)
            torch.cuda.synchronize()
            buf.zero_()
            run_pl(3)
            torch.cuda.synchronize()
            lib.aq_debug_conv_stamp(None, 0)
            asm = os.environ.pop("AQ_PL_ASM", None)
            nbs = os.environ.pop("AQ_PL_NB", "")
            os.environ.pop("AQ_PL_ASM_ABL", None)
            os.environ.pop("AQ_PL_ABL", None)
            t = buf.cpu().view(-1, 8).double()
            t = t[t[:, 6] > 0]
            if t.shape[0] == 0:
                print(f"{c}ch stamped run wrote no rows (variant not built / not selected): skipped", flush=True)
                continue
            names = ["prologue", "chunk-barrier", "stream", "tile-setup", "epilogue", "chunk-top"]
            life, ticks = t[:, 6], t[:, 7]
            print(f"{c}ch stamped {'asm NB=' + nbs if asm else 'hip'} ABL={sab - 16}: waves {t.shape[0]}  lifetime {life.mean():.0f} cycles (min {life.min():.0f} max {life.max():.0f}) = {ticks.mean() * 10:.0f} ns "
                  f"-> clock {(life / ticks).median() * 100:.0f} MHz | " + " ".join(f"{n}={t[:, k].mean():.0f}" for k, n in enumerate(names)), flush=True)
        us = timeit(run_w8)
        print(f"{c}ch {H}x{W} B{B}  planar NB=13 asm, e4m3 weight stream: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
        # fp8 on both MFMA operands (family f8nb13): the input as e4m3 codes
        xq = [(x.float() / 0.01).clamp(-448, 448).to(torch.float8_e4m3fn) for x in xs]
        E._check(lib.aq_pack_conv3x3_pl_f8(wp, bh.ctypes.data_as(C.POINTER(C.c_float)), c, c, 0.01, None, C.byref(n), None, None))
        wf8 = torch.empty(n.value, dtype=torch.uint8, device=dev)
        sbf8 = torch.empty(2048, dtype=torch.float32, device=dev)
        E._check(lib.aq_pack_conv3x3_pl_f8(wp, bh.ctypes.data_as(C.POINTER(C.c_float)), c, c, 0.01, wf8.data_ptr(), C.byref(n), sbf8.data_ptr(), st))

        def run_f8(i):
            o = outs[i % nbuf]
            E._check(lib.aq_conv3x3_pl_f8(xq[i % nbuf].data_ptr(), c, 0, c, o.data_ptr(), 2 * c, 0, c, o.data_ptr(), 2 * c, 0,
                                          wf8.data_ptr(), sbf8.data_ptr(), B, H, W, 1, st))

        r_f8, r_bf = [], []
        for rnd in range(3):                                  # interleaved rounds in one process
            r_f8.append(timeit(run_f8))
            r_bf.append(timeit(run_pl))
        print(f"{c}ch {H}x{W} B{B}  planar fp8 x fp8 (f8nb13): median {np.median(r_f8):8.1f} us (min {min(r_f8):.1f})  {flops / np.median(r_f8) / 1e6:7.1f} TFLOP/s   "
              f"[bf16 planar beside it: {np.median(r_bf):.1f} us]", flush=True)
        if a.stamp:
            for abl_ in [""] + [v for v in a.abl.split(",") if v]:
                buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
                E._check(lib.aq_debug_conv_stamp(buf.data_ptr(), buf.numel() * 8))
                os.environ["AQ_PL_ASM"] = "2"
                if abl_:
                    os.environ["AQ_PL_ASM_ABL"] = abl_
                for i in range(3):
                    run_f8(i)
                torch.cuda.synchronize()
                buf.zero_()
                run_f8(3)
                torch.cuda.synchronize()
                lib.aq_debug_conv_stamp(None, 0)
                os.environ.pop("AQ_PL_ASM", None)
                os.environ.pop("AQ_PL_ASM_ABL", None)
                t = buf.cpu().view(-1, 8).double()
                t = t[t[:, 6] > 0]
                if t.shape[0]:
                    names = ["prologue", "chunk-barrier", "stream", "tile-setup", "epilogue", "chunk-top"]
                    life, ticks = t[:, 6], t[:, 7]
                    print(f"{c}ch fp8 stamped ABL={abl_ or 0}: waves {t.shape[0]}  lifetime {life.mean():.0f} cycles = {ticks.mean() * 10:.0f} ns -> clock "
                          f"{(life / ticks).median() * 100:.0f} MHz | " + " ".join(f"{n_}={t[:, k].mean():.0f}" for k, n_ in enumerate(names)), flush=True)
        us = timeit(run_pl)
        print(f"{c}ch {H}x{W} B{B}  planar auto : {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
        for cfg in [int(v) for v in a.old.split(",") if v]:
            for flag in (0, E.CONV_CFG_ONE_TILE_PER_WG):
                os.environ["AQ_CONV_CFG"] = str(cfg | flag)
                try:
                    us = timeit(run_old)
                    print(f"{c}ch {H}x{W} B{B}  old cfg {cfg | flag:5d}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
                except RuntimeError as err:
                    print(f"{c}ch old cfg {cfg | flag}: {err}")
        os.environ.pop("AQ_CONV_CFG", None)


if __name__ == "__main__":
    main()
