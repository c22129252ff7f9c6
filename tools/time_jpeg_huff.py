#!/usr/bin/env python3
"""GPU entropy decode (aq_jpeg_huffman_decode) alone: time per launch and images/s for super-batches of 1024-px tiles, smooth (0.3 bit per
pixel, the synthetic ocean tiles) and with sensor noise (1.1 bpp), one launch at a time and with several launches in flight on separate
streams; beside it the host decoder's time per tile on one core of this box.

    python tools/time_jpeg_huff.py [--n 512] [--size 1024]
"""
import argparse
import io
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aquaculture_amd import engine, jpeg, tiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=4)
    a = ap.parse_args()
    from PIL import Image
    H = W = a.size
    engine.load_library()
    for label, noise in (("smooth", 0.0), ("noise sigma 10", 10.0)):
        rng = np.random.default_rng(3)
        files = []
        for i in range(16):
            t = tiles.synthetic_tile(i, a.size).astype(np.float32)
            if noise:
                t = np.clip(t + rng.normal(0, noise, t.shape), 0, 255)
            bio = io.BytesIO()
            Image.fromarray(t.astype(np.uint8)).save(bio, format="JPEG", quality=75)
            files.append(bio.getvalue())
        nco = jpeg.coef_count(H, W)
        t0 = time.perf_counter()
        tmp, qt = np.zeros(nco, np.int16), np.zeros((3, 64), np.uint16)
        for d in files:
            jpeg.decode_coeffs(d, tmp, qt)
        host_ms = (time.perf_counter() - t0) / len(files) * 1e3
        b = jpeg.GpuDecodeBatch(a.n, H, W)      # default capacity: 2 bits per pixel
        t0 = time.perf_counter()
        for i in range(a.n):
            assert b.add(i, files[i % len(files)]) == 0
        prep_us = (time.perf_counter() - t0) / a.n * 1e6
        segs, sets, first = b.finish()
        used = int(segs["stream_off"][-1] + segs["stream_len"][-1])
        dev = torch.device("cuda")
        streams_d = torch.from_numpy(b.streams).to(dev)
        segs_d = torch.from_numpy(segs.view(np.uint8).reshape(-1, 32)).to(dev)
        sets_d = torch.from_numpy(sets).to(dev)
        coefs = [torch.zeros(a.n * nco, dtype=torch.int16, device=dev) for _ in range(a.streams)]
        sts = [torch.empty(a.n, dtype=torch.int32, device=dev) for _ in range(a.streams)]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):
            coefs[0].zero_()
            e0.record()
            engine.jpeg_huffman_decode(streams_d, segs_d, sets_d, coefs[0], sts[0])
            e1.record()
            torch.cuda.synchronize()
        one = e0.elapsed_time(e1)
        assert int(sts[0].abs().max()) == 0
        cs = [torch.cuda.Stream() for _ in range(a.streams)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for rep in range(reps):
            for k, st in enumerate(cs):
                with torch.cuda.stream(st):
                    coefs[k].zero_()
                    engine.jpeg_huffman_decode(streams_d, segs_d, sets_d, coefs[k], sts[k])
        torch.cuda.synchronize()
        many = (time.perf_counter() - t0) / (reps * a.streams) * 1e3
        print(f"{label:15s} {a.n} x {a.size}px ({np.mean([len(f) for f in files]) / 1024:.0f} KB per file, {used / a.n / 1024:.0f} KB uploaded per tile): "
              f"one launch {one:7.2f} ms = {a.n / one * 1e3:8.0f} images/s; {a.streams} launches in flight: {many:7.2f} ms per launch = {a.n / many * 1e3:8.0f} images/s; "
              f"host decoder {host_ms:.2f} ms per tile on one core; host preparation {prep_us:.0f} us per tile (Python + C, one thread)", flush=True)


if __name__ == "__main__":
    main()
