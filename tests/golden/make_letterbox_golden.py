#!/usr/bin/env python3
"""Generates tests/golden/g9_letterbox.json: digests and sample pixels of the C oracle's letterbox (oracle/ref_kernels.c
ref_letterbox_u8, a restatement of OpenCV's 8-bit INTER_LINEAR resize + border 114) on seeded inputs.  cv2 is not installed in this
image, so the fixture pins the oracle against regressions, not against OpenCV (UNPINNED, as DESIGN.md says).
Run from the repo root: python tests/golden/make_letterbox_golden.py"""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(1024, 1024), (500, 700), (1000, 600), (640, 640), (97, 33)]       # (h, w): the reference's real tile size first


def oracle_letterbox(lib, im, new=(640, 640), auto=True, scaleup=True, stride=32):
    h, w = im.shape[:2]
    hw = (C.c_int * 2)()
    lib.ref_letterbox_u8(im.ctypes.data_as(C.POINTER(C.c_uint8)), h, w, new[0], new[1], int(auto), int(scaleup), stride, None, hw)
    out = np.empty((hw[0], hw[1], 3), np.uint8)
    lib.ref_letterbox_u8(im.ctypes.data_as(C.POINTER(C.c_uint8)), h, w, new[0], new[1], int(auto), int(scaleup), stride,
                         out.ctypes.data_as(C.POINTER(C.c_uint8)), hw)
    return out


def seeded_image(h, w):
    rng = np.random.Generator(np.random.PCG64(0xA9CA9000 + h * 4099 + w))
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8, 1)))[:h, :w] * 0.7 + rng.integers(0, 77, (h, w, 3))      # blocks + noise: edges and texture
    return np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8))


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libref_kernels.so"))
    out = {"what": "sha256 and sample pixels of oracle ref_letterbox_u8 on seeded images (see this script)", "cases": []}
    for h, w in CASES:
        im = seeded_image(h, w)
        lb = oracle_letterbox(lib, im)
        ys = [0, lb.shape[0] // 3, lb.shape[0] // 2, lb.shape[0] - 1]
        xs = [0, lb.shape[1] // 3, lb.shape[1] // 2, lb.shape[1] - 1]
        out["cases"].append({"h": h, "w": w, "out_shape": list(lb.shape), "sha256": hashlib.sha256(lb.tobytes()).hexdigest(),
                             "input_sha256": hashlib.sha256(im.tobytes()).hexdigest(),
                             "samples": [[y, x, lb[y, x].tolist()] for y in ys for x in xs]})
    with open(os.path.join(ROOT, "tests", "golden", "g9_letterbox.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
