"""fp8 weight quantisation for the `fp8w` precision mode (BASELINE.json configs[3]: "yolov5m fp8 weights", reached through the same
invocation as every other mode, reference README.md:77).

Format: OCP e4m3fn (1-4-3, bias 7, max 448, no infinities; gfx950's fp8, NOT MI300X's fnuz) with one POWER-OF-TWO scale per output
channel (an e8m0 scale, as the MX formats use): w ~ 2^e[co] * q[co, ...], q in e4m3fn, 2^e the smallest power of two with
max|w[co]| / 2^e <= 448.  Because the scale is a power of two and e4m3 has 3 mantissa bits, every dequantised weight is EXACTLY
representable in bf16 (7 mantissa bits): the engine's bf16 MFMA kernels run the fp8-weight model bit for bit, with no scale multiply
in any epilogue and no rounding besides the fp8 one.  Rounding: to nearest, ties to even, subnormals kept (min subnormal 2^-9).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

E4M3_MAX = 448.0


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """fp32/fp64 array with |x| <= 448 -> nearest e4m3fn value (ties to even), returned as float32."""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    if a.size and a.max() > E4M3_MAX:
        raise ValueError("e4m3_round: magnitude above 448 (scale first)")
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.clip(e, -6, 8)                      # normals 2^-6 .. 2^8; below 2^-6 the grid is the subnormal one (spacing 2^-9)
    step = 2.0 ** (e - 3)                      # 3 mantissa bits
    q = np.rint(a / step) * step               # np.rint: ties to even; a mantissa overflow lands on the next binade's first value
    return np.copysign(q, x).astype(np.float32)


def e4m3_encode(v: np.ndarray) -> np.ndarray:
    """e4m3fn VALUES (outputs of e4m3_round) -> their uint8 bit patterns."""
    v = np.asarray(v, dtype=np.float64)
    a = np.abs(v)
    sign = (np.signbit(v)).astype(np.uint8) << 7
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    normal = a >= 2.0 ** -6
    man_n = np.rint(a / 2.0 ** (np.clip(e, -6, 8) - 3)).astype(np.int64) - 8
    exp_n = (np.clip(e, -6, 8) + 7).astype(np.int64)
    man_s = np.rint(a / 2.0 ** -9).astype(np.int64)
    code = np.where(normal, (exp_n << 3) | man_n, man_s).astype(np.uint8)
    return sign | code


def quantize_rows(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """w float32 [cout, ...] -> (dequantised float32 of the same shape, uint8 codes, int32 exponent e[cout] of the scale 2^e)."""
    w = np.asarray(w, dtype=np.float32)
    flat = w.reshape(w.shape[0], -1).astype(np.float64)
    amax = np.abs(flat).max(1)
    e = np.where(amax > 0, np.ceil(np.log2(np.maximum(amax, 1e-300) / E4M3_MAX)), 0).astype(np.int32)
    e = np.clip(e, -100, 100)
    scale = np.ldexp(1.0, e)[:, None]
    q = e4m3_round(flat / scale)
    deq = (q.astype(np.float64) * scale).astype(np.float32).reshape(w.shape)
    return deq, e4m3_encode(q).reshape(w.shape), e


def is_bf16_exact(x: np.ndarray) -> bool:
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return bool(np.all((u & 0xFFFF) == 0))
