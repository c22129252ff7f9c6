"""CPU restatement of the pixel half of a baseline JPEG decode -- TEST INFRASTRUCTURE, not the product path.

What [UPSTREAM detect.py LoadImages -> cv2.imread] gets from libjpeg(-turbo) after the entropy decoder: dequantisation, the
"islow" inverse DCT (jidctint.c: 13-bit fixed point, two passes), h2v2 "fancy" (triangle) chroma upsampling (jdsample.c) and the
fixed-point YCbCr -> RGB conversion (jdcolor.c), restated from the published IJG / libjpeg-turbo algorithm in numpy integer
arithmetic.  PINNED: tests/test_jpeg.py holds it to Pillow's own decoder (libjpeg-turbo 3.x, present in this image) on synthetic and
noise tiles, byte for byte -- so the HIP kernel (csrc/jpeg_idct.hip), which is held to this oracle, is held to the library the reference's
cv2.imread uses.  Only tests/ and tools/ import this module.
"""
from __future__ import annotations

import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(c, shift):
    """jidctint.c's butterfly on the last axis (8 values), int64 in, DESCALE by `shift` out."""
    c = c.astype(np.int64)
    z2, z3 = c[..., 2], c[..., 6]
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    z2, z3 = c[..., 0], c[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = c[..., 7], c[..., 5], c[..., 3], c[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F_1_175875602
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F_0_298631336, tmp1 * F_2_053119869, tmp2 * F_3_072711026, tmp3 * F_1_501321110
    z1, z2, z3, z4 = z1 * (-F_0_899976223), z2 * (-F_2_562915447), z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], -1)
    return _descale(out, shift)


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """coef int16 [..., 64] (natural order, quantised), qt [64] -> samples uint8 [..., 8, 8] (jpeg_idct_islow + range limit)."""
    blk = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(blk, -1, -2), CONST_BITS - PASS1_BITS)        # pass 1: columns
    ws = np.swapaxes(ws, -1, -2)
    px = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                           # pass 2: rows
    return np.clip(px + 128, 0, 255).astype(np.uint8)


def plane_from_blocks(blocks: np.ndarray) -> np.ndarray:
    """[bh][bw][8][8] -> [8 bh][8 bw]."""
    bh, bw = blocks.shape[:2]
    return blocks.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample_h2v2_fancy(c: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: [h][w] -> [2h][2w]; 3/4 nearer + 1/4 further in each direction, edges replicated, the two roundings
    (+8 and +7) of the library."""
    c = c.astype(np.int32)
    up = np.concatenate([c[:1], c[:-1]], 0)                                   # row above (replicated at the top)
    dn = np.concatenate([c[1:], c[-1:]], 0)
    out = np.empty((2 * c.shape[0], 2 * c.shape[1]), np.int32)
    for v, near in ((0, up), (1, dn)):
        s = 3 * c + near                                                      # "thiscolsum"
        last = np.concatenate([s[:, :1], s[:, :-1]], 1)
        nxt = np.concatenate([s[:, 1:], s[:, -1:]], 1)
        ev = (3 * s + last + 8) >> 4
        od = (3 * s + nxt + 7) >> 4
        ev[:, 0] = (4 * s[:, 0] + 8) >> 4
        od[:, -1] = (4 * s[:, -1] + 7) >> 4
        out[v::2, 0::2] = ev
        out[v::2, 1::2] = od
    return out.astype(np.uint8)


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert: SCALEBITS 16 tables, arithmetic right shifts, range limit."""
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode_from_coeffs(coef: np.ndarray, qt: np.ndarray, width: int, height: int, mcu_cols: int, mcu_rows: int) -> np.ndarray:
    """The aq_jpeg_decode_coeffs layout (4:2:0) -> RGB uint8 [height][width][3]."""
    ny = 4 * mcu_cols * mcu_rows
    nc = mcu_cols * mcu_rows
    yb = coef[:ny * 64].reshape(2 * mcu_rows, 2 * mcu_cols, 64)
    cbb = coef[ny * 64:(ny + nc) * 64].reshape(mcu_rows, mcu_cols, 64)
    crb = coef[(ny + nc) * 64:(ny + 2 * nc) * 64].reshape(mcu_rows, mcu_cols, 64)
    y = plane_from_blocks(idct_islow(yb, qt[0]))
    # the library upsamples the component's REAL extent (ceil(h / 2) x ceil(w / 2)) and replicates ITS edge, not the padded block's
    ch, cw = (height + 1) // 2, (width + 1) // 2
    cb = upsample_h2v2_fancy(plane_from_blocks(idct_islow(cbb, qt[1]))[:ch, :cw])
    cr = upsample_h2v2_fancy(plane_from_blocks(idct_islow(crb, qt[2]))[:ch, :cw])
    return ycc_to_rgb(y[:2 * ch, :2 * cw], cb, cr)[:height, :width]
