#!/usr/bin/env python3
"""Generator of the gfx950 assembly of the fused Bottleneck for hidden width C = 48 (bottleneck_asm_c48*; round 4).

    y = (x +) SiLU(cv2_3x3(SiLU(cv1_1x1(x) + b1)) + b2)      [UPSTREAM models/common.py Bottleneck.forward; reference README.md:77]

Same mathematics, tile (16 x 16 output pixels, 18 x 18 patch) and LDS pixel format (6 slots of 16 B, conflict-free for ds_read_b128's real
lane groups) as bottleneck_kernel<1, 3, 12, ...> in bottleneck.hip -- read its header first.  What is different, and why it is assembly:

  * A wave owns ALL THREE 16-row M blocks of its pixels (8 waves x 2 output rows), so one 1 KB B fragment read from LDS feeds three
    MFMAs; the HIP kernel's 12 waves own one M block each and read a fragment per MFMA -- the LDS, not the matrix pipe, paced it
    (profiles/r03_per_op_pmc.txt: 472-482 TFLOP/s, LDS bank-conflict 21.8 %, wait 39.6 %).  The 42 A fragments of the 3x3 (168 registers)
    stay in registers for the life of the workgroup; with 256 registers per wave that leaves 88 for everything else, which a compiler
    does not manage (round 1's "wide" shapes: it parked the weights in AccVGPRs and copied every fragment back before use).
  * K order of the 3x3 without per-k-step address registers: nine k-steps take channels 0-31 of one tap each (lane group g = channel
    block g: one address register, the tap is an immediate offset), five take channels 32-47 of a PAIR of taps (lane groups 0-1: the
    first tap, 2-3: the second): pairs (0,1), (3,4), (6,7) are one pixel apart (address register P1), pair (2,5) one patch row apart
    (P2), tap 8 pairs with zero weights.  14 k-steps for 13.5 of arithmetic, three address registers instead of fourteen.
  * The two waves of a SIMD run HALF A TILE OUT OF STEP (MI355X_MICROARCH.md, two waves per SIMD, item 9: stagger by wave number >= 4):
    in every barrier interval waves 0-3 run phase C of tile k - 1 (MFMA-bound) and then phase B of tile k (SiLU-bound), waves 4-7 the
    other way round, so that one wave's transcendental issue runs under its partner's MFMAs instead of all eight waves doing the same
    thing at the same time.  It needs two t patches and two x patches in LDS (4 x 32 KB) and ONE barrier per tile.
  * The x patch arrives by LDS-DMA through a buffer descriptor (out-of-range lanes write zeros: tools/ubench/lds_dma_buffer_oob.hip), one
    precomputed per-lane offset per instruction; pixels outside the image need no select at all -- whatever they read only feeds t at
    pixels outside the image, which phase B forces to zero (the 3x3 pads t, not x).  The shortcut is re-read from global memory
    (L2-hot) at the top of phase C, so x is dead after phase B.

Register map (VGPRs; no AGPRs -- 256 registers, two waves per SIMD):
  W2   168   A fragments of the 3x3: (k-step s, M block m) at W2 + 4 (3 s + m)
  ADDR  13   cA cP1 cP2 (phase C fragment addresses in the t patch), xb (phase B fragment address in the x patch), tw (t write address),
             bb (bias), w1a (W1 fragments in LDS), pre0-3 (LDS-DMA source offsets inside a patch), vin / vout (lane part of the shortcut / output offset)
  ACC   24   phase C: acc(j, m) = ACC + 4 (3 j + m) for output row j of the wave; phase B: two sets of 12
  F     16   phase C: ring of four B fragments; phase B: the block's two x fragments + eight temporaries
  P     28   phase C: 12 shortcut registers + temporaries; phase B: the six W1 fragments (24) + temporaries
Hazards the assembler does not pad (LLVM GCNHazardRecognizer, gfx940): MFMA result -> VALU read (s_nop 15), transcendental -> consumer
(independent instructions in between), s_mov m0 -> LDS-DMA (s_nop 0), VALU -> v_readfirstlane (s_nop 1), VALU-written SGPR -> VMEM (s_nop 4).

Usage: python gen_bottleneck_asm.py OUT.s   (aquaculture_amd/build.py assembles it and embeds the code object in bottleneck.hip)
"""
import sys

C = 48
CB = 6                       # 16-byte slots per pixel
PXB = 96                     # LDS pixel stride
PW = 18                      # patch width (and height)
PP = PW * PW
BUF = 32768                  # one patch buffer (341 pixels: 21 blocks of 16 = 336 are touched)
X0, T0 = 0, 2 * BUF          # x patches at 0 / 32768, t patches at 65536 / 98304
W1_OFF = 4 * BUF             # six 1 KB A fragments of the 1x1
BIAS_OFF = W1_OFF + 6 * 1024     # b1 | b2 as 96 floats
LDS_BYTES = BIAS_OFF + 512
NBLK1 = 21                   # 16-pixel blocks of the patch (linear pixel order)
KS2 = 14

# k-steps of the 3x3: ("A", tap) = channels 0-31 of the tap; ("P1" | "P2", tap_a, tap_b) = channels 32-47 of two taps (tap_b = -1: zeros)
KSTEPS = [("A", t) for t in range(9)] + [("P1", 0, 1), ("P1", 3, 4), ("P1", 6, 7), ("P2", 2, 5), ("P1", 8, -1)]
assert len(KSTEPS) == KS2

ARG = dict(inp=0, out=8, w=16, bias=24, debug=32, in_ld=40, out_ld=44, B=48, H=52, W=56, tiles_x=60, tpi=64, ntiles=68, shortcut=72, G=76,
           magic_tpi=80, magic_tx=84, in_bytes=88, pad=92)
ARG_BYTES = 96


class Regs:
    def __init__(self, prefix, limit):
        self.prefix, self.limit, self.next, self.names = prefix, limit, 0, {}

    def alloc(self, name, n=1, align=1):
        self.next = (self.next + align - 1) // align * align
        base = self.next
        self.next += n
        assert self.next <= self.limit, f"out of {self.prefix} registers at {name}"
        self.names[name] = (base, n)
        return base


V = Regs("v", 256)
S = Regs("s", 100)

S.alloc("karg", 2)
S.alloc("wg")
S.alloc("pad0")
for nm in ("inp", "out", "w", "bias", "debug"):
    S.alloc(nm, 2, 2)
for nm in ("in_ld", "out_ld", "B", "H", "W", "tiles_x", "tpi", "ntiles", "shortcut", "G", "magic_tpi", "magic_tx", "in_bytes", "pad"):
    S.alloc(nm)
S.alloc("srd", 4, 4)
for nm in ("wave", "group", "tmp0", "tmp1", "tmp2", "tmp3", "tmp4", "tmp5",
           "d_tile", "d_ok", "d_b", "d_y0", "d_x0",            # stage D: the tile whose x patch is being fetched
           "b_ok", "b_b", "b_y0", "b_x0",                      # stage B: phase B runs on it in this interval
           "c_ok", "c_b", "c_y0", "c_x0",                      # stage C: phase C
           "dma_lds", "org", "y0m1", "x0m1", "interior", "wlim", "rowok0", "rowok1", "blk2"):
    S.alloc(nm)
S.alloc("klog2e2", 2, 2)
S.alloc("kone2", 2, 2)
S.alloc("orow0", 2, 2)        # output address of the wave's row 0 / row 1 (pixel x0), 64-bit
S.alloc("orow1", 2, 2)
S.alloc("irow0", 2, 2)        # the same in the input (shortcut)
S.alloc("irow1", 2, 2)
S.alloc("colmask", 2, 2)
S.alloc("mask0", 2, 2)
S.alloc("mask1", 2, 2)
S.alloc("sa", 2, 2)
S.alloc("t64", 2, 2)
S.alloc("st_last", 2, 2)
S.alloc("st_acc", 12, 2)      # stamped build: cycle sums of six phases

V.alloc("tid")
V.alloc("l15")
V.alloc("W2", 4 * 3 * KS2, 4)
for nm in ("cA", "cP1", "cP2", "xb", "tw", "bb", "w1a", "pre0", "pre1", "pre2", "pre3", "vin", "vout"):
    V.alloc(nm)
V.alloc("ACC", 24, 4)
V.alloc("F", 16, 4)
V.alloc("P", 28, 4)


def s(name, i=0):
    b, n = S.names[name]
    assert i < n
    return f"s{b + i}"


def s2(name, i=0):
    b, n = S.names[name]
    assert i + 1 < n and (b + i) % 2 == 0
    return f"s[{b + i}:{b + i + 1}]"


def s4(name):
    b, n = S.names[name]
    assert n == 4 and b % 4 == 0
    return f"s[{b}:{b + 3}]"


def v(name, i=0):
    b, n = V.names[name]
    assert i < n, (name, i)
    return f"v{b + i}"


def vr(name, i, cnt):
    b, n = V.names[name]
    assert i + cnt <= n, (name, i, cnt)
    return f"v[{b + i}:{b + i + cnt - 1}]"


def w2(s_, m):
    return vr("W2", 4 * (3 * s_ + m), 4)


def acc(j, m):
    return vr("ACC", 4 * (3 * j + m), 4)


out = []
_uid = [0]
STAMPED = [False]
PH_PROLOGUE, PH_DMA, PH_B, PH_C_MFMA, PH_C_EPI, PH_BARRIER = range(6)


def E(line="", comment=None):
    out.append(("\t" + line if line and not line.endswith(":") else line) + (f"\t; {comment}" if comment else ""))


def label(name):
    out.append(f"{name}:")


def uid(prefix):
    _uid[0] += 1
    return f".L{prefix}_{_uid[0]}"


def stamp(k):
    if not STAMPED[0]:
        return
    E(f"s_memtime {s2('t64')}")
    E("s_waitcnt lgkmcnt(0)")
    E(f"s_sub_u32 {s('tmp4')}, {s('t64')}, {s('st_last')}")
    E(f"s_subb_u32 {s('tmp5')}, {s('t64', 1)}, {s('st_last', 1)}")
    E(f"s_add_u32 {s('st_acc', 2 * k)}, {s('st_acc', 2 * k)}, {s('tmp4')}")
    E(f"s_addc_u32 {s('st_acc', 2 * k + 1)}, {s('st_acc', 2 * k + 1)}, {s('tmp5')}")
    E(f"s_mov_b64 {s2('st_last')}, {s2('t64')}")


def tap_off(t):
    return ((t // 3) * PW + t % 3) * PXB


def emit_decode(stage):
    """stage D's tile number (d_tile) -> d_ok, d_b, d_y0, d_x0 (scalar; the magic multipliers come from the host)."""
    skip = uid("dec")
    E(f"s_cmp_lt_u32 {s('d_tile')}, {s('ntiles')}")
    E(f"s_cselect_b32 {s('d_ok')}, 1, 0")
    E(f"s_cbranch_scc0 {skip}")
    E(f"s_mul_hi_u32 {s('d_b')}, {s('d_tile')}, {s('magic_tpi')}", "image = tile / tiles per image")
    E(f"s_mul_i32 {s('tmp0')}, {s('d_b')}, {s('tpi')}")
    E(f"s_sub_u32 {s('tmp0')}, {s('d_tile')}, {s('tmp0')}", "tile inside the image")
    E(f"s_mul_hi_u32 {s('tmp1')}, {s('tmp0')}, {s('magic_tx')}", "tile row")
    E(f"s_mul_i32 {s('tmp2')}, {s('tmp1')}, {s('tiles_x')}")
    E(f"s_sub_u32 {s('tmp2')}, {s('tmp0')}, {s('tmp2')}", "tile column")
    E(f"s_lshl_b32 {s('d_y0')}, {s('tmp1')}, 4")
    E(f"s_lshl_b32 {s('d_x0')}, {s('tmp2')}, 4")
    label(skip)


def emit_dma():
    """Stage D's x patch -> LDS buffer dma_lds: this wave's four 1 KB instructions (slots 256 wave .. + 255 of the patch image)."""
    skip = uid("dma")
    E(f"s_cmp_eq_u32 {s('d_ok')}, 0")
    E(f"s_cbranch_scc1 {skip}")
    # byte offset of patch pixel (0, 0) = image pixel (y0 - 1, x0 - 1); may be "negative": wraps beyond the descriptor -> zeros
    E(f"s_mul_i32 {s('tmp0')}, {s('d_b')}, {s('H')}")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('d_y0')}")
    E(f"s_sub_u32 {s('tmp0')}, {s('tmp0')}, 1")
    E(f"s_mul_i32 {s('tmp0')}, {s('tmp0')}, {s('W')}")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('d_x0')}")
    E(f"s_sub_u32 {s('tmp0')}, {s('tmp0')}, 1")
    E(f"s_mul_i32 {s('org')}, {s('tmp0')}, {s('in_ld')}")
    E(f"s_lshl_b32 {s('tmp1')}, {s('wave')}, 12")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('dma_lds')}")
    T = V.names["F"][0] + 8
    for i in range(4):
        E(f"v_add_u32 v{T + i}, {s('org')}, {v('pre' + str(i))}")
    for i in range(4):
        E(f"s_add_u32 m0, {s('tmp1')}, {1024 * i}")
        E("s_nop 0", "hz: m0 write -> LDS-DMA")
        E(f"buffer_load_dwordx4 v{T + i}, {s4('srd')}, 0 offen lds")
    label(skip)


def emit_silu4(a0, y0):
    """In place on v[a0 : a0 + 4): a = a / (1 + 2^(-a log2 e)); temporaries v[y0 : y0 + 4).  (Same sequence as the planar kernels' epilogue.)"""
    E(f"v_pk_mul_f32 v[{y0}:{y0 + 1}], v[{a0}:{a0 + 1}], {s2('klog2e2')}")
    E(f"v_pk_mul_f32 v[{y0 + 2}:{y0 + 3}], v[{a0 + 2}:{a0 + 3}], {s2('klog2e2')}")
    for e in range(4):
        E(f"v_exp_f32 v{y0 + e}, v{y0 + e}")
    E(f"v_pk_add_f32 v[{y0}:{y0 + 1}], v[{y0}:{y0 + 1}], {s2('kone2')}")
    E(f"v_pk_add_f32 v[{y0 + 2}:{y0 + 3}], v[{y0 + 2}:{y0 + 3}], {s2('kone2')}")
    for e in range(4):
        E(f"v_rcp_f32 v{y0 + e}, v{y0 + e}")


def emit_phase_b():
    """t = SiLU(W1 x + b1) for this wave's blocks (wave, wave + 8, wave + 16 < 21) of stage B's patch, zero outside the image."""
    skip = uid("pb")
    E(f"s_cmp_eq_u32 {s('b_ok')}, 0")
    E(f"s_cbranch_scc1 {skip}")
    P0 = V.names["P"][0]
    F0 = V.names["F"][0]
    A0 = V.names["ACC"][0]
    # interior tile: the whole 18 x 18 patch lies inside the image (no zeroing)
    E(f"s_sub_u32 {s('y0m1')}, {s('b_y0')}, 1")
    E(f"s_sub_u32 {s('x0m1')}, {s('b_x0')}, 1")
    E(f"s_add_u32 {s('tmp0')}, {s('b_y0')}, 17")
    E(f"s_add_u32 {s('tmp1')}, {s('b_x0')}, 17")
    E(f"s_cmp_le_u32 {s('tmp0')}, {s('H')}")
    E(f"s_cselect_b32 {s('interior')}, 1, 0")
    E(f"s_cmp_le_u32 {s('tmp1')}, {s('W')}")
    E(f"s_cselect_b32 {s('tmp2')}, 1, 0")
    E(f"s_and_b32 {s('interior')}, {s('interior')}, {s('tmp2')}")
    E(f"s_cmp_gt_u32 {s('b_y0')}, 0")
    E(f"s_cselect_b32 {s('tmp2')}, 1, 0")
    E(f"s_and_b32 {s('interior')}, {s('interior')}, {s('tmp2')}")
    E(f"s_cmp_gt_u32 {s('b_x0')}, 0")
    E(f"s_cselect_b32 {s('tmp2')}, 1, 0")
    E(f"s_and_b32 {s('interior')}, {s('interior')}, {s('tmp2')}")
    # the six W1 fragments: LDS -> P[0 : 24)
    for i in range(6):
        E(f"ds_read_b128 v[{P0 + 4 * i}:{P0 + 4 * i + 3}], {v('w1a')} offset:{1024 * i}")

    def reads(i, aset):
        for ks in range(2):
            E(f"ds_read_b128 v[{F0 + 4 * ks}:{F0 + 4 * ks + 3}], {v('xb')} offset:{12288 * i + 64 * ks}")
        for m in range(3):
            E(f"ds_read_b128 v[{A0 + 12 * aset + 4 * m}:{A0 + 12 * aset + 4 * m + 3}], {v('bb')} offset:{64 * m}", "accumulators start from b1")

    def mfmas(aset):
        for ks in range(2):
            for m in range(3):
                a = A0 + 12 * aset + 4 * m
                E(f"v_mfma_f32_16x16x32_bf16 v[{a}:{a + 3}], v[{P0 + 4 * (3 * ks + m)}:{P0 + 4 * (3 * ks + m) + 3}], v[{F0 + 4 * ks}:{F0 + 4 * ks + 3}], v[{a}:{a + 3}]")

    def silu_store(i, aset):
        """accumulator set aset = block i: SiLU, bf16, zero outside the image, into the t patch."""
        Y = [F0 + 8, F0 + 12]
        nozero = uid("nz")
        # (px, pr, pc) of this lane's pixel and the "inside the image" mask -- border tiles only
        E(f"s_cmp_eq_u32 {s('interior')}, 1")
        E(f"s_cbranch_scc1 {nozero}")
        t0, t1, t2 = P0 + 24, P0 + 25, P0 + 26
        E(f"s_lshl_b32 {s('tmp0')}, {s('wave')}, 4")
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {128 * i}")
        E(f"v_add_u32 v{t0}, {s('tmp0')}, {v('l15')}", "patch pixel")
        E(f"v_mul_u32_u24 v{t1}, 3641, v{t0}")
        E(f"v_lshrrev_b32 v{t1}, 16, v{t1}", "patch row = pixel / 18")
        E(f"v_mul_u32_u24 v{t2}, 18, v{t1}")
        E(f"v_sub_u32 v{t2}, v{t0}, v{t2}", "patch column")
        E(f"v_add_u32 v{t1}, {s('y0m1')}, v{t1}", "image row (wraps below zero)")
        E(f"v_add_u32 v{t2}, {s('x0m1')}, v{t2}")
        E(f"v_cmp_gt_u32 {s2('sa')}, {s('H')}, v{t1}")
        E(f"v_cmp_gt_u32 vcc, {s('W')}, v{t2}")
        E(f"s_and_b64 vcc, vcc, {s2('sa')}")
        label(nozero)
        for m in range(3):
            a = A0 + 12 * aset + 4 * m
            y = Y[m % 2]
            emit_silu4(a, y)
            # (independent instructions between the rcp and its consumer: the next block's address arithmetic is elsewhere; pad)
            E("s_nop 0", "hz: transcendental -> consumer")
            E(f"v_pk_mul_f32 v[{a}:{a + 1}], v[{a}:{a + 1}], v[{y}:{y + 1}]")
            E(f"v_pk_mul_f32 v[{a + 2}:{a + 3}], v[{a + 2}:{a + 3}], v[{y + 2}:{y + 3}]")
            E(f"v_cvt_pk_bf16_f32 v{y}, v{a}, v{a + 1}")
            E(f"v_cvt_pk_bf16_f32 v{y + 1}, v{a + 2}, v{a + 3}")
            z = uid("z")
            E(f"s_cmp_eq_u32 {s('interior')}, 1")
            E(f"s_cbranch_scc1 {z}")
            E(f"v_cndmask_b32 v{y}, 0, v{y}, vcc")
            E(f"v_cndmask_b32 v{y + 1}, 0, v{y + 1}, vcc")
            label(z)
            E(f"ds_write_b64 {v('tw')}, v[{y}:{y + 1}] offset:{12288 * i + 32 * m}")

    # blocks 0 and 1 always exist for every wave (21 blocks, 8 waves); block 2 for waves 0-4
    reads(0, 0)
    E("s_waitcnt lgkmcnt(0)")
    mfmas(0)
    reads(1, 1)
    E("s_waitcnt lgkmcnt(0)")
    mfmas(1)
    E("s_nop 7", "hz: MFMA result -> VALU read (set 0: six MFMAs of set 1 and this pad behind it)")
    silu_store(0, 0)
    no2 = uid("nob2")
    E(f"s_cmp_eq_u32 {s('blk2')}, 0")
    E(f"s_cbranch_scc1 {no2}")
    reads(2, 0)
    E("s_waitcnt lgkmcnt(0)")
    mfmas(0)
    label(no2)
    E("s_nop 7", "hz: MFMA result -> VALU read (set 1)")
    silu_store(1, 1)
    end = uid("pbe")
    E(f"s_cmp_eq_u32 {s('blk2')}, 0")
    E(f"s_cbranch_scc1 {end}")
    E("s_nop 15", "hz: MFMA result -> VALU read (set 0, block 2)")
    silu_store(2, 0)
    label(end)
    label(skip)


def emit_phase_c(shortcut_flag_s):
    """y = (x +) SiLU(W2 (*) t + b2) for the wave's two output rows of stage C's tile."""
    skip = uid("pc")
    E(f"s_cmp_eq_u32 {s('c_ok')}, 0")
    E(f"s_cbranch_scc1 {skip}")
    P0 = V.names["P"][0]
    F0 = V.names["F"][0]
    # ---- scalar: row addresses and store masks ----
    E(f"s_lshl_b32 {s('tmp0')}, {s('wave')}, 1")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('c_y0')}", "image row of the wave's first output row")
    E(f"s_cmp_lt_u32 {s('tmp0')}, {s('H')}")
    E(f"s_cselect_b32 {s('rowok0')}, 1, 0")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp0')}, 1")
    E(f"s_cmp_lt_u32 {s('tmp1')}, {s('H')}")
    E(f"s_cselect_b32 {s('rowok1')}, 1, 0")
    E(f"s_mul_i32 {s('tmp1')}, {s('c_b')}, {s('H')}")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('tmp0')}")
    E(f"s_mul_i32 {s('tmp1')}, {s('tmp1')}, {s('W')}")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('c_x0')}", "pixel index of (row 0, x0)")
    for nm, ld, base in (("orow", "out_ld", "out"), ("irow", "in_ld", "inp")):
        E(f"s_mul_i32 {s('tmp2')}, {s('tmp1')}, {s(ld)}")
        E(f"s_mul_hi_u32 {s('tmp3')}, {s('tmp1')}, {s(ld)}")
        E(f"s_add_u32 {s(nm + '0')}, {s(base)}, {s('tmp2')}")
        E(f"s_addc_u32 {s(nm + '0', 1)}, {s(base, 1)}, {s('tmp3')}")
        E(f"s_mul_i32 {s('tmp2')}, {s('W')}, {s(ld)}")
        E(f"s_add_u32 {s(nm + '1')}, {s(nm + '0')}, {s('tmp2')}")
        E(f"s_addc_u32 {s(nm + '1', 1)}, {s(nm + '0', 1)}, 0")
    E(f"s_sub_u32 {s('wlim')}, {s('W')}, {s('c_x0')}", "columns of the tile inside the image")
    E(f"v_cmp_gt_u32 {s2('colmask')}, {s('wlim')}, {v('l15')}")
    E(f"s_cmp_eq_u32 {s('rowok0')}, 1")
    E(f"s_cselect_b64 {s2('mask0')}, {s2('colmask')}, 0")
    E(f"s_cmp_eq_u32 {s('rowok1')}, 1")
    E(f"s_cselect_b64 {s2('mask1')}, {s2('colmask')}, 0")
    # ---- shortcut values (L2-hot: the x patch of this tile was fetched two intervals ago), accumulators from b2, first fragments ----
    nosc = uid("nosc")
    E(f"s_cmp_eq_u32 {shortcut_flag_s}, 0")
    E(f"s_cbranch_scc1 {nosc}")
    for j in range(2):
        E(f"s_mov_b64 exec, {s2('mask' + str(j))}")
        for m in range(3):
            E(f"global_load_dwordx2 v[{P0 + 2 * (3 * j + m)}:{P0 + 2 * (3 * j + m) + 1}], {v('vin')}, {s2('irow' + str(j))} offset:{32 * m}")
    E("s_mov_b64 exec, -1")
    label(nosc)
    for j in range(2):
        for m in range(3):
            E(f"ds_read_b128 {acc(j, m)}, {v('bb')} offset:{192 + 64 * m}", "accumulators start from b2")

    def frag_reads(k):
        kind = KSTEPS[k]
        for j in range(2):
            dst = F0 + 4 * ((2 * k + j) % 4)
            if kind[0] == "A":
                E(f"ds_read_b128 v[{dst}:{dst + 3}], {v('cA')} offset:{tap_off(kind[1]) + j * PW * PXB}")
            else:
                E(f"ds_read_b128 v[{dst}:{dst + 3}], {v('c' + kind[0])} offset:{tap_off(kind[1]) + j * PW * PXB}")

    frag_reads(0)
    stamp(PH_DMA)
    for k in range(KS2):
        if k + 1 < KS2:
            frag_reads(k + 1)
            E("s_waitcnt lgkmcnt(2)")
        else:
            E("s_waitcnt lgkmcnt(0)")
        for m in range(3):
            for j in range(2):
                f = F0 + 4 * ((2 * k + j) % 4)
                E(f"v_mfma_f32_16x16x32_bf16 {acc(j, m)}, {w2(k, m)}, v[{f}:{f + 3}], {acc(j, m)}")
    stamp(PH_C_MFMA)
    E("s_nop 15", "hz: MFMA result -> VALU read")
    E("s_waitcnt vmcnt(0)", "shortcut values (and, vmcnt being in order, this interval's LDS-DMA and the previous tile's stores)")
    # ---- epilogue: SiLU, shortcut, bf16, store ----
    Y = [P0 + 12, P0 + 16]
    R = [P0 + 20, P0 + 24]
    for j in range(2):
        for m in range(3):
            n = 3 * j + m
            a = V.names["ACC"][0] + 4 * n
            y, r = Y[n % 2], R[n % 2]
            sc = P0 + 2 * n
            emit_silu4(a, y)
            plain = uid("pl")
            done = uid("dn")
            E(f"s_cmp_eq_u32 {shortcut_flag_s}, 0")
            E(f"s_cbranch_scc1 {plain}")
            E(f"v_lshlrev_b32 v{r}, 16, v{sc}")
            E(f"v_and_b32 v{r + 1}, 0xffff0000, v{sc}")
            E(f"v_lshlrev_b32 v{r + 2}, 16, v{sc + 1}")
            E(f"v_and_b32 v{r + 3}, 0xffff0000, v{sc + 1}")
            E(f"v_pk_mul_f32 v[{a}:{a + 1}], v[{a}:{a + 1}], v[{y}:{y + 1}]")
            E(f"v_pk_mul_f32 v[{a + 2}:{a + 3}], v[{a + 2}:{a + 3}], v[{y + 2}:{y + 3}]")
            E(f"v_pk_add_f32 v[{a}:{a + 1}], v[{a}:{a + 1}], v[{r}:{r + 1}]")
            E(f"v_pk_add_f32 v[{a + 2}:{a + 3}], v[{a + 2}:{a + 3}], v[{r + 2}:{r + 3}]")
            E(f"s_branch {done}")
            label(plain)
            E("s_nop 0", "hz: transcendental -> consumer")
            E(f"v_pk_mul_f32 v[{a}:{a + 1}], v[{a}:{a + 1}], v[{y}:{y + 1}]")
            E(f"v_pk_mul_f32 v[{a + 2}:{a + 3}], v[{a + 2}:{a + 3}], v[{y + 2}:{y + 3}]")
            label(done)
            E(f"v_cvt_pk_bf16_f32 v{y}, v{a}, v{a + 1}")
            E(f"v_cvt_pk_bf16_f32 v{y + 1}, v{a + 2}, v{a + 3}")
            E(f"s_mov_b64 exec, {s2('mask' + str(j))}")
            E(f"global_store_dwordx2 {v('vout')}, v[{y}:{y + 1}], {s2('orow' + str(j))} offset:{32 * m}")
            E("s_mov_b64 exec, -1")
    stamp(PH_C_EPI)
    label(skip)


def gen_kernel(name, stamped=False):
    global out
    out = []
    STAMPED[0] = stamped
    _uid[0] = 0 if not stamped else 100000
    E(f"; fused Bottleneck, C = 48, 8 waves: generated by gen_bottleneck_asm.py -- do not edit")
    label(name)
    a0 = S.names["inp"][0]
    assert a0 % 4 == 0 and S.names["in_ld"][0] == a0 + 10
    E(f"s_load_dwordx8 s[{a0}:{a0 + 7}], {s2('karg')}, 0x0", "inp, out, w, bias")
    E(f"s_load_dwordx2 {s2('debug')}, {s2('karg')}, 0x20")
    b0 = S.names["in_ld"][0]
    assert b0 % 2 == 0
    E(f"s_load_dwordx8 s[{b0}:{b0 + 7}], {s2('karg')}, 0x28", "in_ld .. ntiles")
    E(f"s_load_dwordx4 s[{b0 + 8}:{b0 + 11}], {s2('karg')}, 0x48", "shortcut, G, magic_tpi, magic_tx")
    E(f"s_load_dwordx2 s[{b0 + 12}:{b0 + 13}], {s2('karg')}, 0x58", "in_bytes, pad")
    T = [V.names["P"][0] + i for i in range(28)]
    E(f"v_and_b32 v{T[0]}, 63, {v('tid')}", "lane")
    E(f"v_lshrrev_b32 v{T[1]}, 6, {v('tid')}")
    E("s_nop 1", "hz: VALU write -> v_readfirstlane")
    E(f"v_readfirstlane_b32 {s('wave')}, v{T[1]}")
    E(f"v_and_b32 {v('l15')}, 15, v{T[0]}")
    E(f"v_lshrrev_b32 v{T[2]}, 4, v{T[0]}", "g")
    E(f"s_lshr_b32 {s('group')}, {s('wave')}, 2")
    E(f"s_cmp_lt_u32 {s('wave')}, 5")
    E(f"s_cselect_b32 {s('blk2')}, 1, 0", "waves 0-4 own a third block of the patch (21 blocks)")
    E("s_waitcnt lgkmcnt(0)")
    # buffer descriptor of the input slice: raw, num_records = its byte size (lanes beyond it -- and "negative" offsets -- read zeros)
    E(f"s_mov_b32 {s('srd', 0)}, {s('inp')}")
    E(f"s_and_b32 {s('srd', 1)}, {s('inp', 1)}, 0xffff")
    E(f"s_mov_b32 {s('srd', 2)}, {s('in_bytes')}")
    E(f"s_mov_b32 {s('srd', 3)}, 0x00020000")
    E(f"s_mov_b32 {s('klog2e2')}, 0xbfb8aa3b", "-log2(e)")
    E(f"s_mov_b32 {s('klog2e2', 1)}, 0xbfb8aa3b")
    E(f"s_mov_b32 {s('kone2')}, 1.0")
    E(f"s_mov_b32 {s('kone2', 1)}, 1.0")
    if stamped:
        for k in range(12):
            E(f"s_mov_b32 {s('st_acc', k)}, 0")
        E(f"s_memtime {s2('st_last')}")
        E("s_waitcnt lgkmcnt(0)")
    # ---- per-lane addresses ----
    # phase C: t patch 1 (the first phase C follows the first toggle), rows 2 wave, 2 wave + 1
    E(f"s_mul_i32 {s('tmp0')}, {s('wave')}, {2 * PW * PXB}")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {T0 + BUF}")
    E(f"v_mul_u32_u24 v{T[3]}, {PXB}, {v('l15')}")
    E(f"v_add_u32 v{T[3]}, {s('tmp0')}, v{T[3]}", "pixel (2 wave, l15) of t patch 1")
    E(f"v_lshl_add_u32 {v('cA')}, v{T[2]}, 4, v{T[3]}", "+ 16 g: channel block g of the tap")
    E(f"v_and_b32 v{T[4]}, 1, v{T[2]}", "g & 1")
    E(f"v_lshrrev_b32 v{T[5]}, 1, v{T[2]}", "g >> 1: which tap of the pair")
    E(f"v_lshl_add_u32 v{T[6]}, v{T[4]}, 4, v{T[3]}")
    E(f"v_add_u32 v{T[6]}, 64, v{T[6]}", "channel block 4 + (g & 1)")
    E(f"v_mul_u32_u24 v{T[7]}, {PXB}, v{T[5]}")
    E(f"v_add_u32 {v('cP1')}, v{T[6]}, v{T[7]}", "second tap: one pixel to the right")
    E(f"v_mul_u32_u24 v{T[7]}, {PW * PXB}, v{T[5]}")
    E(f"v_add_u32 {v('cP2')}, v{T[6]}, v{T[7]}", "second tap: one patch row down")
    # phase B: x patch 0 / t patch 0, block `wave`: pixel 16 wave + l15
    E(f"s_mul_i32 {s('tmp0')}, {s('wave')}, {16 * PXB}")
    E(f"v_mul_u32_u24 v{T[3]}, {PXB}, {v('l15')}")
    E(f"v_add_u32 v{T[3]}, {s('tmp0')}, v{T[3]}")
    E(f"v_lshl_add_u32 {v('xb')}, v{T[2]}, 4, v{T[3]}", "x patch 0 + 16 g")
    E(f"v_lshl_add_u32 {v('tw')}, v{T[2]}, 3, v{T[3]}")
    E(f"v_add_u32 {v('tw')}, {T0}, {v('tw')}", "t patch 0 + 8 g (four bf16 channels per lane and M block)")
    E(f"v_lshlrev_b32 {v('bb')}, 4, v{T[2]}")
    E(f"v_add_u32 {v('bb')}, {BIAS_OFF}, {v('bb')}")
    E(f"v_lshlrev_b32 {v('w1a')}, 4, v{T[0]}")
    E(f"v_add_u32 {v('w1a')}, {W1_OFF}, {v('w1a')}")
    # LDS-DMA: instruction i of this wave fills slots 64 (4 wave + i) + lane; slot -> (pixel, part) -> (patch row, column)
    for i in range(4):
        E(f"s_lshl_b32 {s('tmp0')}, {s('wave')}, 8")
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {64 * i}")
        E(f"v_add_u32 v{T[3]}, {s('tmp0')}, v{T[0]}", "slot")
        E(f"v_mul_u32_u24 v{T[4]}, 10923, v{T[3]}")
        E(f"v_lshrrev_b32 v{T[4]}, 16, v{T[4]}", "pixel = slot / 6")
        E(f"v_mul_u32_u24 v{T[5]}, 6, v{T[4]}")
        E(f"v_sub_u32 v{T[5]}, v{T[3]}, v{T[5]}", "part")
        E(f"v_mul_u32_u24 v{T[6]}, 3641, v{T[4]}")
        E(f"v_lshrrev_b32 v{T[6]}, 16, v{T[6]}", "patch row")
        E(f"v_mul_u32_u24 v{T[7]}, {PW}, v{T[6]}")
        E(f"v_sub_u32 v{T[7]}, v{T[4]}, v{T[7]}", "patch column")
        E(f"v_mul_lo_u32 v{T[6]}, v{T[6]}, {s('W')}")
        E(f"v_add_u32 v{T[6]}, v{T[6]}, v{T[7]}")
        E(f"v_mul_lo_u32 v{T[6]}, v{T[6]}, {s('in_ld')}")
        E(f"v_lshl_add_u32 v{T[6]}, v{T[5]}, 4, v{T[6]}", "(row W + column) in_ld + 16 part")
        E(f"v_mov_b32 v{T[8]}, 0x80000000")
        E(f"v_cmp_gt_u32 vcc, {PP}, v{T[4]}")
        E(f"v_cndmask_b32 {v('pre' + str(i))}, v{T[8]}, v{T[6]}, vcc", "slots beyond the patch: outside the descriptor -> zeros")
    # shortcut / output: lane part of the offset inside the wave's row
    E(f"v_mul_lo_u32 v{T[3]}, {v('l15')}, {s('in_ld')}")
    E(f"v_lshl_add_u32 {v('vin')}, v{T[2]}, 3, v{T[3]}")
    E(f"v_mul_lo_u32 v{T[3]}, {v('l15')}, {s('out_ld')}")
    E(f"v_lshl_add_u32 {v('vout')}, v{T[2]}, 3, v{T[3]}")
    # ---- first tile: XCD-aware bijective map (blocks sharing an XCD get consecutive tiles), as the planar kernels ----
    E(f"s_lshr_b32 {s('tmp0')}, {s('G')}, 3", "q")
    E(f"s_and_b32 {s('tmp1')}, {s('G')}, 7", "r")
    E(f"s_and_b32 {s('tmp2')}, {s('wg')}, 7", "xcd")
    E(f"s_add_u32 {s('tmp3')}, {s('tmp0')}, 1", "q + 1")
    E(f"s_cmp_lt_u32 {s('tmp2')}, {s('tmp1')}")
    E(f"s_cselect_b32 {s('d_tile')}, {s('tmp2')}, {s('tmp1')}", "min(xcd, r)")
    E(f"s_mul_i32 {s('d_tile')}, {s('d_tile')}, {s('tmp3')}")
    E(f"s_sub_u32 {s('tmp3')}, {s('tmp2')}, {s('tmp1')}")
    E(f"s_cselect_b32 {s('tmp3')}, 0, {s('tmp3')}", "max(xcd - r, 0)   (scc still: xcd < r)")
    E(f"s_mul_i32 {s('tmp3')}, {s('tmp3')}, {s('tmp0')}")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('tmp3')}")
    E(f"s_lshr_b32 {s('tmp3')}, {s('wg')}, 3")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('tmp3')}")
    E(f"s_cmp_ge_u32 {s('d_tile')}, {s('ntiles')}")
    E(f"s_cbranch_scc1 .Lend_{name}")
    # ---- weights: the 42 A fragments of the 3x3 -> registers; W1 (waves 0-5: one fragment each) and the biases (wave 6) -> LDS ----
    W1B = 6 * 1024
    E(f"v_lshlrev_b32 v{T[3]}, 4, v{T[0]}", "16 lane")
    E(f"s_add_u32 {s('sa')}, {s('w')}, {W1B}")
    E(f"s_addc_u32 {s('sa', 1)}, {s('w', 1)}, 0")
    for i in range(3 * KS2):
        if i and i % 4 == 0:
            E(f"s_add_u32 {s('sa')}, {s('sa')}, 4096")
            E(f"s_addc_u32 {s('sa', 1)}, {s('sa', 1)}, 0")
        E(f"global_load_dwordx4 {vr('W2', 4 * i, 4)}, v{T[3]}, {s2('sa')} offset:{1024 * (i % 4)}")
    nw1 = uid("nw1")
    E(f"s_cmp_gt_u32 {s('wave')}, 5")
    E(f"s_cbranch_scc1 {nw1}")
    E(f"s_lshl_b32 {s('tmp0')}, {s('wave')}, 10")
    E(f"v_add_u32 v{T[4]}, {s('tmp0')}, v{T[3]}")
    E(f"global_load_dwordx4 v[{T[8]}:{T[8] + 3}], v{T[4]}, {s2('w')}")
    E("s_waitcnt vmcnt(0)")
    E(f"v_add_u32 v{T[4]}, {W1_OFF}, v{T[4]}")
    E(f"ds_write_b128 v{T[4]}, v[{T[8]}:{T[8] + 3}]")
    label(nw1)
    nb = uid("nb")
    E(f"s_cmp_lg_u32 {s('wave')}, 6")
    E(f"s_cbranch_scc1 {nb}")
    E(f"v_cmp_gt_u32 vcc, 24, v{T[0]}", "96 floats = 24 lanes x 16 B")
    E("s_and_saveexec_b64 s[98:99], vcc")
    E(f"global_load_dwordx4 v[{T[8]}:{T[8] + 3}], v{T[3]}, {s2('bias')}")
    E("s_waitcnt vmcnt(0)")
    E(f"v_add_u32 v{T[4]}, {BIAS_OFF}, v{T[3]}")
    E(f"ds_write_b128 v{T[4]}, v[{T[8]}:{T[8] + 3}]")
    E("s_mov_b64 exec, s[98:99]")
    label(nb)
    # ---- pipeline fill: x patch of the first tile -> x buffer 0; second tile decoded ----
    E(f"s_mov_b32 {s('c_ok')}, 0")
    E(f"s_mov_b32 {s('b_ok')}, 0")
    emit_decode("d")
    E(f"s_mov_b32 {s('dma_lds')}, {X0}")
    emit_dma()
    for nm in ("ok", "b", "y0", "x0"):
        E(f"s_mov_b32 {s('b_' + nm)}, {s('d_' + nm)}")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('G')}")
    emit_decode("d")
    E(f"s_mov_b32 {s('dma_lds')}, {X0 + BUF}")
    E("s_waitcnt vmcnt(0)")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    stamp(PH_PROLOGUE)
    # ---- one barrier interval per tile: [DMA of tile k + 1] ; phase C of tile k - 1 and phase B of tile k, in the order of the wave's half ----
    loop = uid("loop")
    label(loop)
    emit_dma()
    g1 = uid("g1")
    join = uid("join")
    E(f"s_cmp_eq_u32 {s('group')}, 1")
    E(f"s_cbranch_scc1 {g1}")
    emit_phase_c(s("shortcut"))
    stamp(PH_C_EPI)
    emit_phase_b()
    stamp(PH_B)
    E(f"s_branch {join}")
    label(g1)
    emit_phase_b()
    stamp(PH_B)
    emit_phase_c(s("shortcut"))
    stamp(PH_C_EPI)
    label(join)
    # the x patch fetched in this interval must have landed before the barrier: phase C's vmcnt(0) came after its issue; without a phase C, wait here
    nowait = uid("nw")
    E(f"s_cmp_eq_u32 {s('c_ok')}, 1")
    E(f"s_cbranch_scc1 {nowait}")
    E("s_waitcnt vmcnt(0)")
    label(nowait)
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    stamp(PH_BARRIER)
    # rotate the stages, toggle the buffers
    for nm in ("ok", "b", "y0", "x0"):
        E(f"s_mov_b32 {s('c_' + nm)}, {s('b_' + nm)}")
        E(f"s_mov_b32 {s('b_' + nm)}, {s('d_' + nm)}")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('G')}")
    E(f"s_mov_b32 {s('d_ok')}, 0")
    more = uid("more")
    E(f"s_cmp_eq_u32 {s('b_ok')}, 0", "no tile in stage B: none follows either")
    E(f"s_cbranch_scc1 {more}")
    emit_decode("d")
    label(more)
    E(f"s_xor_b32 {s('dma_lds')}, {s('dma_lds')}, {BUF}")
    for nm in ("cA", "cP1", "cP2", "xb", "tw"):
        E(f"v_xor_b32 {v(nm)}, {BUF}, {v(nm)}")
    E(f"s_or_b32 {s('tmp0')}, {s('c_ok')}, {s('b_ok')}")
    E(f"s_cmp_lg_u32 {s('tmp0')}, 0")
    E(f"s_cbranch_scc1 {loop}")
    label(f".Lend_{name}")
    if stamped:
        # per wave: six phase sums -> debug[(wg * 8 + wave) * 8 + k]
        T3 = V.names["P"][0]
        E(f"s_lshl_b32 {s('tmp0')}, {s('wg')}, 3")
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('wave')}")
        E(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 6")
        E(f"v_mov_b32 v{T3 + 2}, {s('tmp0')}")
        E("s_mov_b64 exec, 1")
        for k in range(6):
            E(f"v_mov_b32 v{T3}, {s('st_acc', 2 * k)}")
            E(f"v_mov_b32 v{T3 + 1}, {s('st_acc', 2 * k + 1)}")
            E(f"global_store_dwordx2 v{T3 + 2}, v[{T3}:{T3 + 1}], {s2('debug')} offset:{8 * k}")
        E("s_waitcnt vmcnt(0)")
    E("s_endpgm")
    return list(out)


def descriptor(name):
    total = (V.next + 7) // 8 * 8
    assert total <= 256 and LDS_BYTES <= 160 * 1024
    return f"""
	.rodata
	.p2align 6
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size {LDS_BYTES}
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size {ARG_BYTES}
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr {total}
		.amdhsa_next_free_sgpr 100
		.amdhsa_accum_offset {total}
		.amdhsa_reserve_vcc 1
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text
"""


def metadata_entry(name):
    total = (V.next + 7) // 8 * 8
    return f"""  - .agpr_count:     0
    .args:
      - .offset:         0
        .size:           {ARG_BYTES}
        .value_kind:     by_value
    .group_segment_fixed_size: {LDS_BYTES}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {ARG_BYTES}
    .max_flat_workgroup_size: 512
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     106
    .symbol:         {name}.kd
    .vgpr_count:     {total}
    .wavefront_size: 64
"""


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "bottleneck_asm.s"
    text = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.text"]
    entries = []
    for name, stamped in (("bottleneck_asm_c48", False), ("bottleneck_asm_c48_stamped", True)):
        text += [f"\t.globl\t{name}", "\t.p2align\t8", f"\t.type\t{name},@function"]
        text += gen_kernel(name, stamped)
        entries.append(metadata_entry(name))
        text += [f".Lfend_{name}:", f"\t.size\t{name}, .Lfend_{name}-{name}", descriptor(name)]
    text.append(f"""	.amdgpu_metadata
---
amdhsa.kernels:
{"".join(entries)}amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
""")
    with open(path, "w") as f:
        f.write("\n".join(text) + "\n")
    print(f"bottleneck_asm_c48: {V.next} VGPRs, {S.next} SGPRs, {LDS_BYTES} B LDS; wrote {path}: {sum(1 for l in text if 'v_mfma' in l)} MFMA instructions")


if __name__ == "__main__":
    main()
