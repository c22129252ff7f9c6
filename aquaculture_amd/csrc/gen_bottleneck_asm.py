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

ARG = dict(inp=0, out=8, w=16, bias=24, in_ld=32, out_ld=36, B=40, H=44, W=48, tiles_x=52, tpi=56, ntiles=60, shortcut=64, G=68,
           magic_tpi=72, magic_tx=76, in_bytes=80, pad=84, debug=88)
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


import os as _os
# W16 (experiment, AQ_GEN_BTL_W16=1; parity-green): 16-byte output stores as in the planar 3x3 families (three store instructions per tile and wave
# instead of six).  MEASURED, no gain: HBM-cold 169.8-176.9 / 154.3-155.4 us (shortcut / none) -> 173.0-177.0 / 151.2-152.1; in the engine model.2.m.0 /
# m.1 127.9 / 122.8 -> 128.9 / 123.3 us (three interleaved bench runs each): with two waves per SIMD the partner's work already covers a wave's store issue.
# Off: the shipped kernel keeps its 8-byte stores.  (First build wrong in row 0's channels 2, 3 of every second quad: gfx940+ needs TWO wait states
# between a store of more than 8 bytes and a VALU write of its data registers -- row 1's SiLU starts in them; hence the s_nop.)
W16 = _os.environ.get("AQ_GEN_BTL_W16", "0") == "1"
V = Regs("v", 256)
S = Regs("s", 100)

S.alloc("karg", 2)
S.alloc("wg")
S.alloc("pad0")
for nm in ("inp", "out", "w", "bias"):
    S.alloc(nm, 2, 2)
for nm in ("in_ld", "out_ld", "B", "H", "W", "tiles_x", "tpi", "ntiles", "shortcut", "G", "magic_tpi", "magic_tx", "in_bytes", "pad"):
    S.alloc(nm)
S.alloc("debug", 2, 2)
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
OPT = dict(stagger=True, nosilu=False, nomfma=False, nopk=False, nodma=False, nold=False, nost=False, prio=0, ntst=False, ntld=False)      # experiment switches of a kernel variant (main() sets them per kernel)
PH_PROLOGUE, PH_DMA, PH_B, PH_C_MFMA, PH_C_EPI, PH_BARRIER = range(6)      # (PH_DMA: the vmcnt wait in front of the epilogue)


def E(line="", comment=None):
    if OPT["nomfma"] and line.startswith("v_mfma"):
        return
    if (OPT["nodma"] and line.startswith("buffer_load_dwordx4")) or (OPT["nold"] and line.startswith("global_load_dwordx2")) or (OPT["nost"] and line.startswith("global_store_dwordx")):
        return
    if OPT["nosilu"] and line.split(" ")[0] in ("v_exp_f32", "v_rcp_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_add_f32"):
        return
    if OPT["nopk"] and line.split(" ")[0] in ("v_pk_mul_f32", "v_pk_add_f32"):
        # experiment: the packed fp32 forms as two plain VOP2 / VOP3 instructions (tools/ubench/valu_issue.hip: beside another wave's MFMAs a
        # v_pk_mul_f32 takes 20.7 cycles, a v_mul_f32 8.3)
        op = "v_mul_f32" if "mul" in line else "v_add_f32"
        import re
        m_ = re.match(r"v_pk_\w+ v\[(\d+):\d+\], v\[(\d+):\d+\], (.*)$", line)
        d, a, rest = int(m_.group(1)), int(m_.group(2)), m_.group(3)
        for h in range(2):
            r2 = re.match(r"v\[(\d+):\d+\]$", rest)
            if r2:
                b_ = f"v{int(r2.group(1)) + h}"
            else:
                r3 = re.match(r"s\[(\d+):\d+\]$", rest)
                b_ = f"s{int(r3.group(1)) + h}"
            out.append(f"\t{op}_e64 v{d + h}, v{a + h}, {b_}")
        return
    if (OPT["ntst"] and line.startswith("global_store_dwordx")) or (OPT["ntld"] and line.startswith("global_load_dwordx2")):
        line += " nt"                      # experiment: streaming hint on the output stores / the shortcut loads (neither is read again by this kernel)
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


def emit_dma(temps_base=None):
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
    T = V.names["F"][0] + 8 if temps_base is None else temps_base
    for i in range(4):
        E(f"v_add_u32 v{T + i}, {s('org')}, {v('pre' + str(i))}")
    for i in range(4):
        E(f"s_add_u32 m0, {s('tmp1')}, {1024 * i}")
        E("s_nop 0", "hz: m0 write -> LDS-DMA")
        E(f"buffer_load_dwordx4 v{T + i}, {s4('srd')}, 0 offen lds")
    label(skip)


def emit_silu(regs, temps):
    """SiLU in place on the accumulator registers `regs` (a multiple of 4 numbers, each group of 4 consecutive), one temporary per
    register: a = a / (1 + 2^(-a log2 e)), evaluated STAGE BY STAGE over all of them -- every instruction's operands were produced at least
    len(regs) / 2 instructions earlier, so neither the transcendental unit's latency nor a dependent VALU issue stalls the wave (round 4,
    first build: four registers at a time, 49 % of the wave cycles in SQ_WAIT_INST_ANY).  Leaves the reciprocals in `temps`; the caller
    multiplies (it may have independent work to put in front)."""
    n = len(regs)
    assert n % 4 == 0 and len(temps) >= n
    for i in range(0, n, 2):
        assert regs[i + 1] == regs[i] + 1 and temps[i + 1] == temps[i] + 1 and regs[i] % 2 == 0 and temps[i] % 2 == 0
        E(f"v_pk_mul_f32 v[{temps[i]}:{temps[i] + 1}], v[{regs[i]}:{regs[i] + 1}], {s2('klog2e2')}")
    for i in range(n):
        E(f"v_exp_f32 v{temps[i]}, v{temps[i]}")
    for i in range(0, n, 2):
        E(f"v_pk_add_f32 v[{temps[i]}:{temps[i] + 1}], v[{temps[i]}:{temps[i] + 1}], {s2('kone2')}")
    for i in range(n):
        E(f"v_rcp_f32 v{temps[i]}, v{temps[i]}")


def emit_phase_b(interior):
    """t = SiLU(W1 x + b1) for this wave's blocks (wave, wave + 8, wave + 16 < 21) of stage B's patch, zero outside the image.
    Two copies behind one branch: `interior` tiles (the whole 18 x 18 patch inside the image) need no zeroing at all."""
    P0 = V.names["P"][0]
    F0 = V.names["F"][0]
    A0 = V.names["ACC"][0]
    if not interior:
        E(f"s_sub_u32 {s('y0m1')}, {s('b_y0')}, 1")
        E(f"s_sub_u32 {s('x0m1')}, {s('b_x0')}, 1")
    # the six W1 fragments: LDS -> P[0 : 24)
    for i in range(6):
        E(f"ds_read_b128 v[{P0 + 4 * i}:{P0 + 4 * i + 3}], {v('w1a')} offset:{1024 * i}")
    temps = [F0 + 8 + i for i in range(8)] + [P0 + 24 + i for i in range(4)]

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
        regs = [A0 + 12 * aset + k for k in range(12)]
        if not interior:
            # (patch pixel, row, column) of this lane's pixel and the "inside the image" mask, before the temporaries are taken
            t0, t1, t2 = temps[8], temps[9], temps[10]
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
        emit_silu(regs, temps)
        for k in range(0, 12, 2):
            E(f"v_pk_mul_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
        for k in range(6):
            E(f"v_cvt_pk_bf16_f32 v{temps[k]}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
        if not interior:
            for k in range(6):
                E(f"v_cndmask_b32 v{temps[k]}, 0, v{temps[k]}, vcc")
        for m in range(3):
            E(f"ds_write_b64 {v('tw')}, v[{temps[2 * m]}:{temps[2 * m] + 1}] offset:{12288 * i + 32 * m}")

    # blocks 0 and 1 exist for every wave (21 blocks, 8 waves); block 2 for waves 0-4.  The next block's fragment reads are in flight
    # under the current block's SiLU (the block's own MFMAs have been issued, so its two fragment registers are free again).
    reads(0, 0)
    E("s_waitcnt lgkmcnt(0)")
    mfmas(0)
    reads(1, 1)
    E("s_nop 7", "hz: MFMA result -> VALU read")
    silu_store(0, 0)
    E("s_waitcnt lgkmcnt(3)", "block 1's fragments and b1 (the three t writes behind them may still be on their way)")
    mfmas(1)
    no2 = uid("nob2")
    end = uid("pbe")
    E(f"s_cmp_eq_u32 {s('blk2')}, 0")
    E(f"s_cbranch_scc1 {no2}")
    reads(2, 0)
    E("s_nop 7", "hz: MFMA result -> VALU read")
    silu_store(1, 1)
    E("s_waitcnt lgkmcnt(3)")
    mfmas(0)
    E("s_nop 15", "hz: MFMA result -> VALU read")
    silu_store(2, 0)
    E(f"s_branch {end}")
    label(no2)
    E("s_nop 15", "hz: MFMA result -> VALU read")
    silu_store(1, 1)
    label(end)


def emit_phase_b_dispatch():
    skip = uid("pb")
    border = uid("pbb")
    E(f"s_cmp_eq_u32 {s('b_ok')}, 0")
    E(f"s_cbranch_scc1 {skip}")
    # interior tile: the whole 18 x 18 patch lies inside the image
    E(f"s_add_u32 {s('tmp0')}, {s('b_y0')}, 17")
    E(f"s_add_u32 {s('tmp1')}, {s('b_x0')}, 17")
    E(f"s_cmp_le_u32 {s('tmp0')}, {s('H')}")
    E(f"s_cselect_b32 {s('interior')}, 1, 0")
    E(f"s_cmp_le_u32 {s('tmp1')}, {s('W')}")
    E(f"s_cselect_b32 {s('tmp2')}, 1, 0")
    E(f"s_and_b32 {s('interior')}, {s('interior')}, {s('tmp2')}")
    E(f"s_min_u32 {s('tmp2')}, {s('b_y0')}, {s('b_x0')}")
    E(f"s_min_u32 {s('tmp2')}, {s('tmp2')}, 1", "1 when both origins are > 0")
    E(f"s_and_b32 {s('interior')}, {s('interior')}, {s('tmp2')}")
    E(f"s_cmp_eq_u32 {s('interior')}, 0")
    E(f"s_cbranch_scc1 {border}")
    emit_phase_b(True)
    E(f"s_branch {skip}")
    label(border)
    emit_phase_b(False)
    label(skip)


def emit_phase_c(dma_inside):
    """y = (x +) SiLU(W2 (*) t + b2) for the wave's two output rows of stage C's tile.  dma_inside (waves 0-3, whose interval starts with
    this phase): the next x patch's LDS-DMA is issued BEHIND the shortcut loads, so that the wait in front of the epilogue covers the
    shortcut values only (vmcnt is in order) and the patch has the whole interval to arrive."""
    skip = uid("pc")
    if dma_inside:
        have = uid("pch")
        E(f"s_cmp_eq_u32 {s('c_ok')}, 1")
        E(f"s_cbranch_scc1 {have}")
        emit_dma()
        E(f"s_branch {skip}")
        label(have)
    else:
        E(f"s_cmp_eq_u32 {s('c_ok')}, 0")
        E(f"s_cbranch_scc1 {skip}")
    P0 = V.names["P"][0]
    F0 = V.names["F"][0]
    A0 = V.names["ACC"][0]
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
    E(f"s_cmp_eq_u32 {s('shortcut')}, 0")
    E(f"s_cbranch_scc1 {nosc}")
    for j in range(2):
        E(f"s_mov_b64 exec, {s2('mask' + str(j))}")
        for m in range(3):
            E(f"global_load_dwordx2 v[{P0 + 2 * (3 * j + m)}:{P0 + 2 * (3 * j + m) + 1}], {v('vin')}, {s2('irow' + str(j))} offset:{32 * m}")
    E("s_mov_b64 exec, -1")
    label(nosc)
    if dma_inside:
        emit_dma(temps_base=P0 + 12)
    for j in range(2):
        for m in range(3):
            E(f"ds_read_b128 {acc(j, m)}, {v('bb')} offset:{192 + 64 * m}", "accumulators start from b2")

    # B fragments: ring of eight (F and the sixteen temporaries of the epilogue, idle during the stream): read three k-steps ahead
    ring = [F0 + 4 * i for i in range(4)] + [P0 + 12 + 4 * i for i in range(4)]
    PD = 3

    def frag_reads(k):
        kind = KSTEPS[k]
        for j in range(2):
            dst = ring[(2 * k + j) % 8]
            areg = "cA" if kind[0] == "A" else "c" + kind[0]
            E(f"ds_read_b128 v[{dst}:{dst + 3}], {v(areg)} offset:{tap_off(kind[1]) + j * PW * PXB}")

    for k in range(PD):
        frag_reads(k)
    if OPT["prio"]:
        # the wave in its MFMA stream outranks its partner's SiLU stream (issue is arbitrated by priority, then age: without this the OLDER
        # wave's dense VALU stream starves the younger wave's MFMAs -- stamped build: waves 4-7 spent 5.6 k cycles per tile in a stream that
        # takes 1.5 k alone -- whereas a VALU stream loses little beside MFMAs: one transcendental or two plain instructions fit every gap)
        E(f"s_setprio {OPT['prio']}")
    for k in range(KS2):
        if k + PD < KS2:
            frag_reads(k + PD)
        ahead = 2 * (min(KS2, k + PD + 1) - (k + 1))         # fragment reads issued after this k-step's
        E(f"s_waitcnt lgkmcnt({ahead})")
        for m in range(3):
            for j in range(2):
                f = ring[(2 * k + j) % 8]
                E(f"v_mfma_f32_16x16x32_bf16 {acc(j, m)}, {w2(k, m)}, v[{f}:{f + 3}], {acc(j, m)}")
    if OPT["prio"]:
        E("s_setprio 0")
    stamp(PH_C_MFMA)
    E("s_nop 15", "hz: MFMA result -> VALU read")
    if dma_inside:
        w0, wd = uid("w0"), uid("wd")
        E(f"s_cmp_eq_u32 {s('d_ok')}, 0")
        E(f"s_cbranch_scc1 {w0}")
        E("s_waitcnt vmcnt(4)", "shortcut values (the four LDS-DMA instructions behind them stay in flight)")
        E(f"s_branch {wd}")
        label(w0)
        E("s_waitcnt vmcnt(0)")
        label(wd)
    else:
        E("s_waitcnt vmcnt(0)", "shortcut values (and, vmcnt being in order, this interval's LDS-DMA and the previous tile's stores)")
    stamp(PH_DMA)
    # ---- epilogue, one output row (12 accumulator registers) at a time: SiLU, shortcut, bf16, store.  Two copies behind ONE branch. ----
    temps = [P0 + 12 + i for i in range(12)]
    plain, done = uid("pl"), uid("dn")
    if W16:
        E(f"s_mov_b32 {s('sa')}, 0xffff0000", "lanes with odd q")
        E(f"s_mov_b32 {s('sa', 1)}, 0xffff0000")
        E(f"s_mul_i32 {s('tmp4')}, {s('W')}, {s('out_ld')}")
        E(f"s_add_u32 {s('tmp4')}, {s('tmp4')}, {64 - 8 - 64}", "row 1's block 2 from row 0's address: + one image row - 8 (the store's own offset is 64)")
    E(f"s_cmp_eq_u32 {s('shortcut')}, 0")
    E(f"s_cbranch_scc1 {plain}")
    for with_sc in (True, False):
        if not with_sc:
            label(plain)
        for j in range(2):
            regs = [A0 + 12 * j + k for k in range(12)]
            emit_silu(regs, temps)
            for k in range(0, 12, 2):
                E(f"v_pk_mul_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
            if with_sc:
                for m in range(3):
                    sc = P0 + 2 * (3 * j + m)
                    E(f"v_lshlrev_b32 v{temps[4 * m]}, 16, v{sc}")
                    E(f"v_and_b32 v{temps[4 * m + 1]}, 0xffff0000, v{sc}")
                    E(f"v_lshlrev_b32 v{temps[4 * m + 2]}, 16, v{sc + 1}")
                    E(f"v_and_b32 v{temps[4 * m + 3]}, 0xffff0000, v{sc + 1}")
                for k in range(0, 12, 2):
                    E(f"v_pk_add_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
            if W16:
                # 16-byte stores (gen_conv3x3_pl_asm.py W16): v_permlane16_swap pairs the 8-byte pieces of lanes q and q + 1 of a
                # pixel; M blocks 0 and 1 of a row leave in one store (even-q lanes: block 0's 16 bytes, odd-q lanes: block 1's, 24 bytes beyond
                # their own offset), M block 2 of rows 0 and 1 in one more -- three store instructions per tile instead of six.  Row 0's block 2
                # waits in its own (dead) accumulator registers A0, A0 + 1; row 1's goes to A0 + 2, A0 + 3.
                for k in range(4):
                    E(f"v_cvt_pk_bf16_f32 v{temps[k]}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
                for k in (4, 5):
                    E(f"v_cvt_pk_bf16_f32 v{A0 + 2 * j + k - 4}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
                E(f"v_cndmask_b32 v{temps[6]}, 0, 24, {s2('sa')}")
                E(f"v_add_u32 v{temps[6]}, v{temps[6]}, {v('vout')}")
                E(f"v_permlane16_swap_b32 v{temps[0]}, v{temps[2]}")
                E(f"v_permlane16_swap_b32 v{temps[1]}, v{temps[3]}")
                E(f"s_mov_b64 exec, {s2('mask' + str(j))}")
                E(f"global_store_dwordx4 v{temps[6]}, v[{temps[0]}:{temps[3]}], {s2('orow' + str(j))}")
                E("s_mov_b64 exec, -1")
                if j == 0:
                    E("s_nop 0", "hz: store of more than 8 bytes -> VALU write of its data registers, 2 wait states on gfx940+ (row 1's SiLU starts in them)")
                if j == 1:
                    # even-q lanes: row 0's block 2 at their own offset + 64; odd-q lanes: row 1's, one image row further and 8 bytes back
                    E(f"v_mov_b32 v{temps[7]}, {s('tmp4')}")
                    E(f"v_cndmask_b32 v{temps[7]}, 0, v{temps[7]}, {s2('sa')}")
                    E(f"v_add_u32 v{temps[7]}, v{temps[7]}, {v('vout')}")
                    E(f"v_permlane16_swap_b32 v{A0}, v{A0 + 2}")
                    E(f"v_permlane16_swap_b32 v{A0 + 1}, v{A0 + 3}")
                    E(f"s_andn2_b64 {s2('t64')}, {s2('mask0')}, {s2('sa')}")
                    E(f"s_and_b64 {s2('colmask')}, {s2('mask1')}, {s2('sa')}")
                    E(f"s_or_b64 exec, {s2('t64')}, {s2('colmask')}")
                    E(f"global_store_dwordx4 v{temps[7]}, v[{A0}:{A0 + 3}], {s2('orow0')} offset:64")
                    E("s_mov_b64 exec, -1")
                continue
            for k in range(6):
                E(f"v_cvt_pk_bf16_f32 v{temps[k]}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
            E(f"s_mov_b64 exec, {s2('mask' + str(j))}")
            for m in range(3):
                E(f"global_store_dwordx2 {v('vout')}, v[{temps[2 * m]}:{temps[2 * m] + 1}], {s2('orow' + str(j))} offset:{32 * m}")
            E("s_mov_b64 exec, -1")
        if with_sc:
            E(f"s_branch {done}")
    label(done)
    stamp(PH_C_EPI)
    label(skip)


_kernel_no = [0]


def gen_kernel(name, stamped=False, **opt):
    global out
    out = []
    STAMPED[0] = stamped
    OPT.update(dict(stagger=True, nosilu=False, nomfma=False, nopk=False, nodma=False, nold=False, nost=False, prio=0, ntst=False, ntld=False))
    OPT.update(opt)
    _kernel_no[0] += 1
    _uid[0] = 100000 * _kernel_no[0]
    E(f"; fused Bottleneck, C = 48, 8 waves: generated by gen_bottleneck_asm.py -- do not edit")
    label(name)
    a0 = S.names["inp"][0]
    b0 = S.names["in_ld"][0]
    assert a0 % 4 == 0 and b0 == a0 + 8 and S.names["debug"][0] == b0 + 14
    E(f"s_load_dwordx8 s[{a0}:{a0 + 7}], {s2('karg')}, 0x0", "inp, out, w, bias")
    E(f"s_load_dwordx8 s[{b0}:{b0 + 7}], {s2('karg')}, 0x20", "in_ld .. ntiles")
    E(f"s_load_dwordx4 s[{b0 + 8}:{b0 + 11}], {s2('karg')}, 0x40", "shortcut, G, magic_tpi, magic_tx")
    E(f"s_load_dwordx4 s[{b0 + 12}:{b0 + 15}], {s2('karg')}, 0x50", "in_bytes, pad, debug")
    T = [V.names["P"][0] + i for i in range(28)]
    E(f"v_and_b32 v{T[0]}, 63, {v('tid')}", "lane")
    E(f"v_lshrrev_b32 v{T[1]}, 6, {v('tid')}")
    E("s_nop 1", "hz: VALU write -> v_readfirstlane")
    E(f"v_readfirstlane_b32 {s('wave')}, v{T[1]}")
    E(f"v_and_b32 {v('l15')}, 15, v{T[0]}")
    E(f"v_lshrrev_b32 v{T[2]}, 4, v{T[0]}", "g")
    if OPT["stagger"]:
        E(f"s_lshr_b32 {s('group')}, {s('wave')}, 2")
    else:
        E(f"s_mov_b32 {s('group')}, 0", "experiment: every wave runs phase C first (no stagger)")
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
    g1 = uid("g1")
    join = uid("join")
    E(f"s_cmp_eq_u32 {s('group')}, 1")
    E(f"s_cbranch_scc1 {g1}")
    emit_phase_c(True)
    stamp(PH_C_EPI)
    emit_phase_b_dispatch()
    stamp(PH_B)
    # the x patch fetched in this interval must have landed before the barrier.  Its four instructions are older than the six output
    # stores of a full tile, which may stay in flight; ragged tiles (stores with an empty EXEC) and intervals without a phase C: everything
    full, w0 = uid("full"), uid("w0")
    E(f"s_and_b32 {s('tmp0')}, {s('rowok0')}, {s('rowok1')}")
    E(f"s_and_b32 {s('tmp0')}, {s('tmp0')}, {s('c_ok')}")
    E(f"s_cmp_eq_u32 {s('tmp0')}, 1")
    E(f"s_cbranch_scc1 {full}")
    E("s_waitcnt vmcnt(0)")
    E(f"s_branch {join}")
    label(full)
    E(f"s_waitcnt vmcnt({3 if W16 else 6})")
    E(f"s_branch {join}")
    label(g1)
    emit_dma()
    emit_phase_b_dispatch()
    stamp(PH_B)
    emit_phase_c(False)
    stamp(PH_C_EPI)
    # (phase C's vmcnt(0) came after the DMA's issue; without a phase C, wait here)
    nowait = uid("nw")
    E(f"s_cmp_eq_u32 {s('c_ok')}, 1")
    E(f"s_cbranch_scc1 {nowait}")
    E("s_waitcnt vmcnt(0)")
    label(nowait)
    label(join)
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
    # (name, stamped, options): the shipped kernel and its stamped diagnostic build; with AQ_GEN_EXPERIMENTAL=1 also the timing experiments of
    # round 4 (selected at run time by AQ_BTL_ASM_KERNEL=<name>, tools/time_bottleneck.py; results in profiles/NOTES_r04.md): no stagger, the
    # packed fp32 instructions as plain ones, and ablations that leave out the SiLU arithmetic, the MFMAs, the LDS-DMA, the shortcut loads or
    # the output stores (wrong results: timing only)
    import os
    variants = [("bottleneck_asm_c48", False, {}), ("bottleneck_asm_c48_stamped", True, {})]
    if os.environ.get("AQ_GEN_EXPERIMENTAL") == "1":
        variants += [("bottleneck_asm_c48_nostagger", False, dict(stagger=False)), ("bottleneck_asm_c48_nostagger_stamped", True, dict(stagger=False)),
                     ("bottleneck_asm_c48_prio", False, dict(prio=3)), ("bottleneck_asm_c48_prio_stamped", True, dict(prio=3)),
                     ("bottleneck_asm_c48_nopk", False, dict(nopk=True)), ("bottleneck_asm_c48_nosilu", False, dict(nosilu=True)),
                     ("bottleneck_asm_c48_nomfma", False, dict(nomfma=True)), ("bottleneck_asm_c48_skel", False, dict(nomfma=True, nosilu=True)),
                     ("bottleneck_asm_c48_skel_nodma", False, dict(nomfma=True, nosilu=True, nodma=True)),
                     ("bottleneck_asm_c48_skel_nold", False, dict(nomfma=True, nosilu=True, nold=True)),
                     ("bottleneck_asm_c48_skel_nost", False, dict(nomfma=True, nosilu=True, nost=True)),
                     ("bottleneck_asm_c48_skel_nomem", False, dict(nomfma=True, nosilu=True, nodma=True, nold=True, nost=True)),
                     ("bottleneck_asm_c48_nomem", False, dict(nodma=True, nold=True, nost=True)),
                     ("bottleneck_asm_c48_ntst", False, dict(ntst=True)), ("bottleneck_asm_c48_ntld", False, dict(ntld=True)),
                     ("bottleneck_asm_c48_nt", False, dict(ntst=True, ntld=True))]
    for name, stamped, opt in variants:
        text += [f"\t.globl\t{name}", "\t.p2align\t8", f"\t.type\t{name},@function"]
        text += gen_kernel(name, stamped, **opt)
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
