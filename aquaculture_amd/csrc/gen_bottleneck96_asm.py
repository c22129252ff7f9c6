#!/usr/bin/env python3
"""Generator of the gfx950 assembly of the fused Bottleneck for C = 96 (bottleneck_asm_c96): y = (x +) SiLU(W2 (*) SiLU(W1 x + b1) + b2), bf16.

yolov5m's model.4.m.0-3 and model.17.m.0-1 ([UPSTREAM models/common.py Bottleneck.forward: cv2(cv1(x)), 1x1 then 3x3, hidden width = width])
at 80 x 80 run 104-119 us per launch on bottleneck_kernel<1, 6, 12, 16, 12> = 0.64-0.72 PFLOP/s: twelve waves, ONE 16-row M block per wave, so every
MFMA needs its own 1 KB B fragment from LDS (256 B/clk at full MFMA rate against the LDS's 128).  gen_bottleneck_asm.py cured that for C = 48 by
giving every wave all three M blocks and all the weights; at C = 96 the 3x3's weights are 162 A fragments (648 registers).  Here:

  * FOUR waves, one per SIMD, the whole register file each (240 VGPRs + 256 AGPRs).  Wave (h, q): h = output-channel half (three M blocks, 48
    channels), q = pixel half.  Its 81 A fragments of the 3x3 (tap, k-step, M block) stay in registers for the life of the workgroup -- 64 in the
    accumulator half of the file (MFMA takes srcA from AGPRs), 17 in VGPRs -- and the nine of the 1x1.  Every B fragment read from LDS feeds
    three MFMAs.
  * Tile = 8 rows x 16 columns (patch 10 x 18 = 180 pixels; 12 blocks of 16 in linear patch order).  Pixel stride in LDS 224 bytes (192 + 32:
    every lane group of a ds_read_b128 falls on 16 different bank quads, see PXB).  Two x patches and two t patches + the biases = 163,328 B.
  * Per tile and wave: phase B -- t = SiLU(W1 x + b1) for the wave's six patch blocks and 48 channels, zero outside the image (the 3x3 pads t),
    in two passes of three blocks; ONE barrier; phase C -- the 3x3 for the wave's four output rows: 27 k-steps x (4 fragment reads, 12 MFMAs),
    SiLU, shortcut (re-read from global memory at the top of the phase, as in the C = 48 kernel: x is dead after phase B, so the next-but-one
    tile's patch is fetched into its buffer by LDS-DMA right behind those loads), bf16, 16 + 8 byte stores (weight rows permuted as in the
    wide 1x1: a lane's twelve outputs are 8 + 4 consecutive channels).
  * x patches arrive by buffer-descriptor LDS-DMA, 14 sixteen-byte slots per pixel (the last two are the row padding: their lanes point outside
    the descriptor).  Pixels outside the image feed only t values that phase B forces to zero.

Hazards the assembler does not pad (LLVM GCNHazardRecognizer, gfx940): MFMA result -> VALU read (s_nop 15), transcendental -> consumer,
s_mov m0 -> LDS-DMA (s_nop 0), VALU -> v_readfirstlane (s_nop 1).

Usage: python gen_bottleneck96_asm.py OUT.s   (aquaculture_amd/build.py assembles it and embeds the code object in bottleneck.hip)
"""
import os
import sys

C = 96
TH, TW = 8, 16
PW, PH = TW + 2, TH + 2
PP = PW * PH                 # 180 patch pixels
NBLK = 12                    # 16-pixel blocks of the patch (192 pixels: the last 12 are never read by phase C)
PXB = 224                    # LDS pixel stride: 14 sixteen-byte slots.  ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... together
                             # (MI355X_MICROARCH.md, LDS): with a fragment lane (p, g) at 224 p + 16 g the sixteen lanes of every group fall on
                             # sixteen different bank quads for any pixel shift (brute-forced; 208 gave five 2-way conflicts per group: 65 % MFMA duty)
SLOTS = PXB // 16            # 14
NINST = (PP * SLOTS + 63) // 64      # 40 LDS-DMA instructions per patch (the last 40 slots of the fortieth lie past the patch)
XBUF = NINST * 1024          # an x buffer holds what the DMA writes: 180 pixels + 640 bytes
TBUF = PP * PXB              # a t buffer: 180 pixels (phase B masks its writes beyond)
BIAS_OFF = 0                 # b1 | b2: 192 floats (first: DS instruction offsets are 16 bits)
X0 = 4 * 2 * C
T0 = X0 + 2 * XBUF
LDS_BYTES = T0 + 2 * TBUF
NDMA = 10                    # per wave: i = wave + 4 n
assert NINST == 40 and NDMA * 4 == NINST and LDS_BYTES <= 163840, LDS_BYTES
NF1, NF2 = 9, 81             # A fragments per wave: 1x1 (k-step, M block), 3x3 (tap, k-step, M block)
NAGPR_FRAGS = 64
PD = 6                       # B fragments of the 3x3 read ahead (ring of NRING)
NRING = 8
ROW_GROUP = 2                # phase C: rows per group (4: all rows k-step-major, epilogue behind the loop; 2: rows 0-1, then rows 2-3 with the first pair's epilogue between their MFMAs)
VM_FIRST, VM_EVERY = 3, 6    # phase C issues one of its fourteen vector-memory operations behind elements 3, 9, 15, ... of its 108

ARG_BYTES = 96
PH_B, PH_BARRIER, PH_C_MFMA, PH_C_EPI = range(4)      # (the prologue is booked on the first phase B)
NPH = 4


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
S = Regs("s", 102)

S.alloc("karg", 2)
S.alloc("wg")
S.alloc("pad0")
for nm in ("inp", "out", "w", "bias"):
    S.alloc(nm, 2, 2)
for nm in ("in_ld", "out_ld", "B", "H", "W", "tiles_x", "tpi", "ntiles", "shortcut", "G", "magic_tpi", "magic_tx", "in_bytes", "out_bytes"):
    S.alloc(nm)
S.alloc("debug", 2, 2)
S.alloc("srd_in", 4, 4)
S.alloc("srd_out", 4, 4)
S.alloc("srd_w", 4, 4)
for nm in ("wave", "wh", "wq", "tmp0", "tmp1", "tmp2", "tmp3", "tmp4", "tmp5",
           "d_tile", "d_ok", "d_b", "d_y0", "d_x0",            # stage D: the tile whose x patch the LDS-DMA fetches next (two ahead)
           "n_ok", "n_b", "n_y0", "n_x0",                      # stage N: its patch is on its way
           "b_ok", "b_b", "b_y0", "b_x0",                      # stage B: this iteration's tile (phase B, then phase C)
           "xbuf", "dt", "dx", "org", "y0m1", "x0m1", "pix0", "full", "prev_full", "wlim", "soff", "woff"):
    S.alloc(nm)
S.alloc("klog2e2", 2, 2)
S.alloc("kone2", 2, 2)
S.alloc("colmask", 2, 2)
S.alloc("sa", 2, 2)
S.alloc("rowm", 8, 2)
S.alloc("t64", 2, 2)
S.alloc("st_last", 2, 2)
S.alloc("st_acc", 2 * NPH, 2)

V.alloc("tid")
V.alloc("W2V", 4 * (NF2 - NAGPR_FRAGS), 4)
V.alloc("W1", 4 * NF1, 4)
V.alloc("ACC", 48, 4)
V.alloc("BR", 4 * NRING, 4)
V.alloc("SC", 24, 4)
V.alloc("T", 12, 4)
for nm in ("vxb", "vtw1", "vtw2", "vtc", "vpp0", "vsc1", "vsc2", "vo1", "vo2", "vbl1", "vbl2", "va0", "va1", "va2", "va3", "vl15"):
    V.alloc(nm)
V.alloc("pre", NDMA)
ACCUM_OFFSET = (V.next + 3) // 4 * 4
NAGPR = 4 * NAGPR_FRAGS


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


def w2frag(tap, ks, m):
    """A fragment (tap, k-step, M block) of the 3x3: the first 64 live in AGPRs, the rest in VGPRs."""
    f = (tap * 3 + ks) * 3 + m
    if OPT.get("vgpra"):                   # timing experiment: every srcA from the 17 VGPR fragments (wrong results)
        return vr("W2V", 4 * (f % (NF2 - NAGPR_FRAGS)), 4)
    if f < NAGPR_FRAGS:
        return f"a[{4 * f}:{4 * f + 3}]"
    return vr("W2V", 4 * (f - NAGPR_FRAGS), 4)


def w1frag(ks, m):
    return vr("W1", 4 * (ks * 3 + m), 4)


def acc(m, j):
    """Phase C: M block m, output row j of the wave (4 rows).  Phase B: M block m, block j of the pass (3 blocks)."""
    return vr("ACC", 4 * (3 * j + m), 4)


out = []
_uid = [0]
STAMPED = [False]
PLAIN = [False]              # True while phase C's interleaved epilogue is generated: packed fp32 instructions are written as two plain ones
OPT = dict(nosilu=False, nomfma=False, nodma=False, nold=False, nost=False, vgpra=False, nords=False, pk=False, rg=0, bstamp=False)


def E(line="", comment=None):
    op = line.split(" ")[0]
    if OPT["nomfma"] and op.startswith("v_mfma"):
        return
    if OPT["nodma"] and op == "buffer_load_dwordx4" and line.endswith(" lds"):
        return
    if OPT["nold"] and op in ("buffer_load_dwordx4", "buffer_load_dwordx2") and "SHORTCUT" in (comment or ""):
        return
    if OPT["nost"] and op.startswith("buffer_store"):
        return
    if OPT["nosilu"] and op in ("v_exp_f32", "v_rcp_f32", "v_pk_mul_f32", "v_pk_add_f32"):
        return
    if OPT["nords"] and op == "ds_read_b128" and (comment or "") == "CFRAG":
        return
    if op in ("v_pk_mul_f32", "v_pk_add_f32") and PLAIN[0] and not OPT.get("pk"):
        # the packed fp32 forms as two plain instructions: beside MFMAs in flight a v_pk_mul_f32 takes 20.7 cycles, a v_mul_f32 8.3
        # (tools/ubench/valu_issue.hip).  Only where the instruction sits between MFMAs (phase C's epilogue units): in phase B, where the
        # MFMAs in flight are nine of a block's 570 cycles, the packed forms are the faster ones (stamped build: 66.5 k vs 72 k cycles)
        import re
        m_ = re.match(r"v_pk_(mul|add)_f32 v\[(\d+):\d+\], v\[(\d+):\d+\], (.*)$", line)
        o_, d_, a_, rest = m_.group(1), int(m_.group(2)), int(m_.group(3)), m_.group(4)
        for h_ in range(2):
            r2 = re.match(r"v\[(\d+):\d+\]$", rest)
            if r2:
                b_ = f"v{int(r2.group(1)) + h_}"
            else:
                r3 = re.match(r"s\[(\d+):\d+\]$", rest)
                b_ = f"s{int(r3.group(1)) + h_}"
            out.append(f"\tv_{o_}_f32_e64 v{d_ + h_}, v{a_ + h_}, {b_}")
        return
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


def emit_decode():
    """Stage D's tile number -> d_ok, d_b, d_y0, d_x0 (scalar; the magic multipliers come from the host)."""
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
    E(f"s_lshl_b32 {s('d_y0')}, {s('tmp1')}, 3")
    E(f"s_lshl_b32 {s('d_x0')}, {s('tmp2')}, 4")
    label(skip)


def emit_dma(dst):
    """Stage D's x patch -> the x buffer at LDS byte `dst` (an SGPR name): this wave's ten 1 KB instructions."""
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
    for n in range(NDMA):
        a = v("va0") if n % 2 == 0 else v("va1")
        E(f"v_add_u32 {a}, {s('org')}, {v('pre', n)}")
        # instruction i = wave + 4 n fills LDS bytes 1024 i .. + 1023 of the buffer (woff = 1024 wave)
        E(f"s_add_u32 {s('tmp1')}, {s(dst)}, {s('woff')}")
        E(f"s_add_u32 m0, {s('tmp1')}, {4096 * n}")
        E("s_nop 0", "hz: m0 write -> LDS-DMA")
        E(f"buffer_load_dwordx4 {a}, {s4('srd_in')}, 0 offen lds")
    label(skip)


def emit_silu(regs, temps):
    """temps <- 1 / (1 + exp(-regs)), batched: independent transcendentals back to back."""
    n = len(regs)
    assert n % 2 == 0 and len(temps) >= n
    for i in range(0, n, 2):
        E(f"v_pk_mul_f32 v[{temps[i]}:{temps[i] + 1}], v[{regs[i]}:{regs[i] + 1}], {s2('klog2e2')}")
    for i in range(n):
        E(f"v_exp_f32 v{temps[i]}, v{temps[i]}")
    for i in range(0, n, 2):
        E(f"v_pk_add_f32 v[{temps[i]}:{temps[i] + 1}], v[{temps[i]}:{temps[i] + 1}], {s2('kone2')}")
    for i in range(n):
        E(f"v_rcp_f32 v{temps[i]}, v{temps[i]}")


def emit_acc_init(which, ntiles):
    """Accumulator tiles 0 .. ntiles - 1 (three M blocks each) <- bias vector `which` (0: b1, 1: b2) of this lane's twelve channels."""
    A0 = V.names["ACC"][0]
    off = BIAS_OFF + 4 * C * which
    E(f"ds_read_b128 v[{A0}:{A0 + 3}], {v('vbl1')} offset:{off}")
    E(f"ds_read_b128 v[{A0 + 4}:{A0 + 7}], {v('vbl1')} offset:{off + 16}")
    E(f"ds_read_b128 v[{A0 + 8}:{A0 + 11}], {v('vbl2')} offset:{off}")
    E("s_waitcnt lgkmcnt(0)")
    for j in range(1, ntiles):
        for k in range(12):
            E(f"v_mov_b32 v{A0 + 12 * j + k}, v{A0 + k}")


def emit_phase_b():
    """t = SiLU(W1 x + b1) for this wave's six patch blocks (6 q .. 6 q + 5) and 48 channels, zero outside the image, into the t patch.
    Software-pipelined over the blocks (one wave per SIMD: nothing else hides a latency): block b + 1's nine MFMAs are issued in front of block
    b's SiLU and run under it; block b + 2's three x fragments and its accumulators' start (b1, from LDS) are read in front of that.  Three
    accumulator sets (36 registers), two fragment sets (BR)."""
    A0 = V.names["ACC"][0]
    T0_ = V.names["T"][0]
    BR0 = V.names["BR"][0]
    temps = [T0_ + i for i in range(12)]
    NB6 = 6

    def reads(bi):
        f0 = BR0 + 12 * (bi % 2)
        a0 = A0 + 12 * (bi % 3)
        for ks in range(3):
            E(f"ds_read_b128 v[{f0 + 4 * ks}:{f0 + 4 * ks + 3}], {v('vxb')} offset:{16 * PXB * bi + 64 * ks}")
        E(f"ds_read_b128 v[{a0}:{a0 + 3}], {v('vbl1')} offset:{BIAS_OFF}", "accumulators start from b1")
        E(f"ds_read_b128 v[{a0 + 4}:{a0 + 7}], {v('vbl1')} offset:{BIAS_OFF + 16}")
        E(f"ds_read_b128 v[{a0 + 8}:{a0 + 11}], {v('vbl2')} offset:{BIAS_OFF}")

    def mfmas(bi):
        f0 = BR0 + 12 * (bi % 2)
        a0 = A0 + 12 * (bi % 3)
        for ks in range(3):
            for m in range(3):
                E(f"v_mfma_f32_16x16x32_bf16 v[{a0 + 4 * m}:{a0 + 4 * m + 3}], {w1frag(ks, m)}, v[{f0 + 4 * ks}:{f0 + 4 * ks + 3}], v[{a0 + 4 * m}:{a0 + 4 * m + 3}]")

    def silu_store(bi):
        regs = [A0 + 12 * (bi % 3) + k for k in range(12)]
        # is this lane's patch pixel inside the image?  (the padding of the 3x3 is zero in t, not in x)
        t0, t1, t2 = v("va0"), v("va1"), v("va2")
        E(f"v_add_u32 {t0}, {16 * bi}, {v('vpp0')}", "patch pixel")
        E(f"v_mul_u32_u24 {t1}, 3641, {t0}")
        E(f"v_lshrrev_b32 {t1}, 16, {t1}", "patch row = pixel / 18")
        E(f"v_mul_u32_u24 {t2}, 18, {t1}")
        E(f"v_sub_u32 {t2}, {t0}, {t2}", "patch column")
        E(f"v_add_u32 {t1}, {s('y0m1')}, {t1}", "image row (wraps below zero)")
        E(f"v_add_u32 {t2}, {s('x0m1')}, {t2}")
        E(f"v_cmp_gt_u32 {s2('sa')}, {s('H')}, {t1}")
        E(f"v_cmp_gt_u32 vcc, {s('W')}, {t2}")
        E(f"s_and_b64 vcc, vcc, {s2('sa')}")
        emit_silu(regs, temps)
        for k in range(0, 12, 2):
            E(f"v_pk_mul_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
        for k in range(6):
            E(f"v_cvt_pk_bf16_f32 v{temps[k]}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
        for k in range(6):
            E(f"v_cndmask_b32 v{temps[k]}, 0, v{temps[k]}, vcc")
        o = 16 * PXB * bi
        last = bi == NB6 - 1                       # block 11 of the patch (waves with q = 1) ends 12 pixels past it: those lanes must not write
        if last:
            E(f"v_cmp_gt_u32 vcc, {PP}, {t0}")
            E(f"s_and_saveexec_b64 {s2('sa')}, vcc")
        E(f"ds_write_b128 {v('vtw1')}, v[{temps[0]}:{temps[3]}] offset:{o}")
        E(f"ds_write_b64 {v('vtw2')}, v[{temps[4]}:{temps[5]}] offset:{o}")
        if last:
            E(f"s_mov_b64 exec, {s2('sa')}")

    E(f"s_sub_u32 {s('y0m1')}, {s('b_y0')}, 1")
    E(f"s_sub_u32 {s('x0m1')}, {s('b_x0')}, 1")
    reads(0)
    reads(1)
    E("s_waitcnt lgkmcnt(6)", "block 0's fragments and b1")
    mfmas(0)
    for bi in range(NB6):
        if bi + 1 < NB6:
            # in the LDS queue, in order: reads(bi + 1) [6], then the previous block's two t writes (bi >= 1) -- the reads must be done
            E(f"s_waitcnt lgkmcnt({2 if bi >= 1 else 0})")
            mfmas(bi + 1)
        if bi + 2 < NB6:
            reads(bi + 2)
        if bi == 0:
            E("s_nop 15", "hz: MFMA result -> VALU read (block 0 only: later blocks' MFMAs are a whole SiLU old)")
            E("s_nop 15")
        if OPT["bstamp"]:
            stamp(PH_B)
        silu_store(bi)
        if OPT["bstamp"]:
            stamp(PH_BARRIER)


def tap_off(t):
    return ((t // 3) * PW + t % 3) * PXB


def emit_phase_c():
    """y = (x +) SiLU(W2 (*) t + b2) for the wave's four output rows (4 q .. 4 q + 3 of the tile) and 48 channels."""
    A0 = V.names["ACC"][0]
    T0_ = V.names["T"][0]
    SC0 = V.names["SC"][0]
    BR0 = V.names["BR"][0]
    temps = [T0_ + i for i in range(12)]
    # ---- scalar: the tile's first pixel, row masks ----
    E(f"s_lshl_b32 {s('tmp0')}, {s('wq')}, 2")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('b_y0')}", "image row of the wave's first output row")
    E(f"s_mul_i32 {s('tmp1')}, {s('b_b')}, {s('H')}")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('tmp0')}")
    E(f"s_mul_i32 {s('tmp1')}, {s('tmp1')}, {s('W')}")
    E(f"s_add_u32 {s('pix0')}, {s('tmp1')}, {s('b_x0')}", "pixel index of (row 0 of the wave, x0)")
    E(f"s_sub_u32 {s('wlim')}, {s('W')}, {s('b_x0')}", "columns of the tile inside the image")
    E(f"v_cmp_gt_u32 {s2('colmask')}, {s('wlim')}, {v('vl15')}")
    E(f"s_mov_b32 {s('full')}, 1")
    for j in range(4):
        E(f"s_add_u32 {s('tmp1')}, {s('tmp0')}, {j}")
        E(f"s_cmp_lt_u32 {s('tmp1')}, {s('H')}")
        E(f"s_cselect_b64 {s2('rowm', 2 * j)}, {s2('colmask')}, 0")
        E(f"s_cselect_b32 {s('tmp2')}, 1, 0")
        E(f"s_and_b32 {s('full')}, {s('full')}, {s('tmp2')}")
    # ---- vector-memory operations of the phase: the next-but-one tile's patch (ten LDS-DMA instructions) and the shortcut (four rows x two
    # loads), ONE every few elements of the MFMA loop.  One wave per SIMD: a vector-memory instruction that finds the address unit busy holds
    # the wave, and with it the MFMA issue -- issued back to back at the top of the phase the eighteen cost 2 k cycles per tile (stamped build:
    # set-up 41 k -> 16.5 k cycles without the shortcut loads, the MFMA loop + 10 k with them right behind the DMA). ----
    E(f"s_mul_i32 {s('tmp0')}, {s('d_b')}, {s('H')}")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('d_y0')}")
    E(f"s_sub_u32 {s('tmp0')}, {s('tmp0')}, 1")
    E(f"s_mul_i32 {s('tmp0')}, {s('tmp0')}, {s('W')}")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('d_x0')}")
    E(f"s_sub_u32 {s('tmp0')}, {s('tmp0')}, 1")
    E(f"s_mul_i32 {s('org')}, {s('tmp0')}, {s('in_ld')}", "byte offset of stage D's patch pixel (0, 0); may wrap below zero: outside the descriptor")
    E(f"s_add_u32 {s('tmp3')}, {s('xbuf')}, {s('woff')}")
    E(f"s_mul_i32 {s('soff')}, {s('pix0')}, {s('in_ld')}")
    E(f"s_mul_i32 {s('tmp2')}, {s('W')}, {s('in_ld')}")

    def dma_op(n):
        skip = uid("dma")
        E(f"s_cmp_eq_u32 {s('d_ok')}, 0")
        E(f"s_cbranch_scc1 {skip}")
        a = v("va0") if n % 2 == 0 else v("va1")
        E(f"v_add_u32 {a}, {s('org')}, {v('pre', n)}")
        E(f"s_add_u32 m0, {s('tmp3')}, {4096 * n}")
        E("s_nop 0", "hz: m0 write -> LDS-DMA")
        E(f"buffer_load_dwordx4 {a}, {s4('srd_in')}, 0 offen lds")
        label(skip)

    def sc_op(j):
        skip = uid("nosc")
        E(f"s_cmp_eq_u32 {s('shortcut')}, 0")
        E(f"s_cbranch_scc1 {skip}")
        E(f"s_mov_b64 exec, {s2('rowm', 2 * j)}")
        E(f"v_add_u32 {v('va2')}, {s('soff')}, {v('vsc1')}")
        E(f"v_add_u32 {v('va3')}, {s('soff')}, {v('vsc2')}")
        E(f"buffer_load_dwordx4 v[{SC0 + 6 * j}:{SC0 + 6 * j + 3}], {v('va2')}, {s4('srd_in')}, 0 offen", "SHORTCUT")
        E(f"buffer_load_dwordx2 v[{SC0 + 6 * j + 4}:{SC0 + 6 * j + 5}], {v('va3')}, {s4('srd_in')}, 0 offen", "SHORTCUT")
        E("s_mov_b64 exec, -1")
        E(f"s_add_u32 {s('soff')}, {s('soff')}, {s('tmp2')}")
        label(skip)

    emit_acc_init(1, 4)
    if not OPT["bstamp"]:
        stamp(PH_BARRIER)                          # (stamped build: barrier wait + masks + accumulator start)

    # ---- the 3x3, ROW-major: output row j's 27 k-steps (81 MFMAs), then row j + 1's with row j's epilogue between them -- one wave per SIMD,
    # so the SiLU (v_exp + v_rcp at 16 cycles each) only overlaps MFMAs it is interleaved with.  B-fragment ring of NRING, read PD ahead.
    # Vector memory: the four rows' shortcut loads in row 0, ONE wait at the top of row 1; the ten LDS-DMA instructions spread over rows 1-3. ----
    NEL = 27 * 4

    def frag_read(e):
        j_, kstep = divmod(e, 27)
        tap, ks = divmod(kstep, 3)
        r = BR0 + 4 * (e % NRING)
        E(f"ds_read_b128 v[{r}:{r + 3}], {v('vtc')} offset:{tap_off(tap) + j_ * PW * PXB + 64 * ks}", "CFRAG")

    def epilogue_units(j_):
        """Row j's epilogue as units that may be separated by MFMAs (a unit itself is never split: EXEC changes stay inside one)."""
        global out
        saved, out = out, []
        units = []
        PLAIN[0] = True

        def cut():
            global out
            if out:
                units.append(out)
            out = []
        regs = [A0 + 12 * j_ + k for k in range(12)]
        for i_ in range(0, 12, 2):
            E(f"v_pk_mul_f32 v[{temps[i_]}:{temps[i_] + 1}], v[{regs[i_]}:{regs[i_] + 1}], {s2('klog2e2')}")
            cut()
        for i_ in range(12):
            E(f"v_exp_f32 v{temps[i_]}, v{temps[i_]}")
            cut()
        for i_ in range(0, 12, 2):
            E(f"v_pk_add_f32 v[{temps[i_]}:{temps[i_] + 1}], v[{temps[i_]}:{temps[i_] + 1}], {s2('kone2')}")
            cut()
        for i_ in range(12):
            E(f"v_rcp_f32 v{temps[i_]}, v{temps[i_]}")
            cut()
        for k in range(0, 12, 2):
            E(f"v_pk_mul_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
            cut()
        for k in range(6):                         # the shortcut (zeros in these registers when the layer has none)
            sc = SC0 + 6 * j_ + k
            E(f"v_lshlrev_b32 v{temps[2 * k]}, 16, v{sc}")
            E(f"v_and_b32 v{temps[2 * k + 1]}, 0xffff0000, v{sc}")
            cut()
        for k in range(0, 12, 2):
            E(f"v_pk_add_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
            cut()
        for k in range(6):
            E(f"v_cvt_pk_bf16_f32 v{temps[k]}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
            cut()
        a1, a2 = (v("va0"), v("va1")) if j_ % 2 == 0 else (v("va2"), v("va3"))
        E(f"v_add_u32 {a1}, {s('tmp1')}, {v('vo1')}")
        E(f"v_add_u32 {a2}, {s('tmp1')}, {v('vo2')}")
        E(f"s_mov_b64 exec, {s2('rowm', 2 * j_)}")
        E(f"buffer_store_dwordx4 v[{temps[0]}:{temps[3]}], {a1}, {s4('srd_out')}, 0 offen")
        E(f"buffer_store_dwordx2 v[{temps[4]}:{temps[5]}], {a2}, {s4('srd_out')}, 0 offen")
        E("s_mov_b64 exec, -1")
        E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('tmp0')}")
        cut()
        out = saved
        PLAIN[0] = False
        return units

    E(f"s_mul_i32 {s('tmp1')}, {s('pix0')}, {s('out_ld')}")
    E(f"s_mul_i32 {s('tmp0')}, {s('W')}, {s('out_ld')}")
    # element order: groups of RG rows, k-step-major inside a group (an accumulator is touched every 3 RG MFMAs: RG = 1 puts the next MFMA on
    # the same accumulator 49 cycles behind, at the edge of the MFMA's latency, and every interleaved instruction then stalls it)
    RG = OPT.get("rg") or ROW_GROUP
    NG = 4 // RG
    order = [(g_ * RG + r_, kstep) for g_ in range(NG) for kstep in range(27) for r_ in range(RG)]
    GEL = 27 * RG
    sc_at = {2 + 5 * j_: j_ for j_ in range(4)}                     # elements of the first group
    if NG > 1:
        dma_at = {GEL + 2 + ((NEL - GEL - PD - 4) // NDMA) * n: n for n in range(NDMA)}     # spread over the other groups
    else:
        dma_at = {24 + 8 * n: n for n in range(NDMA)}               # behind the shortcut loads
    assert max(dma_at) < NEL - PD and max(sc_at) < min(dma_at)

    def frag_read2(e):
        j_, kstep = order[e]
        tap, ks = divmod(kstep, 3)
        r = BR0 + 4 * (e % NRING)
        E(f"ds_read_b128 v[{r}:{r + 3}], {v('vtc')} offset:{tap_off(tap) + j_ * PW * PXB + 64 * ks}", "CFRAG")

    for e in range(PD):
        frag_read2(e)
    pending = []
    for e in range(NEL):
        j_, kstep = order[e]
        tap, ks = divmod(kstep, 3)
        g_, ge = divmod(e, GEL)
        if ge == 0 and g_ == 1:
            E("s_waitcnt vmcnt(0)", "the four rows' shortcut values (issued a row group ago; nothing younger is in flight yet)")
        if ge == 2 * RG and g_ >= 1:
            pending = [u for r_ in range(RG) for u in epilogue_units((g_ - 1) * RG + r_)]     # (its last MFMAs are >= six MFMAs old)
        if e + PD < NEL:
            frag_read2(e + PD)
        E(f"s_waitcnt lgkmcnt({min(PD, NEL - 1 - e)})")
        r = BR0 + 4 * (e % NRING)
        for m in range(3):
            E(f"v_mfma_f32_16x16x32_bf16 {acc(m, j_)}, {w2frag(tap, ks, m)}, v[{r}:{r + 3}], {acc(m, j_)}")
            # one epilogue unit behind EACH MFMA: an MFMA holds the issue port for 4 of its 16 cycles, a VALU instruction issued in the other
            # twelve is free (tools/ubench/valu_issue.hip: mfma / exp alternating = 16.9 cycles per pair); three in a row behind the third are not
            if pending:
                left = 3 * (GEL - ge) - m
                take = (len(pending) + left - 1) // left
                for u in pending[:take]:
                    out.extend(u)
                pending = pending[take:]
        if e in sc_at:
            sc_op(sc_at[e])
        if e in dma_at:
            dma_op(dma_at[e])
    assert not pending
    stamp(PH_C_MFMA)
    E("s_nop 15", "hz: MFMA result -> VALU read")
    E("s_nop 3")
    if NG == 1:
        w0, wd = uid("w0"), uid("wd")
        E(f"s_cmp_eq_u32 {s('d_ok')}, 0")
        E(f"s_cbranch_scc1 {w0}")
        E(f"s_waitcnt vmcnt({NDMA})", "shortcut values (the LDS-DMA instructions behind them stay in flight)")
        E(f"s_branch {wd}")
        label(w0)
        E("s_waitcnt vmcnt(0)")
        label(wd)
    for r_ in range(RG):
        for u in epilogue_units((NG - 1) * RG + r_):
            out.extend(u)


_kernel_no = [0]


def count_stores_behind_dma():
    """Dry run of phase C: how many output stores does a full tile issue behind its last LDS-DMA instruction?  (vmcnt is in order: that many may
    still be in flight when the next patch must have landed)"""
    global out
    saved, out = out, []
    u0 = _uid[0]
    st = STAMPED[0]
    STAMPED[0] = False
    saved_opt = dict(OPT)
    for k_ in OPT:
        if k_ not in ("rg", "bstamp"):
            OPT[k_] = False
    emit_phase_c()
    OPT.update(saved_opt)
    body, out = out, saved
    _uid[0] = u0
    STAMPED[0] = st
    last = max(i_ for i_, l in enumerate(body) if l.split(";")[0].rstrip().endswith(" lds"))
    n = sum(1 for l in body[last:] if l.lstrip().startswith("buffer_store"))
    STORES_BEHIND_DMA[0] = n
    return n
STORES_BEHIND_DMA = [8]       # output stores a full tile issues behind its last LDS-DMA instruction (set per kernel from the generated phase C)


def gen_kernel(name, stamped=False, **opt):
    global out
    out = []
    STAMPED[0] = stamped
    for k in OPT:
        OPT[k] = False
    OPT.update(opt)
    _kernel_no[0] += 1
    _uid[0] = 100000 * _kernel_no[0]
    E("; fused Bottleneck, C = 96, 4 waves: generated by gen_bottleneck96_asm.py -- do not edit")
    label(name)
    a0 = S.names["inp"][0]
    b0 = S.names["in_ld"][0]
    assert a0 % 4 == 0 and b0 == a0 + 8 and S.names["debug"][0] == b0 + 14
    E(f"s_load_dwordx8 s[{a0}:{a0 + 7}], {s2('karg')}, 0x0", "inp, out, w, bias")
    E(f"s_load_dwordx8 s[{b0}:{b0 + 7}], {s2('karg')}, 0x20", "in_ld .. ntiles")
    E(f"s_load_dwordx4 s[{b0 + 8}:{b0 + 11}], {s2('karg')}, 0x40", "shortcut, G, magic_tpi, magic_tx")
    E(f"s_load_dwordx4 s[{b0 + 12}:{b0 + 15}], {s2('karg')}, 0x50", "in_bytes, out_bytes, debug")
    T = [V.names["T"][0] + i for i in range(12)]
    lane, g = T[0], T[1]
    E(f"v_and_b32 v{lane}, 63, {v('tid')}", "lane")
    E(f"v_lshrrev_b32 v{T[2]}, 6, {v('tid')}")
    E("s_nop 1", "hz: VALU write -> v_readfirstlane")
    E(f"v_readfirstlane_b32 {s('wave')}, v{T[2]}")
    E(f"v_and_b32 {v('vl15')}, 15, v{lane}")
    E(f"v_lshrrev_b32 v{g}, 4, v{lane}")
    E(f"s_and_b32 {s('wh')}, {s('wave')}, 1", "output-channel half")
    E(f"s_lshr_b32 {s('wq')}, {s('wave')}, 1", "pixel half")
    E(f"s_lshl_b32 {s('woff')}, {s('wave')}, 10")
    E("s_waitcnt lgkmcnt(0)")
    for nm, base, size in (("srd_in", "inp", "in_bytes"), ("srd_out", "out", "out_bytes")):
        E(f"s_mov_b32 {s(nm, 0)}, {s(base)}")
        E(f"s_and_b32 {s(nm, 1)}, {s(base, 1)}, 0xffff")
        E(f"s_mov_b32 {s(nm, 2)}, {s(size)}")
        E(f"s_mov_b32 {s(nm, 3)}, 0x00020000")
    E(f"s_mov_b32 {s('srd_w', 0)}, {s('w')}")
    E(f"s_and_b32 {s('srd_w', 1)}, {s('w', 1)}, 0xffff")
    E(f"s_mov_b32 {s('srd_w', 2)}, {2 * (NF1 + NF2) * 1024}")
    E(f"s_mov_b32 {s('srd_w', 3)}, 0x00020000")
    E(f"s_mov_b32 {s('klog2e2')}, 0xbfb8aa3b", "-log2(e)")
    E(f"s_mov_b32 {s('klog2e2', 1)}, 0xbfb8aa3b")
    E(f"s_mov_b32 {s('kone2')}, 1.0")
    E(f"s_mov_b32 {s('kone2', 1)}, 1.0")
    E(f"s_mov_b32 {s('prev_full')}, 0")
    for k in range(24):
        E(f"v_mov_b32 v{V.names['SC'][0] + k}, 0", "the shortcut registers: zeros unless the layer has one (the epilogue always adds them)" if k == 0 else None)
    if stamped:
        for k in range(2 * NPH):
            E(f"s_mov_b32 {s('st_acc', k)}, 0")
        E(f"s_memtime {s2('st_last')}")
        E("s_waitcnt lgkmcnt(0)")
    # ---- the weights: this half's nine + 81 A fragments, straight into their registers ----
    E(f"v_lshlrev_b32 v{T[2]}, 4, v{lane}", "16 lane")
    E(f"s_mul_i32 {s('tmp0')}, {s('wh')}, {(NF1 + NF2) * 1024}")
    dsts = [w1frag(ks, m) for ks in range(3) for m in range(3)] + [w2frag(t, ks, m) for t in range(9) for ks in range(3) for m in range(3)]
    for i, d in enumerate(dsts):
        if i and i % 4 == 0:
            E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, 4096")
        E(f"buffer_load_dwordx4 {d}, v{T[2]}, {s4('srd_w')}, {s('tmp0')} offen offset:{1024 * (i % 4)}")
    # ---- per-lane offsets ----
    E(f"v_mul_u32_u24 v{T[3]}, {PXB}, {v('vl15')}", "208 p")
    E(f"s_mul_i32 {s('tmp0')}, {s('wq')}, {6 * 16 * PXB}", "the wave's six blocks of the patch")
    E(f"v_add_u32 v{T[4]}, {s('tmp0')}, v{T[3]}")
    E(f"v_lshl_add_u32 {v('vxb')}, v{g}, 4, v{T[4]}")
    E(f"v_add_u32 {v('vxb')}, {X0}, {v('vxb')}", "phase B fragment: x buffer 0 + (96 q + p) 208 + 16 g  (+ 3328 block + 64 k-step)")
    E(f"s_mul_i32 {s('tmp1')}, {s('wh')}, 96")
    E(f"v_add_u32 v{T[5]}, {s('tmp1')}, v{T[4]}")
    E(f"v_lshl_add_u32 {v('vtw1')}, v{g}, 4, v{T[5]}")
    E(f"v_add_u32 {v('vtw1')}, {T0}, {v('vtw1')}", "t write, 16 bytes: t buffer 0 + pixel + 96 h + 16 g")
    E(f"v_lshl_add_u32 {v('vtw2')}, v{g}, 3, v{T[5]}")
    E(f"v_add_u32 {v('vtw2')}, {T0 + 64}, {v('vtw2')}", "t write, 8 bytes: + 64 + 8 g")
    E(f"s_mul_i32 {s('tmp0')}, {s('wq')}, {4 * PW * PXB}", "the wave's four output rows")
    E(f"v_add_u32 v{T[4]}, {s('tmp0')}, v{T[3]}")
    E(f"v_lshl_add_u32 {v('vtc')}, v{g}, 4, v{T[4]}")
    E(f"v_add_u32 {v('vtc')}, {T0}, {v('vtc')}", "phase C fragment: t buffer 0 + (72 q + p) 208 + 16 g  (+ tap + row + k-step)")
    E(f"s_mul_i32 {s('tmp0')}, {s('wq')}, 96")
    E(f"v_add_u32 {v('vpp0')}, {s('tmp0')}, {v('vl15')}", "patch pixel of block 0 of the wave")
    for nm, ld in (("vsc", "in_ld"), ("vo", "out_ld")):
        E(f"v_mul_lo_u32 v{T[4]}, {v('vl15')}, {s(ld)}")
        E(f"v_add_u32 v{T[4]}, {s('tmp1')}, v{T[4]}", "+ 96 h")
        E(f"v_lshl_add_u32 {v(nm + '1')}, v{g}, 4, v{T[4]}")
        E(f"v_lshl_add_u32 {v(nm + '2')}, v{g}, 3, v{T[4]}")
        E(f"v_add_u32 {v(nm + '2')}, 64, {v(nm + '2')}")
    E(f"s_mul_i32 {s('tmp0')}, {s('wh')}, 192", "48 floats")
    E(f"v_lshlrev_b32 {v('vbl1')}, 5, v{g}")
    E(f"v_add_u32 {v('vbl1')}, {s('tmp0')}, {v('vbl1')}", "biases: floats 48 h + 8 g .. + 7")
    E(f"v_lshlrev_b32 {v('vbl2')}, 4, v{g}")
    E(f"v_add_u32 {v('vbl2')}, {s('tmp0')}, {v('vbl2')}")
    E(f"v_add_u32 {v('vbl2')}, 128, {v('vbl2')}", "and 48 h + 32 + 4 g .. + 3")
    # LDS-DMA: instruction i = wave + 4 n fills slots 64 i + lane; slot -> (pixel = slot / 13, part = slot % 13) -> (patch row, column)
    for n in range(NDMA):
        E(f"s_add_u32 {s('tmp0')}, {s('wave')}, {4 * n}")
        E(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 6")
        E(f"v_add_u32 v{T[3]}, {s('tmp0')}, v{lane}", "slot")
        E(f"v_mul_u32_u24 v{T[4]}, 4682, v{T[3]}")
        E(f"v_lshrrev_b32 v{T[4]}, 16, v{T[4]}", "pixel = slot / 14  (exact below 4096)")
        E(f"v_mul_u32_u24 v{T[5]}, {SLOTS}, v{T[4]}")
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
        E(f"v_cmp_gt_u32 {s2('sa')}, 12, v{T[5]}")
        E(f"s_and_b64 vcc, vcc, {s2('sa')}")
        E(f"v_cndmask_b32 {v('pre', n)}, v{T[8]}, v{T[6]}, vcc", "beyond the patch / the row padding: outside the descriptor -> zeros")
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
    # ---- the biases -> LDS (wave 0, lanes 0-47: 192 floats) ----
    nb = uid("nobias")
    E(f"s_cmp_lg_u32 {s('wave')}, 0")
    E(f"s_cbranch_scc1 {nb}")
    E(f"v_lshlrev_b32 v{T[3]}, 4, v{lane}")
    E(f"v_cmp_gt_u32 vcc, 48, v{lane}")
    E(f"s_and_saveexec_b64 {s2('t64')}, vcc")
    E(f"global_load_dwordx4 v[{T[8]}:{T[8] + 3}], v{T[3]}, {s2('bias')}")
    E("s_waitcnt vmcnt(0)")
    E(f"ds_write_b128 v{T[3]}, v[{T[8]}:{T[8] + 3}] offset:{BIAS_OFF}")
    E(f"s_mov_b64 exec, {s2('t64')}")
    label(nb)
    # ---- pipeline fill: x patches of the first two tiles -> x buffers 0 and 1 ----
    emit_decode()
    E(f"s_mov_b32 {s('xbuf')}, {X0}")
    emit_dma("xbuf")
    for nm in ("ok", "b", "y0", "x0"):
        E(f"s_mov_b32 {s('b_' + nm)}, {s('d_' + nm)}")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('G')}")
    emit_decode()
    E(f"s_mov_b32 {s('xbuf')}, {X0 + XBUF}")
    emit_dma("xbuf")
    for nm in ("ok", "b", "y0", "x0"):
        E(f"s_mov_b32 {s('n_' + nm)}, {s('d_' + nm)}")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('G')}")
    emit_decode()
    E(f"s_mov_b32 {s('xbuf')}, {X0}", "stage B's x buffer = the buffer stage D's patch goes to (after this iteration's barrier)")
    E(f"s_mov_b32 {s('dx')}, {XBUF}", "toggles of the per-lane LDS addresses at the end of the iteration")
    E(f"s_mov_b32 {s('dt')}, {TBUF}")
    E("s_waitcnt vmcnt(0)")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    # ---- one iteration per tile: phase B ; [own part of the next patch landed] barrier ; phase C (shortcut loads, DMA of the patch after next) ----
    loop = uid("loop")
    label(loop)
    emit_phase_b()
    stamp(PH_B)
    # the next tile's patch (issued one iteration ago, behind it only that iteration's eight stores) must have landed before the barrier
    pf, join = uid("pf"), uid("join")
    E(f"s_cmp_eq_u32 {s('prev_full')}, 1")
    E(f"s_cbranch_scc1 {pf}")
    E("s_waitcnt vmcnt(0)")
    E(f"s_branch {join}")
    label(pf)
    E(f"s_waitcnt vmcnt({count_stores_behind_dma()})")
    label(join)
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    emit_phase_c()
    stamp(PH_C_EPI)
    E(f"s_mov_b32 {s('prev_full')}, {s('full')}")
    # rotate the stages, toggle the buffers
    for nm in ("ok", "b", "y0", "x0"):
        E(f"s_mov_b32 {s('b_' + nm)}, {s('n_' + nm)}")
        E(f"s_mov_b32 {s('n_' + nm)}, {s('d_' + nm)}")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('G')}")
    E(f"s_mov_b32 {s('d_ok')}, 0")
    more = uid("more")
    E(f"s_cmp_eq_u32 {s('n_ok')}, 0", "no tile in stage N: none follows either")
    E(f"s_cbranch_scc1 {more}")
    emit_decode()
    label(more)
    E(f"s_add_u32 {s('xbuf')}, {s('xbuf')}, {s('dx')}")
    E(f"v_add_u32 {v('vxb')}, {s('dx')}, {v('vxb')}")
    for nm in ("vtw1", "vtw2", "vtc"):
        E(f"v_add_u32 {v(nm)}, {s('dt')}, {v(nm)}")
    E(f"s_sub_u32 {s('dx')}, 0, {s('dx')}")
    E(f"s_sub_u32 {s('dt')}, 0, {s('dt')}")
    E(f"s_cmp_eq_u32 {s('b_ok')}, 1")
    E(f"s_cbranch_scc1 {loop}")
    label(f".Lend_{name}")
    E("s_waitcnt vmcnt(0)", "nothing of this workgroup may still be on its way to LDS or memory")
    E("s_waitcnt lgkmcnt(0)")
    if stamped:
        T3 = V.names["T"][0]
        E(f"s_lshl_b32 {s('tmp0')}, {s('wg')}, 2")
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('wave')}")
        E(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 6")
        E(f"v_mov_b32 v{T3 + 2}, {s('tmp0')}")
        E("s_mov_b64 exec, 1")
        for k in range(NPH):
            E(f"v_mov_b32 v{T3}, {s('st_acc', 2 * k)}")
            E(f"v_mov_b32 v{T3 + 1}, {s('st_acc', 2 * k + 1)}")
            E(f"global_store_dwordx2 v{T3 + 2}, v[{T3}:{T3 + 1}], {s2('debug')} offset:{8 * k}")
        E("s_waitcnt vmcnt(0)")
    E("s_endpgm")
    return list(out)


def descriptor(name):
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
		.amdhsa_next_free_vgpr {ACCUM_OFFSET + NAGPR}
		.amdhsa_next_free_sgpr 102
		.amdhsa_accum_offset {ACCUM_OFFSET}
		.amdhsa_reserve_vcc 1
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text
"""


def metadata_entry(name):
    return f"""  - .agpr_count:     {NAGPR}
    .args:
      - .offset:         0
        .size:           {ARG_BYTES}
        .value_kind:     by_value
    .group_segment_fixed_size: {LDS_BYTES}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {ARG_BYTES}
    .max_flat_workgroup_size: 256
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     108
    .symbol:         {name}.kd
    .vgpr_count:     {ACCUM_OFFSET + NAGPR}
    .wavefront_size: 64
"""


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "bottleneck96_asm.s"
    text = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.text"]
    entries = []
    variants = [("bottleneck_asm_c96", False, {}), ("bottleneck_asm_c96_stamped", True, {})]
    if os.environ.get("AQ_GEN_EXPERIMENTAL") == "1":
        variants += [("bottleneck_asm_c96_nosilu", False, dict(nosilu=True)), ("bottleneck_asm_c96_nomfma", False, dict(nomfma=True)),
                     ("bottleneck_asm_c96_nomem", False, dict(nodma=True, nold=True, nost=True)),
                     ("bottleneck_asm_c96_skel", False, dict(nosilu=True, nomfma=True)),
                     ("bottleneck_asm_c96_nodma", False, dict(nodma=True)), ("bottleneck_asm_c96_nodma_stamped", True, dict(nodma=True)),
                     ("bottleneck_asm_c96_nold", False, dict(nold=True)), ("bottleneck_asm_c96_nold_stamped", True, dict(nold=True)),
                     ("bottleneck_asm_c96_nost", False, dict(nost=True)), ("bottleneck_asm_c96_nost_stamped", True, dict(nost=True)),
                     ("bottleneck_asm_c96_rg4", False, dict(rg=4)), ("bottleneck_asm_c96_rg4_stamped", True, dict(rg=4)),
                     ("bottleneck_asm_c96_rg1", False, dict(rg=1)), ("bottleneck_asm_c96_rg1_stamped", True, dict(rg=1)),
                     ("bottleneck_asm_c96_bstamp", False, dict(bstamp=True)), ("bottleneck_asm_c96_bstamp_stamped", True, dict(bstamp=True)),
                     ("bottleneck_asm_c96_pk", False, dict(pk=True)), ("bottleneck_asm_c96_pk_stamped", True, dict(pk=True)),
                     ("bottleneck_asm_c96_vgpra", False, dict(vgpra=True)), ("bottleneck_asm_c96_vgpra_stamped", True, dict(vgpra=True)),
                     ("bottleneck_asm_c96_nords", False, dict(nords=True)), ("bottleneck_asm_c96_nords_stamped", True, dict(nords=True)),
                     ("bottleneck_asm_c96_vgpra_nords", False, dict(vgpra=True, nords=True)),
                     ("bottleneck_asm_c96_vgpra_nords_stamped", True, dict(vgpra=True, nords=True))]
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
    print(f"bottleneck_asm_c96: {V.next} VGPRs (accum_offset {ACCUM_OFFSET}) + {NAGPR} AGPRs, {S.next} SGPRs, {LDS_BYTES} B LDS; wrote {path}: "
          f"{sum(1 for l in text if 'v_mfma' in l)} MFMA instructions")


if __name__ == "__main__":
    main()
