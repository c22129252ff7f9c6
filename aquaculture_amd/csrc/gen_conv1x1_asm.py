#!/usr/bin/env python3
"""Generator of the gfx950 assembly of the wide 1x1 convolution (conv1x1_asm_*): out[p][n] = SiLU(sum_k x[p][k] w[n][k] + b[n]), bf16.

The K >= 768 1x1 layers of yolov5m (C3 cv1|cv2 / cv3 at 20x20, SPPF cv1 / cv2, model.10, model.13 cv1|cv2: [UPSTREAM models/common.py C3,
SPPF; models/yolov5m.yaml]) ran on round 1's implicit-GEMM kernel at 0.51-0.70 PFLOP/s with 52-59 % of the wave cycles waiting
(profiles/r04_per_op_pmc.txt); the vendor GEMM reaches 0.62-1.0 PFLOP/s on the same shapes (tools/ubench/hipblaslt_1x1.cpp), held back by
tile quantisation: 25,600 pixels x 768 channels is 300 tiles of 256 x 256 on 256 CUs.  This kernel's tile is 208 pixels x 384 channels:
124 x 2 = 248 tiles for those layers -- one round.

  * Eight waves (two per SIMD, 256 registers each).  Wave w owns output channels 48 w .. 48 w + 47 of the tile's 384 (three 16-row M
    blocks) for all 13 pixel blocks: 39 accumulator tiles.  Its weights never touch LDS: the three A fragments of a k-step (32 input
    channels) come straight from L2 into one of three register sets, two k-steps ahead -- the planar 3x3 kernel's recipe.
  * The pixels go through LDS once per workgroup: chunks of 96 channels (three k-steps), three ring buffers.  A k-step's plane is the
    stride-2 / pixel-major families' format: 64 contiguous bytes per pixel (32 channels), 16 pixels = one 1 KB LDS-DMA instruction through a
    buffer descriptor (out-of-range lanes write zeros: ragged last tile), 16-byte channel group q of pixel p at position q ^ 2 ((p >> 2) & 1)
    so that every ds_read_b128 lane group covers all 64 banks.  One DMA instruction fills exactly one B fragment.
  * ONE loop body for every chunk of every tile: the loads run ahead of the arithmetic ACROSS tile boundaries (LDS-DMA two chunks ahead,
    weights two k-steps ahead, B fragments PD elements ahead), so a tile's epilogue is just a block between two bodies and every
    s_waitcnt vmcnt(N) is a constant.  Loads past the workgroup's last tile fetch zeros / the start of the weight image.
  * The barrier of a chunk sits PD + 1 elements before its end: by then every wave has ISSUED all its reads of the chunk (so its
    buffer may be refilled by the DMA issued in the next body) and waited for its own part of the next chunk, whose first fragments
    are read under the last MFMAs of this one.
  * Weight rows are permuted so that a lane's twelve outputs of a pixel are channels 8 g .. 8 g + 7 and 32 + 4 g .. 32 + 4 g + 3 of the
    wave's 48 (g = lane >> 4): one 16-byte and one 8-byte store per pixel block, 96 contiguous bytes per pixel and wave.

Register map (VGPRs only, 2 waves per SIMD):
  ACC  156  acc(m, j) = ACC + 4 (3 j + m)        A   36  three weight sets x three fragments        BR  24  B-fragment ring (six slots)
  T     12  epilogue temporaries                  addresses / lane offsets: see V.alloc below
Hazards the assembler does not pad (LLVM GCNHazardRecognizer, gfx940): MFMA result -> VALU read (s_nop 15), transcendental -> consumer
(independent instructions in between), s_mov m0 -> LDS-DMA (s_nop 0), VALU -> v_readfirstlane (s_nop 1), VALU-written SGPR -> VMEM.

Usage: python gen_conv1x1_asm.py OUT.s   (aquaculture_amd/build.py assembles it and embeds the code object in conv1x1_asm.hip)
"""
import os
import sys

NW = 8                       # waves
MB = 3                       # M blocks per wave
NT = 16 * MB * NW            # output channels per tile (384)
KS = 3                       # k-steps (32 channels) per chunk
RING = 3
MAX_COUT = 1536
BIAS_OFF = 0                 # the layer's biases (floats), then the ring (DS instruction offsets are 16 bits: the biases go first)
RB0 = 4 * MAX_COUT
PD = 4                       # B fragments read ahead
STEP_B = MB * 1024           # weight bytes per (wave, k-step)
NSLOT = 6
MAX_NDMA = 5
# Tile heights.  nb13: 208 pixels -- 25,600 pixels x 768 channels = 248 tiles, one round of 256 CUs.  nb7: 112 pixels, for the layers whose nb13
# grid would leave half the CUs idle (768 -> 384 at 20x20: 124 tiles of 208 pixels, 229 of 112); twice the weight bytes per MFMA.
# B-fragment ring: the body repeats every NE elements and a fragment is live for PD + 1 = 5 of them, so the slot of element e must be periodic
# in NE with any five consecutive elements in different slots: NE = 39 / 21 are not multiples of 5; six slots in runs of 6 and 5 do it (a window
# of five spans at most two runs: the tail t of one -- slots len - t .. len - 1 -- and the first 5 - t slots of the next; len >= 5).
RUNS = {13: (6, 6, 6, 6, 5, 5, 5), 7: (6, 5, 5, 5)}


def configure(nb):
    """Sets the tile constants of one family and allocates its vector registers."""
    g = globals()
    g["NB"] = nb
    g["TPX"] = 16 * nb
    g["HP"] = nb * 1024               # one k-step's plane of a chunk
    g["CH"] = KS * HP                 # one chunk buffer
    g["LDS_BYTES"] = RB0 + RING * CH
    g["NE"] = KS * nb                 # (k-step, pixel block) elements per chunk
    g["BAR_AT"] = NE - PD - 1
    g["NDMA"] = (NE + NW - 1) // NW   # LDS-DMA instructions per wave and chunk (a wave whose last index is past NE repeats its previous one)
    g["SLOT"] = [k for run in RUNS[nb] for k in range(run)]
    assert PD == 4 and NDMA <= MAX_NDMA and NW * (NDMA - 1) < NE <= NW * NDMA
    assert len(SLOT) == NE and all(len({SLOT[(e + d) % NE] for d in range(PD + 1)}) == PD + 1 for e in range(NE))
    V = Regs("v", 256)
    g["V"] = V
    V.alloc("tid")
    V.alloc("ACC", 4 * MB * nb, 4)
    V.alloc("A", 4 * MB * 3, 4)
    V.alloc("BR", 4 * NSLOT, 4)
    V.alloc("T", 12, 4)
    for nm in ("vlrd", "vrd", "vrdn", "vwl", "vdl", "vol1", "vol2", "vbl1", "vbl2", "va0", "va1"):
        V.alloc(nm)
    V.alloc("vdt", MAX_NDMA)


ARG = dict(inp=0, out=8, w=16, bias=24, in_ld=32, out_ld=36, npix=40, nchunks=44, ntiles=48, nct_log2=52, G=56, in_bytes=60,
           out_bytes=64, w_bytes=68, stream_b=72, cout=76, debug=80)
ARG_BYTES = 96
PH_PROLOGUE, PH_STREAM, PH_BARRIER, PH_EPILOGUE = range(4)
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


S = Regs("s", 100)

S.alloc("karg", 2)
S.alloc("wg")
S.alloc("pad0")
for nm in ("inp", "out", "w", "bias"):
    S.alloc(nm, 2, 2)
for nm in ("in_ld", "out_ld", "npix", "nchunks", "ntiles", "nct_log2", "G", "in_bytes", "out_bytes", "w_bytes", "stream_b", "cout"):
    S.alloc(nm)
S.alloc("debug", 2, 2)
S.alloc("srd_in", 4, 4)
S.alloc("srd_out", 4, 4)
S.alloc("srd_w", 4, 4)
for nm in ("wave", "tmp0", "tmp1", "tmp2", "tmp3", "tmp4", "tmp5", "wstream",
           "d_tile", "d_chunk", "d_src", "d_wb", "d_px0", "d_ct", "d_ok",          # stage D: the chunk whose pixels the LDS-DMA fetches (two ahead)
           "n_wb", "n_px0", "n_ct", "n_ok", "n_last",                              # stage N: the next chunk
           "c_wb", "c_px0", "c_ct", "c_ok", "c_last",                              # stage C: the chunk the MFMAs run on
           "dbuf", "rbufn", "orow", "boff", "oguard", "after_epi"):
    S.alloc(nm)
S.alloc("dsrc", MAX_NDMA)
S.alloc("dlds", MAX_NDMA)
S.alloc("klog2e2", 2, 2)
S.alloc("kone2", 2, 2)
S.alloc("t64", 2, 2)
S.alloc("st_last", 2, 2)
S.alloc("st_acc", 2 * NPH, 2)



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


def acc(m, j):
    return vr("ACC", 4 * (MB * j + m), 4)


def afrag(set_, m):
    return vr("A", 4 * (MB * set_ + m), 4)


def bfrag(e):
    return vr("BR", 4 * SLOT[e % NE], 4)


out = []
_uid = [0]
STAMPED = [False]
OPT = dict(nosilu=False, nomfma=False, nodma=False, nowl=False, nost=False, nords=False)


def E(line="", comment=None):
    op = line.split(" ")[0]
    if OPT["nomfma"] and op.startswith("v_mfma"):
        return
    if OPT["nodma"] and op == "buffer_load_dwordx4" and line.endswith(" lds"):
        return
    if OPT["nowl"] and op == "buffer_load_dwordx4" and not line.endswith(" lds"):
        return
    if OPT["nost"] and op.startswith("buffer_store"):
        return
    if OPT["nords"] and op == "ds_read_b128" and "BFRAG" in (comment or ""):
        return
    if OPT["nosilu"] and op in ("v_exp_f32", "v_rcp_f32", "v_pk_mul_f32", "v_pk_add_f32"):
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


def emit_decode_d():
    """Stage D's (d_tile, d_chunk) -> d_ok, d_px0, d_ct, d_src (byte offset of the chunk's first channel of pixel px0 in the input slice),
    d_wb (byte offset of the chunk's first k-step in this wave's weight stream).  No tile left: the pixels come from beyond the descriptor
    (zeros), the weights from the start of the image (never used)."""
    bad, done = uid("dbad"), uid("ddone")
    E(f"s_cmp_lt_u32 {s('d_tile')}, {s('ntiles')}")
    E(f"s_cselect_b32 {s('d_ok')}, 1, 0")
    E(f"s_cbranch_scc0 {bad}")
    E(f"s_lshr_b32 {s('tmp0')}, {s('d_tile')}, {s('nct_log2')}", "pixel tile")
    E(f"s_lshl_b32 {s('tmp1')}, {s('tmp0')}, {s('nct_log2')}")
    E(f"s_sub_u32 {s('d_ct')}, {s('d_tile')}, {s('tmp1')}", "channel tile")
    E(f"s_mul_i32 {s('d_px0')}, {s('tmp0')}, {TPX}")
    E(f"s_mul_i32 {s('d_src')}, {s('d_px0')}, {s('in_ld')}")
    E(f"s_mul_i32 {s('tmp1')}, {s('d_chunk')}, {64 * KS}")
    E(f"s_add_u32 {s('d_src')}, {s('d_src')}, {s('tmp1')}")
    E(f"s_lshl_b32 {s('tmp1')}, {s('d_ct')}, 3")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('wave')}")
    E(f"s_mul_i32 {s('d_wb')}, {s('tmp1')}, {s('stream_b')}")
    E(f"s_mul_i32 {s('tmp1')}, {s('d_chunk')}, {KS * STEP_B}")
    E(f"s_add_u32 {s('d_wb')}, {s('d_wb')}, {s('tmp1')}")
    E(f"s_branch {done}")
    label(bad)
    E(f"s_mov_b32 {s('d_src')}, {s('oguard')}")
    E(f"s_mov_b32 {s('d_wb')}, 0")
    E(f"s_mov_b32 {s('d_px0')}, 0")
    E(f"s_mov_b32 {s('d_ct')}, 0")
    label(done)


def emit_dma(dst):
    """Stage D's chunk -> the ring buffer at LDS byte `dst` (an SGPR name or an integer): this wave's five 1 KB instructions."""
    for n in range(NDMA):
        E(f"s_add_u32 {s('tmp0')}, {s('d_src')}, {s('dsrc', n)}")
        E(f"v_add_u32 {v('vdt', n)}, {s('tmp0')}, {v('vdl')}")
    for n in range(NDMA):
        if isinstance(dst, int):
            E(f"s_add_u32 m0, {s('dlds', n)}, {dst}")
        else:
            E(f"s_add_u32 m0, {s('dlds', n)}, {s(dst)}")
        E("s_nop 0", "hz: m0 write -> LDS-DMA")
        E(f"buffer_load_dwordx4 {v('vdt', n)}, {s4('srd_in')}, 0 offen lds")


def emit_wloads(set_, base, add):
    """The three A fragments of one k-step -> register set `set_`; stream offset = SGPR `base` + `add` bytes."""
    E(f"s_add_u32 {s('tmp0')}, {s(base)}, {add}")
    for m in range(MB):
        E(f"buffer_load_dwordx4 {afrag(set_, m)}, {v('vwl')}, {s4('srd_w')}, {s('tmp0')} offen offset:{1024 * m}")


def elem_off(e):
    return (e // NB) * HP + (e % NB) * 1024


def emit_bread(e):
    """B fragment of element e of stage C's chunk (e >= NE: element e - NE of stage N's chunk) -> ring slot e % (PD + 1)."""
    if e < NE:
        E(f"ds_read_b128 {bfrag(e)}, {v('vrd')} offset:{elem_off(e)}", "BFRAG")
    else:
        E(f"ds_read_b128 {bfrag(e)}, {v('vrdn')} offset:{elem_off(e - NE)}", "BFRAG")


def emit_body():
    for e in range(NE):
        ks, j = divmod(e, NB)
        if j == 0:
            # weights two k-steps ahead; then this k-step's set must have landed.  Vector-memory operations return in order; issued behind the
            # awaited loads: k-step 0: 3 + 3; k-step 1: 3 + NDMA (LDS-DMA) + 3; k-step 2: NDMA + 3 + 3.  After an epilogue its 2 NB stores sit in
            # between: the same waits with + 2 NB, or they would wait for the stores to be written
            if ks == 0:
                emit_wloads(2, "c_wb", 2 * STEP_B)
            else:
                emit_wloads(ks - 1, "n_wb", (ks - 1) * STEP_B)
            n = 6 if ks == 0 else 6 + NDMA
            if ks < 2:
                late, join = uid("late"), uid("join")
                E(f"s_cmp_eq_u32 {s('after_epi')}, 1")
                E(f"s_cbranch_scc1 {late}")
                E(f"s_waitcnt vmcnt({n})")
                E(f"s_branch {join}")
                label(late)
                E(f"s_waitcnt vmcnt({n + 2 * NB})")
                label(join)
            else:
                E(f"s_waitcnt vmcnt({n})")
            if ks == 0:
                emit_dma("dbuf")
        emit_bread(e + PD)
        E(f"s_waitcnt lgkmcnt({PD})")
        for m in range(MB):
            E(f"v_mfma_f32_16x16x32_bf16 {acc(m, j)}, {afrag(ks, m)}, {bfrag(e)}, {acc(m, j)}")
        if e == BAR_AT:
            # stage N's pixels: this wave's part was issued one body ago; behind it 3 + 3, then this body's 3 + NDMA + 3 + 3 (+ the stores)
            stamp(PH_STREAM)
            late, join = uid("late"), uid("join")
            E(f"s_cmp_eq_u32 {s('after_epi')}, 1")
            E(f"s_cbranch_scc1 {late}")
            E(f"s_waitcnt vmcnt({15 + NDMA})")
            E(f"s_branch {join}")
            label(late)
            E(f"s_waitcnt vmcnt({15 + NDMA + 2 * NB})")
            label(join)
            E("s_barrier")
            stamp(PH_BARRIER)
    E(f"s_mov_b32 {s('after_epi')}, 0")


def emit_silu(regs, temps):
    """temps <- 1 / (1 + exp(-regs)) for twelve registers (batched: independent transcendentals back to back)."""
    n = len(regs)
    for i in range(0, n, 2):
        E(f"v_pk_mul_f32 v[{temps[i]}:{temps[i] + 1}], v[{regs[i]}:{regs[i] + 1}], {s2('klog2e2')}")
    for i in range(n):
        E(f"v_exp_f32 v{temps[i]}, v{temps[i]}")
    for i in range(0, n, 2):
        E(f"v_pk_add_f32 v[{temps[i]}:{temps[i] + 1}], v[{temps[i]}:{temps[i] + 1}], {s2('kone2')}")
    for i in range(n):
        E(f"v_rcp_f32 v{temps[i]}, v{temps[i]}")


def emit_epilogue():
    """Stage C's tile is complete: SiLU, bf16, store (13 pixel blocks x (16 + 8 bytes per lane)).  The B ring and the weight sets already hold
    the next tile's first operands, so the temporaries are its own twelve registers."""
    A0 = V.names["ACC"][0]
    T0 = V.names["T"][0]
    temps = [T0 + i for i in range(12)]
    E("s_nop 15", "hz: MFMA result -> VALU read")
    E("s_nop 3")
    # byte offset of (pixel px0, channel 384 ct + 48 wave) in the output slice
    E(f"s_mul_i32 {s('orow')}, {s('c_px0')}, {s('out_ld')}")
    E(f"s_mul_i32 {s('tmp0')}, {s('c_ct')}, {2 * NT}")
    E(f"s_add_u32 {s('orow')}, {s('orow')}, {s('tmp0')}")
    E(f"s_mul_i32 {s('tmp0')}, {s('wave')}, {2 * 16 * MB}")
    E(f"s_add_u32 {s('orow')}, {s('orow')}, {s('tmp0')}")
    E(f"s_lshl_b32 {s('tmp1')}, {s('out_ld')}, 4", "16 pixels")
    for j in range(NB):
        regs = [A0 + 4 * MB * j + k for k in range(12)]
        emit_silu(regs, temps)
        for k in range(0, 12, 2):
            E(f"v_pk_mul_f32 v[{regs[k]}:{regs[k] + 1}], v[{regs[k]}:{regs[k] + 1}], v[{temps[k]}:{temps[k] + 1}]")
        for k in range(6):
            E(f"v_cvt_pk_bf16_f32 v{temps[k]}, v{regs[2 * k]}, v{regs[2 * k] + 1}")
        E(f"v_add_u32 {v('va0')}, {s('orow')}, {v('vol1')}")
        E(f"v_add_u32 {v('va1')}, {s('orow')}, {v('vol2')}")
        E(f"buffer_store_dwordx4 v[{temps[0]}:{temps[3]}], {v('va0')}, {s4('srd_out')}, 0 offen")
        E(f"buffer_store_dwordx2 v[{temps[4]}:{temps[5]}], {v('va1')}, {s4('srd_out')}, 0 offen offset:64")
        if j + 1 < NB:
            E(f"s_add_u32 {s('orow')}, {s('orow')}, {s('tmp1')}")
    E(f"s_mov_b32 {s('after_epi')}, 1")


def emit_acc_init(ct):
    """Accumulators <- the biases of channel tile `ct` (an SGPR name): the wave's 48 floats sit in LDS; lane group g takes 8 g .. 8 g + 7 and 32 + 4 g .. + 3."""
    E(f"s_mul_i32 {s('boff')}, {s(ct)}, {4 * NT}")
    E(f"s_mul_i32 {s('tmp0')}, {s('wave')}, {4 * 16 * MB}")
    E(f"s_add_u32 {s('boff')}, {s('boff')}, {s('tmp0')}")
    E(f"v_add_u32 {v('va0')}, {s('boff')}, {v('vbl1')}")
    E(f"v_add_u32 {v('va1')}, {s('boff')}, {v('vbl2')}")
    E(f"ds_read_b128 {acc(0, 0)}, {v('va0')} offset:{BIAS_OFF}")
    E(f"ds_read_b128 {acc(1, 0)}, {v('va0')} offset:{BIAS_OFF + 16}")
    E(f"ds_read_b128 {acc(2, 0)}, {v('va1')} offset:{BIAS_OFF}")
    E("s_waitcnt lgkmcnt(0)")
    A0 = V.names["ACC"][0]
    for j in range(1, NB):
        for k in range(12):
            E(f"v_mov_b32 v{A0 + 12 * j + k}, v{A0 + k}")


_kernel_no = [0]


def gen_kernel(name, nb, stamped=False, **opt):
    global out
    out = []
    configure(nb)
    STAMPED[0] = stamped
    for k in OPT:
        OPT[k] = False
    OPT.update(opt)
    _kernel_no[0] += 1
    _uid[0] = 100000 * _kernel_no[0]
    E(f"; wide 1x1 convolution, {TPX} pixels x {NT} channels per tile, 8 waves: generated by gen_conv1x1_asm.py -- do not edit")
    label(name)
    a0 = S.names["inp"][0]
    b0 = S.names["in_ld"][0]
    assert a0 % 4 == 0 and b0 == a0 + 8 and S.names["debug"][0] == b0 + 12
    E(f"s_load_dwordx8 s[{a0}:{a0 + 7}], {s2('karg')}, 0x0", "inp, out, w, bias")
    E(f"s_load_dwordx8 s[{b0}:{b0 + 7}], {s2('karg')}, 0x20", "in_ld .. in_bytes")
    E(f"s_load_dwordx4 s[{b0 + 8}:{b0 + 11}], {s2('karg')}, 0x40", "out_bytes, w_bytes, stream_b, cout")
    E(f"s_load_dwordx2 {s2('debug')}, {s2('karg')}, 0x50")
    T = [V.names["T"][0] + i for i in range(12)]
    lane, p15, g = T[0], T[1], T[2]
    E(f"v_and_b32 v{lane}, 63, {v('tid')}", "lane")
    E(f"v_lshrrev_b32 v{T[3]}, 6, {v('tid')}")
    E("s_nop 1", "hz: VALU write -> v_readfirstlane")
    E(f"v_readfirstlane_b32 {s('wave')}, v{T[3]}")
    E(f"v_and_b32 v{p15}, 15, v{lane}")
    E(f"v_lshrrev_b32 v{g}, 4, v{lane}")
    E("s_waitcnt lgkmcnt(0)")
    for nm, base, size in (("srd_in", "inp", "in_bytes"), ("srd_out", "out", "out_bytes"), ("srd_w", "w", "w_bytes")):
        E(f"s_mov_b32 {s(nm, 0)}, {s(base)}")
        E(f"s_and_b32 {s(nm, 1)}, {s(base, 1)}, 0xffff")
        E(f"s_mov_b32 {s(nm, 2)}, {s(size)}")
        E(f"s_mov_b32 {s(nm, 3)}, 0x00020000")
    E(f"s_mov_b32 {s('klog2e2')}, 0xbfb8aa3b", "-log2(e)")
    E(f"s_mov_b32 {s('klog2e2', 1)}, 0xbfb8aa3b")
    E(f"s_mov_b32 {s('kone2')}, 1.0")
    E(f"s_mov_b32 {s('kone2', 1)}, 1.0")
    E(f"s_mov_b32 {s('after_epi')}, 0")
    # pixels past the last tile: an offset beyond the input descriptor (lane parts added on top stay below 2^32: the host keeps in_bytes < 2^31)
    E(f"s_add_u32 {s('oguard')}, {s('in_bytes')}, 0x100")
    if stamped:
        for k in range(2 * NPH):
            E(f"s_mov_b32 {s('st_acc', k)}, 0")
        E(f"s_memtime {s2('st_last')}")
        E("s_waitcnt lgkmcnt(0)")
    # ---- per-lane offsets ----
    # B fragment: pixel p = lane & 15 at 64 p, channel group g = lane >> 4 at position g ^ 2 ((p >> 2) & 1)
    E(f"v_lshrrev_b32 v{T[3]}, 2, v{p15}")
    E(f"v_and_b32 v{T[3]}, 1, v{T[3]}")
    E(f"v_lshlrev_b32 v{T[3]}, 1, v{T[3]}", "2 b")
    E(f"v_xor_b32 v{T[3]}, v{T[3]}, v{g}")
    E(f"v_lshlrev_b32 v{T[3]}, 4, v{T[3]}")
    E(f"v_lshl_add_u32 {v('vlrd')}, v{p15}, 6, v{T[3]}")
    E(f"v_add_u32 {v('vrd')}, {RB0}, {v('vlrd')}", "stage C: ring buffer 0")
    E(f"v_add_u32 {v('vrdn')}, {RB0 + CH}, {v('vlrd')}", "stage N: ring buffer 1")
    E(f"v_lshlrev_b32 {v('vwl')}, 4, v{lane}")
    # LDS-DMA: lane i fills slot i of the instruction's 1 KB: pixel i >> 2, position i & 3 = channel group (i & 3) ^ 2 b of that pixel
    E(f"v_lshrrev_b32 v{T[3]}, 2, v{lane}", "pixel of the block")
    E(f"v_and_b32 v{T[4]}, 3, v{lane}")
    E(f"v_lshrrev_b32 v{T[5]}, 2, v{T[3]}")
    E(f"v_and_b32 v{T[5]}, 1, v{T[5]}")
    E(f"v_lshlrev_b32 v{T[5]}, 1, v{T[5]}")
    E(f"v_xor_b32 v{T[4]}, v{T[4]}, v{T[5]}")
    E(f"v_mul_lo_u32 v{T[3]}, v{T[3]}, {s('in_ld')}")
    E(f"v_lshl_add_u32 {v('vdl')}, v{T[4]}, 4, v{T[3]}")
    # output: pixel p of the block; the 16-byte piece at byte 16 g of the wave's 96, the 8-byte piece at 64 + 8 g
    E(f"v_mul_lo_u32 v{T[3]}, v{p15}, {s('out_ld')}")
    E(f"v_lshl_add_u32 {v('vol1')}, v{g}, 4, v{T[3]}")
    E(f"v_lshl_add_u32 {v('vol2')}, v{g}, 3, v{T[3]}")
    # biases in LDS: floats 8 g .. 8 g + 7 and 32 + 4 g .. + 3 of the wave's 48
    E(f"v_lshlrev_b32 {v('vbl1')}, 5, v{g}")
    E(f"v_lshlrev_b32 {v('vbl2')}, 4, v{g}")
    E(f"v_add_u32 {v('vbl2')}, 128, {v('vbl2')}")
    # this wave's LDS-DMA instructions: i = wave + 8 n (past the chunk's NE: the wave's previous one again); k-step i / NB, pixel block i % NB
    for n in range(NDMA):
        E(f"s_add_u32 {s('tmp0')}, {s('wave')}, {8 * n}")
        if NW * n + NW - 1 >= NE:
            E(f"s_cmp_ge_u32 {s('tmp0')}, {NE}")
            E(f"s_cselect_b32 {s('tmp1')}, 8, 0")
            E(f"s_sub_u32 {s('tmp0')}, {s('tmp0')}, {s('tmp1')}")
        E(f"s_cmp_ge_u32 {s('tmp0')}, {NB}")
        E(f"s_cselect_b32 {s('tmp1')}, 1, 0")
        E(f"s_cmp_ge_u32 {s('tmp0')}, {2 * NB}")
        E(f"s_cselect_b32 {s('tmp2')}, 1, 0")
        E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {s('tmp2')}", "k-step")
        E(f"s_mul_i32 {s('tmp2')}, {s('tmp1')}, {NB}")
        E(f"s_sub_u32 {s('tmp2')}, {s('tmp0')}, {s('tmp2')}", "pixel block")
        E(f"s_mul_i32 {s('tmp3')}, {s('tmp1')}, {HP}")
        E(f"s_lshl_b32 {s('tmp4')}, {s('tmp2')}, 10")
        E(f"s_add_u32 {s('dlds', n)}, {s('tmp3')}, {s('tmp4')}")
        E(f"s_lshl_b32 {s('tmp3')}, {s('tmp2')}, 4")
        E(f"s_mul_i32 {s('tmp3')}, {s('tmp3')}, {s('in_ld')}")
        E(f"s_lshl_b32 {s('tmp4')}, {s('tmp1')}, 6")
        E(f"s_add_u32 {s('dsrc', n)}, {s('tmp3')}, {s('tmp4')}")
    # ---- first tile: XCD-aware bijective map (workgroups sharing an XCD get consecutive tiles: the channel tiles of a pixel tile share its pixels in L2) ----
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
    # ---- pipeline fill: chunk q0 -> stage C (ring buffer 0), q1 -> stage N (buffer 1), q2 -> stage D ----
    E(f"s_mov_b32 {s('d_chunk')}, 0")
    emit_decode_d()
    emit_dma(RB0)
    emit_wloads(0, "d_wb", 0)
    emit_wloads(1, "d_wb", STEP_B)
    emit_stage_copy("c", "d")
    emit_advance_d()
    emit_dma(RB0 + CH)
    emit_stage_copy("n", "d")
    emit_advance_d()
    E(f"s_mov_b32 {s('dbuf')}, {RB0 + 2 * CH}")
    E(f"s_mov_b32 {s('rbufn')}, {RB0 + CH}")
    # ---- the biases -> LDS (all `cout` floats, 16 bytes per thread), fetched behind the first chunks' loads: one memory latency for all ----
    E(f"v_lshlrev_b32 v{T[3]}, 4, {v('tid')}")
    E(f"s_lshl_b32 {s('tmp0')}, {s('cout')}, 2")
    E(f"v_cmp_gt_u32 vcc, {s('tmp0')}, v{T[3]}")
    E(f"s_and_saveexec_b64 {s2('t64')}, vcc")
    E(f"global_load_dwordx4 v[{T[4]}:{T[7]}], v{T[3]}, {s2('bias')}")
    E("s_waitcnt vmcnt(0)")
    E(f"ds_write_b128 v{T[3]}, v[{T[4]}:{T[7]}] offset:{BIAS_OFF}")
    E(f"s_mov_b64 exec, {s2('t64')}")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    emit_acc_init("c_ct")
    for e in range(PD):
        emit_bread(e)
    stamp(PH_PROLOGUE)
    loop = uid("loop")
    label(loop)
    emit_body()
    noepi = uid("noepi")
    E(f"s_cmp_eq_u32 {s('c_last')}, 0")
    E(f"s_cbranch_scc1 {noepi}")
    stamp(PH_STREAM)
    emit_epilogue()
    noinit = uid("noinit")
    E(f"s_cmp_eq_u32 {s('n_ok')}, 0", "no tile follows: nothing to initialise")
    E(f"s_cbranch_scc1 {noinit}")
    emit_acc_init("n_ct")
    label(noinit)
    stamp(PH_EPILOGUE)
    label(noepi)
    # rotate the stages and the ring
    emit_stage_copy("c", "n")
    emit_stage_copy("n", "d")
    emit_advance_d()
    for nm in ("dbuf", "rbufn"):
        E(f"s_add_u32 {s(nm)}, {s(nm)}, {CH}")
        E(f"s_cmp_eq_u32 {s(nm)}, {RB0 + RING * CH}")
        E(f"s_cselect_b32 {s(nm)}, {RB0}, {s(nm)}")
    E(f"v_mov_b32 {v('vrd')}, {v('vrdn')}")
    E(f"v_add_u32 {v('vrdn')}, {s('rbufn')}, {v('vlrd')}")
    E(f"s_cmp_eq_u32 {s('c_ok')}, 1")
    E(f"s_cbranch_scc1 {loop}")
    label(f".Lend_{name}")
    E("s_waitcnt vmcnt(0)", "nothing of this workgroup may still be on its way to LDS or memory")
    E("s_waitcnt lgkmcnt(0)")
    if stamped:
        T3 = V.names["T"][0]
        E(f"s_lshl_b32 {s('tmp0')}, {s('wg')}, 3")
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


def emit_stage_copy(dst, src):
    """Stage `dst` <- stage `src` (c <- n, n <- d, and the pipeline fill's c <- d)."""
    E(f"s_mov_b32 {s(dst + '_wb')}, {s(src + '_wb')}")
    E(f"s_mov_b32 {s(dst + '_px0')}, {s(src + '_px0')}")
    E(f"s_mov_b32 {s(dst + '_ct')}, {s(src + '_ct')}")
    E(f"s_mov_b32 {s(dst + '_ok')}, {s(src + '_ok')}")
    if src == "d":
        E(f"s_add_u32 {s('tmp0')}, {s('d_chunk')}, 1")
        E(f"s_cmp_eq_u32 {s('tmp0')}, {s('nchunks')}")
        E(f"s_cselect_b32 {s(dst + '_last')}, 1, 0")
        E(f"s_and_b32 {s(dst + '_last')}, {s(dst + '_last')}, {s('d_ok')}")
    else:
        E(f"s_mov_b32 {s(dst + '_last')}, {s(src + '_last')}")


def emit_advance_d():
    """Stage D moves on one chunk: the next 96 channels of its tile, or the first chunk of this workgroup's next tile."""
    same, done = uid("same"), uid("adv")
    E(f"s_add_u32 {s('d_chunk')}, {s('d_chunk')}, 1")
    E(f"s_cmp_lt_u32 {s('d_chunk')}, {s('nchunks')}")
    E(f"s_cbranch_scc1 {same}")
    E(f"s_mov_b32 {s('d_chunk')}, 0")
    E(f"s_add_u32 {s('d_tile')}, {s('d_tile')}, {s('G')}")
    emit_decode_d()
    E(f"s_branch {done}")
    label(same)
    E(f"s_cmp_eq_u32 {s('d_ok')}, 0")
    E(f"s_cbranch_scc1 {done}")
    E(f"s_add_u32 {s('d_src')}, {s('d_src')}, {64 * KS}")
    E(f"s_add_u32 {s('d_wb')}, {s('d_wb')}, {KS * STEP_B}")
    label(done)


def descriptor(name):
    total = (V.next + 7) // 8 * 8
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
    path = sys.argv[1] if len(sys.argv) > 1 else "conv1x1_asm.s"
    text = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.text"]
    entries = []
    # the shipped kernel and its stamped build; with AQ_GEN_EXPERIMENTAL=1 also timing-only ablations (AQ_C1_ASM_KERNEL=<name>; wrong results)
    variants = [(f"conv1x1_asm_nb{nb}{'_stamped' if st else ''}", nb, st, {}) for nb in (13, 7) for st in (False, True)]
    if os.environ.get("AQ_GEN_EXPERIMENTAL") == "1":
        variants += [("conv1x1_asm_nb13_nosilu", 13, False, dict(nosilu=True)), ("conv1x1_asm_nb13_nomfma", 13, False, dict(nomfma=True)),
                     ("conv1x1_asm_nb13_nodma", 13, False, dict(nodma=True)), ("conv1x1_asm_nb13_nowl", 13, False, dict(nowl=True)),
                     ("conv1x1_asm_nb13_nost", 13, False, dict(nost=True)), ("conv1x1_asm_nb13_nords", 13, False, dict(nords=True)),
                     ("conv1x1_asm_nb13_mfmaonly", 13, False, dict(nosilu=True, nodma=True, nowl=True, nost=True, nords=True))]
    report = []
    for name, nb, stamped, opt in variants:
        text += [f"\t.globl\t{name}", "\t.p2align\t8", f"\t.type\t{name},@function"]
        text += gen_kernel(name, nb, stamped, **opt)
        entries.append(metadata_entry(name))
        text += [f".Lfend_{name}:", f"\t.size\t{name}, .Lfend_{name}-{name}", descriptor(name)]
        if not stamped and not opt:
            report.append(f"{name}: {V.next} VGPRs, {LDS_BYTES} B LDS")
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
    print("; ".join(report) + f"; {S.next} SGPRs; wrote {path}: {sum(1 for l in text if 'v_mfma' in l)} MFMA instructions")


if __name__ == "__main__":
    main()
