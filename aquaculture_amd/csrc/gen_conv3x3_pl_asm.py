#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 assembly of the planar 3x3/s1 convolution (conv3x3_pl_asm_*).

Same algorithm, data layout and LDS-DMA ring as the HIP-source kernel in conv3x3_pl.hip (read its header first) -- that kernel is the
specification and the fallback.  Why assembly: in the HIP kernel every vector-memory operation is inline asm with hand-counted
s_waitcnt, and the compiler's register allocator then sets the limits (it spills or copies registers that an asm load is still
filling, tests/test_build_isa.py): the residual cannot be prefetched under the last chunk, waits cannot depend on run-time flags, and
the epilogue waits for memory twice per tile (stamped build: epilogue 15 k of a tile's 75 k cycles, chunk tops 6 k).
Here every register is assigned by hand:

  a[0 : 12 NB)           accumulators, acc(i, j) = a[4 (3 j + i) : +4]  (M block i of the wave, pixel block j)
  v[A]   72 registers    weight fragments of tap-steps T, T + 1, T + 2 (loaded two tap-steps ahead, straight from L2)
  v[B]   4 (PD + 1)      rotating B fragments, read PD elements ahead of their MFMAs
  v[R]   6 NB            the tile's whole residual, fetched at the top of its LAST chunk (lands under that chunk's MFMAs)
  v[addr] NB             LDS byte address of this lane's B fragment per pixel block, advanced in place from tap to tap
  (pixel-major families since round 4's second session: the 72 weight registers sit BEHIND the accumulators in the accumulator file --
   MFMA reads srcA from there at no cost -- and the residual in VGPRs: A_ACC below; v[cst] 4 registers hold M block 2's packed outputs of
   two pixel blocks for their shared 16-byte store: W16 below)

and every s_waitcnt vmcnt(N) is exact (N = vector-memory operations issued after the awaited one; the only run-time dependence, "+3 NB
at taps 0 and 1 after an epilogue or a residual fetch", is a scalar branch between two immediates).

Hazards the assembler does not pad (LLVM GCNHazardRecognizer, gfx940 family): s_mov m0 -> LDS-DMA 1 state; VALU write -> v_readfirstlane
1; VALU-written SGPR -> VMEM 5 / -> VALU 2; transcendental -> consumer 1; MFMA result -> VALU read up to 19 (s_nop 15 twice before the
epilogue); v_accvgpr_write -> MFMA 2.  Each is marked `hz:` where it is handled.

Usage: python gen_conv3x3_pl_asm.py OUT.s   (aquaculture_amd/build.py assembles OUT.s with clang and embeds the code object)
"""
import sys

# Two families.  "nb13": one workgroup per CU (one wave per SIMD, the whole register file), three ring buffers, the residual in its own
# registers fetched under the last chunk.  "nb7" / "nb8": TWO workgroups per CU (<= 256 registers per wave, <= 80 KB LDS), so that one
# workgroup's chunk tops, epilogue and prologue run under the other's MFMAs; two ring buffers of 256 region rows, and the residual is
# fetched at the start of the epilogue into registers the stream has finished with (weight set 2 and the B ring).
# (Options measured in round 2 and REMOVED from the generator in round 3 -- the notes are the record of why.)
# SPLIT = s > 0: a tile's LAST chunk has its own code: all 18 half-taps for pixel blocks 0 .. s - 1 first, then again for blocks s .. NB - 1
# with the epilogue of the first group (accumulators final) issued between those MFMAs -- the epilogue is VALU-bound (v_exp + v_rcp at
# quarter rate, 10 k cycles per tile) and one wave per SIMD has nothing else to overlap it with; this way it also pays for workgroups
# that own a single tile, which a second accumulator set (epilogue under the NEXT tile) would not.  Costs: the last chunk's weights are
# streamed twice, and its own copy of the stream code.
# MEASURED (SPLIT = 7, parity-green on all ten geometries; stamped build, cycles per wave): 192 ch @ 40x40: epilogue 19.5 k -> 9.9 k but
# stream 99.2 k -> 113.1 k, lifetime 140.1 k -> 143.9 k; 384 ch @ 20x20: epilogue 10.4 k -> 5.1 k, stream 93.0 k -> 99.2 k, lifetime
# 117.2 k -> 118.0 k.  The instructions put between the MFMAs are not free: with one wave per SIMD the stream is bound by what the wave
# can issue and by in-order waits, so every filler lengthens it by about its own issue time (the same reason the fp8 weight stream's
# conversions lose).  Off; kept as a generator option because it is the measurement behind "the epilogue cannot be hidden here".
# DMA_FRONT: all 2 NG LDS-DMA instructions of a chunk in its FIRST tap, behind that tap's weight loads (one per element), instead of two per
# tap over the first NG taps.  Loads return in order: a weight load issued behind an LDS-DMA (HBM / MALL latency) cannot land before it,
# so with the DMA spread out every tap's weights could be held back; in front, only the weights of one tap per chunk sit behind it.
# MEASURED: stream 92.8 k vs 90.1 k cycles (192 ch), 83.1 k vs 80.7 k (384 ch) -- worse; twelve requests at once queue behind each other.
# Also measured on the weight loads: cache policy nt 94.4 k (worse), sc1 90.2 k (same) against 90.1 k (AQ_GEN_A_POLICY at build time).
# LOOK = how many taps ahead the weights are loaded (LOOK + 1 fragment sets of 24 registers).  Loads return in order, so every LDS-DMA or
# residual load (HBM / MALL latency) holds back the weight loads issued behind it: with LOOK = 2 such a load has two taps (2.5 k cycles)
# before it stalls the stream, and the stamped builds showed exactly that stall; the one-workgroup family keeps its FOUR sets in the
# accumulator half of the register file (100 AGPRs are free there), which also frees 72 VGPRs for a deeper B ring.
# KS = k-steps (of 32 channels) per chunk: 2 = the 64-channel chunks of the stride-1 families; 1 = 32-channel chunks.
# S2 = the stride-2 family (round 3).  A 3x3 / stride 2 / pad 1 convolution reads input pixel (2 oy + dy - 1, 2 ox + dx - 1): split the input
# into its four PARITY planes (y odd / even x x odd / even), each an image of the OUTPUT's size, and tap (dy, dx) reads plane
# (dy != 1, dx != 1) at output pixel (oy - (dy == 0), ox - (dx == 0)) -- offsets -1 and 0 only.  So the region uses the stride-1 families'
# padded OUTPUT coordinates (one zero pixel after every row, one zero row after every image; region row 0 = the upper left neighbour
# of the tile's first pixel), once per parity plane, and wave w loads parity plane w.
# The first build kept the stride-1 families' slot-major planes (one LDS-DMA instruction = 64 rows x 16 B of one channel group): correct,
# and bound by the vector-memory front end -- every lane of every instruction touches a different cache line, a chunk needs 20
# instructions per wave for 351 MFMAs, and the stamped build without the DMA ran its stream in half the time (284 k -> 146 k cycles,
# model.7).  Hence the layout of THIS family: a region row is the 64 contiguous bytes of a pixel's 32-channel chunk; one LDS-DMA
# instruction moves 16 rows x 64 B (four lanes per pixel: a quarter of the cache-line look-ups), through a buffer descriptor whose
# out-of-range lanes write ZEROS to LDS (measured: tools/ubench/lds_dma_buffer_oob.hip) -- the padding rows cost no zero page, no 64-bit
# address arithmetic and no select: per instruction one s_add m0 and one buffer_load ... lds with a per-tile offset register.  A
# pixel-major row would make the fragment reads 4-way bank-conflicted (16 lanes, 64 B apart), so the 16-byte channel group q of region
# row r sits at position q ^ 2 ((r >> 2) & 1) of its row (byte address bit 5 ^= bit 8): with ds_read_b128's real lane groups
# ({0-3, 12-15, 20-27}, ...: two channel groups per group) every group then covers all 64 banks exactly once, for any starting row.
# The LDS-DMA writes lane-linearly, so the XOR is applied on the SOURCE side (which channel group a lane fetches) and on the read side
# (four address registers per pixel block -- one per (row, column) offset of the taps, since an XOR does not commute with the add).
# ROWS = 304 = 208 pixels + row pads + an image seam + one-sided halo for W = 20 and 40; two ring buffers (2 x 76 KB + bias = 156 KB).
CONFIGS = {
    13: dict(NB=13, PD=8, ROWS=384, RING=3, OCC=1, RES_EARLY=True, LOOK=2, KS=2, S2=False),
    7: dict(NB=7, PD=8, ROWS=256, RING=2, OCC=2, RES_EARLY=False, LOOK=2, KS=2, S2=False),
    8: dict(NB=8, PD=7, ROWS=256, RING=2, OCC=2, RES_EARLY=False, LOOK=2, KS=2, S2=False),
    "s2nb13": dict(NB=13, PD=8, ROWS=304, RING=2, OCC=1, RES_EARLY=True, LOOK=2, KS=1, S2=True),
    # F8 = the fp8 family of the stride-1 kernel (BASELINE.json configs[3], "fp8 weights (CDNA4 fp8 MFMA)"): OCP e4m3 on BOTH MFMA
    # operands, v_mfma_f32_16x16x128_f8f6f4 (33.5 cycles for 4x the K of the 16-cycle bf16 form: tools/ubench/mfma_f8_layout.hip).
    # A 64-channel chunk is 64 BYTES per pixel, so K = 128 spans two TAPS: MFMA step p multiplies taps 2 p and 2 p + 1 (step 4: tap 8
    # and zeros), operand lanes 0-31 holding the first tap's 64 channels and lanes 32-63 the second's -- the B fragment is simply read
    # at a per-lane tap offset.  Five steps per chunk instead of eighteen half-taps.  Region rows are the stride-2 family's: the pixel's
    # 64 contiguous bytes, 16 rows per LDS-DMA instruction through a buffer descriptor, byte address bit 5 ^= bit 8 (a lane's 32 bytes
    # stay contiguous; the two 16-byte reads of a fragment are 2-way bank conflicted, which the 96-cycle MFMA triple covers).  Each
    # fragment address is computed where it is read (add, shift, and, xor): nine tap offsets would cost 117 address registers.
    # LOOK = 4: five weight sets, one per step, each reloaded for the next chunk as soon as its step is done.  The residual waits in
    # the accumulator half of the register file (78 of the 100 registers free there).
    "f8nb13": dict(NB=13, PD=4, ROWS=384, RING=3, OCC=1, RES_EARLY=True, LOOK=4, KS=1, S2=False, F8=True),
    # PM = the PIXEL-MAJOR build of the stride-1 bf16 kernel (round 3).  Family nb13's slot-major planes make every lane of an LDS-DMA
    # instruction touch a different 128-byte line (64 tag look-ups for 1 KB); the stamped ablations that round the source rows to
    # multiples of eight / to one row (abl 16 / 32: same instructions, 8 / 1 lines each) run the stream in 83.0 k / 81.6 k cycles
    # against 90.3 k (no LDS-DMA at all: 81.3 k) and the prologue in 4.3 k against 9.9 k: the look-ups, not the bytes, are the cost.
    # Here a chunk is two half-planes (32 channels each) of 64-byte pixel-major region rows -- the stride-2 / fp8 families' row format:
    # one buffer-descriptor LDS-DMA instruction moves 16 rows x 64 B (16 look-ups), out-of-range lanes write the zero padding, channel
    # group q of a row sits at position q ^ 2 b (byte address bit 5 ^= b).  b is bit 2 of the pixel's COLUMN x (not of the region row,
    # as in those families): consecutive pixels of an image row still spread every ds_read_b128 lane group over all 64 banks, and a
    # kernel-row offset (dy Wp rows) no longer changes b -- so a pixel block needs three swizzled addresses (dx = 0, 1, 2), moved
    # down a kernel row by one add each, and the k-step is an immediate offset (the other half-plane).
    # WFIX = W: builds for one image width (yolov5m at 640 px: 40 and 20).  The first pixel-major build moved each address down a kernel row
    # with a v_add in front of the read that uses it -- 117 of them per chunk, each stalling its read: the stream WITHOUT any LDS-DMA took
    # 85.3 k cycles against the slot-major build's 81.8 k.  With the width known, dy Wp rows is an immediate offset like the k-step.
    "pm13": dict(NB=13, PD=8, ROWS=384, RING=3, OCC=1, RES_EARLY=True, LOOK=2, KS=2, S2=False, PM=True),
    "pm13w20": dict(NB=13, PD=8, ROWS=384, RING=3, OCC=1, RES_EARLY=True, LOOK=2, KS=2, S2=False, PM=True, WFIX=20),
    "pm13w40": dict(NB=13, PD=8, ROWS=384, RING=3, OCC=1, RES_EARLY=True, LOOK=2, KS=2, S2=False, PM=True, WFIX=40),
}
STEP_B = 6 * 1024          # weight bytes per (wave, tap-step): six 1 KB bf16 fragments (KS = 1: three) ...
W8 = False                 # ... or, in the fp8-weight kernels (set per kernel by gen_kernel), three 1 KB pairs of e4m3 fragments


def configure(nb):
    """Sets the module-level tile constants and (re)allocates the registers of one family."""
    g = globals()
    g["F8"] = False
    g["PM"] = False
    g["WFIX"] = 0
    g.update(CONFIGS[nb])
    import os as _os0
    # A_ACC (round 4; AQ_GEN_A_ACC=0 restores the old assignment; pixel-major bf16 families): the weight sets live in the accumulator half of
    # the register file (MFMA takes srcA from there at no cost) and the residual in VGPRs, instead of the other way round: the epilogue reads
    # the residual without v_accvgpr_read (78 instructions per tile) and the weight loads' returns no longer write the VGPR banks the MFMAs
    # read B from.  Stamped, with the 16-byte stores: 192 ch @ 40x40 lifetime 114.1 k -> 112.6 k cycles (stream 84.3 k -> 83.5 k, epilogue
    # 14.76 k -> 14.17 k), 384 ch @ 20x20 96.4 k -> 96.2 k.
    g["A_ACC"] = bool(PM and not F8 and _os0.environ.get("AQ_GEN_A_ACC", "1") == "1")
    g["RES_ACC"] = bool((F8 or PM) and not A_ACC)          # the residual waits in the accumulator file
    # W16 (round 4, AQ_GEN_W16=0 restores the 8-byte stores): the epilogue's output stores as 16 bytes per lane.  A wave's three stores per
    # pixel block (8 bytes per lane: this lane's four channels of an M block) are store-ISSUE bound, not bandwidth bound -- the stamped
    # ablations: epilogue 20.1 k cycles per wave (two tiles), without the stores 12.6 k, the stores alone 18.7 k = 240 cycles per 512-byte
    # instruction (cdna_hip_programming.md T21 saw the same on an attention tail).  v_permlane16_swap_b32 (lanes 16-31 / 48-63 of vdst <->
    # lanes 0-15 / 32-47 of src; tools/ubench/permlane16_swap.hip) pairs the 8-byte pieces of lanes q and q + 1 (same pixel, adjacent
    # channels): M blocks 0 and 1 of a pixel block leave in ONE 16-byte store (even-q lanes: block 0's 16 bytes, odd-q lanes: block 1's),
    # M block 2 of pixel blocks j and j + 1 in one more: 20 store instructions per tile instead of 39, same bytes, same addresses.
    g["W16"] = bool((PM or S2 or F8) and _os0.environ.get("AQ_GEN_W16", "1") == "1")
    # EP3 (experiment, AQ_GEN_EP3=1): the epilogue runs the three M blocks of a pixel block as three interleaved chains instead of two + one
    g["EP3"] = bool(PM and not F8 and _os0.environ.get("AQ_GEN_EP3", "0") == "1")
    # EARLY_SETUP (round 4; AQ_GEN_EARLY_SETUP=0 restores the old order): a workgroup's FIRST tile computes its addresses (some 700 VALU
    # instructions) between the prologue's loads and the wait for them instead of behind the first barrier.  Stamped, cycles per wave:
    # 192 ch @ 40x40 lifetime 122.1 k -> 119.1 k (tile set-up 6.9 k -> 3.9 k, prologue 5.6 k -> 6.8 k), 384 ch @ 20x20 100.1 k -> 98.8 k.
    g["EARLY_SETUP"] = bool(_os0.environ.get("AQ_GEN_EARLY_SETUP", "1") == "1")
    g["NT"] = 5 if F8 else 9                   # MFMA steps per chunk that take their own weight fragments: taps, or (fp8) tap pairs
    g["FAMILY"] = f"nb{nb}" if isinstance(nb, int) else nb
    g["NE"] = NT * KS * NB
    g["GROUPS"] = list(range(0, ROWS - 63, 64)) + ([ROWS - 64] if ROWS % 64 else [])      # first region row of every LDS-DMA group
    g["NG"] = len(GROUPS)
    g["PS"] = ROWS * 16
    g["NPAR"] = 4 if S2 else 1
    g["SL"] = 4 * KS                           # slot planes (8 channels each) per chunk and parity plane
    g["PPW"] = NPAR * SL // 4                  # planes each wave loads per chunk
    g["CHUNK"] = NPAR * SL * PS
    g["BIAS_OFF"] = RING * CHUNK
    g["LDS_BYTES"] = BIAS_OFF + 4096
    # where in a tap's element list its LDS-DMA instructions go (one region-row group per tap, PPW planes): behind the weight loads
    g["DPOS"] = [12, 12 + (2 * NB - 12) // 2]
    import os as _os
    if PM:
        g["DPOS"] = [int(x) for x in _os.environ.get("AQ_GEN_DPOS", "14,22").split(",")]
    if S2:
        g["GROUPS"] = list(range(0, ROWS, 64))             # groups of 64 rows whose source pixels are computed (then dealt to the instructions)
        g["NG"] = len(GROUPS)
        g["PSTR"] = ROWS * 64                              # one parity plane: ROWS pixel-major rows of 64 B
        g["CHUNK"] = NPAR * PSTR
        g["BIAS_OFF"] = RING * CHUNK
        g["LDS_BYTES"] = BIAS_OFF + 4096
        g["NDMA"] = ROWS // 16                             # LDS-DMA instructions per wave (= parity plane) and chunk
        # which of them go into which tap: all within taps 0 .. 5, so that the wait for tap 8's weights (loaded in tap 6) covers them
        per = [-(-NDMA // 6)] * (NDMA % 6 or 6) + [NDMA // 6] * (6 - (NDMA % 6 or 6))
        g["DMA_TAPS"] = [list(range(sum(per[:t]), sum(per[:t + 1]))) for t in range(6)] + [[], [], []]
        g["DPOS"] = [6 + 2 * i for i in range(max(per))]
        assert ROWS % 16 == 0 and sum(len(x) for x in DMA_TAPS) == NDMA and DPOS[-1] < NB and 3 * PSTR + 16 < 65536
    if F8:
        g["GROUPS"] = [0, 64]                              # per wave: the 96 region rows it loads, as two groups whose source pixels are computed
        g["NG"] = 2
        g["PSTR"] = ROWS * 64
        g["CHUNK"] = PSTR
        g["BIAS_OFF"] = RING * CHUNK
        g["LDS_BYTES"] = BIAS_OFF + 8192                   # bias / scale, then scale * ... (two 4 KB vectors, as the fp8-weight kernels)
        g["NDMA"] = ROWS // 16 // 4                        # LDS-DMA instructions per wave and chunk (the waves split the rows)
        g["DMA_TAPS"] = [[0, 1], [2, 3], [4, 5], [], []]
        g["DPOS"] = [6, 12]
        g["RPOS"] = [0, 2, 4, 8, 10]                       # elements of a step whose residual loads may go (steps 0 .. 3 of a tile's last chunk)
        assert ROWS % 64 == 0 and NDMA == 6 and NB == 13 and 2 * PD <= 15
    if PM:
        g["GROUPS"] = [0, 64]                              # per wave: the 96 region rows it loads (both half-planes), as two groups of computed source pixels
        g["NG"] = 2
        g["PSTR"] = ROWS * 64                              # one half-plane (32 channels): ROWS pixel-major rows of 64 B
        g["CHUNK"] = 2 * PSTR
        g["BIAS_OFF"] = RING * CHUNK
        g["LDS_BYTES"] = BIAS_OFF + 4096
        g["NDMA"] = ROWS // 16 // 4                        # LDS-DMA instructions per wave, chunk AND half-plane (the waves split the rows)
        assert ROWS % 64 == 0 and NDMA == 6 and PSTR + 16 < 65536
    assert (KS == 1) == bool(S2 or F8)
    assert PS % 256 == 0 and NB >= 6 and NG + 2 <= 8
    assert KS == 2 or RES_EARLY
    assert NT % (LOOK + 1) == 0, "the weight sets must come round at the end of a chunk"
    allocate_registers()


# ---- kernel argument block (must match PlAsmArgs in conv3x3_pl.hip) ----
ARG = dict(inp=0, in_sp=8, in_ss=16, out=24, res=32, w=40, bias=48, zero=56, out_ld=64, res_ld=68, B=72, H=76, W=80, npix=84, cout=88, act=92,
           CC=96, mt_log2=100, ntiles=104, G=108, inv_hw=112, inv_w=116, inv_hpwp=120, inv_wp=124, debug=128, in_row=136)
ARG_BYTES = 144


class Regs:
    """Named register allocation; errors on overlap."""

    def __init__(self, prefix, limit):
        self.prefix, self.limit, self.next, self.names = prefix, limit, 0, {}

    def alloc(self, name, n=1, align=1):
        self.next = (self.next + align - 1) // align * align
        base = self.next
        self.next += n
        assert self.next <= self.limit, f"out of {self.prefix} registers at {name}"
        self.names[name] = (base, n)
        return base


V = S = None


def allocate_registers():
    global V, S
    V = Regs("v", 256)
    S = Regs("s", 102)

    # ---------------- SGPRs ----------------
    S.alloc("karg", 2)            # s[0:1]
    S.alloc("wg", 1)              # s2
    for nm in ("inp", "in_sp", "in_ss", "out", "res", "w", "bias", "zero"):
        S.alloc(nm, 2, 2)
    for nm in ("out_ld", "res_ld", "B", "H", "W", "npix", "cout", "act", "CC", "mt_log2", "ntiles", "G", "inv_hw", "inv_w", "inv_hpwp", "inv_wp"):
        S.alloc(nm)
    S.alloc("debug", 2, 2)
    if S2:
        S.alloc("in_row", 2, 2)   # stride-2 family: bytes per INPUT image row (H, W, npix describe the OUTPUT there)
    for nm in ("HW", "Wp", "HpWp", "lead", "Hpad", "tile", "next_tile", "has_next", "n0", "cbase", "c", "buf", "cd", "bd", "lastc", "extra",
               "rs", "rs_dma", "delta0", "dRow", "wave", "first", "tmp0", "tmp1", "tmp2", "tmp3", "dlds", "lim") + (("coff",) if S2 else ("coff", "rlim") if F8 else ("coff", "coff1", "rlim") if PM else ("par", "rlim")):
        S.alloc(nm)
    S.alloc("klog2e2", 2, 2)      # (-log2 e, -log2 e) for v_pk_mul_f32
    S.alloc("kone2", 2, 2)        # (1.0, 1.0)
    S.alloc("a_cur", 2, 2)
    S.alloc("a_nxt", 2, 2)
    S.alloc("a_ld", 2, 2)         # base of the weight loads being issued
    if S2 or F8 or PM:
        S.alloc("srd", 4, 4)      # buffer descriptor of (this wave's parity plane of) the input (constant for the kernel)
    else:
        S.alloc("dbase", 2, 2)    # LDS-DMA: input base of (chunk, this wave's first plane) ...
    if KS == 2 and not F8 and not PM:
        S.alloc("dbase1", 2, 2)   # ... and of its second plane (32-channel chunks: one base, the planes are immediate offsets)
    S.alloc("t64", 2, 2)
    S.alloc("actm", 2, 2)         # prologue: all ones when the layer has an activation (read by the removed SPLIT epilogue only); W16 epilogue:
                                  # the lanes with odd q (0xffff0000 twice), the select mask of the paired stores' per-lane addresses
    S.alloc("st_acc", 12, 2)      # stamped build only: cycle sums of six phases
    if S2:                        # (arguments the stride-2 kernels never read: no shortcut, channel groups 16 bytes apart)
        S.names["st_last"], S.names["st_rt0"] = S.names["res"], S.names["in_ss"]
    elif F8 or PM:                # (the zero page's address lives on in two VGPRs; channel groups are contiguous)
        S.names["st_last"], S.names["st_rt0"] = S.names["zero"], S.names["in_ss"]
    else:
        S.alloc("st_last", 2, 2)
        S.alloc("st_rt0", 2, 2)
    # ---------------- VGPRs ----------------
    V.alloc("tid")                # v0 on entry
    V.alloc("lane")
    V.alloc("l15")
    V.alloc("q")
    V.alloc("aoff")               # lane * 16
    V.alloc("qps")                # q * PS
    V.alloc("insp")               # in_sp (low 32 bits) as a VGPR operand of v_mad_u64_u32
    V.alloc("zero_lo")
    V.alloc("zero_hi")
    if not A_ACC:                 # (kept as a block: the weight sets come first among the big arrays)
        V.alloc("A", (24 if F8 else 12 * KS) * (LOOK + 1), 4)
    V.alloc("B", (8 if F8 else 4) * (PD + 1), 4)
    V.alloc("addr", 4 * NB if S2 else 3 * NB if PM else NB)      # (stride 2: per pixel block one swizzled address per (row, column) offset of the taps; pixel-major: per column offset)
    V.alloc("prow", NG)
    if F8:
        V.alloc("tapoff", 5)      # per MFMA step: 64 x the region-row offset of the tap THIS lane's operand half belongs to (lanes 0-31: tap 2 p, 32-63: 2 p + 1)
        V.alloc("SC", 12, 4)      # epilogue: act_scale x weight_scale of this lane's 3 x 4 output channels
    if S2 or F8 or PM:
        V.alloc("voff", NDMA)     # per LDS-DMA instruction of a chunk: this lane's byte offset into the parity plane, or beyond the descriptor
        V.alloc("bpa", 4)         # ds_bpermute addresses: lane (16 m + lane / 4) * 4, m = 0 .. 3
        V.alloc("qoff")           # 16 x the channel group this lane fetches: (lane & 3) ^ 2 ((lane >> 4) & 1)  (pixel-major: 16 (lane & 3), the swizzle bit comes with the row)
    if RES_EARLY and not S2 and not RES_ACC:
        V.alloc("R", 6 * NB, 2)
    V.alloc("oo", NB)
    V.alloc("t", 36 if EP3 else 24, 4)           # temporaries
    if W16:
        V.alloc("cst", 4, 4)      # packed outputs of M block 2 of an even / odd pixel block, waiting for their 16-byte store


def s(name, i=0):
    b, n = S.names[name]
    assert i < n
    return f"s{b + i}"


def s2(name, i=0):
    b, n = S.names[name]
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


def rreg(b):
    """First register of residual pair b (0 .. 3 NB - 1): its own block, or (two-workgroup families) weight set 2 and then the B ring."""
    if RES_ACC:
        return 12 * NB + 2 * b                             # (accumulator-file register number)
    if RES_EARLY:
        return V.names["R"][0] + 2 * b
    assert 3 * NB <= 12 + 2 * (PD + 1) and LOOK == 2
    return V.names["A"][0] + 48 + 2 * b if b < 12 else V.names["B"][0] + 2 * (b - 12)


def areg(set_idx, k):
    """The four registers of fragment k (0 .. 5) of weight set set_idx."""
    if F8:                                                 # fragment k (0 .. 2) of a step: 8 registers (loads 2 k and 2 k + 1)
        assert 0 <= set_idx <= LOOK and 0 <= k < 3
        return vr("A", 24 * set_idx + 8 * k, 8)
    assert 0 <= set_idx <= LOOK and 0 <= k < 3 * KS
    if A_ACC:                                              # behind the accumulators
        b = 12 * NB + 12 * KS * set_idx + 4 * k
        return f"a[{b}:{b + 3}]"
    return vr("A", 12 * KS * set_idx + 4 * k, 4)


# fp8-weight kernels: the 72 weight registers hold three RAW sets of 12 (e4m3 codes as loaded), two converted bf16 half-sets of 12
# (k-step 0 and k-step 1 of the tap being multiplied) and the 12 per-channel scales of the epilogue
def rawreg(set_idx, pair):
    assert W8 and 0 <= set_idx < 3 and 0 <= pair < 3
    return vr("A", 12 * set_idx + 4 * pair, 4)


def bfreg(half, i):
    assert W8 and half in (0, 1) and 0 <= i < 3
    return vr("A", 36 + 12 * half + 4 * i, 4)


def emit_convert(tap, ks, i, step):
    """Quarter `step` (two VALU instructions) of converting fragment i of (tap, k-step ks) from raw e4m3 codes to bf16."""
    k = 3 * ks + i
    raw = V.names["A"][0] + 12 * (tap % 3) + 4 * (k // 2) + 2 * (k % 2) + step // 2
    dst = V.names["A"][0] + 36 + 12 * ks + 4 * i + 2 * (step // 2)
    t0 = V.names["t"][0] + 12
    if step % 2 == 0:
        E(f"v_cvt_pk_f32_fp8_e32 v[{t0}:{t0 + 1}], v{raw}")
        E(f"v_cvt_pk_f32_fp8_sdwa v[{t0 + 2}:{t0 + 3}], v{raw} src0_sel:WORD_1")
    else:
        E(f"v_cvt_pk_bf16_f32 v{dst}, v{t0}, v{t0 + 1}")
        E(f"v_cvt_pk_bf16_f32 v{dst + 1}, v{t0 + 2}, v{t0 + 3}")


def n_acc():
    return 12 * NB + (6 * NB if RES_ACC else 0) + (12 * KS * (LOOK + 1) if A_ACC else 0)


def acc(i, j):
    b = 4 * (3 * j + i)
    return f"a[{b}:{b + 3}]"


out = []


def E(line="", comment=None):
    out.append(("\t" + line if line and not line.endswith(":") else line) + (f"\t; {comment}" if comment else ""))


def label(name):
    out.append(f"{name}:")


_uid = [0, ""]


def uid(prefix):
    _uid[0] += 1
    return f".L{prefix}_{_uid[1]}_{_uid[0]}"


# ------------------------------------------------------------------------------------------------------------------------------
# macros
# ------------------------------------------------------------------------------------------------------------------------------
def emit_pp_of(dst, src, t0, t1, t2, t3):
    """dst = padded coordinate of pixel index src (VGPRs; dst may equal src): P + lead + b (H + W + 1) + y."""
    E(f"v_cvt_f32_i32 {t0}, {src}")
    E(f"v_mul_f32 {t0}, {s('inv_hw')}, {t0}")
    E(f"v_cvt_i32_f32 {t0}, {t0}", "b")
    E(f"v_mul_lo_u32 {t1}, {t0}, {s('HW')}")
    E(f"v_sub_u32 {t1}, {src}, {t1}", "rem")
    E(f"v_cmp_gt_i32 vcc, 0, {t1}")
    E(f"v_cndmask_b32 {t2}, 0, -1, vcc", "rem < 0 -> -1")
    E(f"v_cmp_le_i32 vcc, {s('HW')}, {t1}")
    E(f"v_cndmask_b32 {t3}, 0, 1, vcc", "rem >= HW -> +1")
    E(f"v_add_u32 {t2}, {t2}, {t3}", "correction of b")
    E(f"v_add_u32 {t0}, {t0}, {t2}")
    E(f"v_mul_lo_u32 {t2}, {t2}, {s('HW')}")
    E(f"v_sub_u32 {t1}, {t1}, {t2}", "rem in [0, HW)")
    E(f"v_cvt_f32_i32 {t2}, {t1}")
    E(f"v_mul_f32 {t2}, {s('inv_w')}, {t2}")
    E(f"v_cvt_i32_f32 {t2}, {t2}", "y")
    E(f"v_mul_lo_u32 {t3}, {t2}, {s('W')}")
    E(f"v_sub_u32 {t3}, {t1}, {t3}", "x")
    E(f"v_cmp_gt_i32 vcc, 0, {t3}")
    E(f"v_subbrev_co_u32 {t2}, vcc, 0, {t2}, vcc", "x < 0 -> y - 1")
    E(f"v_cmp_le_i32 vcc, {s('W')}, {t3}")
    E(f"v_addc_co_u32 {t2}, vcc, 0, {t2}, vcc", "x >= W -> y + 1 (x was >= 0, so at most one of the two fires)")
    E(f"v_mul_lo_u32 {t0}, {t0}, {s('Hpad')}", "b (H + W + 1)")
    E(f"v_add3_u32 {dst}, {src}, {t0}, {t2}")
    E(f"v_add_u32 {dst}, {s('lead')}, {dst}")


def emit_unpad(dst, src, t0, t1, t2, t3, t4):
    """dst = pixel index of padded coordinate src, or -1 (padding / outside the batch)."""
    E(f"v_subrev_u32 {t4}, {s('lead')}, {src}", "Pq")
    E(f"v_cvt_f32_i32 {t0}, {t4}")
    E(f"v_mul_f32 {t0}, {s('inv_hpwp')}, {t0}")
    E(f"v_cvt_i32_f32 {t0}, {t0}", "b")
    E(f"v_mul_lo_u32 {t1}, {t0}, {s('HpWp')}")
    E(f"v_sub_u32 {t1}, {t4}, {t1}", "rem")
    E(f"v_cmp_gt_i32 vcc, 0, {t1}")
    E(f"v_cndmask_b32 {t2}, 0, -1, vcc")
    E(f"v_cmp_le_i32 vcc, {s('HpWp')}, {t1}")
    E(f"v_cndmask_b32 {t3}, 0, 1, vcc")
    E(f"v_add_u32 {t2}, {t2}, {t3}")
    E(f"v_add_u32 {t0}, {t0}, {t2}")
    E(f"v_mul_lo_u32 {t2}, {t2}, {s('HpWp')}")
    E(f"v_sub_u32 {t1}, {t1}, {t2}", "rem in [0, HpWp)")
    E(f"v_cvt_f32_i32 {t2}, {t1}")
    E(f"v_mul_f32 {t2}, {s('inv_wp')}, {t2}")
    E(f"v_cvt_i32_f32 {t2}, {t2}", "y")
    E(f"v_mul_lo_u32 {t3}, {t2}, {s('Wp')}")
    E(f"v_sub_u32 {t3}, {t1}, {t3}", "x")
    # x < 0 -> y -= 1, x += Wp ; x >= Wp -> y += 1, x -= Wp
    E(f"v_cmp_gt_i32 vcc, 0, {t3}")
    E(f"v_cndmask_b32 {t1}, 0, -1, vcc")
    E(f"v_cmp_le_i32 vcc, {s('Wp')}, {t3}")
    E(f"v_addc_co_u32 {t1}, vcc, 0, {t1}, vcc", "dy in {-1, 0, +1}")
    E(f"v_add_u32 {t2}, {t2}, {t1}", "y")
    E(f"v_mul_lo_u32 {t1}, {t1}, {s('Wp')}")
    E(f"v_sub_u32 {t3}, {t3}, {t1}", "x in [0, Wp)")
    # P = b HW + y W + x
    E(f"v_mul_lo_u32 {t1}, {t0}, {s('HW')}")
    E(f"v_mad_u32_u24 {t1}, {t2}, {s('W')}, {t1}")
    E(f"v_add_u32 {dst}, {t1}, {t3}")
    if S2:                                   # the INPUT pixel of parity plane (0, 0): (b, 2 y, 2 x) = 4 P - 2 x  (input image = 2 H x 2 W)
        E(f"v_lshlrev_b32 {t1}, 1, {t3}")
        E(f"v_lshl_add_u32 {dst}, {dst}, 2, 0")
        E(f"v_sub_u32 {dst}, {dst}, {t1}")
    # valid: Pq >= 0 and b < B and x < W and y < H   (b >= 0 follows from Pq >= 0)
    E(f"v_cmp_gt_i32 vcc, 0, {t4}")
    E(f"v_cndmask_b32 {dst}, {dst}, -1, vcc")
    E(f"v_cmp_le_i32 vcc, {s('B')}, {t0}")
    E(f"v_cndmask_b32 {dst}, {dst}, -1, vcc")
    E(f"v_cmp_le_i32 vcc, {s('W')}, {t3}")
    E(f"v_cndmask_b32 {dst}, {dst}, -1, vcc")
    E(f"v_cmp_le_i32 vcc, {s('H')}, {t2}")
    E(f"v_cndmask_b32 {dst}, {dst}, -1, vcc")


def emit_rs_of_tile(tile_s, dst_s):
    """dst_s = pp_of(n0 of tile) - lead as an SGPR (n0 = (tile >> mt_log2) * BN)."""
    T = [v("t", i) for i in range(6)]
    E(f"s_lshr_b32 {s('tmp0')}, {tile_s}, {s('mt_log2')}")
    E(f"s_mul_i32 {s('tmp0')}, {s('tmp0')}, {NB * 16}")
    E(f"v_mov_b32 {T[4]}, {s('tmp0')}")
    emit_pp_of(T[5], T[4], T[0], T[1], T[2], T[3])
    E(f"v_subrev_u32 {T[5]}, {s('lead')}, {T[5]}")
    E("s_nop 1", "hz: VALU write -> v_readfirstlane")
    E(f"v_readfirstlane_b32 {dst_s}, {T[5]}")


def emit_region_rows(tile_s):
    """prow[k] = unpad(rs(tile) + GROUPS[k] + lane) for k = 0 .. NG - 1."""
    emit_rs_of_tile(tile_s, s("rs_dma"))
    T = [v("t", i) for i in range(6)]
    if F8 or PM:
        E(f"s_mul_i32 {s('tmp0')}, {s('wave')}, {16 * NDMA}", "this wave loads region rows 96 wave .. + 95")
        E(f"s_add_u32 {s('rs_dma')}, {s('rs_dma')}, {s('tmp0')}")
    for k in range(NG):
        E(f"v_add_u32 {T[5]}, {s('rs_dma')}, {v('lane')}")
        if k:
            E(f"v_add_u32 {T[5]}, {GROUPS[k]}, {T[5]}")
        emit_unpad(v("prow", k), T[5], T[0], T[1], T[2], T[3], T[4])
        # timing-only ablations of the slot-major LDS-DMA's cache-line look-ups (64 distinct 128-byte lines per instruction):
        if ABL[0] & 16:                                    # 16: eight distinct lines per instruction (what a pixel-major layout would touch)
            E(f"v_and_b32 {v('prow', k)}, -8, {v('prow', k)}")
        if ABL[0] & 32:                                    # 32: one line per instruction
            E("s_nop 1")
            E(f"v_readfirstlane_b32 {s('tmp0')}, {v('prow', k)}")
            E(f"v_mov_b32 {v('prow', k)}, {s('tmp0')}")
        if PM:                                             # the row's swizzle bit b = bit 2 of its column x travels with the pixel: 2 pixel + b (-1 stays negative)
            E(f"v_bfe_u32 {T[3]}, {T[3]}, 2, 1")
            E(f"v_lshl_or_b32 {v('prow', k)}, {v('prow', k)}, 1, {T[3]}")
    if PM:
        # deal the 64-row groups to the 16-row instructions (lane L of instruction g: region row 16 g + L / 4), then
        # offset = pixel * in_sp + 16 ((L & 3) ^ 2 b), or out of the descriptor's range for a padding row (the DMA writes zeros)
        for g_ in range(NDMA):
            E(f"ds_bpermute_b32 {v('voff', g_)}, {v('bpa', g_ & 3)}, {v('prow', g_ >> 2)}")
        E("s_waitcnt lgkmcnt(0)")
        E(f"v_bfrev_b32 {T[0]}, 1", "0x80000000: beyond num_records")
        for g_ in range(NDMA):
            E(f"v_cmp_gt_i32 vcc, 0, {v('voff', g_)}")
            E(f"v_and_b32 {T[2]}, 1, {v('voff', g_)}", "b")
            E(f"v_ashrrev_i32 {T[3]}, 1, {v('voff', g_)}", "pixel")
            E(f"v_lshlrev_b32 {T[2]}, 5, {T[2]}")
            E(f"v_xor_b32 {T[2]}, {T[2]}, {v('qoff')}", "16 ((L & 3) ^ 2 b)")
            E(f"v_mad_u32_u24 {v('voff', g_)}, {T[3]}, {v('insp')}, {T[2]}")
            E(f"v_cndmask_b32 {v('voff', g_)}, {v('voff', g_)}, {T[0]}, vcc")
    if S2 or F8:
        # deal the 64-row groups to the 16-row instructions: lane L of instruction g fetches for region row 16 g + L / 4, whose source pixel
        # sits in lane 16 (g & 3) + L / 4 of prow[g >> 2]; then offset = pixel * in_sp + 16 * (channel group), or out of the descriptor's range
        for g_ in range(NDMA):
            E(f"ds_bpermute_b32 {v('voff', g_)}, {v('bpa', g_ & 3)}, {v('prow', g_ >> 2)}")
        E("s_waitcnt lgkmcnt(0)")
        E(f"v_bfrev_b32 {T[0]}, 1", "0x80000000: beyond num_records")
        for g_ in range(NDMA):
            E(f"v_cmp_gt_i32 vcc, 0, {v('voff', g_)}")
            E(f"v_mad_u32_u24 {v('voff', g_)}, {v('voff', g_)}, {v('insp')}, {v('qoff')}")
            E(f"v_cndmask_b32 {v('voff', g_)}, {v('voff', g_)}, {T[0]}, vcc")


def emit_no_rows():
    if S2 or F8 or PM:
        for g_ in range(NDMA):
            E(f"v_bfrev_b32 {v('voff', g_)}, 1")
        return
    for k in range(NG):
        E(f"v_mov_b32 {v('prow', k)}, -1")


def emit_dma_base(cd_s, bd_s):
    """Per chunk: dbase = inp + (8 cd + 2 wave) * in_ss (plane 2 wave of chunk cd) and dlds = bd * CHUNK + 2 wave * PS, the operands every
    LDS-DMA instruction of that chunk starts from.  (Computing them inside each of the 12 instructions' sequences cost 13 scalar
    instructions apiece, in a stream that is bound by instruction issue.)"""
    if PM:
        # chunk cd = bytes 128 cd .. + 127 of every pixel, half-plane h the 64 bytes from 64 h (the descriptor's soffset); this wave's rows of ring buffer bd
        E(f"s_lshl_b32 {s('coff')}, {cd_s}, 7")
        E(f"s_add_u32 {s('coff1')}, {s('coff')}, 64")
        E(f"s_mul_i32 {s('tmp2')}, {bd_s}, {CHUNK}")
        E(f"s_mul_i32 {s('tmp1')}, {s('wave')}, {1024 * NDMA}")
        E(f"s_add_u32 {s('dlds')}, {s('tmp2')}, {s('tmp1')}", "the ring starts at LDS address 0")
        return
    if S2 or F8:
        # chunk cd = bytes 64 cd .. + 63 of every pixel (the descriptor's soffset); this wave's parity plane (fp8: its rows) of ring buffer bd
        E(f"s_lshl_b32 {s('coff')}, {cd_s}, 6")
        E(f"s_mul_i32 {s('tmp2')}, {bd_s}, {CHUNK}")
        E(f"s_mul_i32 {s('tmp1')}, {s('wave')}, {1024 * NDMA if F8 else PSTR}")
        E(f"s_add_u32 {s('dlds')}, {s('tmp2')}, {s('tmp1')}", "the ring starts at LDS address 0")
        return
    E(f"s_lshl_b32 {s('tmp0')}, {cd_s}, 3")
    E(f"s_lshl_b32 {s('tmp1')}, {s('wave')}, 1")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('tmp1')}", "slot of this wave's first plane")
    E(f"s_mul_i32 {s('tmp2')}, {s('tmp0')}, {s('in_ss')}")
    E(f"s_mul_hi_u32 {s('tmp3')}, {s('tmp0')}, {s('in_ss')}")
    E(f"s_add_u32 {s('dbase')}, {s('inp')}, {s('tmp2')}")
    E(f"s_addc_u32 {s('dbase', 1)}, {s('inp', 1)}, {s('tmp3')}")
    E(f"s_mul_i32 {s('tmp2')}, {bd_s}, {CHUNK}")
    E(f"s_mul_i32 {s('tmp1')}, {s('tmp1')}, {PS}")
    E(f"s_add_u32 {s('dlds')}, {s('tmp2')}, {s('tmp1')}", "the ring starts at LDS address 0")
    E(f"s_add_u32 {s('dbase1')}, {s('dbase')}, {s('in_ss')}")
    E(f"s_addc_u32 {s('dbase1', 1)}, {s('dbase', 1)}, 0")


def emit_dma(k, s2i, cd_s=None, bd_s=None):
    """One LDS-DMA instruction: plane 2 wave + s2i of the chunk emit_dma_base was called for, region rows 64 k .. 64 k + 63."""
    if S2 or F8 or PM:
        # (k = the instruction's number 0 .. NDMA - 1; in the stream the two halves sit around an element's MFMAs)
        emit_dma_m0(k, s2i)
        E("s_nop 0", "hz: m0 write -> LDS-DMA")
        emit_dma_issue(k, s2i)
        return
    T = [v("t", i) for i in range(8, 12)]
    base = s2("dbase1") if s2i else s2("dbase")
    E(f"s_add_u32 m0, {s('dlds')}, {s2i * PS + GROUPS[k] * 16}", "hz: m0 write -> LDS-DMA: the address arithmetic below sits in between")
    # (an invalid row, prow = -1, multiplies out to a wild address that the select below replaces: no clamp needed)
    E(f"v_mad_u64_u32 v[{V.names['t'][0] + 8}:{V.names['t'][0] + 9}], vcc, {v('prow', k)}, {v('insp')}, {base}")
    E(f"v_cmp_gt_i32 vcc, 0, {v('prow', k)}")
    E(f"v_cndmask_b32 {T[0]}, {T[0]}, {v('zero_lo')}, vcc")
    E(f"v_cndmask_b32 {T[1]}, {T[1]}, {v('zero_hi')}, vcc")
    E(f"global_load_lds_dwordx4 v[{V.names['t'][0] + 8}:{V.names['t'][0] + 9}], off")


def emit_dma_m0(g_, _=0):
    """Stride-2 family, first half of LDS-DMA instruction g_ (region rows 16 g_ .. + 15 of this wave's parity plane): its LDS address."""
    E(f"s_add_u32 m0, {s('dlds')}, {1024 * g_ + (_ * PSTR if PM else 0)}", "hz: m0 write -> LDS-DMA: at least one instruction before the DMA")


def emit_dma_issue(g_, _=0):
    E(f"buffer_load_dwordx4 {v('voff', g_)}, {s4('srd')}, {s('coff1') if PM and _ else s('coff')} offen lds")


def emit_load_a(set_idx, k, base_s2, extra_off):
    """fragment k (0..5) of a tap-step into A set set_idx: 1 KB at base + extra_off + 1024 k (base is the tap-step's stream + 3072);
    fp8 weights: fragment PAIR k (0..2), lane = 8 codes of fragment 2 k then 8 of fragment 2 k + 1, into raw set set_idx."""
    off = 1024 * k - 3072 + extra_off
    assert -4096 <= off <= 4095
    import os as _os
    pol = _os.environ.get("AQ_GEN_A_POLICY", "")          # experiment: cache policy of the weight loads ("nt", "sc0", "sc1", ...)
    dst = vr("A", 24 * set_idx + 4 * k, 4) if F8 else (rawreg(set_idx, k) if W8 else areg(set_idx, k))
    E(f"global_load_dwordx4 {dst}, {v('aoff')}, {base_s2} offset:{off}" + (f" {pol}" if pol else ""))


def emit_set_a_base(dst, tap):
    """dst (SGPR pair) = stream address (+3072) of tap-step `tap` of this chunk (0..8) or of the next chunk / tile (9, 10)."""
    if tap < NT:
        E(f"s_add_u32 {s(dst)}, {s('a_cur')}, {tap * STEP_B}")
        E(f"s_addc_u32 {s(dst, 1)}, {s('a_cur', 1)}, 0")
    else:
        E(f"s_add_u32 {s(dst)}, {s('a_nxt')}, {(tap - NT) * STEP_B}")
        E(f"s_addc_u32 {s(dst, 1)}, {s('a_nxt', 1)}, 0")


def emit_a_stream_base(dst, tile_s, c_s):
    """dst = w + (((mt * 4 + wave) * CC + c) * 9) * STEP_B + 3072, mt = tile & ((1 << mt_log2) - 1)."""
    E(f"s_lshl_b32 {s('tmp0')}, 1, {s('mt_log2')}")
    E(f"s_sub_u32 {s('tmp0')}, {s('tmp0')}, 1")
    E(f"s_and_b32 {s('tmp0')}, {tile_s}, {s('tmp0')}", "mt")
    E(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 2")
    if not ABL[0] & 64:                                    # (timing-only ablation 64: all four waves stream wave 0's weights -- three of four loads hit the CU's L1)
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('wave')}")
    E(f"s_mul_i32 {s('tmp0')}, {s('tmp0')}, {s('CC')}")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {c_s}")
    E(f"s_mul_i32 {s('tmp1')}, {s('tmp0')}, {NT * STEP_B}")
    E(f"s_mul_hi_u32 {s('tmp2')}, {s('tmp0')}, {NT * STEP_B}")
    E(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, 3072", "keeps every fragment offset inside the 13-bit immediate")
    E(f"s_addc_u32 {s('tmp2')}, {s('tmp2')}, 0")
    E(f"s_add_u32 {s(dst)}, {s('w')}, {s('tmp1')}")
    E(f"s_addc_u32 {s(dst, 1)}, {s('w', 1)}, {s('tmp2')}")


def emit_barrier():
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")


STAMPED = [False]
ABL = [0]
PH_PROLOGUE, PH_BARRIER, PH_STREAM, PH_SETUP, PH_EPILOGUE, PH_TOP = range(6)


def stamp(k):
    """Stamped build: add the shader clocks since the previous stamp to phase k's sum (t64, tmp0, tmp1 are dead at every stamp point)."""
    if not STAMPED[0]:
        return
    E(f"s_memtime {s2('t64')}")
    E("s_waitcnt lgkmcnt(0)")
    E(f"s_sub_u32 {s('tmp0')}, {s('t64')}, {s('st_last')}")
    E(f"s_subb_u32 {s('tmp1')}, {s('t64', 1)}, {s('st_last', 1)}")
    E(f"s_add_u32 {s('st_acc', 2 * k)}, {s('st_acc', 2 * k)}, {s('tmp0')}")
    E(f"s_addc_u32 {s('st_acc', 2 * k + 1)}, {s('st_acc', 2 * k + 1)}, {s('tmp1')}")
    E(f"s_mov_b64 {s2('st_last')}, {s2('t64')}")


def gen_kernel(name, RES, stamped=False, abl=0, w8=False):
    """abl: timing-only ablations of the stamped build (wrong results): 1 = no weight loads in the stream, 2 = no LDS-DMA in the stream,
    4 = no B fragment reads, 8 = no MFMAs, 128 = no output stores, 256 = no epilogue arithmetic (stores only)."""
    global out, W8, STEP_B
    out = []
    STAMPED[0] = stamped
    ABL[0] = abl
    NLOAD = 6 if F8 else 3 if (w8 or KS == 1) else 6       # weight load instructions per tap: 1 KB fragments (or e4m3 fragment pairs)
    W8, STEP_B = w8, NLOAD * 1024
    assert not w8 or (RES_EARLY and LOOK == 2 and NB >= 12 and KS == 2)
    assert not (S2 and RES), "the stride-2 layers have no shortcut"
    _uid[0], _uid[1] = 0, name.split("asm_", 1)[1].replace("_", "")
    E(f"; conv3x3_pl assembly, family {FAMILY}, NB = {NB}, {OCC} workgroup(s) per CU, RES = {int(RES)}: generated by gen_conv3x3_pl_asm.py -- do not edit")
    label(name)
    # ---- arguments ----
    E(f"s_load_dwordx8 s[{S.names['inp'][0]}:{S.names['inp'][0] + 7}], {s2('karg')}, 0x0", "inp, in_sp, in_ss, out")
    E(f"s_load_dwordx8 s[{S.names['res'][0]}:{S.names['res'][0] + 7}], {s2('karg')}, 0x20", "res, w, bias, zero")
    E(f"s_load_dwordx8 s[{S.names['out_ld'][0]}:{S.names['out_ld'][0] + 7}], {s2('karg')}, 0x40", "out_ld .. act")
    E(f"s_load_dwordx8 s[{S.names['CC'][0]}:{S.names['CC'][0] + 7}], {s2('karg')}, 0x60", "CC .. inv_wp")
    E(f"s_load_dwordx2 {s2('debug')}, {s2('karg')}, 0x80")
    if S2:
        E(f"s_load_dwordx2 {s2('in_row')}, {s2('karg')}, 0x88")
    assert S.names['in_sp'][0] == S.names['inp'][0] + 2 and S.names['out'][0] == S.names['inp'][0] + 6
    assert S.names['zero'][0] == S.names['res'][0] + 6 and S.names['act'][0] == S.names['out_ld'][0] + 7
    assert S.names['inv_wp'][0] == S.names['CC'][0] + 7 and S.names['inp'][0] % 4 == 0 and S.names['res'][0] % 4 == 0
    if OCC > 1:
        E(f"s_getreg_b32 {s('par')}, hwreg(HW_REG_HW_ID, 0, 1)", "parity of this wave's slot on its SIMD: differs between the two co-resident waves")
    E(f"v_and_b32 {v('lane')}, 63, {v('tid')}")
    E(f"v_lshrrev_b32 {v('t', 0)}, 6, {v('tid')}")
    E("s_nop 1", "hz: VALU write -> v_readfirstlane")
    E(f"v_readfirstlane_b32 {s('wave')}, {v('t', 0)}")
    E(f"v_and_b32 {v('l15')}, 15, {v('lane')}")
    E(f"v_lshrrev_b32 {v('q')}, 4, {v('lane')}")
    E(f"v_lshlrev_b32 {v('aoff')}, 4, {v('lane')}")
    if F8:
        E(f"v_and_b32 {v('qps')}, 1, {v('q')}")
        E(f"v_lshlrev_b32 {v('qps')}, 5, {v('qps')}", "fp8 family: operand lanes 16 q .. hold the 32 channels 32 (q & 1) .. of their tap")
    elif S2 or PM:
        E(f"v_lshlrev_b32 {v('qps')}, 4, {v('q')}", "stride-2 / pixel-major families: channel group q is 16 q bytes into a (pixel-major) region row")
    else:
        E(f"v_mul_u32_u24 {v('qps')}, {PS}, {v('q')}")
    E("s_waitcnt lgkmcnt(0)")
    E(f"v_mov_b32 {v('insp')}, {s('in_sp')}")
    E(f"v_mov_b32 {v('zero_lo')}, {s('zero')}")
    E(f"v_mov_b32 {v('zero_hi')}, {s('zero', 1)}")
    if S2 or F8 or PM:
        # descriptor of this wave's parity plane (py, px) = (wave >> 1, wave & 1): base = inp + py in_row + px in_sp, raw buffer (stride 0),
        # num_records 2^31 (every valid offset is below it -- the host checks the tensor's size --, the padding rows' 0x80000000 is not)
        # (fp8 family: one plane, base = inp)
        lp, lq = uid("py"), uid("px")
        E(f"s_mov_b32 {s('srd', 0)}, {s('inp')}")
        E(f"s_mov_b32 {s('srd', 1)}, {s('inp', 1)}")
    if S2:
        E(f"s_bitcmp0_b32 {s('wave')}, 1")
        E(f"s_cbranch_scc1 {lp}")
        E(f"s_add_u32 {s('srd', 0)}, {s('srd', 0)}, {s('in_row')}")
        E(f"s_addc_u32 {s('srd', 1)}, {s('srd', 1)}, {s('in_row', 1)}")
        label(lp)
        E(f"s_bitcmp0_b32 {s('wave')}, 0")
        E(f"s_cbranch_scc1 {lq}")
        E(f"s_add_u32 {s('srd', 0)}, {s('srd', 0)}, {s('in_sp')}")
        E(f"s_addc_u32 {s('srd', 1)}, {s('srd', 1)}, {s('in_sp', 1)}")
        label(lq)
    if S2 or F8 or PM:
        E(f"s_and_b32 {s('srd', 1)}, {s('srd', 1)}, 0xffff")
        E(f"s_mov_b32 {s('srd', 2)}, 0x80000000")
        E(f"s_mov_b32 {s('srd', 3)}, 0x00020000")
        E(f"v_lshrrev_b32 {v('t', 0)}, 2, {v('lane')}", "lane / 4: the row of an instruction's 16 this lane fetches for")
        for m_ in range(4):
            E(f"v_add_u32 {v('bpa', m_)}, {16 * m_}, {v('t', 0)}")
            E(f"v_lshlrev_b32 {v('bpa', m_)}, 2, {v('bpa', m_)}")
        E(f"v_bfe_u32 {v('t', 1)}, {v('lane')}, 4, 1", "bit 2 of that row (instructions start at multiples of 16 rows)")
        E(f"v_and_b32 {v('qoff')}, 3, {v('lane')}")
        E(f"v_lshl_add_u32 {v('t', 1)}, {v('t', 1)}, 1, 0")
        if not PM:                                         # (pixel-major family: the swizzle bit is a property of the row's pixel, applied per tile)
            E(f"v_xor_b32 {v('qoff')}, {v('qoff')}, {v('t', 1)}", "position (lane & 3) of a row holds channel group (lane & 3) ^ 2 bit2(row)")
        E(f"v_lshlrev_b32 {v('qoff')}, 4, {v('qoff')}")
    E(f"s_mul_i32 {s('HW')}, {s('H')}, {s('W')}")
    E(f"s_add_u32 {s('Wp')}, {s('W')}, 1")
    E(f"s_add_u32 {s('tmp0')}, {s('H')}, 1")
    E(f"s_mul_i32 {s('HpWp')}, {s('tmp0')}, {s('Wp')}")
    E(f"s_add_u32 {s('lead')}, {s('W')}, 2")
    E(f"s_add_u32 {s('Hpad')}, {s('tmp0')}, {s('W')}", "H + W + 1")
    E(f"s_lshl_b32 {s('tmp0')}, {s('Wp')}, {6 if S2 or PM else 4}")
    E(f"s_mov_b32 {s('dRow')}, {s('tmp0')}", "kernel row r -> r + 1: Wp * 16 bytes (stride-2 family: Wp * 64)")
    if F8:
        # tapoff[p]: lanes 0-31 carry tap 2 p, lanes 32-63 tap 2 p + 1 (step 4: tap 8 again -- its weights are zeros there);
        # tap t = (dy, dx) sits dy Wp + dx region rows (of 64 bytes) below / right of the upper left neighbour
        E(f"v_lshrrev_b32 {v('t', 0)}, 5, {v('lane')}", "0 / 1: which tap of the pair")
        for p_ in range(5):
            ta, tb = 2 * p_, min(2 * p_ + 1, 8)
            offs = []
            for t_ in (ta, tb):
                E(f"s_mul_i32 {s('tmp0')}, {s('Wp')}, {t_ // 3}")
                E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {t_ % 3}")
                E(f"s_lshl_b32 {s('tmp1' if t_ == ta else 'tmp2')}, {s('tmp0')}, 6")
            E(f"v_mov_b32 {v('tapoff', p_)}, {s('tmp1')}")
            E(f"v_mov_b32 {v('t', 1)}, {s('tmp2')}")
            E(f"v_cmp_eq_u32 vcc, 1, {v('t', 0)}")
            E(f"v_cndmask_b32 {v('tapoff', p_)}, {v('tapoff', p_)}, {v('t', 1)}, vcc")
    E(f"s_mov_b32 {s('klog2e2')}, 0xbfb8aa3b", "-log2(e)")
    E(f"s_mov_b32 {s('klog2e2', 1)}, 0xbfb8aa3b")
    E(f"s_mov_b32 {s('kone2')}, 1.0")
    E(f"s_mov_b32 {s('kone2', 1)}, 1.0")
    E(f"s_cmp_lg_u32 {s('act')}, 0")
    E(f"s_cselect_b64 {s2('actm')}, -1, 0")
    if stamped:
        for k in range(12):
            E(f"s_mov_b32 {s('st_acc', k)}, 0")
        E(f"s_memrealtime {s2('st_rt0')}")
        E(f"s_memtime {s2('st_last')}")
        E("s_waitcnt lgkmcnt(0)")
    # ---- first tile: XCD-aware bijective map (blocks sharing an XCD get consecutive tiles) ----
    E(f"s_lshr_b32 {s('tmp0')}, {s('G')}, 3", "q")
    E(f"s_and_b32 {s('tmp1')}, {s('G')}, 7", "r")
    E(f"s_and_b32 {s('tmp2')}, {s('wg')}, 7", "xcd")
    E(f"s_add_u32 {s('tmp3')}, {s('tmp0')}, 1", "q + 1")
    E(f"s_cmp_lt_u32 {s('tmp2')}, {s('tmp1')}")
    E(f"s_cselect_b32 {s('tile')}, {s('tmp2')}, {s('tmp1')}", "min(xcd, r)")
    E(f"s_mul_i32 {s('tile')}, {s('tile')}, {s('tmp3')}", "min(xcd, r) * (q + 1)")
    E(f"s_sub_u32 {s('tmp3')}, {s('tmp2')}, {s('tmp1')}", "xcd - r")
    E(f"s_cselect_b32 {s('tmp3')}, 0, {s('tmp3')}", "max(xcd - r, 0)   (scc still: xcd < r)")
    E(f"s_mul_i32 {s('tmp3')}, {s('tmp3')}, {s('tmp0')}")
    E(f"s_add_u32 {s('tile')}, {s('tile')}, {s('tmp3')}")
    E(f"s_lshr_b32 {s('tmp3')}, {s('wg')}, 3")
    E(f"s_add_u32 {s('tile')}, {s('tile')}, {s('tmp3')}")
    E(f"s_cmp_ge_u32 {s('tile')}, {s('ntiles')}")
    E("s_cbranch_scc1 .Lend_" + name)
    def emit_setup_addr():
        """Per tile: n0, cbase, next_tile, rs, and this lane's fragment addresses addr[] and output offsets oo[] (VALU only, no memory)."""
        E(f"s_lshr_b32 {s('tmp1')}, {s('tile')}, {s('mt_log2')}")
        E(f"s_lshl_b32 {s('tmp0')}, {s('tmp1')}, {s('mt_log2')}")
        E(f"s_sub_u32 {s('tmp2')}, {s('tile')}, {s('tmp0')}")
        E(f"s_mul_i32 {s('n0')}, {s('tmp1')}, {NB * 16}")
        E(f"s_add_u32 {s('next_tile')}, {s('tile')}, {s('G')}")
        E(f"s_cmp_lt_u32 {s('next_tile')}, {s('ntiles')}")
        E(f"s_cselect_b32 {s('has_next')}, 1, 0")
        E(f"s_mul_i32 {s('cbase')}, {s('tmp2')}, 192")
        E(f"s_mul_i32 {s('tmp0')}, {s('wave')}, 48")
        E(f"s_add_u32 {s('cbase')}, {s('cbase')}, {s('tmp0')}")
        emit_rs_of_tile(s("tile"), s("rs"))
        # addr[j] = q PS + (pp_of(P_j) - rs - Wp - 1) * 16 + buf * CHUNK ;  oo[j] = P_j * out_ld + (cbase + 4 q) * 2 ; omask[j]
        E(f"s_sub_u32 {s('tmp1')}, {s('npix')}, 1")
        E(f"s_add_u32 {s('tmp2')}, {s('rs')}, {s('Wp')}")
        E(f"s_add_u32 {s('tmp2')}, {s('tmp2')}, 1", "rs + Wp + 1")
        E(f"s_mul_i32 {s('tmp3')}, {s('buf')}, {CHUNK}")
        if S2:
            E(f"s_add_u32 {s('tmp0')}, {s('dRow')}, 64", "one row down and one column right")
        E(f"v_lshl_add_u32 {T[7]}, {v('q')}, 2, {s('cbase')}", "cbase + 4 q")
        E(f"v_lshlrev_b32 {T[7]}, 1, {T[7]}", "bytes")
        for j in range(NB):
            E(f"v_add_u32 {T[4]}, {s('n0')}, {v('l15')}")
            if j:
                E(f"v_add_u32 {T[4]}, {16 * j}, {T[4]}")
            E(f"v_min_i32 {T[4]}, {s('tmp1')}, {T[4]}", "clamp")
            E(f"v_mul_lo_u32 {v('oo', j)}, {T[4]}, {s('out_ld')}")
            E(f"v_add_u32 {v('oo', j)}, {v('oo', j)}, {T[7]}")
            emit_pp_of(T[5], T[4], T[0], T[1], T[2], T[3])
            E(f"v_subrev_u32 {T[5]}, {s('tmp2')}, {T[5]}")
            if PM:
                # three swizzled addresses per pixel block, one per column offset dx of the taps: region row (upper left neighbour) + dx, 64 B per
                # row, channel group q at position q ^ 2 b with b = bit 2 of that row's column x - 1 + dx (a padding column holds zeros at every
                # position); + the ring buffer.  emit_pp_of leaves rem = y W + x in T[1] and y in T[2].
                E(f"v_mul_lo_u32 {T[0]}, {T[2]}, {s('W')}")
                E(f"v_sub_u32 {T[0]}, {T[1]}, {T[0]}", "x")
                E(f"v_lshl_add_u32 {T[5]}, {T[5]}, 6, {s('tmp3')}", "byte address of the row in this tile's first ring buffer")
                for dx in range(3):
                    E(f"v_add_u32 {T[1]}, {dx - 1}, {T[0]}")
                    E(f"v_bfe_u32 {T[1]}, {T[1]}, 2, 1", "b")
                    E(f"v_lshlrev_b32 {T[1]}, 5, {T[1]}")
                    E(f"v_xor_b32 {T[1]}, {T[1]}, {v('qps')}", "16 (q ^ 2 b)")
                    E(f"v_add_u32 {T[1]}, {T[5]}, {T[1]}")
                    E(f"v_add_u32 {v('addr', 3 * j + dx)}, {64 * dx}, {T[1]}")
                continue
            if F8:
                # linear byte address of (upper left neighbour's row, this lane's 32-byte channel half), in the tile's first ring buffer;
                # the tap offset and the swizzle are applied where the fragment is read
                E(f"v_lshl_add_u32 {T[5]}, {T[5]}, 6, {v('qps')}", "qps = 32 (q & 1) here")
                E(f"v_add_u32 {v('addr', j)}, {s('tmp3')}, {T[5]}")
                continue
            if S2:
                # four swizzled addresses per pixel block: region row (upper left neighbour) + (ry Wp + rx), 64 B per row, channel group q at
                # position q ^ 2 bit2(row): byte bit 5 ^= byte bit 8; + the ring buffer
                E(f"v_lshl_add_u32 {T[5]}, {T[5]}, 6, {v('qps')}", "linear byte address of (row, group q): qps = 16 q here")
                for o in range(4):
                    src = T[5]
                    if o:
                        E(f"v_add_u32 {T[0]}, {s('dRow') if o == 2 else (64 if o == 1 else s('tmp0'))}, {T[5]}")
                        src = T[0]
                    E(f"v_lshrrev_b32 {T[1]}, 3, {src}")
                    E(f"v_and_b32 {T[1]}, 32, {T[1]}")
                    E(f"v_xad_u32 {v('addr', 4 * j + o)}, {src}, {T[1]}, {s('tmp3')}")
                continue
            E(f"v_lshl_add_u32 {v('addr', j)}, {T[5]}, 4, {v('qps')}")
            E(f"v_add_u32 {v('addr', j)}, {s('tmp3')}, {v('addr', j)}")
    # ---- prologue: weights of taps 0 and 1, chunks 0 and 1 of the region, the bias ----
    E(f"s_mov_b32 {s('c')}, 0")
    emit_a_stream_base("a_cur", s("tile"), s("c"))
    for tap in range(LOOK):                        # offsets beyond a tap-step exceed the immediate range, so move the base
        emit_set_a_base("a_ld", tap)
        for k in range(NLOAD):
            emit_load_a(tap, k, s2("a_ld"), 0)
    emit_region_rows(s("tile"))
    E(f"s_mov_b32 {s('cd')}, 0")
    E(f"s_mov_b32 {s('bd')}, 0")
    emit_dma_base(s("cd"), s("bd"))
    if S2 or F8 or PM:
        for g_ in range(NDMA):
            for h_ in range(2 if PM else 1):
                emit_dma(g_, h_)
    for k in range(NG if not (S2 or F8 or PM) else 0):
        for s2i in range(PPW):
            emit_dma(k, s2i, s("cd"), s("bd"))
    # bias: 256 floats per wave by LDS-DMA (lane: floats wave * 256 + 4 lane .. + 3, or zeros beyond cout)
    T = [v("t", i) for i in range(8)]
    E(f"s_lshl_b32 {s('tmp0')}, {s('wave')}, 8")
    E(f"v_lshl_add_u32 {T[2]}, {v('lane')}, 2, {s('tmp0')}", "f0")
    E(f"v_add_u32 {T[3]}, 4, {T[2]}")
    E(f"v_lshlrev_b32 {T[2]}, 2, {T[2]}", "byte offset")
    E(f"v_mov_b32 {T[4]}, {s('bias')}")
    E(f"v_mov_b32 {T[5]}, {s('bias', 1)}")
    E(f"v_add_co_u32 {T[4]}, vcc, {T[4]}, {T[2]}")
    E(f"v_addc_co_u32 {T[5]}, vcc, 0, {T[5]}, vcc")
    E(f"v_cmp_lt_u32 vcc, {s('cout')}, {T[3]}", "f0 + 4 > cout")
    E(f"v_cndmask_b32 {T[4]}, {T[4]}, {v('zero_lo')}, vcc")
    E(f"v_cndmask_b32 {T[5]}, {T[5]}, {v('zero_hi')}, vcc")
    E(f"s_lshl_b32 {s('tmp0')}, {s('wave')}, 10")
    E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {BIAS_OFF}")
    E(f"s_mov_b32 m0, {s('tmp0')}")
    E("s_nop 0", "hz: s_mov m0 -> LDS-DMA")
    E(f"global_load_lds_dwordx4 v[{V.names['t'][0] + 4}:{V.names['t'][0] + 5}], off")
    if w8 or F8:
        # fp8 weights: `bias` holds bias x 2^-e (1024 floats) and then the per-channel scales 2^e (1024 floats)
        # (fp8 family: bias / (act_scale x weight_scale), then act_scale x weight_scale)
        E(f"v_add_co_u32 {T[4]}, vcc, 4096, {T[4]}")
        E(f"v_addc_co_u32 {T[5]}, vcc, 0, {T[5]}, vcc")
        E(f"v_cmp_lt_u32 vcc, {s('cout')}, {T[3]}")
        E(f"v_cndmask_b32 {T[4]}, {T[4]}, {v('zero_lo')}, vcc")
        E(f"v_cndmask_b32 {T[5]}, {T[5]}, {v('zero_hi')}, vcc")
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, 4096")
        E(f"s_mov_b32 m0, {s('tmp0')}")
        E("s_nop 0", "hz: s_mov m0 -> LDS-DMA")
        E(f"global_load_lds_dwordx4 v[{V.names['t'][0] + 4}:{V.names['t'][0] + 5}], off")
    if EARLY_SETUP:
        # the first tile's address arithmetic (some 700 VALU instructions) between the prologue's loads and the wait for them
        E(f"s_mov_b32 {s('buf')}, 0")
        E(f"s_mov_b32 {s('first')}, 1")
        emit_setup_addr()
    E("s_waitcnt vmcnt(0)", "weights of the first taps, chunk 0, the bias")
    emit_barrier()
    if w8:
        for i in range(3):                         # k-step 0 of the very first tap; every later half-tap is converted under MFMAs
            for step in range(4):
                emit_convert(0, 0, i, step)
    if RING == 3:
        # chunk 1's LDS-DMA only now: issued together with chunk 0's, every workgroup of the launch asked for twice the bytes at once and
        # the first barrier came 2-3 k cycles later; it has the whole first chunk to land
        E(f"s_mov_b32 {s('cd')}, 1")
        E(f"s_mov_b32 {s('bd')}, 1")
        emit_dma_base(s("cd"), s("bd"))
        if F8 or PM:
            for g_ in range(NDMA):
                for h_ in range(2 if PM else 1):
                    emit_dma(g_, h_)
        for k in range(NG if not (F8 or PM) else 0):
            for s2i in range(PPW):
                emit_dma(k, s2i, s("cd"), s("bd"))
    stamp(PH_PROLOGUE)
    if not EARLY_SETUP:
        E(f"s_mov_b32 {s('buf')}, 0")
        E(f"s_mov_b32 {s('first')}, 1")
    else:
        E(f"s_branch .Ltile_bias_{name}", "the first tile's addresses were computed under the prologue's loads")

    # =========================================== tile loop ===========================================
    label(".Ltile_" + name)
    emit_setup_addr()
    if EARLY_SETUP:
        label(".Ltile_bias_" + name)
    # accumulators start from the bias: LDS reads straight into the accumulator registers (3 NB reads instead of 12 NB register moves)
    E(f"v_lshl_add_u32 {T[6]}, {v('q')}, 2, {s('cbase')}")
    E(f"v_lshlrev_b32 {T[6]}, 2, {T[6]}")
    E(f"v_add_u32 {T[6]}, {BIAS_OFF}, {T[6]}", "the bias sits behind the ring (beyond the 16-bit DS offset)")
    for j in range(NB):
        for i in range(3):
            E(f"ds_read_b128 {acc(i, j)}, {T[6]} offset:{64 * i}")
    E("s_waitcnt lgkmcnt(0)")
    E(f"s_mov_b32 {s('c')}, 0")
    E(f"s_mov_b32 {s('delta0')}, 0", "the first chunk of a tile starts at tap 0 of its own buffer")
    stamp(PH_SETUP)

    # =========================================== chunk loop ===========================================
    label(".Lchunk_" + name)
    emit_barrier()
    stamp(PH_BARRIER)
    E(f"s_add_u32 {s('tmp0')}, {s('c')}, 1")
    E(f"s_cmp_eq_u32 {s('tmp0')}, {s('CC')}")
    E(f"s_cselect_b32 {s('lastc')}, 1, 0")
    # DMA target: chunk c + RING - 1 of this tile or c + RING - 1 - CC of the next one, into buffer (buf + RING - 1) % RING
    E(f"s_add_u32 {s('cd')}, {s('c')}, {RING - 1}")
    E(f"s_add_u32 {s('bd')}, {s('buf')}, {RING - 1}")
    E(f"s_cmp_ge_u32 {s('bd')}, {RING}")
    E(f"s_cselect_b32 {s('tmp0')}, {RING}, 0")
    E(f"s_sub_u32 {s('bd')}, {s('bd')}, {s('tmp0')}")
    lskip = uid("cd")
    E(f"s_cmp_lt_u32 {s('cd')}, {s('CC')}")
    E(f"s_cbranch_scc1 {lskip}")
    E(f"s_sub_u32 {s('cd')}, {s('cd')}, {s('CC')}")
    E(f"s_cmp_lg_u32 {s('cd')}, 0")
    E(f"s_cbranch_scc1 {lskip}")
    lno, ldone = uid("norows"), uid("rowsdone")
    E(f"s_cmp_eq_u32 {s('has_next')}, 0")
    E(f"s_cbranch_scc1 {lno}")
    emit_region_rows(s("next_tile"))
    E(f"s_branch {ldone}")
    label(lno)
    emit_no_rows()
    label(ldone)
    label(lskip)
    emit_dma_base(s("cd"), s("bd"))
    # weight stream of this chunk and of the next (chunk c + 1, or chunk 0 of the next tile; the own tile again when there is none)
    emit_a_stream_base("a_cur", s("tile"), s("c"))
    lnx, lnd = uid("anxt"), uid("anxtd")
    E(f"s_cmp_eq_u32 {s('lastc')}, 1")
    E(f"s_cbranch_scc1 {lnx}")
    E(f"s_add_u32 {s('tmp3')}, {s('c')}, 1")
    emit_a_stream_base("a_nxt", s("tile"), s("tmp3"))
    E(f"s_branch {lnd}")
    label(lnx)
    E(f"s_cmp_eq_u32 {s('has_next')}, 1")
    E(f"s_cselect_b32 {s('t64')}, {s('next_tile')}, {s('tile')}")
    E(f"s_mov_b32 {s('tmp3')}, 0")
    emit_a_stream_base("a_nxt", s("t64"), s("tmp3"))
    label(lnd)
    # extra = vector-memory operations younger than the weights of taps 0 and 1: the epilogue's stores (and, two-workgroup families,
    # its residual loads) in the first chunk of a tile that is not the wave's first
    E(f"s_cmp_eq_u32 {s('c')}, 0")
    E(f"s_cselect_b32 {s('extra')}, 1, 0")
    E(f"s_cmp_eq_u32 {s('first')}, 1")
    E(f"s_cselect_b32 {s('extra')}, 0, {s('extra')}")
    n_extra = 3 * NB if (RES_EARLY or not RES) else 6 * NB
    if W16 and not w8:
        n_extra = NB + (NB + 1) // 2      # one 16-byte store per pixel block (M blocks 0, 1) + one per pair of pixel blocks (M block 2)
    if abl & 128:
        n_extra = 0
    if RES and RES_EARLY:
        # The tile's whole residual rides in the LAST chunk's stream (3 loads per pixel block, spread over taps 0 .. 6, landing under
        # the MFMAs); the other chunks branch over the loads, and every weight wait of the stream picks between two hand-counted
        # immediates on `lastc`.  (Issuing them in every chunk with EXEC = 0 keeps one immediate per wait, and is correct -- an
        # instruction without active lanes still counts -- but each still costs its issue slot: + 4 k cycles per six chunks.)
        E(f"s_sub_u32 {s('rlim')}, {s('npix')}, 1")
        E(f"v_lshl_add_u32 {T[7]}, {v('q')}, 2, {s('cbase')}")
        E(f"v_lshlrev_b32 {T[7]}, 1, {T[7]}", "t7 = (cbase + 4 q) * 2: lives through the stream")
    if OCC > 1:
        # Issue priority against the co-resident workgroup: the wave in the later half of its tile wins (ties: slot parity), so that
        # the two workgroups of a CU fall out of step -- one streams MFMAs at full rate while the other is in its epilogue / tile
        # set-up / chunk top -- instead of sharing the MFMA pipe evenly and then idling it together.
        l_late, l_p1, l_p3, l_pd = uid("plate"), uid("p1"), uid("p3"), uid("pd")
        E(f"s_lshl_b32 {s('tmp0')}, {s('c')}, 1")
        E(f"s_cmp_ge_u32 {s('tmp0')}, {s('CC')}")
        E(f"s_cbranch_scc1 {l_late}")
        E(f"s_cmp_eq_u32 {s('par')}, 1")
        E(f"s_cbranch_scc1 {l_p1}")
        E("s_setprio 0")
        E(f"s_branch {l_pd}")
        label(l_p1)
        E("s_setprio 1")
        E(f"s_branch {l_pd}")
        label(l_late)
        E(f"s_cmp_eq_u32 {s('par')}, 1")
        E(f"s_cbranch_scc1 {l_p3}")
        E("s_setprio 2")
        E(f"s_branch {l_pd}")
        label(l_p3)
        E("s_setprio 3")
        label(l_pd)
    stamp(PH_TOP)

    cold = []                             # out-of-line code: (label, immediate or instruction list, label to return to)
    T7RES = RES and RES_EARLY

    def epilogue_block(i, j, mode, lines, tb=0, dst=None):
        """Appends the instructions of one 16 x 16 output block (M block i, pixel block j): accumulators -> (x scale) -> SiLU -> + residual ->
        bf16 -> store.  mode: "act" / "noact" (the two copies behind a branch after the stream).  The caller has set EXEC to the pixel
        block's store mask."""
        X = V.names["t"][0] + tb + 0
        Y = V.names["t"][0] + tb + 4
        Rr = V.names["t"][0] + tb + 8
        a0 = 4 * (3 * j + i)
        r0 = rreg(3 * j + i) if RES else 0
        L = lines.append
        for e in range(4):
            L(f"v_accvgpr_read_b32 v{X + e}, a{a0 + e}")
        if F8:                                            # accumulators are sums of code products: x act_scale x weight_scale[channel]
            sc = V.names["SC"][0] + 4 * i
            L(f"v_pk_mul_f32 v[{X}:{X + 1}], v[{X}:{X + 1}], v[{sc}:{sc + 1}]")
            L(f"v_pk_mul_f32 v[{X + 2}:{X + 3}], v[{X + 2}:{X + 3}], v[{sc + 2}:{sc + 3}]")

        def unpack():
            if RES_ACC:                                   # the residual waits in the accumulator file
                L(f"v_accvgpr_read_b32 v{Rr + 1}, a{r0}")
                L(f"v_accvgpr_read_b32 v{Rr + 3}, a{r0 + 1}")
                L(f"v_lshlrev_b32 v{Rr}, 16, v{Rr + 1}")
                L(f"v_and_b32 v{Rr + 1}, 0xffff0000, v{Rr + 1}")
                L(f"v_lshlrev_b32 v{Rr + 2}, 16, v{Rr + 3}")
                L(f"v_and_b32 v{Rr + 3}, 0xffff0000, v{Rr + 3}")
                return
            L(f"v_lshlrev_b32 v{Rr}, 16, v{r0}")
            L(f"v_and_b32 v{Rr + 1}, 0xffff0000, v{r0}")
            L(f"v_lshlrev_b32 v{Rr + 2}, 16, v{r0 + 1}")
            L(f"v_and_b32 v{Rr + 3}, 0xffff0000, v{r0 + 1}")

        if abl & 256:                                     # (timing-only ablation 256: no epilogue arithmetic, the stores alone)
            L(f"global_store_dwordx2 {v('oo', j)}, v[{X}:{X + 1}], {s2('out')} offset:{32 * i}")
            return
        if mode == "act":
            L(f"v_pk_mul_f32 v[{Y}:{Y + 1}], v[{X}:{X + 1}], {s2('klog2e2')}")
            L(f"v_pk_mul_f32 v[{Y + 2}:{Y + 3}], v[{X + 2}:{X + 3}], {s2('klog2e2')}")
            for e in range(4):
                L(f"v_exp_f32 v{Y + e}, v{Y + e}")
            L(f"v_pk_add_f32 v[{Y}:{Y + 1}], v[{Y}:{Y + 1}], {s2('kone2')}")
            L(f"v_pk_add_f32 v[{Y + 2}:{Y + 3}], v[{Y + 2}:{Y + 3}], {s2('kone2')}")
            for e in range(4):
                L(f"v_rcp_f32 v{Y + e}, v{Y + e}")
            if RES:                                       # independent work between the rcp and its consumer (hz: transcendental -> consumer)
                unpack()
            else:
                L("s_nop 0")
            L(f"v_pk_mul_f32 v[{X}:{X + 1}], v[{X}:{X + 1}], v[{Y}:{Y + 1}]")
            L(f"v_pk_mul_f32 v[{X + 2}:{X + 3}], v[{X + 2}:{X + 3}], v[{Y + 2}:{Y + 3}]")
        elif RES:
            unpack()
        if RES:
            L(f"v_pk_add_f32 v[{X}:{X + 1}], v[{X}:{X + 1}], v[{Rr}:{Rr + 1}]")
            L(f"v_pk_add_f32 v[{X + 2}:{X + 3}], v[{X + 2}:{X + 3}], v[{Rr + 2}:{Rr + 3}]")
        if dst is not None:                               # wide stores: the caller pairs this block's 8 bytes with a neighbour's
            L(f"v_cvt_pk_bf16_f32 v{dst}, v{X}, v{X + 1}")
            L(f"v_cvt_pk_bf16_f32 v{dst + 1}, v{X + 2}, v{X + 3}")
            return
        L(f"v_cvt_pk_bf16_f32 v{Y}, v{X}, v{X + 1}")
        L(f"v_cvt_pk_bf16_f32 v{Y + 1}, v{X + 2}, v{X + 3}")
        if not abl & 128:                                 # (timing-only ablation 128: no output stores)
            import os as _os1
            pol = _os1.environ.get("AQ_GEN_ST_POLICY", "")    # experiment: cache policy of the output stores ("nt", "sc1", "sc0 sc1")
            L(f"global_store_dwordx2 {v('oo', j)}, v[{Y}:{Y + 1}], {s2('out')} offset:{32 * i}" + (f" {pol}" if pol else ""))

    # first B fragments
    def b_read(n):
        h, j = divmod(n, NB)
        t, ks = divmod(h, KS)
        # the pixel block's address moves once per kernel ROW (taps 0, 3, 6); the column and the k-step are immediate offsets
        if F8:
            # step t = tap pair: this lane's tap offset, then the swizzle (byte bit 5 ^= bit 8), two reads of 16 bytes
            bq = V.names["B"][0] + 8 * (n % (PD + 1))
            if t == 0:
                E(f"v_add_u32 {v('addr', j)}, {s('delta0')}, {v('addr', j)}")
            E(f"v_add_u32 {T[2]}, {v('tapoff', t)}, {v('addr', j)}")
            E(f"v_lshrrev_b32 {T[3]}, 3, {T[2]}")
            E(f"v_and_b32 {T[3]}, 32, {T[3]}")
            E(f"v_xor_b32 {T[2]}, {T[2]}, {T[3]}")
            if not abl & 4:
                E(f"ds_read_b128 v[{bq}:{bq + 3}], {T[2]}")
                E(f"ds_read_b128 v[{bq + 4}:{bq + 7}], {T[2]} offset:16")
            return
        if PM:
            # tap (dy, dx) = divmod(t, 3): the block's address for column offset dx, which moves to the chunk's ring buffer at its first
            # use (taps 0, 1, 2) and one kernel row down at each later one; the k-step is the other half-plane
            dy, dx = divmod(t, 3)
            if ks == 0 and (dy == 0 or not WFIX):
                E(f"v_add_u32 {v('addr', 3 * j + dx)}, {s('delta0') if dy == 0 else s('dRow')}, {v('addr', 3 * j + dx)}")
            if not abl & 4:
                E(f"ds_read_b128 {vr('B', 4 * (n % (PD + 1)), 4)}, {v('addr', 3 * j + dx)} offset:{ks * PSTR + (dy * (WFIX + 1) * 64 if WFIX else 0)}")
            return
        if S2:
            # tap (dy, dx) = divmod(t, 3): parity plane (dy != 1, dx != 1) -- plane index as the waves load them, 2 py + px -- as the
            # immediate offset; the (row, column) offset (dy >= 1, dx >= 1) picks one of the block's four swizzled addresses, each of
            # which moves to the chunk's ring buffer at its first use (taps 0, 1, 3, 4)
            dy, dx = divmod(t, 3)
            o = 2 * (dy >= 1) + (dx >= 1)
            if t in (0, 1, 3, 4):
                E(f"v_add_u32 {v('addr', 4 * j + o)}, {s('delta0')}, {v('addr', 4 * j + o)}")
            if not abl & 4:
                E(f"ds_read_b128 {vr('B', 4 * (n % (PD + 1)), 4)}, {v('addr', 4 * j + o)} offset:{(2 * (dy != 1) + (dx != 1)) * PSTR}")
            return
        if ks == 0 and t == 0:
            E(f"v_add_u32 {v('addr', j)}, {s('delta0')}, {v('addr', j)}")
        elif ks == 0 and t % 3 == 0:
            E(f"v_add_u32 {v('addr', j)}, {s('dRow')}, {v('addr', j)}")
        off = 16 * (t % 3) + (4 * PS if ks else 0)
        if not abl & 4:
            E(f"ds_read_b128 {vr('B', 4 * (n % (PD + 1)), 4)}, {v('addr', j)} offset:{off}")

    for n in range(PD):
        b_read(n)
    # ---- the element stream ----
    # Vector-memory operations of one tap, in issue order: (element, kind, ...).  Weights of tap t + 2 at the odd elements 1 .. 11,
    # two LDS-DMA instructions in the first NG taps, residual loads (one or two pixel blocks of three) in taps 0 .. 6.
    kD0, kD1 = DPOS[0], DPOS[-1]
    in_stream_res = RES and RES_EARLY
    res_groups = []                       # res_groups[t] = pixel blocks whose residual is loaded in tap t
    if in_stream_res and F8:
        nxt = 0
        for t in range(NT):
            cnt = min(NB - nxt, -(-(NB - nxt) // (4 - t)), len(RPOS)) if t < 4 else 0
            res_groups.append(list(range(nxt, nxt + cnt)))
            nxt += cnt
        assert nxt == NB
    elif in_stream_res:
        nxt = 0
        for t in range(9):
            cnt = min(NB - nxt, -(-(NB - nxt) // (7 - t))) if t < 7 else 0
            res_groups.append(list(range(nxt, nxt + cnt)))
            nxt += cnt
        assert nxt == NB and kD1 + 2 < 2 * NB

    def tap_ops(t, last=True):
        import os as _os
        # spacing of a tap's weight loads, in elements.  Pixel-major family, stream cycles (192 ch / 384 ch): every element 89.6 k / 81.1 k,
        # every 2nd 87.1 k / 80.4 k, every 3rd 85.5 k / 80.7 k, every 4th (with the LDS-DMA at elements 14 and 22) 84.6 k / 79.8 k
        astride = int(_os.environ.get("AQ_GEN_ASTRIDE", "4")) if PM else int(_os.environ.get("AQ_GEN_ASTRIDE_S2", "2")) if S2 else 2
        ops = [(astride * k + 1, "A", (t + LOOK) % NT, k) for k in range(NLOAD) if not abl & 1]
        if (S2 or F8) and not abl & 2:
            ops += [(DPOS[i_], "D", g_, 0) for i_, g_ in enumerate(DMA_TAPS[t])]
        elif not S2 and not F8 and t < (NDMA if PM else NG) and not abl & 2:
            ops += [(DPOS[h], "D", t, h) for h in range(PPW)]
        if in_stream_res and last:
            for g, j in enumerate(res_groups[t]):
                ops += [((RPOS[g] if F8 else (kD0, kD1)[g] + 2), "R", j, i) for i in range(3)]
        return sorted(ops, key=lambda o: o[0])

    def younger_than(tap, k_last, t_wait, e_wait, last):
        """Vector-memory operations issued after weight load k_last of tap `tap` and before element e_wait of tap t_wait, in a tile's
        last chunk (with the residual loads) or another one; the chunk before is never a last chunk that matters: its residual loads
        are older than the weights of taps 0 and 1."""
        seq = [o for tt in range(NT) for o in tap_ops(tt, False)] + [o for tt in range(t_wait) for o in tap_ops(tt, last)]
        seq += [o for o in tap_ops(t_wait, last) if o[0] < e_wait]
        idx = max(i for i, o in enumerate(seq) if o[1] == "A" and o[2] == tap and o[3] == k_last)
        return len(seq) - 1 - idx

    def wait_weights(tap, k_last, t_wait, e_wait):
        """Weight loads 0 .. k_last of tap `tap` have landed, at element e_wait of tap t_wait."""
        if abl & 1:
            return
        kN = younger_than(tap, k_last, t_wait, e_wait, False)
        kL = younger_than(tap, k_last, t_wait, e_wait, True)
        assert kN <= kL <= 63
        cases = []                                        # (flag register, immediate)
        if kL != kN:
            cases.append((s("lastc"), kL))
        if tap < LOOK and t_wait < LOOK:                  # loaded before the previous tile's epilogue
            cases.append((s("extra"), min(63, kN + n_extra)))
        # the common case falls through (a taken branch costs the wave its instruction buffer); the others wait out of line
        ld = uid("wd")
        for flag, imm in cases:
            lx = uid("wx")
            E(f"s_cmp_eq_u32 {flag}, 1")
            E(f"s_cbranch_scc1 {lx}")
            cold.append((lx, imm, ld))
        E(f"s_waitcnt vmcnt({kN})", f"weights of tap {tap}, loads 0 .. {k_last}")
        label(ld)

    for n in range(NE):
        h, j = divmod(n, NB)
        t, ks = divmod(h, KS)
        e = ks * NB + j
        post = []                                         # KS = 1 families: the LDS-DMA itself, behind this element's MFMAs
        if n + PD < NE:
            b_read(n + PD)
        if j == 0 and F8:
            wait_weights(t, 5, t, e)                      # the step's three 8-register fragments: six loads
        elif j == 0 and not w8:
            wait_weights(t, 2 + 3 * ks, t, e)             # bf16 fragments: three per k-step
        elif j == 0 and ks == 0:
            wait_weights(t, 2, t, e)                      # raw pair 2 of this tap: converted (k-step 1) under this k-step's MFMAs
        elif j == 0:
            wait_weights((t + 1) % 9, 1, t, e)            # raw pairs 0, 1 of the next tap: its k-step 0 is converted under this one
        for op in tap_ops(t):
            if op[0] != e:
                continue
            if op[1] == "A":
                if op[3] == 0:
                    emit_set_a_base("a_ld", t + LOOK)
                emit_load_a((t + LOOK) % (LOOK + 1), op[3], s2("a_ld"), 0)
            elif op[1] == "D" and (KS == 1 or PM):
                emit_dma_m0(op[2], op[3])
                post.append((op[2], op[3]))
            elif op[1] == "D":
                emit_dma(op[2], op[3], s("cd"), s("bd"))
            elif op[1] == "R" and op[3] == 0:
                # out of line as well: only a tile's last chunk takes the branch
                jr = op[2]
                lx, ld = uid("res"), uid("resd")
                E(f"s_cmp_eq_u32 {s('lastc')}, 1")
                E(f"s_cbranch_scc1 {lx}")
                label(ld)
                body = [f"v_add_u32 {T[4]}, {s('n0')}, {v('l15')}"]
                if jr:
                    body.append(f"v_add_u32 {T[4]}, {16 * jr}, {T[4]}")
                body += [f"v_min_i32 {T[4]}, {s('rlim')}, {T[4]}", f"v_mul_lo_u32 {T[4]}, {T[4]}, {s('res_ld')}", f"v_add_u32 {T[4]}, {T[4]}, {T[7]}"]
                body += [f"global_load_dwordx2 {'a' if RES_ACC else 'v'}[{rreg(3 * jr + i)}:{rreg(3 * jr + i) + 1}], {T[4]}, {s2('res')} offset:{32 * i}" for i in range(3)]
                cold.append((lx, body, ld))
        # one wait per PAIR of elements (fragments n and n + 1 have landed): the stream is bound by instruction issue
        allowed = min(PD, NE - 1 - n)
        if F8:
            E(f"s_waitcnt lgkmcnt({2 * allowed})", "two reads per fragment")
        elif n % 2 == 0:
            E(f"s_waitcnt lgkmcnt({max(allowed - 1, 0) if n + 1 < NE else allowed})")
        for i in range(3 if not abl & 8 else 0):
            if F8:
                E(f"v_mfma_f32_16x16x128_f8f6f4 {acc(i, j)}, {areg(t % (LOOK + 1), i)}, {vr('B', 8 * (n % (PD + 1)), 8)}, {acc(i, j)}")
                continue
            srca = bfreg(ks, i) if w8 else areg(t % (LOOK + 1), 3 * ks + i)
            E(f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {srca}, {vr('B', 4 * (n % (PD + 1)), 4)}, {acc(i, j)}")
        for k_, h_ in post:
            if abl & 8:
                E("s_nop 0", "hz: m0 write -> LDS-DMA (no MFMAs in between in this ablation)")
            emit_dma_issue(k_, h_)
        if w8 and j < 12:
            # one quarter of a fragment of the NEXT half-tap per element: two VALU instructions in the shadow of three MFMAs
            if ks == 0:
                emit_convert(t, 1, j // 4, j % 4)
            else:
                emit_convert((t + 1) % 9, 0, j // 4, j % 4)
    # ---- chunk end ----
    stamp(PH_STREAM)
    # next chunk: delta0 = (next buffer - this buffer) * CHUNK - 2 Wp * 16
    E(f"s_add_u32 {s('tmp0')}, {s('buf')}, 1")
    E(f"s_cmp_ge_u32 {s('tmp0')}, {RING}")
    E(f"s_cselect_b32 {s('tmp0')}, 0, {s('tmp0')}", "next buffer")
    E(f"s_sub_i32 {s('tmp1')}, {s('tmp0')}, {s('buf')}")
    E(f"s_mul_i32 {s('tmp1')}, {s('tmp1')}, {CHUNK}")
    if S2 or F8 or (PM and WFIX):
        E(f"s_mov_b32 {s('delta0')}, {s('tmp1')}", "stride-2 / fp8 / fixed-width families: the row offsets are not accumulated in the address registers")
    else:
        E(f"s_lshl_b32 {s('tmp2')}, {s('Wp')}, {7 if PM else 5}", "2 Wp * 16 (pixel-major rows: * 64): back from kernel row 2 to row 0")
        E(f"s_sub_i32 {s('delta0')}, {s('tmp1')}, {s('tmp2')}")
    E(f"s_mov_b32 {s('buf')}, {s('tmp0')}")
    E(f"s_mov_b32 {s('first')}, 0")
    E(f"s_add_u32 {s('c')}, {s('c')}, 1")
    E(f"s_cmp_lt_u32 {s('c')}, {s('CC')}")
    E(f"s_cbranch_scc1 .Lchunk_{name}")

    # =========================================== epilogue ===========================================
    if OCC > 1:
        E("s_setprio 3", "short instructions that free the way: ahead of the other workgroup's MFMA stream")
    if RES and not RES_EARLY:
        # the tile's residual, into registers the stream has finished with (every MFMA that read them has issued)
        E(f"s_sub_u32 {s('tmp1')}, {s('npix')}, 1")
        E(f"v_lshl_add_u32 {T[7]}, {v('q')}, 2, {s('cbase')}")
        E(f"v_lshlrev_b32 {T[7]}, 1, {T[7]}")
        for j in range(NB):
            E(f"v_add_u32 {T[4]}, {s('n0')}, {v('l15')}")
            if j:
                E(f"v_add_u32 {T[4]}, {16 * j}, {T[4]}")
            E(f"v_min_i32 {T[4]}, {s('tmp1')}, {T[4]}")
            E(f"v_mul_lo_u32 {T[4]}, {T[4]}, {s('res_ld')}")
            E(f"v_add_u32 {T[4]}, {T[4]}, {T[7]}")
            for i in range(3):
                E(f"global_load_dwordx2 v[{rreg(3 * j + i)}:{rreg(3 * j + i) + 1}], {T[4]}, {s2('res')} offset:{32 * i}")
    else:
        E("s_nop 15", "hz: MFMA result -> VALU read")
        E("s_nop 15")
    if RES and RES_EARLY:
        tail = [o for tt in range(NT) for o in tap_ops(tt, True)]
        k_res = len(tail) - 1 - max(i for i, o in enumerate(tail) if o[1] == "R")
        E(f"s_waitcnt vmcnt({k_res})", "the residual (only weight loads of the next tile's first taps are younger)")
    SC = V.names["SC"][0] if F8 else V.names["A"][0] + 60 if not A_ACC else 0    # fp8 weights: per-channel scales 2^e of this lane's 3 x 4 output channels
    if w8 or F8:
        for i in range(3):
            E(f"v_lshl_add_u32 {T[6]}, {v('q')}, 2, {s('cbase')}")
            E(f"v_lshlrev_b32 {T[6]}, 2, {T[6]}")
            E(f"v_add_u32 {T[6]}, {BIAS_OFF + 4096}, {T[6]}")
            E(f"ds_read_b128 v[{SC + 4 * i}:{SC + 4 * i + 3}], {T[6]} offset:{64 * i}")
        E("s_waitcnt lgkmcnt(0)")
    lact, lepd = uid("noact"), uid("epd")
    X = V.names["t"][0] + 0      # x[0:3]
    Y = V.names["t"][0] + 4      # work
    Rr = V.names["t"][0] + 8     # residual as f32
    PAIRED = not w8 and (RES_EARLY or not RES)
    for ACT in (True, False) if PAIRED else ():
        # Two blocks of a pixel block at a time, their instructions interleaved (the v_exp -> add -> v_rcp -> mul chains are dependent, the
        # VALU issues in order: a second independent chain fills the gaps), on two sets of temporaries; EXEC is switched once per pixel
        # block -- the masked-off lanes' arithmetic is never stored.
        if ACT:
            E(f"s_cmp_eq_u32 {s('act')}, 0")
            E(f"s_cbranch_scc1 {lact}")
        else:
            label(lact)
        E(f"s_sub_i32 {s('lim')}, {s('npix')}, {s('n0')}")
        if W16:
            # lanes with odd q (16-31, 48-63): the select mask of the per-lane store addresses.  SGPR pairs that are dead between the last
            # weight load of a tile and the next chunk top: actm = the mask, a_ld = the previous pixel block's store mask, a_cur = scratch
            E(f"s_mov_b32 {s('actm')}, 0xffff0000")
            E(f"s_mov_b32 {s('actm', 1)}, 0xffff0000")
        for j in range(NB if W16 else 0):
            mode = "act" if ACT else "noact"
            P = V.names["t"][0] + 16                      # chain 1's work registers: dead when the two chains reach their conversions
            CS = V.names["cst"][0]
            TA, TB = V.names["t"][0], V.names["t"][0] + 1
            E(f"v_cmp_gt_i32 {s2('t64')}, {s('lim')}, {v('l15')}", "pixel n0 + 16 j + l15 inside the batch?")
            E(f"s_sub_i32 {s('lim')}, {s('lim')}, 16")
            E(f"s_mov_b64 exec, {s2('t64')}")
            la, lb, lc = [], [], []
            epilogue_block(0, j, mode, la, 0, dst=P)
            epilogue_block(1, j, mode, lb, 12, dst=P + 2)
            epilogue_block(2, j, mode, lc, 0, dst=CS + 2 * (j & 1))
            for k in range(max(len(la), len(lb))):
                if k < len(la):
                    E(la[k])
                if k < len(lb):
                    E(lb[k])
            # M blocks 0 and 1: even-q lanes store bytes 0 .. 15 of block 0 (their own 8 and lane q + 1's), odd-q lanes bytes 0 .. 15 of block 1
            # (lane q - 1's 8 and their own), i.e. 32 - 8 = 24 bytes beyond their own offset.  (hz: VALU write -> v_permlane read, 2 wait
            # states: the two address instructions sit in between)
            E(f"v_cndmask_b32 v{TA}, 0, 24, {s2('actm')}")
            E(f"v_add_u32 v{TA}, v{TA}, {v('oo', j)}")
            E(f"v_permlane16_swap_b32 v{P}, v{P + 2}")
            E(f"v_permlane16_swap_b32 v{P + 1}, v{P + 3}")
            if not abl & 128:
                E(f"global_store_dwordx4 v{TA}, v[{P}:{P + 3}], {s2('out')}")
            for l in lc:
                E(l)
            if j % 2 == 0 and j + 1 < NB:
                E(f"s_mov_b64 {s2('a_ld')}, {s2('t64')}", "this block's mask: its M block 2 leaves with the next block's")
            elif j % 2 == 0:
                if not abl & 128:                         # the last pixel block of an odd count: its M block 2 alone, 8 bytes per lane
                    E(f"global_store_dwordx2 {v('oo', j)}, v[{CS}:{CS + 1}], {s2('out')} offset:64")
            else:
                # M block 2 of pixel blocks j - 1 (even-q lanes) and j (odd-q lanes).  Every lane pair of a pixel that either block stores
                # must swap: block j - 1's mask is the superset
                E(f"s_mov_b64 exec, {s2('a_ld')}")
                E(f"v_add_u32 v{TB}, 8, {v('oo', j - 1)}")
                E(f"v_cndmask_b32 v{TA}, v{TB}, {v('oo', j)}, {s2('actm')}", "even q: own offset + 64; odd q: own offset + 64 - 8 (lane q - 1's piece first)")
                E(f"v_permlane16_swap_b32 v{CS}, v{CS + 2}")
                E(f"v_permlane16_swap_b32 v{CS + 1}, v{CS + 3}")
                E(f"s_andn2_b64 {s2('a_cur')}, {s2('a_ld')}, {s2('actm')}")
                E(f"s_and_b64 {s2('t64')}, {s2('t64')}, {s2('actm')}")
                E(f"s_or_b64 exec, {s2('a_cur')}, {s2('t64')}")
                if not abl & 128:
                    E(f"global_store_dwordx4 v{TA}, v[{CS}:{CS + 3}], {s2('out')} offset:56")
            E("s_mov_b64 exec, -1")
        for j in range(0 if W16 else NB):
            E(f"v_cmp_gt_i32 {s2('t64')}, {s('lim')}, {v('l15')}", "pixel n0 + 16 j + l15 inside the batch?")
            E(f"s_sub_i32 {s('lim')}, {s('lim')}, 16")
            E(f"s_mov_b64 exec, {s2('t64')}")
            la, lb, lc = [], [], []
            epilogue_block(0, j, "act" if ACT else "noact", la, 0)
            epilogue_block(1, j, "act" if ACT else "noact", lb, 12)
            epilogue_block(2, j, "act" if ACT else "noact", lc, 24 if EP3 else 0)
            for k in range(max(len(la), len(lb))):
                if k < len(la):
                    E(la[k])
                if k < len(lb):
                    E(lb[k])
                if EP3 and k < len(lc):                  # all three M blocks of the pixel block in lock step, on three sets of temporaries
                    E(lc[k])
            for l in () if EP3 else lc:
                E(l)
            E("s_mov_b64 exec, -1")
        if ACT:
            E(f"s_branch {lepd}")
    for ACT in () if PAIRED else (True, False):
        if ACT:
            E(f"s_cmp_eq_u32 {s('act')}, 0")
            E(f"s_cbranch_scc1 {lact}")
        else:
            label(lact)
        for j in range(NB):
            E(f"s_sub_i32 {s('lim')}, {s('npix')}, {s('n0')}")
            E(f"s_sub_i32 {s('lim')}, {s('lim')}, {16 * j}", "pixels of this block inside the batch")
            E(f"v_cmp_gt_i32 {s2('t64')}, {s('lim')}, {v('l15')}", "pixel n0 + 16 j + l15 inside the batch?")
            for i in range(3):
                a0 = 4 * (3 * j + i)
                r0 = rreg(3 * j + i)
                for e in range(4):
                    E(f"v_accvgpr_read_b32 v{X + e}, a{a0 + e}")
                if RES and not RES_EARLY:
                    E(f"s_waitcnt vmcnt({3 * NB - 1})", "this block's residual: the younger loads and the stores issued so far stay in flight")
                if w8:                                   # exact: the scales are powers of two
                    E(f"v_pk_mul_f32 v[{X}:{X + 1}], v[{X}:{X + 1}], v[{SC + 4 * i}:{SC + 4 * i + 1}]")
                    E(f"v_pk_mul_f32 v[{X + 2}:{X + 3}], v[{X + 2}:{X + 3}], v[{SC + 4 * i + 2}:{SC + 4 * i + 3}]")

                def unpack_residual():
                    E(f"v_lshlrev_b32 v{Rr}, 16, v{r0}")
                    E(f"v_and_b32 v{Rr + 1}, 0xffff0000, v{r0}")
                    E(f"v_lshlrev_b32 v{Rr + 2}, 16, v{r0 + 1}")
                    E(f"v_and_b32 v{Rr + 3}, 0xffff0000, v{r0 + 1}")

                if ACT:
                    E(f"v_pk_mul_f32 v[{Y}:{Y + 1}], v[{X}:{X + 1}], {s2('klog2e2')}")
                    E(f"v_pk_mul_f32 v[{Y + 2}:{Y + 3}], v[{X + 2}:{X + 3}], {s2('klog2e2')}")
                    for e in range(4):
                        E(f"v_exp_f32 v{Y + e}, v{Y + e}")
                    E(f"v_pk_add_f32 v[{Y}:{Y + 1}], v[{Y}:{Y + 1}], {s2('kone2')}")       # hz: 3 instructions after the exp that wrote v[Y]
                    E(f"v_pk_add_f32 v[{Y + 2}:{Y + 3}], v[{Y + 2}:{Y + 3}], {s2('kone2')}")
                    for e in range(4):
                        E(f"v_rcp_f32 v{Y + e}, v{Y + e}")
                    if RES:       # independent work between the rcp and its consumer (hz: transcendental -> consumer)
                        unpack_residual()
                    else:
                        E("s_nop 0")
                    E(f"v_pk_mul_f32 v[{X}:{X + 1}], v[{X}:{X + 1}], v[{Y}:{Y + 1}]")
                    E(f"v_pk_mul_f32 v[{X + 2}:{X + 3}], v[{X + 2}:{X + 3}], v[{Y + 2}:{Y + 3}]")
                elif RES:
                    unpack_residual()
                if RES:
                    E(f"v_pk_add_f32 v[{X}:{X + 1}], v[{X}:{X + 1}], v[{Rr}:{Rr + 1}]")
                    E(f"v_pk_add_f32 v[{X + 2}:{X + 3}], v[{X + 2}:{X + 3}], v[{Rr + 2}:{Rr + 3}]")
                E(f"v_cvt_pk_bf16_f32 v{Y}, v{X}, v{X + 1}")
                E(f"v_cvt_pk_bf16_f32 v{Y + 1}, v{X + 2}, v{X + 3}")
                E(f"s_mov_b64 exec, {s2('t64')}")
                E(f"global_store_dwordx2 {v('oo', j)}, v[{Y}:{Y + 1}], {s2('out')} offset:{32 * i}")
                E("s_mov_b64 exec, -1")
        if ACT:
            E(f"s_branch {lepd}")
    label(lepd)
    if stamped:
        E("s_waitcnt vmcnt(0)", "stamped build: the stores' drain belongs to the epilogue")
    stamp(PH_EPILOGUE)
    E(f"s_cmp_eq_u32 {s('has_next')}, 0")
    E(f"s_cbranch_scc1 .Lend_{name}")
    E(f"s_mov_b32 {s('tile')}, {s('next_tile')}")
    E(f"s_branch .Ltile_{name}")
    for lx, what, ld in cold:
        label(lx)
        for line in ([f"s_waitcnt vmcnt({what})"] if isinstance(what, int) else what):
            E(line)
        E(f"s_branch {ld}")
    label(".Lend_" + name)
    E("s_waitcnt vmcnt(0)")
    if stamped:
        # row (wg * 4 + wave) of the stamp buffer: six phase sums, their total, elapsed 100 MHz ticks
        E(f"s_memrealtime {s2('t64')}")
        E("s_waitcnt lgkmcnt(0)")
        E(f"s_sub_u32 {s('st_rt0')}, {s('t64')}, {s('st_rt0')}")
        E(f"s_subb_u32 {s('st_rt0', 1)}, {s('t64', 1)}, {s('st_rt0', 1)}")
        E(f"s_mov_b64 {s2('st_last')}, 0")
        for k in range(6):
            E(f"s_add_u32 {s('st_last')}, {s('st_last')}, {s('st_acc', 2 * k)}")
            E(f"s_addc_u32 {s('st_last', 1)}, {s('st_last', 1)}, {s('st_acc', 2 * k + 1)}")
        E(f"s_lshl_b32 {s('tmp0')}, {s('wg')}, 2")
        E(f"s_add_u32 {s('tmp0')}, {s('tmp0')}, {s('wave')}")
        E(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 6")
        E(f"v_mov_b32 {v('t', 2)}, {s('tmp0')}")
        E("s_mov_b64 exec, 1")
        vals = [(s('st_acc', 2 * k), s('st_acc', 2 * k + 1)) for k in range(6)] + [(s('st_last'), s('st_last', 1)), (s('st_rt0'), s('st_rt0', 1))]
        for k, (lo, hi) in enumerate(vals):
            E(f"v_mov_b32 {v('t', 0)}, {lo}")
            E(f"v_mov_b32 {v('t', 1)}, {hi}")
            E(f"global_store_dwordx2 {v('t', 2)}, {vr('t', 0, 2)}, {s2('debug')} offset:{8 * k}")
            E("s_waitcnt vmcnt(0)")
    E("s_endpgm")
    return list(out)


def lds_bytes():
    return LDS_BYTES + (4096 if W8 else 0)     # (the fp8 family's LDS_BYTES already holds both vectors)


def register_budget():
    acc_off = (V.next + 3) // 4 * 4
    total = (acc_off + n_acc() + 7) // 8 * 8
    assert total <= 512 // OCC, f"NB = {NB}: {total} registers do not leave room for {OCC} wave(s) per SIMD"
    assert OCC * lds_bytes() <= 160 * 1024
    return acc_off, total


def descriptor(name):
    acc_off, total = register_budget()
    return f"""
	.rodata
	.p2align 6
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size {lds_bytes()}
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size {ARG_BYTES}
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr {total}
		.amdhsa_next_free_sgpr {S.next}
		.amdhsa_accum_offset {acc_off}
		.amdhsa_reserve_vcc 1
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text
"""


def metadata_entry(name):
    acc_off, total = register_budget()
    return f"""  - .agpr_count:     {total - acc_off}
    .args:
      - .offset:         0
        .size:           {ARG_BYTES}
        .value_kind:     by_value
    .group_segment_fixed_size: {lds_bytes()}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {ARG_BYTES}
    .max_flat_workgroup_size: 256
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     {S.next + 6}
    .symbol:         {name}.kd
    .vgpr_count:     {total}
    .wavefront_size: 64
"""


def metadata(entries):
    return f"""	.amdgpu_metadata
---
amdhsa.kernels:
{"".join(entries)}amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
"""


DIAG = True      # also emit the timing-only ablations of the stamped build (tools/time_conv3x3.py --stamp --abl)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "conv3x3_pl_asm.s"
    text = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.text"]
    entries = []
    import os
    experimental = os.environ.get("AQ_GEN_EXPERIMENTAL") == "1"
    for nb in sorted(CONFIGS, key=str):
        if nb in (7, 8) and not experimental:      # two workgroups per CU: parity-green, no faster on any BASELINE geometry (DESIGN.md 4.1a point 2);
            continue                               # kept as a generator option, out of the shipped code object
        configure(nb)
        variants = [(False, False, 0, False), (True, False, 0, False), (True, True, 0, False)]
        if S2:
            variants = [(False, False, 0, False), (False, True, 0, False)] + ([(False, True, a, False) for a in (1, 2, 3, 4, 7, 8)] if DIAG else [])
        if F8 and DIAG:
            variants += [(True, True, a, False) for a in (1, 2, 4, 7, 8)]
        if nb == 13:
            variants += [(False, False, 0, True), (True, False, 0, True), (True, True, 0, True)]     # fp8-weight stream
        if DIAG and nb == "pm13w40":
            variants += [(True, True, a, False) for a in (1, 2, 4, 8, 64, 128, 256)]
        if DIAG and nb in (7, 13):
            variants += [(True, True, a, False) for a in (1, 2, 3, 4, 7, 8) + ((16, 32) if nb == 13 else ())]
        for RES, stamped, abl, w8 in variants:
            name = f"conv3x3_pl_asm_{FAMILY}_res{int(RES)}" + ("_w8" if w8 else "") + ("_stamped" if stamped else "") + (f"_abl{abl}" if abl else "")
            text += [f"\t.globl\t{name}", "\t.p2align\t8", f"\t.type\t{name},@function"]
            text += gen_kernel(name, RES, stamped, abl, w8)
            entries.append(metadata_entry(name))
            text += [f".Lfend_{name}:", f"\t.size\t{name}, .Lfend_{name}-{name}", descriptor(name)]
        print(f"{FAMILY}: {V.next} VGPRs + {n_acc()} AGPRs, {S.next} SGPRs, {LDS_BYTES} B LDS, {OCC} workgroup(s) per CU")
    text.append(metadata(entries))
    with open(path, "w") as f:
        f.write("\n".join(text) + "\n")
    print(f"wrote {path}: {sum(1 for l in text if 'v_mfma' in l)} MFMAs")


if __name__ == "__main__":
    main()
