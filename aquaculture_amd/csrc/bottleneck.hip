// Fused Bottleneck for gfx950, hidden widths C = 16, 32, 48, 64 and 96 (bf16):
//     y = (x +) SiLU(cv2_3x3(SiLU(cv1_1x1(x) + b1)) + b2)        [UPSTREAM models/common.py Bottleneck.forward]
// (BN folded into both convolutions; reached through reference README.md:77 -> yolov5/detect.py -> C3.m).
//
// At 160x160 / 80x80 with 48 / 96 channels (yolov5m's C3 stages 2, 4 and 17) the two-kernel form is bound by HBM round
// trips and by the 3x3 kernel re-gathering nine shifted copies of its input through L2: cv1 writes t, cv2 reads it back
// nine times plus the shortcut.  Here one workgroup owns a small output tile and keeps everything on chip:
//   A. the input patch x (tile + 1-pixel halo, zero outside the image) goes to LDS once, by LDS-DMA (global_load_lds,
//      16 B per lane, no registers) into the other half of a double buffer while the previous tile is being computed;
//   B. t = SiLU(W1 x + b1) is evaluated for EVERY patch pixel (the halo is recomputed) on MFMA and written to a second LDS
//      patch as bf16 -- exactly the rounding the two-kernel path applies when it stores t; pixels outside the image are
//      forced to 0 (the 3x3 convolution pads t, not x);
//   C. the 3x3 convolution reads its MFMA B fragments straight from the t patch (a pixel's 8-channel block of one tap is a
//      contiguous, 16-byte aligned run), adds the shortcut from the x patch and stores y.
// The 3x3 weights live in REGISTERS as MFMA A fragments for the life of the persistent workgroup (v_mfma_f32_16x16x32_bf16,
// M = C in 16-row blocks, no channel padding); LDS serves activation fragments (and, where registers are short, the last
// k-steps of W2).  Two families of tile shapes (table at btl_shape()):
//   * "one M block per wave" (C = 48, 96: 12 waves, 3 per SIMD, <= 168 registers, weights in plain VGPRs): every MFMA needs
//     its own 1 KB fragment read, which is exactly the LDS's 256 B/clk at the full MFMA rate -- LDS-bound;
//   * "wide" (3-4 M blocks per wave, 4 waves, ~500 registers, one wave per SIMD; C = 64 and AQ_BTL_WIDE=1): a third of the
//     LDS traffic, but the compiler parks most of W2 in AccVGPRs and copies each fragment back before use, and a single
//     instruction stream per SIMD serialises MFMA issue, SiLU and LDS latency -- issue-bound.
// Both run at ~28 cycles per MFMA (16 is the pipe rate); the 12-wave shapes are a few percent faster on yolov5m.
// Round 4: yolov5m's two widths run in generated assembly instead -- C = 48: gen_bottleneck_asm.py (8 waves, every wave all three M blocks and
// all the weights, the two waves of a SIMD half a tile out of step); C = 96: gen_bottleneck96_asm.py (4 waves, 81 + 9 fragments per wave
// resident, 64 of them in AGPRs) -- the "wide" idea with the registers and the instruction order assigned by hand.  Their host side is at the
// end of this file; the HIP kernel stays as the specification and the fallback.
// The output must not alias the input (neighbouring tiles read each other's halo): the plan ping-pongs Bottleneck pairs
// between the C3 concat buffer and its temporary.
#include "conv_device.h"

using namespace aqdev;

namespace {

struct BtlParams {
    const char* in;
    char* out;
    const char* w;           // A-fragment image, see aq_pack_bottleneck_weights
    const float* bias;       // [2C]: b1 | b2
    int in_ld_b, out_ld_b;
    int B, H, W;
    int tiles_x, tiles_y, n_tiles;
    int shortcut;
    unsigned long long* debug;   // STAMP builds only (tools/stamp_conv.py)
    const char* zero;            // >= 16 zero bytes: DMA source for pixels outside the image
};

// MBW: 16-row M blocks per wave; MSPLIT: wave groups that split the output channels; NW: waves per workgroup (NW / MSPLIT
// pixel groups of 4 output rows each); TW: tile width in pixels; KT: last k-steps of W2 kept in LDS instead of registers;
// W1REG: the 1x1 weights stay in registers too (else they are re-read from LDS every tile); RESG: the shortcut is re-read from
// global memory (L2-hot) instead of the x patch, so x is dead after phase B and needs ONE LDS buffer instead of two.
template <int MBW, int MSPLIT, int NW, int TW, int KT, bool W1REG, bool RESG> struct BtlGeom {
    static constexpr int C = 16 * MBW * MSPLIT;
    static constexpr int CB = C / 8;                       // 8-channel (16-byte) blocks per pixel
    static constexpr int PG = NW / MSPLIT;                 // pixel groups (waves that own different rows)
    static constexpr int TH = 4 * PG;                      // every wave owns 4 output rows
    static constexpr int PH = TH + 2, PW = TW + 2, PP = PH * PW;
    // 16-byte LDS slots per pixel (channel blocks + pad).  ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): with lanes (pixel = lane & 15, K block = lane >> 4) a group is
    // conflict-free when the slot stride is 2 mod 4 -- 6 for C = 48 (no padding at all), 14 for C = 96; a stride of CB + 1
    // would cost two LDS cycles per group on every fragment read.  (C = 64: stride 10 does not fit the LDS, 9 is 2-way.)
    static constexpr int SPP = C == 16 ? 2 : C == 32 ? 6 : C == 48 ? 6 : C == 64 ? 9 : 14;
    static constexpr int PXB = SPP * 16;                   // LDS pixel stride in bytes
    static constexpr int KS1 = (CB + 3) / 4;               // k-steps (32 K each) of the 1x1
    static constexpr int NBLK2 = 9 * CB;                   // K blocks of the 3x3, tap-major
    static constexpr int KS2 = (NBLK2 + 3) / 4;
    static constexpr bool UNIFORM_K = CB % 4 == 0;         // a k-step never straddles taps: K offsets are compile-time + 16 g
    static constexpr int NQ = (PP * SPP + 63) / 64;        // LDS-DMA wave instructions per x patch (64 slots each)
    static constexpr int PATCHB = NQ * 1024;               // patch buffer, rounded up to whole DMA instructions
    static constexpr int W1B = W1REG ? 0 : MSPLIT * KS1 * MBW * 1024;  // A fragments of the 1x1 (reloaded into registers every phase B)
    static constexpr int KTAIL = KT;                       // last k-steps of W2 that stay in LDS instead of
    static constexpr int KREG = KS2 - KTAIL;               // registers (C = 96: 3 of 27, C = 64: 3 of 18): keeps the kernel spill-free
    static constexpr int W2TB = MSPLIT * KTAIL * MBW * 1024;
    static constexpr int XBUFS = RESG ? 1 : 2;
    static constexpr int LDS = (XBUFS + 1) * PATCHB + W1B + W2TB + 2 * C * 4;   // x patch(es) | t patch | W1 | W2 tail | b1 | b2
    static constexpr int NBLK1 = (PP + 15) / 16;           // 16-pixel MFMA column blocks of phase B
    static constexpr int NBR = TW / 16;                    // 16-pixel blocks per output row
    static constexpr int RB = 4 / NBR;                     // rows per phase-C step (always 4 pixel blocks per step)
    static constexpr int STEPS = 4 / RB;                   // phase-C steps per tile (a wave owns 4 rows)
    static constexpr int WFRAGS = (KS1 + KS2) * MBW;       // A fragments per wave group
    static constexpr int NST = 4 * MBW;                    // output store instructions per wave and phase-C step (full tile)
    static constexpr int THREADS = NW * 64;
    static constexpr bool ROWREUSE = UNIFORM_K && NBR == 1 && MBW == 1;   // phase C loads each fragment once per (row, dx, quad)
    static constexpr bool TIGHT = NW >= 12 && KT > 0;      // 168-register budget with part of W2 resident: smallest working set
    static_assert(TW == 16 || TW == 32, "tile width");
    static_assert(NW % MSPLIT == 0 && KT < KS2, "shape");
    static_assert(LDS <= 160 * 1024, "LDS");
};

// LDS write the compiler cannot see as one: LLVM's waitcnt pass makes every visible LDS store wait for ALL outstanding
// LDS-DMA (vmcnt(0)), which would drain the next tile's prefetch at the first write of phase B.  Ordering against the
// reads of other waves is by lds_barrier() below.
__device__ __forceinline__ void lds_write_b64(char* dst, uint2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)dst), "v"(v));
}
// All of this wave's LDS operations are done, then the workgroup barrier -- without the vmcnt(0) a __syncthreads() adds.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 silu4(f32x4 v) {          // same sequence as the shared conv epilogue (bf16 mode)
    const f32x4 t = v * -1.44269504f;
    f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    d = d + 1.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    return v * r;
}

template <int MBW, int MSPLIT, int NW, int TW, int KT, bool W1REG, bool RESG, bool STAMP = false>
__global__ __launch_bounds__(NW * 64) void bottleneck_kernel(const BtlParams p) {
    using G = BtlGeom<MBW, MSPLIT, NW, TW, KT, W1REG, RESG>;
    constexpr int CB = G::CB, PXB = G::PXB, KS1 = G::KS1, KS2 = G::KS2, PW = G::PW, PP = G::PP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_t = smem + G::XBUFS * G::PATCHB;                 // x patch(es) at smem + {0, PATCHB}
    char* s_w1 = s_t + G::PATCHB;
    char* s_w2t = s_w1 + G::W1B;
    float* s_b = (float*)(s_w2t + G::W2TB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave % MSPLIT, pg = wave / MSPLIT;       // this wave's channel group / pixel group
    const int g = lane >> 4, l15 = lane & 15;
    const int H = p.H, W = p.W;
    const int cbase = mh * MBW * 16 + g * 4;                 // first of this lane's 4 output channels in M block 0

    // diagnostic build only (STAMP): per-wave cycle sums of 8 phases -> p.debug; the shipped kernels contain no stamp.
    // 0 prologue | 1 vmcnt wait | 2 tile barrier | 3 DMA issue | 4 phase B | 5 mid barrier | 6 phase C MFMA | 7 phase C epilogue
    unsigned long long ph_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = 0;
    if constexpr (STAMP) ph_t = clock64();
    auto stamp = [&](int ph) {
        if constexpr (STAMP) {
            const unsigned long long t = clock64();
            ph_sum[ph] += t - ph_t;
            ph_t = t;
        }
    };

    // ---- once per workgroup: this wave group's rows of both weight sets to registers, biases to LDS ----
    constexpr int KREG = G::KREG;
    bf16x8 a2[KREG][MBW];
    bf16x8 a1r[W1REG ? KS1 : 1][MBW];                        // W1REG: the 1x1 weights, resident
    {
        const bf16x8* wsrc = (const bf16x8*)p.w + (size_t)mh * G::WFRAGS * 64 + lane;
        if constexpr (!W1REG) {
            if (pg == 0) {                                   // one wave per channel group copies its W1 fragments to LDS
#pragma unroll
                for (int i = 0; i < KS1 * MBW; ++i) *(bf16x8*)(s_w1 + (mh * KS1 * MBW + i) * 1024 + lane * 16) = wsrc[i * 64];
            }
        }
        if constexpr (W1REG) {
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int m = 0; m < MBW; ++m) a1r[s][m] = wsrc[(s * MBW + m) * 64];
        }
#pragma unroll
        for (int s = 0; s < KREG; ++s)
#pragma unroll
            for (int m = 0; m < MBW; ++m) a2[s][m] = wsrc[((KS1 + s) * MBW + m) * 64];
        if (pg == 0) {
#pragma unroll
            for (int i = 0; i < G::KTAIL * MBW; ++i)
                *(bf16x8*)(s_w2t + (mh * G::KTAIL * MBW + i) * 1024 + lane * 16) = wsrc[((KS1 + KREG) * MBW + i) * 64];
        }
    }
    for (int i = tid; i < 2 * G::C; i += G::THREADS) s_b[i] = p.bias[i];
    // per-lane byte offset of this lane's K block: inside a pixel (1x1) / relative to the tap-(0,0) pixel (3x3).  K blocks
    // past the end carry zero weights and read block 0 (any initialised address would do).
    int koff1[G::UNIFORM_K ? 1 : KS1], koff2[G::UNIFORM_K ? 1 : KS2];
    if constexpr (G::UNIFORM_K) {
        koff1[0] = koff2[0] = g * 16;
    } else {
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const int blk = 4 * s + g;
            koff1[s] = blk < CB ? blk * 16 : 0;
        }
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            int blk = 4 * s + g;
            if (blk >= G::NBLK2) blk = 0;
            const int tap = blk / CB, cb = blk - tap * CB;
            const int dy = tap / 3, dx = tap - 3 * dy;
            koff2[s] = (dy * PW + dx) * PXB + cb * 16;
        }
    }
    auto k1 = [&](int s) -> int {                            // s is a compile-time constant after unrolling
        if constexpr (G::UNIFORM_K) return koff1[0] + 4 * s * 16;
        else return koff1[s];
    };
    auto k2 = [&](int s) -> int {
        if constexpr (G::UNIFORM_K) {
            const int tap = (4 * s) / CB, cb0 = 4 * s - tap * CB;
            return koff2[0] + ((tap / 3) * PW + tap % 3) * PXB + cb0 * 16;
        } else return koff2[s];
    };

    const int tiles_per_img = p.tiles_y * p.tiles_x;
    // x patch of `tile` -> LDS buffer xb, asynchronously: wave w issues DMA instructions w, w+4, ...; instruction q fills the
    // 64 consecutive 16-byte slots [64 q, 64 q + 64) of the patch image (slot = pixel * SPP + part; parts >= CB are padding).
    // Pixels outside the image, pad slots and slots past the patch read the zero page.
    struct PatchOrg { const char* org; int y0, x0; };      // address of patch pixel (0, 0) (may lie outside the image) + tile origin
    auto patch_org = [&](int tile) -> PatchOrg {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int y0 = ty0 * G::TH, x0 = tx0 * TW;
        return {p.in + ((long long)(b * H + y0 - 1) * W + (x0 - 1)) * p.in_ld_b, y0, x0};
    };
    auto dma_one = [&](const PatchOrg& o, int q, char* xb) {  // branch-free: it is issued from inside the MFMA loop
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                     // opaque: keeps the compiler from hoisting the per-lane slot decode of
                                                             // every call site out of the tile loop (dozens of registers for ~20 VALU ops)
        const int slot = q * 64 + lane_o;
        const int px = slot / G::SPP, part = slot - px * G::SPP;
        const int pr = px / PW, pc = px - pr * PW;
        const int iy = o.y0 - 1 + pr, ix = o.x0 - 1 + pc;
        const bool valid = px < PP && part < CB && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const char* src = o.org + (pr * W + pc) * p.in_ld_b + part * 16;
        glds16(valid ? src : p.zero, xb + q * 1024);
    };
    auto issue_dma = [&](int tile, char* xb) {
        const PatchOrg o = patch_org(tile);
#pragma unroll 1
        for (int q = wave; q < G::NQ; q += NW) dma_one(o, q, xb);
    };
    // The next tile's DMA instructions are normally issued from INSIDE the phase-C MFMA loop (their address arithmetic fills
    // VALU slots the matrix pipe leaves free); PER of them per phase-C step, one every SP k-steps.
    constexpr int STEPS = G::STEPS, NQW = (G::NQ + NW - 1) / NW, PER = (NQW + STEPS - 1) / STEPS;
    constexpr int SP = KS2 / PER >= 1 ? KS2 / PER : 1;
    static_assert(PER * SP <= KS2, "not enough k-steps to carry the DMA issue");
    constexpr int RSTEPS = 3 * (CB / 4 > 0 ? CB / 4 : 1) * 6, RSP = RSTEPS / PER >= 1 ? RSTEPS / PER : 1;   // row-reuse form of phase C

    int tile = first_tile(gridDim.x, blockIdx.x);
    if (tile < p.n_tiles) issue_dma(tile, smem);
    stamp(0);
    int cur = 0;
    bool prev_full = false;                                  // first tile: nothing but the DMA is outstanding, wait for all of it
    for (; tile < p.n_tiles; tile += gridDim.x, cur ^= 1) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int y0 = ty0 * G::TH, x0 = tx0 * TW;
        const char* s_x = smem + (RESG ? 0 : cur * G::PATCHB);
        // ---- A. this tile's x patch has landed (own DMA: vmcnt, the other waves': barrier); every wave is also done with
        //      the previous tile, so the other x buffer and the t patch are free: start the next tile's DMA ----
        // (vmcnt is in-order.  After a full tile the youngest DMA instruction is older than the last phase-C step's
        //  4 * MBW output stores, which may stay in flight; after a ragged one everything is drained.)
        if (prev_full) wait_vmcnt<G::NST>(); else wait_vmcnt<0>();
        stamp(1);
        lds_barrier();
        stamp(2);
        const bool has_next = tile + (int)gridDim.x < p.n_tiles;
        prev_full = y0 + G::TH <= H && x0 + TW <= W;         // every store instruction of this tile has active lanes
        char* xbn = smem + (RESG ? 0 : (cur ^ 1) * G::PATCHB);   // RESG: x is dead once phase B is over (mid barrier)
        PatchOrg on = {nullptr, 0, 0};
        if (has_next) on = patch_org(tile + (int)gridDim.x);
        stamp(3);
        // ---- B. t = SiLU(W1 x + b1) on all patch pixels: one or two 16-pixel blocks per step, this wave's channel group ----
        {
            f32x4 b1v[MBW];
            bf16x8 a1[KS1][MBW];                             // live in this phase only: phase C needs the registers
#pragma unroll
            for (int m = 0; m < MBW; ++m) b1v[m] = *(const f32x4*)(s_b + cbase + m * 16);
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    if constexpr (W1REG) a1[s][m] = a1r[s][m];
                    else a1[s][m] = *(const bf16x8*)(s_w1 + ((mh * KS1 + s) * MBW + m) * 1024 + lane * 16);
                }
            constexpr int NBB = G::TIGHT ? 1 : 2;            // 16-pixel blocks per iteration
#pragma unroll 1
            for (int nb0 = pg; nb0 < G::NBLK1; nb0 += NBB * G::PG) {
                int px[NBB], pxc[NBB];
                f32x4 acc[NBB][MBW];
                int l15o = l15;
                asm volatile("" : "+v"(l15o));               // opaque, as in dma_one: no per-block state hoisted across tiles
#pragma unroll
                for (int j = 0; j < NBB; ++j) {
                    px[j] = (nb0 + j * G::PG) * 16 + l15o;    // the second block may lie past the patch: clamped reads, no writes
                    pxc[j] = px[j] < PP ? px[j] : PP - 1;
#pragma unroll
                    for (int m = 0; m < MBW; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                bf16x8 f[KS1][NBB];
#pragma unroll
                for (int s = 0; s < KS1; ++s)
#pragma unroll
                    for (int j = 0; j < NBB; ++j) f[s][j] = *(const bf16x8*)(s_x + pxc[j] * PXB + k1(s));
#pragma unroll
                for (int s = 0; s < KS1; ++s)
#pragma unroll
                    for (int m = 0; m < MBW; ++m)
#pragma unroll
                        for (int j = 0; j < NBB; ++j) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s][m], f[s][j], acc[j][m], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBB; ++j) {
                    const int pr = pxc[j] / PW, pc = pxc[j] - pr * PW;
                    const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
                    const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
                    for (int m = 0; m < MBW; ++m) {
                        const f32x4 v = silu4(acc[j][m] + b1v[m]);
                        uint2 o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                        if (!inside) o = make_uint2(0, 0);   // zero padding of the 3x3 applies to t
                        if (px[j] < PP) lds_write_b64(s_t + px[j] * PXB + (cbase + m * 16) * 2, o);
                    }
                }
            }
        }
        stamp(4);
        lds_barrier();
        stamp(5);
        // ---- C. y = (x +) SiLU(W2 (*) t + b2): FOUR 16-pixel blocks per k-step (4 rows of 16 or 2 rows of 32).  The
        //      compiler keeps most of W2 in AccVGPRs and copies each fragment to VGPRs before use (4 x v_accvgpr_read);
        //      with four MFMAs per fragment the loop is bound by the matrix pipe (64 cycles per fragment against 48 issue
        //      cycles), and the spare issue slots carry the next tile's DMA address arithmetic. ----
        {
#pragma unroll 1
            for (int st = 0; st < G::STEPS; ++st) {
                const int ty = 4 * pg + st * G::RB;
                // block j: row ty + j / NBR, columns 16 * (j % NBR) ..
                int boff[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) boff[j] = ((j / G::NBR) * PW + 16 * (j % G::NBR)) * PXB;
                const char* base = s_t + (ty * PW + l15) * PXB;
                uint2 xg[RESG ? 4 : 1][MBW];                 // RESG: shortcut values, fetched from global before the MFMA loop
                if constexpr (RESG) {
                    if (p.shortcut) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int y = y0 + ty + j / G::NBR, x = x0 + 16 * (j % G::NBR) + l15;
                            const char* src = p.in + ((long long)(b * H + y) * W + x) * p.in_ld_b + cbase * 2;
                            // Loaded by asm: for a load the compiler can see it puts s_waitcnt vmcnt(0) in front of the MFMA loop below
                            // (twice), which also waits for the next patch's LDS-DMA issued there -- even in launches without a
                            // shortcut, because the wait sits after the join.  The hand-written wait is after the MFMA loop.
                            const char* ok_src = (y < H && x < W) ? src : (const char*)p.zero;
#pragma unroll
                            for (int m = 0; m < MBW; ++m)
                                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xg[j][m]) : "v"(ok_src + ((y < H && x < W) ? m * 32 : 0)) : "memory");
                        }
                    }
                }
                f32x4 acc[4][MBW];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MBW; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                auto w2frag = [&](int s, int m) -> bf16x8 {  // A fragment of k-step s (compile-time after unrolling)
                    if constexpr (G::KTAIL > 0) {
                        if (s >= KREG) return *(const bf16x8*)(s_w2t + ((mh * G::KTAIL + (s - KREG)) * MBW + m) * 1024 + lane * 16);
                        return a2[s < KREG ? s : 0][m];
                    } else return a2[s][m];
                };
                if constexpr (G::ROWREUSE) {
                    // The four blocks are four consecutive output rows: fragment (patch row r, dx, channel quad) is the B operand of
                    // every (output row j, dy) with j + dy = r -- load it ONCE and issue all of them (up to 3 MFMAs per M block).
                    // Halves the LDS reads of the shapes that are LDS-bound (one M block per wave: 1 KB of LDS per MFMA otherwise).
                    constexpr int CQ = CB / 4;
                    auto frag = [&](int dx, int cq, int r) -> bf16x8 {
                        return *(const bf16x8*)(base + (r * PW + dx) * PXB + cq * 64 + koff2[0]);
                    };
                    // two fragments in flight ahead of the one being multiplied (rotating variables: an indexed buffer would
                    // keep the loop nest from unrolling and push the weight arrays to scratch)
                    bf16x8 f0 = frag(0, 0, 0), f1 = frag(0, 0, 1);
                    int step = 0;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int cq = 0; cq < CQ; ++cq) {
                            // the three taps (dy = 0..2) of this (dx, quad): fetched ONCE here -- a W2-tail fragment read inside
                            // the row loop would be re-read from LDS for each of its four output rows
                            bf16x8 wf[3][MBW];
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                                for (int m = 0; m < MBW; ++m) wf[dy][m] = w2frag((dy * 3 + dx) * CQ + cq, m);
#pragma unroll
                            for (int r = 0; r < 6; ++r, ++step) {
                                int r2 = r + 2, cq2 = cq, dx2 = dx;        // the fragment two steps ahead
                                if (r2 >= 6) { r2 -= 6; if (++cq2 == CQ) { cq2 = 0; ++dx2; } }
                                bf16x8 f2 = f0;
                                if (dx2 < 3) f2 = frag(dx2, cq2, r2);
                                if (step % RSP == 0 && step / RSP < PER) {   // one DMA instruction of the next tile's x patch
                                    const int q = wave + NW * (step / RSP);
                                    if (has_next && q < G::NQ) dma_one(on, q, xbn);
                                }
#pragma unroll
                                for (int dy = 0; dy < 3; ++dy) {
                                    const int j = r - dy;
                                    if (j < 0 || j > 3) continue;
#pragma unroll
                                    for (int m = 0; m < MBW; ++m)
                                        acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy][m], f0, acc[j][m], 0, 0, 0);
                                }
                                f0 = f1; f1 = f2;
                                if constexpr (G::TIGHT) __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                } else {
                // fragments are fetched one k-step (12+ MFMAs, >= 192 cycles) ahead, except in the register-tight shape, where
                // the other two waves of the SIMD cover the LDS latency
                constexpr int D = G::TIGHT ? 0 : 1;
                bf16x8 fq[D > 0 ? D : 1][4];
                if constexpr (D > 0) {
#pragma unroll
                    for (int s = 0; s < D; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j) fq[s][j] = *(const bf16x8*)(base + boff[j] + k2(s));
                }
#pragma unroll
                for (int s = 0; s < KS2; ++s) {
                    bf16x8 f[4];
                    if constexpr (D > 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = fq[s % D][j];
                        if (s + D < KS2) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) fq[s % D][j] = *(const bf16x8*)(base + boff[j] + k2(s + D));
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = *(const bf16x8*)(base + boff[j] + k2(s));
                    }
                    if (s % SP == 0 && s / SP < PER) {       // one DMA instruction of the next tile's x patch
                        const int q = wave + NW * (st * PER + s / SP);
                        if (has_next && q < G::NQ) dma_one(on, q, xbn);
                    }
#pragma unroll
                    for (int m = 0; m < MBW; ++m) {
                        const bf16x8 am = w2frag(s, m);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, f[j], acc[j][m], 0, 0, 0);
                    }
                    // register-tight shape: stop the scheduler from hoisting later k-steps' LDS loads (fragments and W2 tail)
                    // above this point -- it would otherwise keep dozens of them live and spill
                    if constexpr (G::TIGHT) __builtin_amdgcn_sched_barrier(0);
                }
                }
                if constexpr (STAMP) asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[3][MBW - 1][3]));
                stamp(6);
                if constexpr (RESG) {
                    if (p.shortcut) {       // the shortcut values (and, vmcnt being in order, the DMA issued during the MFMA loop) have landed
                        static_assert(!RESG || MBW == 1, "shortcut registers of the RESG shape");
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xg[0][0]), "+v"(xg[1][0]), "+v"(xg[2][0]), "+v"(xg[3][0])::"memory");
                    }
                }
                f32x4 b2v[MBW];
#pragma unroll
                for (int m = 0; m < MBW; ++m) b2v[m] = *(const f32x4*)(s_b + G::C + cbase + m * 16);
                const char* xc = s_x + ((ty + 1) * PW + l15 + 1) * PXB + cbase * 2;   // shortcut: centre pixel of the x patch
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = y0 + ty + j / G::NBR, x = x0 + 16 * (j % G::NBR) + l15;
                    const bool ok = y < H && x < W;
                    char* o = p.out + ((long long)(b * H + y) * W + x) * p.out_ld_b + cbase * 2;
                    uint2 xr[MBW];
                    if (p.shortcut) {
                        if constexpr (RESG) {
#pragma unroll
                            for (int m = 0; m < MBW; ++m) xr[m] = xg[j][m];
                        } else {
                            // Read by asm, wait included: for a VISIBLE read of the x patch the compiler cannot tell the buffer being
                            // read from the one the next tile's LDS-DMA is filling and waits with vmcnt(0) -- which, vmcnt being in
                            // order, also waits for the output store of the previous block: the four stores of a step were serialised.
                            const uint32_t xa = (uint32_t)(uintptr_t)(xc + boff[j]);
                            if constexpr (MBW == 1)
                                asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=&v"(xr[0]) : "v"(xa) : "memory");
                            else if constexpr (MBW == 2)
                                asm volatile("ds_read_b64 %0, %2\nds_read_b64 %1, %2 offset:32\ns_waitcnt lgkmcnt(0)"
                                             : "=&v"(xr[0]), "=&v"(xr[1]) : "v"(xa) : "memory");
                            else if constexpr (MBW == 3)
                                asm volatile("ds_read_b64 %0, %3\nds_read_b64 %1, %3 offset:32\nds_read_b64 %2, %3 offset:64\ns_waitcnt lgkmcnt(0)"
                                             : "=&v"(xr[0]), "=&v"(xr[1]), "=&v"(xr[2]) : "v"(xa) : "memory");
                            else {
                                static_assert(MBW == 4, "shortcut read");
                                asm volatile("ds_read_b64 %0, %4\nds_read_b64 %1, %4 offset:32\nds_read_b64 %2, %4 offset:64\n"
                                             "ds_read_b64 %3, %4 offset:96\ns_waitcnt lgkmcnt(0)"
                                             : "=&v"(xr[0]), "=&v"(xr[1]), "=&v"(xr[2]), "=&v"(xr[3]) : "v"(xa) : "memory");
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MBW; ++m) {
                        f32x4 v = silu4(acc[j][m] + b2v[m]);
                        if (p.shortcut) {
                            v[0] += __uint_as_float(xr[m].x << 16);
                            v[1] += __uint_as_float(xr[m].x & 0xffff0000u);
                            v[2] += __uint_as_float(xr[m].y << 16);
                            v[3] += __uint_as_float(xr[m].y & 0xffff0000u);
                        }
                        if (ok) *(uint2*)(o + m * 32) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    }
                }
                stamp(7);
            }
        }
    }
    if constexpr (STAMP) {
        if (lane == 0 && p.debug)
            for (int i = 0; i < 8; ++i) p.debug[((long long)blockIdx.x * NW + wave) * 8 + i] = ph_sum[i];
    }
}

int g_btl_cus = 0;

struct BtlShape { int mbw, msplit, nw, tw, ktail; bool w1reg, resg; };

// Tile shapes of the HIP-source kernel (since round 4 the fallback for C = 48 / 96: images narrower than two tiles, AQ_BTL_ASM=0 -- the
// generated-assembly kernels further down take yolov5m's launches).  C = 48 and 96 run "one M block per wave": 12 waves (3 per SIMD, <= 168
// registers each, weights in plain VGPRs), every wave computes 16 output channels for the 4 rows x 16 pixels of its pixel group.  AQ_BTL_WIDE=1 selects
// the earlier 4-wave shapes (3 M blocks per wave, ~500 registers, one wave per SIMD) for A/B runs; the packed weight image
// depends on the shape, so the choice is made once per process.
bool btl_wide() {
    static const bool wide = [] { const char* e = getenv("AQ_BTL_WIDE"); return e && atoi(e) != 0; }();
    return wide;
}

bool btl_shape(int C, BtlShape* s) {
    switch (C) {
        case 16: *s = {1, 1, 4, 32, 0, false, false}; return true;
        case 32: *s = {2, 1, 4, 16, 0, false, false}; return true;
        case 48: *s = btl_wide() ? BtlShape{3, 1, 4, 16, 0, false, false} : BtlShape{1, 3, 12, 16, 0, true, false}; return true;
        case 64: *s = {4, 1, 4, 16, 3, false, false}; return true;
        case 96: *s = btl_wide() ? BtlShape{3, 2, 4, 16, 3, false, false} : BtlShape{1, 6, 12, 16, 12, true, true}; return true;
    }
    return false;
}

template <int MBW, int MSPLIT, int NW, int TW, int KT, bool W1REG, bool RESG>
int launch_btl(BtlParams p, hipStream_t stream) {
    using G = BtlGeom<MBW, MSPLIT, NW, TW, KT, W1REG, RESG>;
    static int occ = 0;                                      // resident workgroups per CU (registers + LDS)
    auto fn = bottleneck_kernel<MBW, MSPLIT, NW, TW, KT, W1REG, RESG>;
    constexpr size_t lds = G::LDS;
    p.tiles_x = (p.W + TW - 1) / TW; p.tiles_y = (p.H + G::TH - 1) / G::TH;
    AQ_REQUIRE((long long)p.B * p.tiles_x * p.tiles_y < (1LL << 30), "bottleneck: batch too large");
    p.n_tiles = p.B * p.tiles_x * p.tiles_y;
    if constexpr (MBW * MSPLIT == 3 || MBW * MSPLIT == 6) {  // stamped diagnostic builds exist for C = 48 and C = 96
        size_t sbytes = 0;
        unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
        if (sbuf && (size_t)g_btl_cus * NW * 64 <= sbytes) {
            auto sfn = bottleneck_kernel<MBW, MSPLIT, NW, TW, KT, W1REG, RESG, true>;
            AQ_CHECK_HIP(hipFuncSetAttribute((const void*)sfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
            p.debug = sbuf;
            const long long sgrid = g_btl_cus < p.n_tiles ? g_btl_cus : p.n_tiles;
            hipLaunchKernelGGL(sfn, dim3((unsigned)sgrid), dim3(G::THREADS), G::LDS, stream, p);
            AQ_CHECK_HIP(hipGetLastError());
            return AQ_OK;
        }
    }
    if (!occ) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int o = 0;
        AQ_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)fn, G::THREADS, lds));
        occ = o > 0 ? o : 1;                                 // a persistent grid must be fully resident
    }
    long long grid = (long long)g_btl_cus * occ;
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(G::THREADS), lds, stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

// ---- C = 48: the hand-scheduled assembly build (gen_bottleneck_asm.py; round 4) -- 8 waves, every wave owns all three M blocks of its
// two output rows, the two waves of a SIMD run half a tile out of step.  Code object embedded at build time. ----
struct BtlAsmArgs {                // must match ARG in gen_bottleneck_asm.py
    const char* in; char* out; const char* w; const float* bias;
    int in_ld_b, out_ld_b, B, H, W, tiles_x, tpi, ntiles, shortcut, G;
    unsigned magic_tpi, magic_tx, in_bytes, pad;
    unsigned long long* debug;
};
static_assert(sizeof(BtlAsmArgs) == 96, "kernel argument block");
const unsigned char kBtlAsmCode[] = {
#include "bottleneck_asm_hsaco.inc"
};
hipModule_t g_btl_asm_mod[64];
hipFunction_t g_btl_asm_fn[64][2];           // plain, stamped
constexpr size_t kBtlAsmWBytes = (size_t)(6 + 42) * 1024;     // six A fragments of the 1x1, 42 of the 3x3

int btl_asm_load(int dev) {
    if (g_btl_asm_mod[dev]) return AQ_OK;
    hipModule_t mod = nullptr;
    AQ_CHECK_HIP(hipModuleLoadData(&mod, kBtlAsmCode));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_btl_asm_fn[dev][0], mod, "bottleneck_asm_c48"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_btl_asm_fn[dev][1], mod, "bottleneck_asm_c48_stamped"));
    g_btl_asm_mod[dev] = mod;
    return AQ_OK;
}

// AQ_BTL_ASM=0: the HIP-source kernel everywhere (A/B and fallback).
bool btl_asm_enabled() {
    static const bool on = [] { const char* e = getenv("AQ_BTL_ASM"); return !(e && *e == '0'); }();
    return on;
}

// Does the assembly kernel take this launch?  (16 x 16 tiles; 32-bit buffer offsets; magic-number tile decode needs >= 2 tiles per row and image.)
bool btl_asm_fits(int C, int B, int H, int W, int in_ld, int out_ld) {
    if (C != 48 || !btl_asm_enabled() || btl_wide()) return false;
    const long long tx = (W + 15) / 16, ty = (H + 15) / 16, tpi = tx * ty, nt = tpi * B;
    if (tx < 2 || tpi < 2 || nt >= (1LL << 24) || nt * tpi >= (1LL << 32)) return false;
    if ((long long)B * H * W * in_ld * 2 >= (1LL << 30) || (long long)B * H * W * out_ld * 2 >= (1LL << 31)) return false;
    return true;
}

int launch_btl_asm(const BtlParams& p, hipStream_t stream) {
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "bottleneck: device ordinal %d", dev);
    { const int rc = btl_asm_load(dev); if (rc) return rc; }
    BtlAsmArgs a{};
    a.in = p.in; a.out = p.out; a.bias = p.bias;
    a.w = p.w;                                                 // (the caller passes the assembly image: it follows the HIP kernel's in the packed buffer)
    a.in_ld_b = p.in_ld_b; a.out_ld_b = p.out_ld_b; a.B = p.B; a.H = p.H; a.W = p.W;
    a.tiles_x = (p.W + 15) / 16;
    a.tpi = a.tiles_x * ((p.H + 15) / 16);
    a.ntiles = a.tpi * p.B;
    a.shortcut = p.shortcut;
    long long grid = g_btl_cus;
    if (grid > a.ntiles) grid = a.ntiles;
    a.G = (int)grid;
    a.magic_tpi = (unsigned)((1ULL << 32) / (unsigned)a.tpi + 1);
    a.magic_tx = (unsigned)((1ULL << 32) / (unsigned)a.tiles_x + 1);
    // bytes of the input slice from its first channel to the end of its last pixel: lanes beyond it (and "negative" offsets) read zeros
    a.in_bytes = (unsigned)(((long long)p.B * p.H * p.W - 1) * p.in_ld_b + 96);
    int which = 0;
    size_t sbytes = 0;
    unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
    if (sbuf && (size_t)grid * 8 * 64 <= sbytes) { a.debug = sbuf; which = 1; }
    hipFunction_t fn = g_btl_asm_fn[dev][which];
    const char* exp_kernel = getenv("AQ_BTL_ASM_KERNEL");      // timing experiments: another kernel of the code object, by name (tools/time_bottleneck.py)
    if (exp_kernel && *exp_kernel) {
        char name[96];
        snprintf(name, sizeof name, "%s%s", exp_kernel, which ? "_stamped" : "");
        AQ_CHECK_HIP(hipModuleGetFunction(&fn, g_btl_asm_mod[dev], name));
    }
    size_t asz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    AQ_CHECK_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 512, 1, 1, 0, stream, nullptr, extra));
    return AQ_OK;
}

// A-fragment image of the assembly kernel (C = 48): six fragments of the 1x1 [k-step 2][M block 3], then 42 of the 3x3
// [k-step 14][M block 3], 64 lanes x 8 bf16 each.  Lane (r = lane & 15, g = lane >> 4) of M block m holds output channel 16 m + r.
// 1x1, k-step s: input channels 8 (4 s + g) .. + 7 (zero past 48).  3x3: k-steps 0-8 = channels 8 g .. + 7 of tap s; k-steps 9-13 =
// channels 32 + 8 (g & 1) .. + 7 of tap pa[g >> 1] for the pairs (0,1), (3,4), (6,7), (2,5), (8,-)  (gen_bottleneck_asm.py KSTEPS).
void btl_asm_pack(const float* w1, const float* w2, bf16_t* dst) {
    const int C = 48;
    static const int pairs[5][2] = {{0, 1}, {3, 4}, {6, 7}, {2, 5}, {8, -1}};
    for (int s = 0; s < 2; ++s)
        for (int m = 0; m < 3; ++m)
            for (int lane = 0; lane < 64; ++lane, dst += 8) {
                const int co = 16 * m + (lane & 15), blk = 4 * s + (lane >> 4);
                for (int e = 0; e < 8; ++e) dst[e] = blk < 6 ? aq_f2bf(w1[(size_t)co * C + blk * 8 + e]) : aq_f2bf(0.0f);
            }
    for (int s = 0; s < 14; ++s)
        for (int m = 0; m < 3; ++m)
            for (int lane = 0; lane < 64; ++lane, dst += 8) {
                const int co = 16 * m + (lane & 15), g = lane >> 4;
                int tap, c0;
                if (s < 9) { tap = s; c0 = 8 * g; }
                else { tap = pairs[s - 9][g >> 1]; c0 = 32 + 8 * (g & 1); }
                for (int e = 0; e < 8; ++e) dst[e] = tap >= 0 ? aq_f2bf(w2[((size_t)co * 9 + tap) * C + c0 + e]) : aq_f2bf(0.0f);
            }
}

// ---- C = 96: the assembly build of round 4 (gen_bottleneck96_asm.py) -- 4 waves, one per SIMD; wave (h, q) owns three M blocks (48 output
// channels) of one pixel half and keeps its 81 + 9 A fragments in registers (64 of them in AGPRs); 8 x 16-pixel tiles. ----
const unsigned char kBtl96AsmCode[] = {
#include "bottleneck96_asm_hsaco.inc"
};
hipModule_t g_btl96_mod[64];
hipFunction_t g_btl96_fn[64][2];             // plain, stamped
constexpr size_t kBtl96AsmWBytes = (size_t)2 * (9 + 81) * 1024;     // per channel half: nine A fragments of the 1x1, 81 of the 3x3

int btl96_load(int dev) {
    if (g_btl96_mod[dev]) return AQ_OK;
    hipModule_t mod = nullptr;
    AQ_CHECK_HIP(hipModuleLoadData(&mod, kBtl96AsmCode));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_btl96_fn[dev][0], mod, "bottleneck_asm_c96"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_btl96_fn[dev][1], mod, "bottleneck_asm_c96_stamped"));
    g_btl96_mod[dev] = mod;
    return AQ_OK;
}

// Does the C = 96 assembly kernel take this launch?  (8 x 16 tiles; 32-bit buffer offsets; magic-number tile decode needs >= 2 tiles per row and image.)
bool btl96_asm_fits(int C, int B, int H, int W, int in_ld, int out_ld) {
    static const bool off = [] { const char* e = getenv("AQ_BTL96_ASM"); return e && *e == '0'; }();
    if (C != 96 || off || !btl_asm_enabled() || btl_wide()) return false;
    const long long tx = (W + 15) / 16, ty = (H + 7) / 8, tpi = tx * ty, nt = tpi * B;
    if (tx < 2 || tpi < 2 || nt >= (1LL << 24) || nt * tpi >= (1LL << 32)) return false;
    if ((long long)B * H * W * in_ld * 2 >= (1LL << 30) || (long long)B * H * W * out_ld * 2 >= (1LL << 31)) return false;
    return true;
}

int launch_btl96_asm(const BtlParams& p, hipStream_t stream) {
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "bottleneck: device ordinal %d", dev);
    { const int rc = btl96_load(dev); if (rc) return rc; }
    BtlAsmArgs a{};
    a.in = p.in; a.out = p.out; a.bias = p.bias;
    a.w = p.w;                                                 // (the caller passes the assembly image: it follows the HIP kernel's in the packed buffer)
    a.in_ld_b = p.in_ld_b; a.out_ld_b = p.out_ld_b; a.B = p.B; a.H = p.H; a.W = p.W;
    a.tiles_x = (p.W + 15) / 16;
    a.tpi = a.tiles_x * ((p.H + 7) / 8);
    a.ntiles = a.tpi * p.B;
    a.shortcut = p.shortcut;
    long long grid = g_btl_cus;
    if (grid > a.ntiles) grid = a.ntiles;
    a.G = (int)grid;
    a.magic_tpi = (unsigned)((1ULL << 32) / (unsigned)a.tpi + 1);
    a.magic_tx = (unsigned)((1ULL << 32) / (unsigned)a.tiles_x + 1);
    // bytes of the slices from their first channel to the end of their last pixel: loads beyond read zeros, stores beyond are dropped
    a.in_bytes = (unsigned)(((long long)p.B * p.H * p.W - 1) * p.in_ld_b + 192);
    a.pad = (unsigned)(((long long)p.B * p.H * p.W - 1) * p.out_ld_b + 192);          // (out_bytes in this kernel's argument block)
    int which = 0;
    size_t sbytes = 0;
    unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
    if (sbuf && (size_t)grid * 4 * 64 <= sbytes) { a.debug = sbuf; which = 1; }
    hipFunction_t fn = g_btl96_fn[dev][which];
    const char* exp_kernel = getenv("AQ_BTL96_ASM_KERNEL");    // timing experiments: another kernel of the code object, by name (tools/time_bottleneck.py)
    if (exp_kernel && *exp_kernel) {
        char name[96];
        snprintf(name, sizeof name, "%s%s", exp_kernel, which ? "_stamped" : "");
        AQ_CHECK_HIP(hipModuleGetFunction(&fn, g_btl96_mod[dev], name));
    }
    size_t asz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    AQ_CHECK_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, stream, nullptr, extra));
    return AQ_OK;
}

// A-fragment image of the C = 96 assembly kernel: per output-channel half h, nine fragments of the 1x1 [k-step 3][M block 3], then 81 of the 3x3
// [tap 9][k-step 3][M block 3], 64 lanes x 8 bf16 each.  Lane (r = lane & 15, kg = lane >> 4) of M block m holds input channels
// 32 kstep + 8 kg .. + 7 of output channel 48 h + (m < 2 ? 8 (r >> 2) + 4 m + (r & 3) : 32 + 4 (r >> 2) + (r & 3)) -- the rows are permuted so
// that a lane's twelve results of a pixel are 8 + 4 consecutive channels (16- and 8-byte stores / LDS writes).
void btl96_asm_pack(const float* w1, const float* w2, bf16_t* dst) {
    const int C = 96;
    for (int h = 0; h < 2; ++h) {
        auto chan = [&](int m, int r) { return 48 * h + (m < 2 ? 8 * (r >> 2) + 4 * m + (r & 3) : 32 + 4 * (r >> 2) + (r & 3)); };
        for (int s = 0; s < 3; ++s)
            for (int m = 0; m < 3; ++m)
                for (int lane = 0; lane < 64; ++lane, dst += 8) {
                    const int co = chan(m, lane & 15), c0 = 32 * s + 8 * (lane >> 4);
                    for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(w1[(size_t)co * C + c0 + e]);
                }
        for (int tap = 0; tap < 9; ++tap)
            for (int s = 0; s < 3; ++s)
                for (int m = 0; m < 3; ++m)
                    for (int lane = 0; lane < 64; ++lane, dst += 8) {
                        const int co = chan(m, lane & 15), c0 = 32 * s + 8 * (lane >> 4);
                        for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(w2[((size_t)co * 9 + tap) * C + c0 + e]);
                    }
    }
}

}  // namespace

// Packs the fused fp32 weights of one Bottleneck -- w1 KRSC (C,1,1,C), w2 KRSC (C,3,3,C) -- into the A-fragment image the
// kernel loads once per workgroup: [channel group][k-step][M block][lane] x 8 bf16.  Lane (m = lane & 15, g = lane >> 4) of
// M block mb in channel group mh holds output channel 16 * (mh * MBW + mb) + m and K block 4 * kstep + g (8 consecutive
// input channels of one tap, tap-major); first the 1x1's k-steps, then the 3x3's.  K blocks past the end are zero.
extern "C" int aq_pack_bottleneck_weights(const float* w1_host, const float* w2_host, int C, void* packed_dev, size_t* bytes, void* stream) {
    BtlShape sh;
    AQ_REQUIRE(w1_host && w2_host && bytes && btl_shape(C, &sh), "pack_bottleneck: C must be 16, 32, 48, 64 or 96 (got %d)", C);
    const int cb = C / 8, ks1 = (cb + 3) / 4, ks2 = (9 * cb + 3) / 4;
    const size_t frags = (size_t)sh.msplit * (ks1 + ks2) * sh.mbw;
    const size_t hip_bytes = frags * 64 * 16;
    *bytes = hip_bytes + (C == 48 ? kBtlAsmWBytes : C == 96 ? kBtl96AsmWBytes : 0);      // C = 48 / 96: the assembly kernel's image follows the HIP kernel's
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_bottleneck: out of host memory");
    for (int mh = 0; mh < sh.msplit; ++mh)
        for (int s = 0; s < ks1 + ks2; ++s)
            for (int m = 0; m < sh.mbw; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = (mh * sh.mbw + m) * 16 + (lane & 15), g = lane >> 4;
                    bf16_t* dst = host + ((((size_t)mh * (ks1 + ks2) + s) * sh.mbw + m) * 64 + lane) * 8;
                    if (s < ks1) {
                        const int blk = 4 * s + g;
                        if (blk < cb)
                            for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(w1_host[(size_t)co * C + blk * 8 + e]);
                    } else {
                        const int blk = 4 * (s - ks1) + g;
                        if (blk < 9 * cb) {
                            const int tap = blk / cb, c8 = blk % cb;
                            for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(w2_host[((size_t)co * 9 + tap) * C + c8 * 8 + e]);
                        }
                    }
                }
    if (C == 48) btl_asm_pack(w1_host, w2_host, (bf16_t*)((char*)host + hip_bytes));
    if (C == 96) btl96_asm_pack(w1_host, w2_host, (bf16_t*)((char*)host + hip_bytes));
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

// in/out: bf16 NHWC [B][H][W][ld] with the C channels at ch_off; out must not overlap in.  bias_dev: [2C] fp32 (b1 | b2).
extern "C" int aq_bottleneck(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int C,
                             const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int shortcut, void* stream) {
    BtlShape sh;
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "bottleneck: null pointer");
    AQ_REQUIRE(btl_shape(C, &sh), "bottleneck: C must be 16, 32, 48, 64 or 96 (got %d)", C);
    AQ_REQUIRE(B > 0 && H > 0 && W > 0, "bottleneck: empty input");
    AQ_REQUIRE(in_ld % 8 == 0 && out_ld % 8 == 0 && in_choff % 8 == 0 && out_choff % 8 == 0 && in_choff + C <= in_ld && out_choff + C <= out_ld,
               "bottleneck: channel slices must be 8-aligned and inside their rows");
    AQ_REQUIRE((long long)B * H * W < (1LL << 31), "bottleneck: batch too large");
    BtlParams p{};
    p.in = (const char*)in_dev + (size_t)in_choff * 2; p.in_ld_b = in_ld * 2;
    p.out = (char*)out_dev + (size_t)out_choff * 2; p.out_ld_b = out_ld * 2;
    {   // the kernel reads halos of pixels other workgroups write: refuse aliasing buffers
        const char* i0 = (const char*)in_dev; const char* i1 = i0 + (size_t)B * H * W * in_ld * 2;
        const char* o0 = (const char*)out_dev; const char* o1 = o0 + (size_t)B * H * W * out_ld * 2;
        AQ_REQUIRE(o1 <= i0 || i1 <= o0, "bottleneck: output overlaps input");
    }
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.B = B; p.H = H; p.W = W; p.shortcut = shortcut;
    p.zero = aq_zero_page();
    AQ_REQUIRE(p.zero, "bottleneck: zero page allocation failed");
    if (g_btl_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_btl_cus = cus;
    }
    const hipStream_t st = (hipStream_t)stream;
    if (btl_asm_fits(C, B, H, W, in_ld, out_ld)) {
        BtlShape sh48;
        btl_shape(48, &sh48);
        p.w += (size_t)sh48.msplit * ((48 / 8 + 3) / 4 + (9 * 48 / 8 + 3) / 4) * sh48.mbw * 64 * 16;      // skip the HIP kernel's image
        return launch_btl_asm(p, st);
    }
    if (btl96_asm_fits(C, B, H, W, in_ld, out_ld)) {
        const int cb = 96 / 8, ks1 = (cb + 3) / 4, ks2 = (9 * cb + 3) / 4;
        p.w += (size_t)sh.msplit * (ks1 + ks2) * sh.mbw * 64 * 16;                                       // skip the HIP kernel's image
        return launch_btl96_asm(p, st);
    }
    switch (C) {
        case 16: return launch_btl<1, 1, 4, 32, 0, false, false>(p, st);
        case 32: return launch_btl<2, 1, 4, 16, 0, false, false>(p, st);
        case 48: return btl_wide() ? launch_btl<3, 1, 4, 16, 0, false, false>(p, st) : launch_btl<1, 3, 12, 16, 0, true, false>(p, st);
        case 64: return launch_btl<4, 1, 4, 16, 3, false, false>(p, st);
        default: return btl_wide() ? launch_btl<3, 2, 4, 16, 3, false, false>(p, st) : launch_btl<1, 6, 12, 16, 12, true, true>(p, st);
    }
}
