// Fused down-sampling block for gfx950 (bf16):
//     y = SiLU(Wb (1x1) . SiLU(Wa (3x3, stride 2, pad 1) (*) x + ba) + bb),      48 -> 96 -> 96 channels
// = yolov5m's model.1 (Conv(48, 96, 3, 2)) followed by model.2.cv1|cv2 (the two 1x1 convs of the first C3, stacked along Cout;
// both read model.1's output and nothing else does) [UPSTREAM models/common.py Conv.forward_fuse, C3.forward; reached through
// reference README.md:77 -> yolov5/detect.py].  As two launches these were the most expensive pair of the network (0.34 + 0.15 ms
// per 64-tile batch): the 3x3 gathers nine shifted copies of a 629 MB input through L2, writes 315 MB, and the 1x1 reads them back.
//
// Same skeleton as csrc/bottleneck.hip: a persistent workgroup of 12 waves (3 per SIMD) owns an 8 x 16 output tile;
//   A. the 17 x 33-pixel input patch (zeros from a zero page outside the image) arrives by LDS-DMA into one half of a double
//      buffer -- the DMA instructions of the NEXT tile are issued from inside this tile's MFMA loop;
//   1. wave (M block mb, pixel group pg) computes 16 of the 96 intermediate channels for its 4 rows x 16 pixels: MFMA B fragments
//      are read straight from the patch (pixel stride 7 x 16 B: conflict-free for the stride-2 pixel walk of ds_read_b128's lane
//      groups), weights (14 k-steps) live in registers; t = SiLU(. + ba) goes to LDS as bf16 -- the rounding the two-launch
//      form applies when it stores model.1's output;
//   2. the 1x1 (3 k-steps, weights in registers) reads its fragments from the t tile and stores y.
// Per wave 68 weight registers; LDS 2 x 62 KB (x) + 28 KB (t).
//
// The same kernel, without the 1x1 stage, also runs a plain 3x3/s2 96 -> 192 (yolov5m's model.3): 12 M blocks x 1 pixel group,
// 4 x 16 tiles, k-steps that never straddle taps -- so each (input row, dx, channel quad) fragment is loaded once and used for
// both output rows it serves -- and the last 3 of the 27 weight k-steps in LDS.  The engine's autotuner times that form against
// the implicit-GEMM tile shapes under the config id AQ_CONV_CFG_DIRECT3X3S2.
#include "conv_device.h"

using namespace aqdev;

namespace {

struct DownParams {
    const char* in;
    char* out;
    const char* w;           // A-fragment image, see aq_pack_downblock_weights
    const float* bias;       // [96 + 96]: ba | bb
    const char* zero;        // >= 16 zero bytes
    int in_ld_b, out_ld_b;
    int B, H, W, Ho, Wo;     // input / output spatial size (Ho = H / 2)
    int tiles_x, tiles_y, n_tiles;
    int act;                 // plain 3x3/s2 form only (the fused form always applies SiLU twice)
    // stem-fused form (STEM): the input patch is COMPUTED from the uint8 tile instead of loaded
    const uint8_t* tiles;    // [B][Hi][Wi][3]
    const char* stem_w;      // aq_pack_stem_weights image (bf16): A fragments [k-step 0..4][M block 0..2][lane]
    const float* stem_b;     // [64]: stem bias, zero padded
    int Hi, Wi;              // tile size in pixels (H = Hi / 2, W = Wi / 2 are the stem's output size)
};

constexpr int kNW = 12;                                  // waves per workgroup (3 per SIMD)
constexpr int kTW = 16;                                  // output tile width; every wave owns 4 rows x 16 pixels

// CIN -> CMID by the 3x3/s2; FUSE: followed by a 1x1 CMID -> CMID; KT: last k-steps of the 3x3 weights kept in LDS.
//   <48, 96, true, 0>:   yolov5m model.1 + model.2.cv1|cv2   (6 M blocks x 2 pixel groups, 8 x 16 tile)
//   <96, 192, false, 3>: yolov5m model.3                     (12 M blocks x 1 pixel group, 4 x 16 tile)
// STEM (with <48, 96, true, 0> only): model.0 in front -- uint8 tile -> / 255 -> Conv(3, 48, 6, 2, 2) + SiLU is evaluated for every pixel of the
// 17 x 33 patch (the halo is recomputed, 561 / 512 = 1.10 x), so the 629 MB stem output of a 64-tile batch is never written or read.
template <int CIN, int CMID, bool FUSE, int KT, bool STEM = false> struct DownGeom {
    static constexpr int MB = CMID / 16;                     // M blocks = wave groups along M
    static constexpr int PG = kNW / MB;                      // pixel groups
    static constexpr int TH = 4 * PG;
    static constexpr int PH = 2 * TH + 1, PW = 2 * kTW + 1;  // input patch (stride 2, pad 1)
    static constexpr int PP = PH * PW;
    static constexpr int CBI = CIN / 8;                      // 16-byte channel blocks of an input pixel
    static constexpr int SPPX = CBI | 1;                     // x patch slots per pixel: odd => conflict-free stride-2 fragment reads
    static constexpr int PXB = SPPX * 16;
    static constexpr int NQ = (PP * SPPX + 63) / 64;         // LDS-DMA wave instructions per patch
    static constexpr int XPB = NQ * 1024;
    static constexpr int NBLKA = 9 * CBI;
    static constexpr int KSA = (NBLKA + 3) / 4;              // k-steps (32 K each) of the 3x3; K blocks are tap-major
    static constexpr bool UNIFORM_K = CBI % 4 == 0;          // a k-step never straddles taps
    static constexpr int CQ = CBI / 4;                       // channel quads per tap (UNIFORM_K)
    static constexpr int KREG = KSA - KT;
    static constexpr int WTB = MB * KT * 1024;               // W tail in LDS
    static constexpr int KSB = FUSE ? CMID / 32 : 0;         // k-steps of the 1x1
    static constexpr int SPPT = CMID / 8 + 2, TPXB = SPPT * 16;   // t tile (FUSE): channel slots + 2 pad (conflict-free stride-1 reads)
    static constexpr int TPB = FUSE ? TH * kTW * TPXB : 0;
    static constexpr int NBIAS = FUSE ? 2 * CMID : CMID;
    // STEM: raw patch = image rows 2 iy0 - 2 .. + 2 PH + 1 (6 x 6 / stride 2 / pad 2 window of PH stem rows), bytes 6 ix0 - 6 .. of each
    // (RGB interleaved; a pixel's 18 values of one ky are consecutive, the K padding to 24 reads on into its neighbour against zero weights)
    static constexpr int RAWROWS = 2 * PH + 4;
    static constexpr int RAWDW = (6 * (PW - 1) + 24) / 4;    // dwords (= 4 bytes = 4 patch elements) per raw row
    static constexpr int RAWPROWB = RAWDW * 8 + 16;          // bf16 patch row stride
    static constexpr int RAWB = STEM ? RAWROWS * RAWPROWB : 0;
    static constexpr int NRAW = RAWROWS * RAWDW;
    static constexpr int NITR = (NRAW + kNW * 64 - 1) / (kNW * 64);
    static constexpr int STAGEB = STEM ? NITR * kNW * 64 * 4 : 0;
    static constexpr int WSTEMB = STEM ? 5 * 3 * 1024 : 0;
    static constexpr int NBIAS_ALL = NBIAS + (STEM ? 64 : 0);
    static constexpr int LDS = (STEM ? 1 : 2) * XPB + RAWB + STAGEB + TPB + WTB + WSTEMB + NBIAS_ALL * 4;
    static_assert(!STEM || (CIN == 48 && FUSE && RAWB % 16 == 0), "the stem-fused form is yolov5m's first block");
    static constexpr int WFRAGS = KSA + KSB;                 // A fragments per M block
    static_assert(kNW % MB == 0 && LDS <= 160 * 1024, "shape");
    static_assert(KT == 0 || UNIFORM_K, "the LDS tail is only wired into the uniform-K loop");
};


__device__ __forceinline__ f32x4 down_silu4(f32x4 v) {   // same sequence as the shared conv epilogue (bf16 mode)
    const f32x4 t = v * -1.44269504f;
    f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    d = d + 1.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    return v * r;
}
// LDS write / barrier that do not make the compiler drain the in-flight LDS-DMA (see csrc/bottleneck.hip)
__device__ __forceinline__ void down_lds_write_b64(char* dst, uint2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)dst), "v"(v));
}
__device__ __forceinline__ void down_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int CIN, int CMID, bool FUSE, int KT, bool STEM = false>
__global__ __launch_bounds__(kNW * 64) void downblock_kernel(const DownParams p) {
    using G = DownGeom<CIN, CMID, FUSE, KT, STEM>;
    constexpr int PW = G::PW, PXB = G::PXB, KSA = G::KSA, KSB = G::KSB, KREG = G::KREG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_raw = smem + (STEM ? 1 : 2) * G::XPB;            // STEM: the x patch has ONE buffer (it is computed, not loaded ahead)
    char* s_stage = s_raw + G::RAWB;
    char* s_t = s_stage + G::STAGEB;
    char* s_wt = s_t + G::TPB;
    char* s_ws = s_wt + G::WTB;
    float* s_b = (float*)(s_ws + G::WSTEMB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave % G::MB, pg = wave / G::MB;
    const int g = lane >> 4, l15 = lane & 15;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
    const int cbase = mb * 16 + g * 4;                       // this lane's 4 channels (of both the intermediate and the output)

    // ---- once per workgroup: this wave's rows of the weight sets to registers (3x3 tail: LDS), biases to LDS ----
    bf16x8 wa[KREG], wb[KSB > 0 ? KSB : 1];
    {
        const bf16x8* wsrc = (const bf16x8*)p.w + (size_t)mb * G::WFRAGS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KREG; ++s) wa[s] = wsrc[s * 64];
        if (pg == 0) {
#pragma unroll
            for (int s = 0; s < KT; ++s) *(bf16x8*)(s_wt + (mb * KT + s) * 1024 + lane * 16) = wsrc[(KREG + s) * 64];
        }
#pragma unroll
        for (int s = 0; s < KSB; ++s) wb[s] = wsrc[(KSA + s) * 64];
    }
    for (int i = tid; i < G::NBIAS; i += kNW * 64) s_b[i] = p.bias[i];
    if constexpr (STEM) {
        for (int i = tid; i < 64; i += kNW * 64) s_b[G::NBIAS + i] = p.stem_b[i];
        for (int i = tid; i < G::WSTEMB / 16; i += kNW * 64) ((bf16x8*)s_ws)[i] = ((const bf16x8*)p.stem_w)[i];
    }
    // per-lane byte offset of this lane's K block relative to the tap-(0,0) pixel of an output pixel's 3x3 window
    int koffa[G::UNIFORM_K ? 1 : KSA];
    if constexpr (G::UNIFORM_K) koffa[0] = g * 16;
    else {
#pragma unroll
        for (int s = 0; s < KSA; ++s) {
            int blk = 4 * s + g;
            if (blk >= G::NBLKA) blk = 0;                    // zero weights: any initialised address
            const int tap = blk / G::CBI, cb = blk - tap * G::CBI;
            const int dy = tap / 3, dx = tap - 3 * dy;
            koffa[s] = (dy * PW + dx) * PXB + cb * 16;
        }
    }

    const int tiles_per_img = p.tiles_y * p.tiles_x;
    struct PatchOrg { const char* org; int iy0, ix0; };     // address + image coordinates of patch pixel (0, 0)
    auto patch_org = [&](int tile) -> PatchOrg {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int iy0 = 2 * ty0 * G::TH - 1, ix0 = 2 * tx0 * kTW - 1;
        return {p.in + ((long long)(b * H + iy0) * W + ix0) * p.in_ld_b, iy0, ix0};
    };
    auto dma_one = [&](const PatchOrg& o, int q, char* xb) {  // branch-free: issued from inside the MFMA loop
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                     // opaque: no per-call-site slot decode hoisted out of the tile loop
        const int slot = q * 64 + lane_o;
        const int px = slot / G::SPPX, part = slot - px * G::SPPX;
        const int pr = px / PW, pc = px - pr * PW;
        const int iy = o.iy0 + pr, ix = o.ix0 + pc;
        const bool valid = px < G::PP && part < G::CBI && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const char* src = o.org + (pr * W + pc) * p.in_ld_b + part * 16;
        glds16(valid ? src : p.zero, xb + q * 1024);
    };
    // STEM: the raw uint8 dwords of a tile's patch travel by LDS-DMA into a lane-linear staging area (slot i = tid + 768 it), are
    // converted to bf16 activations (v * (1 / 255), RNE: the stem kernel's arithmetic) by the lanes that loaded them, and stage 0 below
    // turns them into the x patch.
    auto dma_raw = [&](int tile) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int iy0 = 2 * ty0 * G::TH - 1, ix0 = 2 * tx0 * kTW - 1;
        const uint8_t* img = p.tiles + (size_t)b * p.Hi * p.Wi * 3;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                      // opaque: no per-slot row / column decode hoisted out of the tile loop
#pragma unroll
        for (int it = 0; it < G::NITR; ++it) {
            const int i = tid_o + it * (kNW * 64);
            const int r = i / G::RAWDW, d = i - r * G::RAWDW;
            const int iy = 2 * iy0 - 2 + r, byte0 = 6 * ix0 - 6 + 4 * d;
            const bool ok = tile < p.n_tiles && i < G::NRAW && (unsigned)iy < (unsigned)p.Hi && byte0 >= 0 && byte0 < p.Wi * 3;
            const char* src = ok ? (const char*)img + (size_t)iy * p.Wi * 3 + byte0 : p.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(s_stage + (it * (kNW * 64) + wave * 64) * 4), 4, 0, 0);
        }
    };
    auto convert_stage = [&]() {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                      // opaque: no per-slot row / column decode hoisted out of the tile loop
#pragma unroll
        for (int it = 0; it < G::NITR; ++it) {
            const int i = tid_o + it * (kNW * 64);
            const int r = i / G::RAWDW, d = i - r * G::RAWDW;
            if (i < G::NRAW) {
                const uint32_t v = *(const uint32_t*)(s_stage + i * 4);
                constexpr float k = 1.0f / 255.0f;
                const float f0 = (float)(v & 255u) * k, f1 = (float)((v >> 8) & 255u) * k, f2 = (float)((v >> 16) & 255u) * k,
                            f3 = (float)(v >> 24) * k;
                down_lds_write_b64(s_raw + r * G::RAWPROWB + d * 8, make_uint2(pack_bf16x2(f0, f1), pack_bf16x2(f2, f3)));
            }
        }
    };
    constexpr int NQW = (G::NQ + kNW - 1) / kNW;             // DMA instructions per wave and tile
    constexpr int NSTEP = G::UNIFORM_K ? 3 * G::CQ * 9 : KSA;    // MFMA-loop steps that can carry one each
    constexpr int SP = NSTEP / NQW;
    static_assert(SP >= 1 && NQW * SP <= NSTEP, "not enough MFMA-loop steps to carry the DMA issue");

    int tile = first_tile(gridDim.x, blockIdx.x);
    if constexpr (STEM) {
        dma_raw(tile);
    } else if (tile < p.n_tiles) {
        const PatchOrg o = patch_org(tile);
#pragma unroll 1
        for (int q = wave; q < G::NQ; q += kNW) dma_one(o, q, smem);
    }
    int cur = 0;
    bool prev_full = false;
    for (; tile < p.n_tiles; tile += gridDim.x, cur ^= 1) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int y0 = ty0 * G::TH, x0 = tx0 * kTW;
        const char* s_x = smem + (STEM ? 0 : cur) * G::XPB;
        // this tile's patch has landed (own DMA: vmcnt; the other waves': barrier).  vmcnt is in-order: after a full tile the
        // youngest DMA instruction is older than that tile's 4 output stores, which may stay in flight.
        if (prev_full) wait_vmcnt<4>(); else wait_vmcnt<0>();
        if constexpr (STEM) {
            // ---- 0. the x patch = SiLU(stem(raw patch)), zero outside the stem's output image (the 3x3 pads x, not the tile) ----
            convert_stage();                                 // own slots only: no barrier between the DMA and here
            down_lds_barrier();                              // raw patch complete; everyone is done with the previous tile's x and t
            dma_raw(tile + (int)gridDim.x);                  // the staging area is free again (zeros past the last tile)
            const int iy0 = 2 * y0 - 1, ix0 = 2 * x0 - 1;
            int g_o = g;
            asm volatile("" : "+v"(g_o));                    // opaque: the K offsets are rebuilt per tile instead of living in 5 registers
            int koffs[5];                                    // per-lane byte offset of K block 4 s + g inside the raw patch
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int blk = 4 * s + g_o;
                const int ky = blk / 3 < 6 ? blk / 3 : 5;    // K blocks 18, 19 carry zero weights: read any valid row (csrc/stem_conv.hip)
                koffs[s] = ky * G::RAWPROWB + (blk - 3 * (blk / 3)) * 16;
            }
            // a wave takes every 12th block of 16 patch pixels and ALL three M blocks of it: the B fragment (twenty 4-byte LDS reads --
            // a pixel's values start on a 12-byte grid) is read once, not once per M block; the stem's A fragments stay in LDS
            const char* wsl = s_ws + lane * 16;
#pragma unroll 1
            for (int blk = wave; blk < (G::PP + 15) / 16; blk += kNW) {
                const int L = blk * 16 + l15;
                const int Lc = L < G::PP ? L : G::PP - 1;
                const int pr = Lc / PW, pc = Lc - pr * PW;
                const char* base = s_raw + (2 * pr) * G::RAWPROWB + (6 * pc) * 2;
                bf16x8 bf[5];
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const uint32_t* src = (const uint32_t*)(base + koffs[s]);        // 4-byte aligned
                    const uint4 u = make_uint4(src[0], src[1], src[2], src[3]);
                    __builtin_memcpy(&bf[s], &u, 16);
                }
                f32x4 a0[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) a0[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 5; ++s)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        a0[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(wsl + (s * 3 + m) * 1024), bf[s], a0[m], 0, 0, 0);
                const bool inside = (unsigned)(iy0 + pr) < (unsigned)H && (unsigned)(ix0 + pc) < (unsigned)W;
#pragma unroll
                for (int m = 0; m < 3; ++m) {                // lane = (patch pixel L, stem channels 12 g + 4 m .. + 3)
                    f32x4 v = down_silu4(a0[m] + *(const f32x4*)(s_b + G::NBIAS + 12 * g + 4 * m));
                    if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (L < G::PP)
                        down_lds_write_b64((char*)s_x + Lc * PXB + (12 * g + 4 * m) * 2, make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
                }
            }
        }
        down_lds_barrier();
        const bool has_next = tile + (int)gridDim.x < p.n_tiles;
        prev_full = y0 + G::TH <= Ho && x0 + kTW <= Wo;
        char* xbn = smem + (cur ^ 1) * G::XPB;
        PatchOrg on = {nullptr, 0, 0};
        if (has_next) on = patch_org(tile + (int)gridDim.x);

        const int ty = 4 * pg;                               // first of this wave's 4 output rows inside the tile
        // ---- 1. SiLU(Wa (*) x + ba): 4 blocks of 16 pixels (4 rows), one M block ----
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (G::UNIFORM_K) {
            // k-steps do not straddle taps: fragment (input row ri, dx, channel quad) serves output row j with dy = ri - 2 j, i.e.
            // two rows when ri is even -- load it once (9 loads for 12 (row, dy) pairs), two fragments ahead of the MFMAs
            constexpr int CQ = G::CQ;
            const char* base = s_x + ((2 * ty) * PW + 2 * l15) * PXB + koffa[0];
            auto frag = [&](int dx, int cq, int ri) -> bf16x8 { return *(const bf16x8*)(base + (ri * PW + dx) * PXB + cq * 64); };
            auto wfrag = [&](int s) -> bf16x8 {
                if constexpr (KT > 0) {
                    if (s >= KREG) return *(const bf16x8*)(s_wt + (mb * KT + (s - KREG)) * 1024 + lane * 16);
                    return wa[s < KREG ? s : 0];
                } else return wa[s];
            };
            bf16x8 f0 = frag(0, 0, 0), f1 = frag(0, 0, 1);
            int step = 0;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int cq = 0; cq < CQ; ++cq) {
                    bf16x8 wf[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) wf[dy] = wfrag((dy * 3 + dx) * CQ + cq);
#pragma unroll
                    for (int ri = 0; ri < 9; ++ri, ++step) {
                        int r2 = ri + 2, cq2 = cq, dx2 = dx;    // the fragment two steps ahead
                        if (r2 >= 9) { r2 -= 9; if (++cq2 == CQ) { cq2 = 0; ++dx2; } }
                        bf16x8 f2 = f0;
                        if (dx2 < 3) f2 = frag(dx2, cq2, r2);
                        if (!STEM && step % SP == 0 && step / SP < NQW) {
                            const int q = wave + kNW * (step / SP);
                            if (has_next && q < G::NQ) dma_one(on, q, xbn);
                        }
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int d = ri - dy;
                            if (d < 0 || (d & 1) || d / 2 > 3) continue;
                            acc[d / 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy], f0, acc[d / 2], 0, 0, 0);
                        }
                        f0 = f1; f1 = f2;
                        __builtin_amdgcn_sched_barrier(0);   // keep later steps' LDS loads from being hoisted into spills
                    }
                }
        } else {
            const char* base = s_x + ((2 * ty) * PW + 2 * l15) * PXB;
            bf16x8 fn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fn[j] = *(const bf16x8*)(base + (2 * j * PW) * PXB + koffa[0]);
#pragma unroll
            for (int s = 0; s < KSA; ++s) {
                bf16x8 f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = fn[j];
                if (s + 1 < KSA) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fn[j] = *(const bf16x8*)(base + (2 * j * PW) * PXB + koffa[s + 1 < KSA ? s + 1 : 0]);
                }
                if (!STEM && s % SP == 0 && s / SP < NQW) {  // one DMA instruction of the next tile's patch
                    const int q = wave + kNW * (s / SP);
                    if (has_next && q < G::NQ) dma_one(on, q, xbn);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s], f[j], acc[j], 0, 0, 0);
            }
        }
        const f32x4 bav = *(const f32x4*)(s_b + cbase);
        if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = down_silu4(acc[j] + bav);
                down_lds_write_b64(s_t + ((ty + j) * kTW + l15) * G::TPXB + cbase * 2, make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
            }
            down_lds_barrier();
            // ---- 2. y = SiLU(Wb t + bb) ----
            const char* tb = s_t + (ty * kTW + l15) * G::TPXB + g * 16;
            f32x4 acc2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSB; ++s) {
                bf16x8 f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = *(const bf16x8*)(tb + j * kTW * G::TPXB + s * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s], f[j], acc2[j], 0, 0, 0);
            }
            const f32x4 bbv = *(const f32x4*)(s_b + CMID + cbase);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc2[j] + bbv;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + bav;
        }
        const int x = x0 + l15;
        // 16-byte stores (round 4): 8-byte stores per lane are store-ISSUE bound (cdna_hip_programming.md T21; measured on the planar 3x3
        // kernels: 240 cycles per 512-byte instruction).  v_permlane16_swap pairs the 8-byte pieces of lanes g and g + 1 (same pixel,
        // adjacent channel quads) of output rows j and j + 1: afterwards even-g lanes hold row j's 16 bytes (their own quad and lane
        // g + 1's), odd-g lanes row j + 1's (lane g - 1's quad and their own) -- two store instructions per tile instead of four.
        uint2 pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[j];
            if (FUSE || p.act) v = down_silu4(v);
            pk[j] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
        const int odd = g & 1;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const auto r0 = __builtin_amdgcn_permlane16_swap(pk[j].x, pk[j + 1].x, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pk[j].y, pk[j + 1].y, false, false);
            const int y = y0 + ty + j + odd;
            if (y < Ho && x < Wo)
                *(uint4*)(p.out + ((long long)(b * Ho + y) * Wo + x) * p.out_ld_b + (cbase - 4 * odd) * 2) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
    }
}

int g_down_cus = 0;
int down_common(DownParams& p, const void* in_dev, int in_ld, int in_choff, int cin, void* out_dev, int out_ld, int out_choff, int cout,
                const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int th) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "downblock: null pointer");
    AQ_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "downblock: H and W must be even (got %dx%d)", H, W);
    AQ_REQUIRE(in_ld % 8 == 0 && out_ld % 8 == 0 && in_choff % 8 == 0 && out_choff % 8 == 0 && in_choff + cin <= in_ld && out_choff + cout <= out_ld,
               "downblock: channel slices must be 8-aligned and inside their rows");
    AQ_REQUIRE((long long)B * H * W < (1LL << 31), "downblock: batch too large");
    p.in = (const char*)in_dev + (size_t)in_choff * 2; p.in_ld_b = in_ld * 2;
    p.out = (char*)out_dev + (size_t)out_choff * 2; p.out_ld_b = out_ld * 2;
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.B = B; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = (p.Wo + kTW - 1) / kTW; p.tiles_y = (p.Ho + th - 1) / th;
    AQ_REQUIRE((long long)B * p.tiles_x * p.tiles_y < (1LL << 30), "downblock: batch too large");
    p.n_tiles = B * p.tiles_x * p.tiles_y;
    p.zero = aq_zero_page();
    AQ_REQUIRE(p.zero, "downblock: zero page allocation failed");
    if (g_down_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_down_cus = cus;
    }
    return AQ_OK;
}

template <int CIN, int CMID, bool FUSE, int KT, bool STEM = false>
int launch_down(const DownParams& p, hipStream_t stream) {
    using G = DownGeom<CIN, CMID, FUSE, KT, STEM>;
    static bool attr = false;
    auto fn = downblock_kernel<CIN, CMID, FUSE, KT, STEM>;
    if (!attr) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        attr = true;
    }
    long long grid = g_down_cus;                             // > 150 KB of LDS: one persistent workgroup per CU
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kNW * 64), G::LDS, stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

// A-fragment image of a 3x3 (tap-major K blocks of 8 input channels) optionally followed by a 1x1:
// [M block][k-step ksa + ksb][lane 64] x 8 bf16; lane (m = lane & 15, g = lane >> 4) holds output channel 16 * Mblock + m and
// K block 4 * kstep + g.
int pack_down(const float* wa_host, const float* wb_host, int cin, int cmid, void* packed_dev, size_t* bytes, void* stream) {
    const int cbi = cin / 8, nblk = 9 * cbi, ksa = (nblk + 3) / 4, ksb = wb_host ? cmid / 32 : 0, mbn = cmid / 16;
    *bytes = (size_t)mbn * (ksa + ksb) * 64 * 16;
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_downblock: out of host memory");
    for (int mb = 0; mb < mbn; ++mb)
        for (int s = 0; s < ksa + ksb; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = mb * 16 + (lane & 15), g = lane >> 4;
                bf16_t* dst = host + (((size_t)mb * (ksa + ksb) + s) * 64 + lane) * 8;
                if (s < ksa) {
                    const int blk = 4 * s + g;
                    if (blk < nblk) {
                        const int tap = blk / cbi, c8 = blk % cbi;
                        for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(wa_host[((size_t)co * 9 + tap) * cin + c8 * 8 + e]);
                    }
                } else {
                    const int blk = 4 * (s - ksa) + g;
                    for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(wb_host[(size_t)co * cmid + blk * 8 + e]);
                }
            }
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

}  // namespace

// Fused form: wa KRSC (96,3,3,48), wb KRSC (96,1,1,96).
extern "C" int aq_pack_downblock_weights(const float* wa_host, const float* wb_host, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(wa_host && wb_host && bytes, "pack_downblock: null pointer");
    return pack_down(wa_host, wb_host, 48, 96, packed_dev, bytes, stream);
}

// in: bf16 NHWC [B][H][W][in_ld] with the 48 channels at in_choff; out: [B][H/2][W/2][out_ld] with the 96 channels at out_choff.
// bias_dev: [192] fp32 (ba | bb).
extern "C" int aq_downblock(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff,
                            const void* packed_w_dev, const float* bias_dev, int B, int H, int W, void* stream) {
    DownParams p{};
    const int rc = down_common(p, in_dev, in_ld, in_choff, 48, out_dev, out_ld, out_choff, 96, packed_w_dev, bias_dev, B, H, W, DownGeom<48, 96, true, 0>::TH);
    if (rc) return rc;
    return launch_down<48, 96, true, 0>(p, (hipStream_t)stream);
}

// model.0 + model.1 + model.2.cv1|cv2 in one launch (yolov5m, bf16): tiles uint8 [B][Hi][Wi][3] -> [B][Hi/4][Wi/4][out_ld] with the 96
// channels at out_choff.  stem_w / stem_bias: aq_pack_stem_weights' bf16 image of Conv(3, 48, 6, 2, 2) and its bias padded to 64 floats
// (exactly what aq_stem_conv takes); packed_w / bias: aq_downblock's.  Bit-identical to aq_stem_conv followed by aq_downblock.
extern "C" int aq_stemdown_supported(int Hi, int Wi) { return Hi > 0 && Wi > 0 && Hi % 4 == 0 && Wi % 4 == 0; }

extern "C" int aq_stemdown(const uint8_t* tiles_dev, void* out_dev, int out_ld, int out_choff, const void* stem_w_dev, const float* stem_bias_dev,
                           const void* packed_w_dev, const float* bias_dev, int B, int Hi, int Wi, void* stream) {
    AQ_REQUIRE(tiles_dev && stem_w_dev && stem_bias_dev, "stemdown: null pointer");
    AQ_REQUIRE(aq_stemdown_supported(Hi, Wi) && ((uintptr_t)tiles_dev & 3) == 0, "stemdown: tile size %dx%d must be a multiple of 4, tiles 4-byte aligned", Hi, Wi);
    AQ_REQUIRE((long long)B * Hi * Wi * 3 < (1LL << 40), "stemdown: batch too large");
    DownParams p{};
    // `in` is unused by the stem-fused form; down_common only offsets it
    const int rc = down_common(p, tiles_dev, 48, 0, 48, out_dev, out_ld, out_choff, 96, packed_w_dev, bias_dev, B, Hi / 2, Wi / 2, DownGeom<48, 96, true, 0, true>::TH);
    if (rc) return rc;
    p.tiles = tiles_dev; p.stem_w = (const char*)stem_w_dev; p.stem_b = stem_bias_dev; p.Hi = Hi; p.Wi = Wi;
    return launch_down<48, 96, true, 0, true>(p, (hipStream_t)stream);
}

// Plain 3x3 / stride 2 / pad 1 convolution + bias (+ SiLU), 96 -> 192 channels (the autotuner's AQ_CONV_CFG_DIRECT3X3S2 candidate).
extern "C" int aq_conv3x3s2_direct_supported(int cin, int cout) { return cin == 96 && cout == 192; }

extern "C" int aq_pack_conv3x3s2_direct(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_host && bytes && aq_conv3x3s2_direct_supported(cin, cout), "pack_conv3x3s2_direct: unsupported %d -> %d", cin, cout);
    return pack_down(w_host, nullptr, cin, cout, packed_dev, bytes, stream);
}

extern "C" int aq_conv3x3s2_direct(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int cin, int cout,
                                   const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int act, void* stream) {
    AQ_REQUIRE(aq_conv3x3s2_direct_supported(cin, cout), "conv3x3s2_direct: unsupported %d -> %d", cin, cout);
    DownParams p{};
    const int rc = down_common(p, in_dev, in_ld, in_choff, cin, out_dev, out_ld, out_choff, cout, packed_w_dev, bias_dev, B, H, W, DownGeom<96, 192, false, 3>::TH);
    if (rc) return rc;
    p.act = act;
    return launch_down<96, 192, false, 3>(p, (hipStream_t)stream);
}
