// Fused stem for gfx950: uint8 RGB tiles -> (/255) -> Conv(3, C, k=6, s=2, p=2) + bias + SiLU -> NHWC activations.
//
// Replaces `im = im.float() / 255` + model.0 (Conv(3, 48, 6, 2, 2) fused with its BN) of the reference's yolov5 dependency
// ([UPSTREAM detect.py run(), models/common.py Conv.forward_fuse]; reached through reference README.md:77).
// The generic gather kernel is a poor fit for 3 input channels (its 2x2 space-to-depth staging buffer costs 3.3 MB per
// tile, written and read back).  This kernel never materialises an im2col matrix:
//   * a workgroup owns an output tile of 8 rows x 64 columns; the 20 x 132-pixel input patch is read once (dword loads),
//     converted through a 256-entry table of v / 255 (rounded to the activation type, so the values are exactly those of
//     the two-kernel path) and kept in LDS as activations, RGB interleaved, one padded row per input row;
//   * K is ordered ky-major with each ky row padded from 18 (= 6 kx * 3 c) to 24: the 18 inputs of one (pixel, ky) are 18
//     CONSECUTIVE elements of a patch row, so every MFMA B fragment (8 consecutive K of one pixel) is a contiguous
//     16-byte run of the patch -- the padded K positions read the neighbouring pixel's (finite) values against zero weights;
//   * M = cout is tiled with 16-row MFMA blocks (v_mfma_f32_16x16x32_bf16 / 16x16x4_f32), so yolov5m's 48 channels are
//     exactly 3 blocks; the weights (A fragments) stay in registers for the life of the persistent workgroup;
//   * in the 16x16 C/D layout a lane holds rows 4 g .. 4 g + 3 of every M block; the weight rows are PERMUTED at packing time
//     (block m, row 4 g + i = channel 4 MB g + 4 m + i) so that those are 4 MB consecutive channels: bias + SiLU in registers, then
//     a lane's stores of one pixel form one contiguous run (24 B for 48 bf16 channels), the four lane groups cover the pixel and
//     the 16 pixels of a wave are contiguous in NHWC -- no LDS round trip in the epilogue.
#include "conv_device.h"
#include <type_traits>

using namespace aqdev;

namespace {

constexpr int kK = 108;                        // 6 * 6 * 3
constexpr int kKyPad = 24;                     // K positions per ky row (18 real + 6 zero-weight)
constexpr int kTH = 8, kTW = 64;               // output tile per workgroup: 8 rows x 64 columns, wave w owns rows 2w, 2w+1
constexpr int kPH = 2 * kTH + 4;               // 20 patch rows
constexpr int kPRowDw = 104;                   // input dwords (= 416 bytes = 416 patch elements) loaded per patch row:
                                               // bytes [6*x0 - 8, 6*x0 + 408) of the image row; pixel tx's 18+6 values start
                                               // at element 2 + 6*tx, the last one read is element 2 + 6*63 + 23 = 403
constexpr int kMaxMB = 4;                      // cout <= 64

template <bool F32> struct StemGeom {
    static constexpr int EB = F32 ? 4 : 2;
    static constexpr int PROWB = kPRowDw * 4 * EB + 16;           // patch row stride in bytes (+16: rows start on shifted banks)
    static constexpr int KS = F32 ? 6 * kKyPad / 4 : 5;           // MFMA k-steps: 36 x (K=4) fp32, 5 x (K=32) bf16 (K 144 -> 160)
    static constexpr int PATCHB = kPH * PROWB;
    static constexpr int LUTB = 256 * EB;
    static constexpr int STAGEB = ((kPH * kPRowDw + 255) / 256) * 1024;   // raw input dwords of one patch, lane-linear (DMA variant)
    static constexpr int LDS = 2 * PATCHB + LUTB + STAGEB;        // two patch buffers + value table + raw staging
};

// Host/device agreement on the A-fragment image: [k-step][M block][lane] -> 8 bf16 (16 B) or 1 float.
//   bf16: lane (m = lane & 15, g = lane >> 4) holds K block blk = 4 * s + g (8 consecutive K): ky = blk / 3, j = 8 * (blk % 3) + e
//   fp32: lane holds K = 4 * s + g: ky = s / 6, j = 4 * (s % 6) + g
// j < 18 maps to (kx, c) = (j / 3, j % 3); everything else is zero.

// DMA (bf16, image rows whose byte length is a multiple of 4): the raw input dwords of the NEXT tile's patch travel by LDS-DMA into a
// staging area instead of through registers, all waits are hand-counted and the barrier is a bare s_barrier.  With register loads the
// compiler cannot count the output stores issued since (the block loop has wave-uniform skips), waits with vmcnt(0) before the
// conversion, and __syncthreads adds another vmcnt(0): every tile then ended by draining ALL of its output stores -- 88 of the
// kernel's 237 us (measured by compiling the load out).  Here a full tile waits with vmcnt(stores of this tile), i.e. for the DMA only.
template <bool F32, int MB, bool DMA = false>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const ConvParams p, const uint8_t* __restrict__ tiles, int tiles_x, int tiles_y) {
    static_assert(!(DMA && F32), "the DMA variant converts arithmetically: bf16 only");
    using G = StemGeom<F32>;
    using elem_t = typename std::conditional<F32, float, bf16_t>::type;
    using afrag_t = typename std::conditional<F32, float, bf16x8>::type;
    constexpr int EB = G::EB, PROWB = G::PROWB, KS = G::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_patch = smem;
    elem_t* s_lut = (elem_t*)(smem + 2 * G::PATCHB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;

    // ---- once per workgroup: weights and bias to registers, value table to LDS ----
    afrag_t a[KS][MB];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < MB; ++m) a[s][m] = ((const afrag_t*)p.w)[(s * MB + m) * 64 + lane];
    f32x4 bias[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) bias[m] = *(const f32x4*)(p.bias + (4 * MB) * g + 4 * m);      // bias is padded to 64 floats
    {
        const float v = (float)tid / 255.0f;                 // [UPSTREAM detect.py]: im.float() / 255
        if constexpr (F32) s_lut[tid] = v; else s_lut[tid] = aq_f2bf(v);
    }
    // per-lane K offsets inside the patch (bytes, relative to the pixel's first element of patch row 2*ty)
    int koff[F32 ? 1 : KS];
    if constexpr (F32) koff[0] = g * 4;
    else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int blk = 4 * s + g;
            const int ky = blk / 3 < 6 ? blk / 3 : 5;       // K blocks 18, 19 carry zero weights: read any valid row
            koff[s] = ky * PROWB + (blk - 3 * (blk / 3)) * 16;
        }
    }
    const bool aligned_rows = (W & 3) == 0 && ((uintptr_t)tiles & 3) == 0;

    const int tiles_per_img = tiles_y * tiles_x;
    constexpr int NIT = (kPH * kPRowDw + 255) / 256;
    // Raw input dwords of one tile's patch: image rows 2*y0-2 .. 2*y0+17, image-row bytes 6*x0-8 .. 6*x0+407, zero outside.
    auto load_raw = [&](int tile, uint32_t (&raw)[NIT]) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / tiles_x, tx0 = tr - ty0 * tiles_x;
        const int y0 = ty0 * kTH, x0 = tx0 * kTW;
        const uint8_t* img = tiles + (size_t)b * H * W * 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int r = i / kPRowDw, d = i - r * kPRowDw;
            const int iy = 2 * y0 - 2 + r, byte0 = 6 * x0 - 8 + 4 * d;
            uint32_t v = 0;
            if (tile < p.n_tiles_n && i < kPH * kPRowDw && iy >= 0 && iy < H) {
                const uint8_t* row = img + (size_t)iy * W * 3;
                if (aligned_rows) {                          // 3W % 4 == 0: a dword is wholly inside or outside the row
                    if (byte0 >= 0 && byte0 < W * 3) v = *(const uint32_t*)(row + byte0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int bo = byte0 + e;
                        if (bo >= 0 && bo < W * 3) v |= (uint32_t)row[bo] << (8 * e);
                    }
                }
            }
            raw[it] = v;
        }
    };
    // uint8 -> activation through the value table, into one of the two patch buffers
    auto convert = [&](const uint32_t (&raw)[NIT], char* patch) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int r = i / kPRowDw, d = i - r * kPRowDw;
            if (i < kPH * kPRowDw) {
                const uint32_t v = raw[it];
                char* dst = patch + r * PROWB + d * 4 * EB;
                if constexpr (F32) {
                    const elem_t e0 = s_lut[v & 255], e1 = s_lut[(v >> 8) & 255], e2 = s_lut[(v >> 16) & 255], e3 = s_lut[v >> 24];
                    *(f32x4*)dst = f32x4{e0, e1, e2, e3};
                } else {
                    // bf16: RNE(v * (1/255)) == RNE(v / 255) for all 256 byte values (checked exhaustively, tests/test_host_logic.py),
                    // so the four table look-ups (dependent LDS reads) become v_cvt_f32_ubyte + v_mul + v_cvt_pk_bf16_f32
                    constexpr float k = 1.0f / 255.0f;
                    const float f0 = (float)(v & 255u) * k, f1 = (float)((v >> 8) & 255u) * k, f2 = (float)((v >> 16) & 255u) * k,
                                f3 = (float)(v >> 24) * k;
                    *(uint2*)dst = make_uint2(pack_bf16x2(f0, f1), pack_bf16x2(f2, f3));
                }
            }
        }
    };

    // ---- DMA variant: raw dwords -> staging (lane-linear: slot i = tid + 256 it), then staging -> patch ----
    char* s_stage = smem + 2 * G::PATCHB + G::LUTB;
    auto dma_raw = [&](int tile) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / tiles_x, tx0 = tr - ty0 * tiles_x;
        const int y0 = ty0 * kTH, x0 = tx0 * kTW;
        const uint8_t* img = tiles + (size_t)b * H * W * 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int r = i / kPRowDw, d = i - r * kPRowDw;
            const int iy = 2 * y0 - 2 + r, byte0 = 6 * x0 - 8 + 4 * d;
            const bool ok = tile < p.n_tiles_n && i < kPH * kPRowDw && iy >= 0 && iy < H && byte0 >= 0 && byte0 < W * 3;
            const char* src = ok ? (const char*)img + (size_t)iy * W * 3 + byte0 : p.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(s_stage + (it * 256 + wave * 64) * 4), 4, 0, 0);
        }
    };
    auto convert_stage = [&](char* patch) {       // each lane converts the slots it loaded itself: no barrier between DMA and here
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int r = i / kPRowDw, d = i - r * kPRowDw;
            if (i < kPH * kPRowDw) {
                const uint32_t v = *(const uint32_t*)(s_stage + i * 4);
                constexpr float k = 1.0f / 255.0f;
                const float f0 = (float)(v & 255u) * k, f1 = (float)((v >> 8) & 255u) * k, f2 = (float)((v >> 16) & 255u) * k,
                            f3 = (float)(v >> 24) * k;
                const uint2 o = make_uint2(pack_bf16x2(f0, f1), pack_bf16x2(f2, f3));
                // written by asm: a visible LDS store would make the compiler wait for every outstanding LDS-DMA and store first
                asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(patch + r * PROWB + d * 4 * EB)), "v"(o) : "memory");
            }
        }
    };
    constexpr int NST = 8 * (MB / 2 + (MB & 1));             // output stores one wave issues for a full tile

    // Pipeline over this workgroup's tiles t0, t0+G, ...: while tile t is multiplied out of patch buffer `cur`, the raw
    // dwords of tile t+G are in flight; they are converted into the other buffer after the compute phase, and one barrier
    // per tile separates "buffer written by all waves" from "buffer read" (and the reads of two tiles ago from the rewrite).
    int tile = first_tile(gridDim.x, blockIdx.x);
    uint32_t raw[NIT];
    if constexpr (DMA) {
        __syncthreads();                                     // (the value table is not used by this variant; keeps the prologue alike)
        dma_raw(tile);
        wait_vmcnt<0>();
        convert_stage(s_patch);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // staging read, patch written
        dma_raw(tile + (int)gridDim.x);
    } else {
        load_raw(tile, raw);
        __syncthreads();                                     // value table is complete
        convert(raw, s_patch);
        load_raw(tile + (int)gridDim.x, raw);
    }
    int cur = 0;
    for (; tile < p.n_tiles_n; tile += gridDim.x, cur ^= 1) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / tiles_x, tx0 = tr - ty0 * tiles_x;
        const int y0 = ty0 * kTH, x0 = tx0 * kTW;
        const char* patch = s_patch + cur * G::PATCHB;
        if constexpr (DMA) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
        // ---- 8 blocks of 16 pixels per wave: rows 2*wave, 2*wave+1 x 4 column blocks ----
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            const int ty = 2 * wave + (q >> 2), txb = (q & 3) * 16;
            const int y = y0 + ty;
            if (y >= Ho || x0 + txb >= Wo) continue;         // wave-uniform
            const char* base = patch + (2 * ty) * PROWB + (2 + 6 * (txb + l15)) * EB;
            f32x4 acc[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!F32) {
                bf16x8 bf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const uint32_t* src = (const uint32_t*)(base + koff[s]);         // 4-byte aligned
                    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
                    const uint4 u = make_uint4(w0, w1, w2, w3);
                    __builtin_memcpy(&bf[s], &u, 16);
                }
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][m], bf[s], acc[m], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float bv = *(const float*)(base + koff[0] + (s / 6) * PROWB + (s % 6) * 16);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][m], bv, acc[m], 0, 0, 0);
                }
            }
            // ---- epilogue: lane = (pixel l15, channels 4 MB g .. 4 MB g + 4 MB - 1): one contiguous run per lane ----
            const int x = x0 + txb + l15;
            char* orow = p.out + (long long)((b * Ho + y) * Wo + x) * p.out_ld_b + (4 * MB) * g * EB;
            uint2 pk[MB];                                    // bf16 mode: packed results of the MB blocks
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                f32x4 v = acc[m] + bias[m];
                if (p.act) {
                    if constexpr (F32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu<true>(v[e]);
                    } else {
                        const f32x4 t = v * -1.44269504f;
                        f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]),
                                   __builtin_amdgcn_exp2f(t[3])};
                        d = d + 1.0f;
                        const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]),
                                         __builtin_amdgcn_rcpf(d[3])};
                        v = v * r;
                    }
                }
                if constexpr (F32) {
                    if (x < Wo && (4 * MB) * g + 4 * m < p.cout) *(f32x4*)(orow + m * 16) = v;
                } else pk[m] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
            if constexpr (!F32) {
                // 8 MB bytes per lane (24 for 48 channels) at an 8-byte aligned address: 16-byte pieces, then an 8-byte rest.
                // (cout is a multiple of 8, so a 16-byte piece is written whole or not at all.)
                struct __attribute__((packed, aligned(8))) U16 { uint32_t v[4]; };
                if (x < Wo) {
#pragma unroll
                    for (int m = 0; m + 1 < MB; m += 2)
                        if ((4 * MB) * g + 4 * m < p.cout) *(U16*)(orow + m * 8) = U16{{pk[m].x, pk[m].y, pk[m + 1].x, pk[m + 1].y}};
                    if constexpr (MB & 1)
                        if ((4 * MB) * g + 4 * (MB - 1) < p.cout) *(uint2*)(orow + (MB - 1) * 8) = pk[MB - 1];
                }
            }
        }
        if constexpr (DMA) {
            // the DMA of the next tile's raw dwords is older than this tile's output stores: a full tile issued exactly NST of them
            const bool full = y0 + kTH <= Ho && x0 + kTW <= Wo;
            if (full) wait_vmcnt<NST>(); else wait_vmcnt<0>();
            convert_stage(s_patch + (cur ^ 1) * G::PATCHB);   // next tile's patch (all zeros past the last tile)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_raw(tile + 2 * (int)gridDim.x);
        } else {
            convert(raw, s_patch + (cur ^ 1) * G::PATCHB);   // next tile's patch (all zeros past the last tile)
            load_raw(tile + 2 * (int)gridDim.x, raw);
        }
    }
}

int g_stem_occ[4][kMaxMB + 1];                // resident workgroups per CU, 0 = not queried yet ([F32 + 2 * DMA])
int g_stem_cus = 0;

size_t stem_packed_bytes(bool f32, int mb) {
    return f32 ? (size_t)StemGeom<true>::KS * mb * 64 * 4 : (size_t)StemGeom<false>::KS * mb * 64 * 16;
}

template <bool F32, int MB, bool DMA = false>
int launch_stem(const ConvParams& p, const uint8_t* tiles, int tiles_x, int tiles_y, hipStream_t stream) {
    auto fn = stem_conv_kernel<F32, MB, DMA>;
    constexpr size_t lds = StemGeom<F32>::LDS;
    if (!g_stem_occ[F32 + 2 * DMA][MB]) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0;
        AQ_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)fn, 256, lds));
        g_stem_occ[F32 + 2 * DMA][MB] = occ > 0 ? occ : 1;
    }
    long long grid = (long long)g_stem_cus * g_stem_occ[F32 + 2 * DMA][MB];
    if (grid > p.n_tiles_n) grid = p.n_tiles_n;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, stream, p, tiles, tiles_x, tiles_y);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

}  // namespace

// Packs fused fp32 stem weights KRSC (cout, 6, 6, 3) into the A-fragment image the kernel loads once per workgroup
// (layout: see the comment above stem_conv_kernel).  Rows >= cout and the padded K positions are zero.
extern "C" int aq_pack_stem_weights(const float* w_krsc_host, int cout, int precision, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_krsc_host && bytes && cout > 0 && cout <= 16 * kMaxMB && cout % 8 == 0, "pack_stem: cout must be a multiple of 8, at most %d", 16 * kMaxMB);
    const bool f32 = precision == AQ_FP32;
    const int mb = (cout + 15) / 16;
    *bytes = stem_packed_bytes(f32, mb);
    if (!packed_dev) return AQ_OK;
    unsigned char* host = (unsigned char*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_stem: out of host memory");
    auto weight = [&](int co, int ky, int j) -> float {      // j = kx * 3 + c inside a ky row
        return (co < cout && ky < 6 && j < 18) ? w_krsc_host[(size_t)co * kK + ky * 18 + j] : 0.0f;
    };
    const int ks = f32 ? StemGeom<true>::KS : StemGeom<false>::KS;
    for (int s = 0; s < ks; ++s)
        for (int m = 0; m < mb; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                const int co = (4 * mb) * (r >> 2) + 4 * m + (r & 3);   // row permutation, see the kernel header
                const size_t slot = ((size_t)s * mb + m) * 64 + lane;
                if (f32) ((float*)host)[slot] = weight(co, s / 6, 4 * (s % 6) + g);
                else {
                    const int blk = 4 * s + g;
                    for (int e = 0; e < 8; ++e) ((bf16_t*)host)[slot * 8 + e] = aq_f2bf(weight(co, blk / 3, 8 * (blk % 3) + e));
                }
            }
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

extern "C" int aq_stem_conv(const uint8_t* tiles_dev, void* out_dev, int out_ld, int out_choff, int cout,
                            const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int act, int precision,
                            void* stream) {
    AQ_REQUIRE(tiles_dev && out_dev && packed_w_dev && bias_dev, "stem_conv: null pointer");
    AQ_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_conv: H and W must be even (got %dx%d)", H, W);
    AQ_REQUIRE(cout > 0 && cout <= 16 * kMaxMB && cout % 8 == 0 && out_choff % 8 == 0 && out_ld % 8 == 0, "stem_conv: bad channel layout");
    const bool f32 = precision == AQ_FP32;
    const int eb = aq_elem_bytes(precision);
    ConvParams p{};
    p.out = (char*)out_dev + (size_t)out_choff * eb; p.out_ld_b = out_ld * eb;
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.B = B; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.cout = cout; p.act = act; p.npix = B * p.Ho * p.Wo;
    AQ_REQUIRE((long long)B * H * W * 3 < (1LL << 31) * 2 && (long long)B * p.Ho * p.Wo < (1LL << 31), "stem_conv: batch too large");
    const int tiles_x = (p.Wo + kTW - 1) / kTW, tiles_y = (p.Ho + kTH - 1) / kTH;
    AQ_REQUIRE((long long)B * tiles_x * tiles_y < (1LL << 31), "stem_conv: batch too large");
    p.n_tiles_n = B * tiles_y * tiles_x;
    p.n_tiles_m = 1;
    if (g_stem_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_stem_cus = cus;
    }
    const hipStream_t st = (hipStream_t)stream;
    const int mb = (cout + 15) / 16;
    // DMA variant: bf16, every image row a whole number of dwords from a 4-byte aligned base (a dword is inside or outside the row)
    const bool dma = !f32 && (W % 4) == 0 && ((uintptr_t)tiles_dev & 3) == 0 && getenv("AQ_STEM_NO_DMA") == nullptr;
    if (dma) {
        p.zero = aq_zero_page();
        AQ_REQUIRE(p.zero, "stem_conv: zero page allocation failed");
    }
#define AQ_STEM_CASE(M) case M: return f32 ? launch_stem<true, M>(p, tiles_dev, tiles_x, tiles_y, st) \
                                        : (dma ? launch_stem<false, M, true>(p, tiles_dev, tiles_x, tiles_y, st) : launch_stem<false, M>(p, tiles_dev, tiles_x, tiles_y, st))
    switch (mb) {
        AQ_STEM_CASE(1);
        AQ_STEM_CASE(2);
        AQ_STEM_CASE(3);
        AQ_STEM_CASE(4);
    }
#undef AQ_STEM_CASE
    AQ_REQUIRE(false, "stem_conv: unsupported cout %d", cout);
    return AQ_OK;
}
