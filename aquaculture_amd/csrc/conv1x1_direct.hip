// Direct 1x1 convolution for gfx950 (bf16), small K: y = SiLU(W x + b) with (Cin, Cout) in {(96, 96), (192, 192), (384, 192), (384, 384)}.
//
// The pipelined implicit-GEMM kernel (csrc/conv_igemm.hip) streams K in 32-channel chunks through a multi-stage LDS pipeline;
// for a 1x1 with K = 96..384 a tile is over after 3-12 chunks and the pipeline's prologue/epilogue dominate (the 160x160 / 80x80
// 1x1 layers of yolov5m run at 50-65 % of their HBM floor there).  This kernel has no K pipeline at all:
//   * pixels are a flat list (a 1x1 needs no geometry); a persistent workgroup of 12 waves owns TP consecutive pixels;
//   * the whole [TP][Cin] input tile arrives by LDS-DMA into one half of a double buffer (the next tile's DMA is issued as soon
//     as the barrier that frees the other half is passed);
//   * a wave owns a PAIR of 16-row M blocks (32 output channels) and a pixel group; it keeps those weights in registers (Cin / 4
//     VGPRs), reads MFMA B fragments straight from the tile (pixel stride 2 mod 4 sixteen-byte slots: conflict-free for
//     ds_read_b128's lane groups), each fragment feeding both blocks;
//   * the rows of a pair are permuted at packing time -- block m, row 4 g + i holds channel 8 g + 4 m + i -- so that the two C
//     fragments of a lane are 8 CONSECUTIVE output channels: one 16-byte store per pixel and lane, 64-byte runs per wave.
// One barrier per tile.  The engine's autotuner times it against the implicit-GEMM tile shapes per layer (config id
// AQ_CONV_CFG_DIRECT1X1) and keeps whichever is faster.
#include "conv_device.h"

using namespace aqdev;

namespace {

struct C1Params {
    const char* in;
    char* out;
    const char* w;
    const float* bias;
    const char* zero;
    int in_ld_b, out_ld_b;
    int npix, n_tiles, act;
    float out_inv_scale;       // F8OUT: the output leaves as OCP e4m3fn codes of y * out_inv_scale (one byte per channel)
};

constexpr int kNW = 12;

// KS: k-steps of 32 input channels; MBT: 16-row M blocks (Cout / 16); MBW: M blocks per wave; NBW: 16-pixel blocks per wave and tile
template <int KS, int MBT, int MBW, int NBW> struct C1Geom {
    static constexpr int CIN = 32 * KS, COUT = 16 * MBT, CB = CIN / 8;
    static constexpr int MG = MBT / MBW;                     // wave groups along M
    static constexpr int PG = kNW / MG;                      // pixel groups
    static constexpr int TP = PG * NBW * 16;                 // pixels per tile
    static constexpr int SPP = CB + 2;                       // CB is a multiple of 4: CB + 2 is 2 mod 4 (conflict-free reads)
    static constexpr int PXB = SPP * 16;
    static constexpr int NQ = (TP * SPP + 63) / 64;
    static constexpr int XPB = NQ * 1024;
    static constexpr int LDS = 2 * XPB + COUT * 4;
    static_assert(MBT % MBW == 0 && kNW % MG == 0 && LDS <= 160 * 1024, "shape");
};

__device__ __forceinline__ f32x4 c1_silu4(f32x4 v) {        // same sequence as the shared conv epilogue (bf16 mode)
    const f32x4 t = v * -1.44269504f;
    f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    d = d + 1.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    return v * r;
}

// F8OUT (round 3, the fp8 path of BASELINE.json configs[3]): the consumer is the fp8 planar 3x3 kernel (conv3x3_pl.hip, family f8nb13), so
// the activation is quantised HERE, in the producer's epilogue: y / scale, clamped to e4m3's range, v_cvt_pk_fp8_f32 (round to nearest
// even), 8 codes = one 8-byte store per lane and pixel; out_ld_b is the pixel pitch in bytes, the codes of channel c sit at byte c.
template <int KS, int MBT, int MBW, int NBW, bool F8OUT = false>
__global__ __launch_bounds__(kNW * 64) void conv1x1_direct_kernel(const C1Params p) {
    using G = C1Geom<KS, MBT, MBW, NBW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_b = (float*)(smem + 2 * G::XPB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(MBW == 2, "the paired-row layout below is written for two M blocks per wave");
    const int mb0 = (wave % G::MG) * MBW, pg = wave / G::MG;  // first of this wave's M blocks; pixel group
    const int g = lane >> 4, l15 = lane & 15;
    const int cbase = mb0 * 16 + g * 8;                      // this lane's 8 consecutive output channels

    bf16x8 wv[KS][MBW];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < MBW; ++m) wv[s][m] = ((const bf16x8*)p.w)[((size_t)(mb0 + m) * KS + s) * 64 + lane];
    for (int i = tid; i < G::COUT; i += kNW * 64) s_b[i] = p.bias[i];

    auto issue_dma = [&](int tile, char* xb) {               // the [TP][Cin] tile of `tile`, 64 sixteen-byte slots per instruction
        const long long n0 = (long long)tile * G::TP;
#pragma unroll 1
        for (int q = wave; q < G::NQ; q += kNW) {
            const int slot = q * 64 + lane;
            const int px = slot / G::SPP, part = slot - px * G::SPP;
            const bool valid = px < G::TP && part < G::CB && n0 + px < p.npix;
            const char* src = p.in + (n0 + px) * p.in_ld_b + part * 16;
            glds16(valid ? src : p.zero, xb + q * 1024);
        }
    };

    int tile = first_tile(gridDim.x, blockIdx.x);
    if (tile < p.n_tiles) issue_dma(tile, smem);
    int cur = 0;
    bool prev_full = false;
    for (; tile < p.n_tiles; tile += gridDim.x, cur ^= 1) {
        const long long n0 = (long long)tile * G::TP;
        const char* s_x = smem + cur * G::XPB;
        // this tile has landed (own DMA: vmcnt; other waves': barrier), and every wave is done with the other buffer.  vmcnt is
        // in-order: after a full tile the DMA is older than that tile's NBW output stores, which may stay in flight.
        if (prev_full) wait_vmcnt<NBW>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        prev_full = n0 + G::TP <= p.npix;
        if (tile + (int)gridDim.x < p.n_tiles) issue_dma(tile + (int)gridDim.x, smem + (cur ^ 1) * G::XPB);

        f32x4 bv[MBW];
#pragma unroll
        for (int m = 0; m < MBW; ++m) bv[m] = *(const f32x4*)(s_b + cbase + m * 4);
#pragma unroll 2
        for (int j = 0; j < NBW; ++j) {
            const int px = (pg * NBW + j) * 16 + l15;
            const char* base = s_x + px * G::PXB + g * 16;
            bf16x8 f[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) f[s] = *(const bf16x8*)(base + s * 64);
            f32x4 acc[MBW];
#pragma unroll
            for (int m = 0; m < MBW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int m = 0; m < MBW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[s][m], f[s], acc[m], 0, 0, 0);
            f32x4 v0 = acc[0] + bv[0], v1 = acc[1] + bv[1];
            if (p.act) { v0 = c1_silu4(v0); v1 = c1_silu4(v1); }
            if constexpr (F8OUT) {
                f32x4 q0 = v0 * p.out_inv_scale, q1 = v1 * p.out_inv_scale;
#pragma unroll
                for (int e = 0; e < 4; ++e) {                // saturate: a tile brighter than the calibration set must not become a NaN code
                    q0[e] = __builtin_amdgcn_fmed3f(q0[e], -448.0f, 448.0f);
                    q1[e] = __builtin_amdgcn_fmed3f(q1[e], -448.0f, 448.0f);
                }
                int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q0[0], q0[1], 0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q0[2], q0[3], w0, true);
                int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q1[0], q1[1], 0, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q1[2], q1[3], w1, true);
                if (n0 + px < p.npix) *(uint2*)(p.out + (n0 + px) * p.out_ld_b + cbase) = make_uint2((unsigned)w0, (unsigned)w1);
            } else if (n0 + px < p.npix)
                *(uint4*)(p.out + (n0 + px) * p.out_ld_b + cbase * 2) =
                    make_uint4(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3]));
        }
    }
}

int g_c1_cus = 0;

template <int KS, int MBT, int MBW, int NBW, bool F8OUT = false>
int launch_c1(C1Params p, hipStream_t stream) {
    using G = C1Geom<KS, MBT, MBW, NBW>;
    static bool attr = false;
    auto fn = conv1x1_direct_kernel<KS, MBT, MBW, NBW, F8OUT>;
    if (!attr) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        attr = true;
    }
    p.n_tiles = (p.npix + G::TP - 1) / G::TP;
    long long grid = g_c1_cus;                               // > 80 KB of LDS: one persistent workgroup per CU
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kNW * 64), G::LDS, stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

}  // namespace

extern "C" int aq_conv1x1_direct_supported(int cin, int cout) {
    return (cin == 96 && cout == 96) || (cin == 192 && cout == 192) || (cin == 384 && (cout == 192 || cout == 384));
}

// Packs fused fp32 weights KRSC (cout, 1, 1, cin) into the A-fragment image the kernel loads once per workgroup:
// [M block][k-step][lane] x 8 bf16; M blocks come in pairs covering 32 channels: block 2 P + m, row r = lane & 15 holds output
// channel 32 P + 8 (r >> 2) + 4 m + (r & 3) (see the kernel header); lane group g = lane >> 4 holds input channels
// 32 * kstep + 8 * g .. + 7.
extern "C" int aq_pack_conv1x1_direct(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_host && bytes && aq_conv1x1_direct_supported(cin, cout), "pack_conv1x1_direct: unsupported %d -> %d", cin, cout);
    const int ks = cin / 32, mbt = cout / 16;
    *bytes = (size_t)mbt * ks * 64 * 16;
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_conv1x1_direct: out of host memory");
    for (int mb = 0; mb < mbt; ++mb)
        for (int s = 0; s < ks; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                const int co = (mb >> 1) * 32 + 8 * (r >> 2) + 4 * (mb & 1) + (r & 3);
                bf16_t* dst = host + (((size_t)mb * ks + s) * 64 + lane) * 8;
                for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(w_host[(size_t)co * cin + 32 * s + 8 * g + e]);
            }
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

// in / out: bf16, `npix` pixels with row lengths in_ld / out_ld (elements) and the channels at in_choff / out_choff.
extern "C" int aq_conv1x1_direct(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int cin, int cout,
                                 const void* packed_w_dev, const float* bias_dev, long long npix, int act, void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "conv1x1_direct: null pointer");
    AQ_REQUIRE(aq_conv1x1_direct_supported(cin, cout), "conv1x1_direct: unsupported %d -> %d", cin, cout);
    AQ_REQUIRE(npix > 0 && npix < (1LL << 31), "conv1x1_direct: bad pixel count");
    AQ_REQUIRE(in_ld % 8 == 0 && out_ld % 8 == 0 && in_choff % 8 == 0 && out_choff % 8 == 0 && in_choff + cin <= in_ld && out_choff + cout <= out_ld,
               "conv1x1_direct: channel slices must be 8-aligned and inside their rows");
    C1Params p{};
    p.in = (const char*)in_dev + (size_t)in_choff * 2; p.in_ld_b = in_ld * 2;
    p.out = (char*)out_dev + (size_t)out_choff * 2; p.out_ld_b = out_ld * 2;
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.npix = (int)npix; p.act = act;
    p.zero = aq_zero_page();
    AQ_REQUIRE(p.zero, "conv1x1_direct: zero page allocation failed");
    if (g_c1_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_c1_cus = cus;
    }
    const hipStream_t st = (hipStream_t)stream;
    if (cin == 96) return launch_c1<3, 6, 2, 4>(p, st);      // 3 channel pairs x 4 pixel groups, 256-pixel tiles
    if (cin == 192) return launch_c1<6, 12, 2, 4>(p, st);    // 6 x 2, 128-pixel tiles
    if (cout == 192) return launch_c1<12, 12, 2, 2>(p, st);  // 6 x 2, 64-pixel tiles
    return launch_c1<12, 24, 2, 4>(p, st);                   // 12 x 1, 64-pixel tiles
}


// The same convolution writing OCP e4m3fn codes of y / out_scale (192 -> 192 and 384 -> 384: the Bottleneck cv1 layers whose consumer is the
// fp8 planar 3x3 kernel).  out_pitch_bytes: bytes per output pixel row; the codes of channel c go to byte out_byte_off + c.
extern "C" int aq_conv1x1_direct_f8out(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_pitch_bytes, int out_byte_off, int cin,
                                       int cout, const void* packed_w_dev, const float* bias_dev, long long npix, int act, float out_scale,
                                       void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "conv1x1_direct_f8out: null pointer");
    AQ_REQUIRE((cin == 192 && cout == 192) || (cin == 384 && cout == 384), "conv1x1_direct_f8out: unsupported %d -> %d", cin, cout);
    AQ_REQUIRE(npix > 0 && npix < (1LL << 31) && out_scale > 0.0f, "conv1x1_direct_f8out: bad pixel count or scale");
    AQ_REQUIRE(in_ld % 8 == 0 && in_choff % 8 == 0 && in_choff + cin <= in_ld && out_pitch_bytes % 8 == 0 && out_byte_off % 8 == 0 &&
                   out_byte_off + cout <= out_pitch_bytes, "conv1x1_direct_f8out: slices must be 8-byte aligned and inside their rows");
    C1Params p{};
    p.in = (const char*)in_dev + (size_t)in_choff * 2; p.in_ld_b = in_ld * 2;
    p.out = (char*)out_dev + out_byte_off; p.out_ld_b = out_pitch_bytes;
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.npix = (int)npix; p.act = act; p.out_inv_scale = 1.0f / out_scale;
    p.zero = aq_zero_page();
    AQ_REQUIRE(p.zero, "conv1x1_direct_f8out: zero page allocation failed");
    if (g_c1_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_c1_cus = cus;
    }
    const hipStream_t st = (hipStream_t)stream;
    if (cin == 192) return launch_c1<6, 12, 2, 4, true>(p, st);
    return launch_c1<12, 24, 2, 4, true>(p, st);
}
