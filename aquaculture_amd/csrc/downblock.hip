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
};

constexpr int kCin = 48, kCmid = 96, kCout = 96;
constexpr int kNW = 12, kPG = 2;                         // 6 M blocks x 2 pixel groups
constexpr int kTH = 4 * kPG, kTW = 16;                   // output tile: every wave owns 4 rows x 16 pixels
constexpr int kPH = 2 * kTH + 1, kPW = 2 * kTW + 1;      // input patch 17 x 33 (stride 2, pad 1)
constexpr int kPP = kPH * kPW;
constexpr int kCBI = kCin / 8;                           // 16-byte channel blocks of an input pixel
constexpr int kSPPX = 7, kPXB = kSPPX * 16;              // x patch: 6 channel slots + 1 pad per pixel
constexpr int kNQ = (kPP * kSPPX + 63) / 64;             // LDS-DMA wave instructions per patch
constexpr int kXPB = kNQ * 1024;
constexpr int kKSA = (9 * kCBI + 3) / 4;                 // 14 k-steps (32 K each) of the 3x3; K blocks are tap-major
constexpr int kNBLKA = 9 * kCBI;
constexpr int kKSB = kCmid / 32;                         // 3 k-steps of the 1x1
constexpr int kSPPT = 14, kTPXB = kSPPT * 16;            // t tile: 12 channel slots + 2 pad (conflict-free stride-1 reads)
constexpr int kTPB = kTH * kTW * kTPXB;
constexpr int kLDS = 2 * kXPB + kTPB + (kCmid + kCout) * 4;
constexpr int kWFrags = kKSA + kKSB;                     // A fragments per M block
static_assert(kLDS <= 160 * 1024, "LDS");
static_assert(kCmid == kCout, "one M block index per wave serves both convolutions");

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

__global__ __launch_bounds__(kNW * 64) void downblock_kernel(const DownParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_t = smem + 2 * kXPB;
    float* s_b = (float*)(s_t + kTPB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave % 6, pg = wave / 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
    const int cbase = mb * 16 + g * 4;                       // this lane's 4 channels (of both the intermediate and the output)

    // ---- once per workgroup: this wave's rows of both weight sets to registers, biases to LDS ----
    bf16x8 wa[kKSA], wb[kKSB];
    {
        const bf16x8* wsrc = (const bf16x8*)p.w + (size_t)mb * kWFrags * 64 + lane;
#pragma unroll
        for (int s = 0; s < kKSA; ++s) wa[s] = wsrc[s * 64];
#pragma unroll
        for (int s = 0; s < kKSB; ++s) wb[s] = wsrc[(kKSA + s) * 64];
    }
    for (int i = tid; i < kCmid + kCout; i += kNW * 64) s_b[i] = p.bias[i];
    // per-lane byte offset of this lane's K block relative to the tap-(0,0) pixel of an output pixel's 3x3 window
    int koffa[kKSA];
#pragma unroll
    for (int s = 0; s < kKSA; ++s) {
        int blk = 4 * s + g;
        if (blk >= kNBLKA) blk = 0;                          // zero weights: any initialised address
        const int tap = blk / kCBI, cb = blk - tap * kCBI;
        const int dy = tap / 3, dx = tap - 3 * dy;
        koffa[s] = (dy * kPW + dx) * kPXB + cb * 16;
    }

    const int tiles_per_img = p.tiles_y * p.tiles_x;
    struct PatchOrg { const char* org; int iy0, ix0; };     // address + image coordinates of patch pixel (0, 0)
    auto patch_org = [&](int tile) -> PatchOrg {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int iy0 = 2 * ty0 * kTH - 1, ix0 = 2 * tx0 * kTW - 1;
        return {p.in + ((long long)(b * H + iy0) * W + ix0) * p.in_ld_b, iy0, ix0};
    };
    auto dma_one = [&](const PatchOrg& o, int q, char* xb) {  // branch-free: issued from inside the MFMA loop
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                     // opaque: no per-call-site slot decode hoisted out of the tile loop
        const int slot = q * 64 + lane_o;
        const int px = slot / kSPPX, part = slot - px * kSPPX;
        const int pr = px / kPW, pc = px - pr * kPW;
        const int iy = o.iy0 + pr, ix = o.ix0 + pc;
        const bool valid = px < kPP && part < kCBI && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const char* src = o.org + (pr * W + pc) * p.in_ld_b + part * 16;
        glds16(valid ? src : p.zero, xb + q * 1024);
    };
    constexpr int NQW = (kNQ + kNW - 1) / kNW;               // DMA instructions per wave and tile (6)
    constexpr int SP = kKSA / NQW;                           // one every SP k-steps of the 3x3 loop
    static_assert(SP >= 1 && NQW * SP <= kKSA, "not enough k-steps to carry the DMA issue");

    int tile = first_tile(gridDim.x, blockIdx.x);
    if (tile < p.n_tiles) {
        const PatchOrg o = patch_org(tile);
#pragma unroll 1
        for (int q = wave; q < kNQ; q += kNW) dma_one(o, q, smem);
    }
    int cur = 0;
    bool prev_full = false;
    for (; tile < p.n_tiles; tile += gridDim.x, cur ^= 1) {
        const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
        const int ty0 = tr / p.tiles_x, tx0 = tr - ty0 * p.tiles_x;
        const int y0 = ty0 * kTH, x0 = tx0 * kTW;
        const char* s_x = smem + cur * kXPB;
        // this tile's patch has landed (own DMA: vmcnt; the other waves': barrier).  vmcnt is in-order: after a full tile the
        // youngest DMA instruction is older than that tile's 4 output stores, which may stay in flight.
        if (prev_full) wait_vmcnt<4>(); else wait_vmcnt<0>();
        down_lds_barrier();
        const bool has_next = tile + (int)gridDim.x < p.n_tiles;
        prev_full = y0 + kTH <= Ho && x0 + kTW <= Wo;
        char* xbn = smem + (cur ^ 1) * kXPB;
        PatchOrg on = {nullptr, 0, 0};
        if (has_next) on = patch_org(tile + (int)gridDim.x);

        const int ty = 4 * pg;                               // first of this wave's 4 output rows inside the tile
        // ---- 1. t = SiLU(Wa (*) x + ba): 4 blocks of 16 pixels (4 rows), one M block ----
        {
            const char* base = s_x + ((2 * ty) * kPW + 2 * l15) * kPXB;
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 fn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fn[j] = *(const bf16x8*)(base + (2 * j * kPW) * kPXB + koffa[0]);
#pragma unroll
            for (int s = 0; s < kKSA; ++s) {
                bf16x8 f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = fn[j];
                if (s + 1 < kKSA) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fn[j] = *(const bf16x8*)(base + (2 * j * kPW) * kPXB + koffa[s + 1]);
                }
                if (s % SP == 0 && s / SP < NQW) {           // one DMA instruction of the next tile's patch
                    const int q = wave + kNW * (s / SP);
                    if (has_next && q < kNQ) dma_one(on, q, xbn);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s], f[j], acc[j], 0, 0, 0);
            }
            const f32x4 bav = *(const f32x4*)(s_b + cbase);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = down_silu4(acc[j] + bav);
                down_lds_write_b64(s_t + ((ty + j) * kTW + l15) * kTPXB + cbase * 2, make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
            }
        }
        down_lds_barrier();
        // ---- 2. y = SiLU(Wb t + bb) ----
        {
            const char* base = s_t + (ty * kTW + l15) * kTPXB + g * 16;
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < kKSB; ++s) {
                bf16x8 f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = *(const bf16x8*)(base + j * kTW * kTPXB + s * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s], f[j], acc[j], 0, 0, 0);
            }
            const f32x4 bbv = *(const f32x4*)(s_b + kCmid + cbase);
            const int x = x0 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = y0 + ty + j;
                const f32x4 v = down_silu4(acc[j] + bbv);
                if (y < Ho && x < Wo)
                    *(uint2*)(p.out + ((long long)(b * Ho + y) * Wo + x) * p.out_ld_b + cbase * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
}

int g_down_cus = 0;

}  // namespace

// Packs the fused fp32 weights -- wa KRSC (96,3,3,48), wb KRSC (96,1,1,96) -- into the A-fragment image the kernel loads once per
// workgroup: [M block 6][k-step 14 + 3][lane 64] x 8 bf16; lane (m = lane & 15, g = lane >> 4) holds output channel 16 * Mblock + m
// and K block 4 * kstep + g (3x3: 8 consecutive input channels of one tap, tap-major; K blocks past 54 are zero).
extern "C" int aq_pack_downblock_weights(const float* wa_host, const float* wb_host, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(wa_host && wb_host && bytes, "pack_downblock: null pointer");
    *bytes = (size_t)6 * kWFrags * 64 * 16;
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_downblock: out of host memory");
    for (int mb = 0; mb < 6; ++mb)
        for (int s = 0; s < kWFrags; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = mb * 16 + (lane & 15), g = lane >> 4;
                bf16_t* dst = host + (((size_t)mb * kWFrags + s) * 64 + lane) * 8;
                if (s < kKSA) {
                    const int blk = 4 * s + g;
                    if (blk < kNBLKA) {
                        const int tap = blk / kCBI, c8 = blk % kCBI;
                        for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(wa_host[((size_t)co * 9 + tap) * kCin + c8 * 8 + e]);
                    }
                } else {
                    const int blk = 4 * (s - kKSA) + g;
                    for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(wb_host[(size_t)co * kCmid + blk * 8 + e]);
                }
            }
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

// in: bf16 NHWC [B][H][W][in_ld] with the 48 channels at in_choff; out: [B][H/2][W/2][out_ld] with the 96 channels at out_choff.
// bias_dev: [192] fp32 (ba | bb).
extern "C" int aq_downblock(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff,
                            const void* packed_w_dev, const float* bias_dev, int B, int H, int W, void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "downblock: null pointer");
    AQ_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "downblock: H and W must be even (got %dx%d)", H, W);
    AQ_REQUIRE(in_ld % 8 == 0 && out_ld % 8 == 0 && in_choff % 8 == 0 && out_choff % 8 == 0 && in_choff + kCin <= in_ld && out_choff + kCout <= out_ld,
               "downblock: channel slices must be 8-aligned and inside their rows");
    AQ_REQUIRE((long long)B * H * W < (1LL << 31), "downblock: batch too large");
    DownParams p{};
    p.in = (const char*)in_dev + (size_t)in_choff * 2; p.in_ld_b = in_ld * 2;
    p.out = (char*)out_dev + (size_t)out_choff * 2; p.out_ld_b = out_ld * 2;
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.B = B; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = (p.Wo + kTW - 1) / kTW; p.tiles_y = (p.Ho + kTH - 1) / kTH;
    AQ_REQUIRE((long long)B * p.tiles_x * p.tiles_y < (1LL << 30), "downblock: batch too large");
    p.n_tiles = B * p.tiles_x * p.tiles_y;
    static void* zero_page = nullptr;                        // allocated once per process
    static bool attr = false;
    if (!zero_page) {
        AQ_CHECK_HIP(hipMalloc(&zero_page, 256));
        AQ_CHECK_HIP(hipMemset(zero_page, 0, 256));
    }
    p.zero = (const char*)zero_page;
    if (g_down_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        g_down_cus = cus;
    }
    if (!attr) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)downblock_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS));
        attr = true;
    }
    long long grid = g_down_cus;                             // 153 KB of LDS: one persistent workgroup per CU
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(downblock_kernel, dim3((unsigned)grid), dim3(kNW * 64), kLDS, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
