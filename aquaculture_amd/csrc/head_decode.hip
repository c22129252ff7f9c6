// Detect head + decode in one kernel, gfx950, bf16 engines, the `infer` (NMS) path:
//     raw = W_l (1x1, K -> na * (nc + 5)) x + b_l ;  obj = sigmoid(raw[4]) ;  rows with obj > conf_thres are decoded and appended to the
//     image's compact candidate list
// = one level of [UPSTREAM models/yolo.py Detect.forward] followed by the candidate filter of [UPSTREAM utils/general.py
// non_max_suppression] (`xc = prediction[..., 4] > conf_thres`), reached through reference README.md:77 -> yolov5/detect.py.
//
// As three implicit-GEMM launches + aq_detect_decode these cost 94 + 31 us per 64-tile batch: with 30 output channels the GEMM tile is
// almost empty (70-92 TFLOP/s) and the work is reading the three feature maps once (264 MB), then the fp32 head maps go to HBM and come
// back (2 x 64 MB).  Here the GEMM is shaped for what it is -- a skinny matrix product bound by its B operand:
//   * the weights (32 rows x K, two 16-row MFMA blocks) live in REGISTERS for the life of the wave;
//   * the activations never touch LDS: lane (pixel = lane & 15, k quarter = lane >> 4) of v_mfma_f32_16x16x32_bf16's B operand holds 8
//     consecutive channels of its pixel, which is one 16-byte load straight from the NHWC row; the K / 32 loads of a 16-pixel block are
//     issued together (each byte of the feature map is read exactly once, by exactly one lane);
//   * the 16 x 32 fp32 results go through a wave-private LDS tile so that lane (pixel, anchor = lane >> 4) owns one candidate: same
//     arithmetic, in the same order, as decode_kernel (csrc/detect_nms.hip; this file is compiled with -ffp-contract=off like it).
// The head maps are not written: `forward_raw` (the full prediction tensor) keeps the conv + decode path.
#include "conv_device.h"
#include <vector>

using namespace aqdev;

namespace {

struct HeadDecParams {
    const char* in;          // first channel of the level's feature-map slice (bf16 NHWC)
    int in_ld_b;             // bytes per pixel row of that tensor
    const char* w;           // aq_pack_head_weights image: A fragments [2][K / 32][64] x 16 B, then 32 floats of bias
    int npix, ny, nx;        // B * ny * nx
    int off;                 // candidate index of this level's first candidate within an image
    int N;                   // candidates per image over all levels (unused here, kept for symmetry with decode)
    int nc, na, no;
    float stride, anchor[8][2];
    float conf_thres;
    int32_t* cand;           // [B][cap]
    float* cand_rows;        // [B][cap][no]
    int32_t* cand_count;     // image b's counter at cand_count[b * count_stride]
    int count_stride;        // the engine keeps these counters 4 KB apart: 64 adjacent ints share an L2 channel, whose atomic unit then
                             // serialises every workgroup of the launch (measured: 73 us for the 80 x 80 level against 49.5 spread out)
    int cap;
};

__global__ void head_counts_gather_kernel(const int32_t* wide, int stride, int32_t* compact, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) compact[b] = wide[(size_t)b * stride];
}

// same sequence as decode_kernel's (detect_nms.hip): IEEE divide + expf
__device__ __forceinline__ float head_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// KSW = k-steps (of 32 channels) per wave, KSPLIT = waves that share a 16-pixel block, each multiplying its own K range (K = 32 KSW
// KSPLIT): the wide levels (K = 384, 768) would otherwise hold 100-200 weight registers per wave and run one or two waves per SIMD with
// a handful of blocks each; split this way every level has the registers (and the loads in flight) of the K = 192 one.
template <int KSW, int KSPLIT, int OCC = 4>
__global__ __launch_bounds__(256, OCC) void head_decode_kernel(const HeadDecParams p) {
    constexpr int KS = KSW * KSPLIT, NG = 4 / KSPLIT, PXI = 16 * NG;      // pixel groups and pixels per workgroup iteration
    __shared__ __attribute__((aligned(16))) float s_tile[4][16][32];
    __shared__ int s_wcnt[4], s_base;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / KSPLIT, part = wave % KSPLIT;
    const int px = lane & 15, q = lane >> 4;
    bf16x8 a[2][KSW];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) a[mb][ks] = ((const bf16x8*)p.w)[(mb * KS + part * KSW + ks) * 64 + lane];
    const float* bias = (const float*)(p.w + (size_t)2 * KS * 1024);
    f32x4 bias0 = *(const f32x4*)(bias + 4 * q), bias1 = *(const f32x4*)(bias + 16 + 4 * q);
    if (part != 0) { bias0 = f32x4{0.f, 0.f, 0.f, 0.f}; bias1 = bias0; }
    const int hw = p.ny * p.nx;
    // A workgroup iteration = PXI consecutive pixels.  Its passing candidates take ONE global atomic (same-address atomics serialise:
    // with one per candidate the 80 x 80 level took 239 us, 4.5 x the unfused conv; with one per 64 pixels but the 64 images' counters
    // in one 256-byte line, 73 us -- hence count_stride); an iteration that straddles two images -- pixel counts that are not multiples
    // of PXI -- falls back to one atomic per candidate.
    const int nit = (p.npix + PXI - 1) / PXI;
    for (int it = blockIdx.x; it < nit; it += gridDim.x) {
        const int P = it * PXI + grp * 16 + px;
        const int Pc = P < p.npix ? P : p.npix - 1;
        const char* src = p.in + (size_t)Pc * p.in_ld_b + (size_t)part * KSW * 64 + q * 16;
        bf16x8 b[KSW];                                       // (prefetching the next block's fragments under this one costs 24 registers:
#pragma unroll                                               //  spills at four workgroups per CU, slower at three -- measured 60.9 vs 48.0 us)
        for (int ks = 0; ks < KSW; ++ks) b[ks] = *(const bf16x8*)(src + ks * 64);
        f32x4 acc0 = bias0, acc1 = bias1;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ks], b[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][ks], b[ks], acc1, 0, 0, 0);
        }
        // D layout: lane holds output channels 16 mb + 4 q .. + 3 of pixel px (its wave's share of K)
        *(f32x4*)&s_tile[wave][px][4 * q] = acc0;
        *(f32x4*)&s_tile[wave][px][16 + 4 * q] = acc1;
        if constexpr (KSPLIT > 1) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // lane (pixel px, anchor q) of the group's first wave owns one candidate; the K parts are summed in a fixed order
        const bool mine = part == 0 && q < p.na && P < p.npix;
        float raw[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) raw[j] = 0.f;
        if (mine) {
            for (int j = 0; j < p.no; ++j) {
                float v = s_tile[grp * KSPLIT][px][q * p.no + j];
#pragma unroll
                for (int s2 = 1; s2 < KSPLIT; ++s2) v += s_tile[grp * KSPLIT + s2][px][q * p.no + j];
                raw[j] = v;
            }
        }
        float obj = 0.f;
        bool pass = false;
        if (mine) {
            obj = head_sigmoid(raw[4]);
            pass = obj > p.conf_thres;
        }
        const int first = it * PXI, last = first + PXI - 1 < p.npix ? first + PXI - 1 : p.npix - 1;
        const int b_first = first / hw;
        const bool one_image = b_first == last / hw;            // uniform over the workgroup
        const unsigned long long votes = __ballot(pass);
        int pos = -1;
        const int bimg = one_image ? b_first : Pc / hw;
        if (one_image) {
            if (lane == 0) s_wcnt[wave] = __popcll(votes);
            __syncthreads();
            const int c0 = s_wcnt[0], c1 = s_wcnt[1], c2 = s_wcnt[2], c3 = s_wcnt[3];
            if (threadIdx.x == 0 && c0 + c1 + c2 + c3 > 0) s_base = atomicAdd(p.cand_count + (size_t)bimg * p.count_stride, c0 + c1 + c2 + c3);
            __syncthreads();
            const int wbase = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
            if (pass) pos = s_base + wbase + __popcll(votes & ((1ull << lane) - 1));
        } else {
            if (pass) pos = atomicAdd(p.cand_count + (size_t)bimg * p.count_stride, 1);
            if constexpr (KSPLIT > 1) __syncthreads();           // the tiles are rewritten in the next iteration
        }
        if (pass && pos < p.cap) {
            const int pix = Pc - bimg * hw;
            const int y = pix / p.nx, x = pix - y * p.nx;
            p.cand[(long long)bimg * p.cap + pos] = p.off + q * hw + pix;      // upstream candidate index: a * ny * nx + y * nx + x
            float* dst = p.cand_rows + ((long long)bimg * p.cap + pos) * p.no;
            // xy = (xy * 2 + grid) * stride, grid = index - 0.5 ; wh = (wh * 2) ** 2 * anchor_grid   (decode_row, detect_nms.hip)
            const float s0 = head_sigmoid(raw[0]), s1 = head_sigmoid(raw[1]);
            const float s2 = head_sigmoid(raw[2]), s3 = head_sigmoid(raw[3]);
            const float gx = (float)x - 0.5f, gy = (float)y - 0.5f;
            dst[0] = (s0 * 2.0f + gx) * p.stride;
            dst[1] = (s1 * 2.0f + gy) * p.stride;
            const float tw = s2 * 2.0f, th = s3 * 2.0f;
            dst[2] = (tw * tw) * p.anchor[q][0];
            dst[3] = (th * th) * p.anchor[q][1];
            dst[4] = obj;
            for (int c = 0; c < p.nc && c < 11; ++c) dst[5 + c] = head_sigmoid(raw[5 + c]);
        }
        __builtin_amdgcn_wave_barrier();                 // (KSPLIT = 1) the tile is rewritten by this wave's next block
    }
}

struct HeadKernel { int ks, split; void (*fn)(const HeadDecParams); };
const HeadKernel kHead[] = {
    {4, 1, head_decode_kernel<4, 1>},  {6, 1, head_decode_kernel<6, 1>},  {8, 1, head_decode_kernel<8, 1>},
    {10, 2, head_decode_kernel<5, 2>}, {12, 2, head_decode_kernel<6, 2>}, {16, 2, head_decode_kernel<8, 2>},
    {20, 4, head_decode_kernel<5, 4>}, {24, 4, head_decode_kernel<6, 4>}, {32, 4, head_decode_kernel<8, 4>},
    {40, 4, head_decode_kernel<10, 4, 2>},
};
int g_head_cus[64];

}  // namespace

extern "C" int aq_head_decode_supported(int cin, int na, int nc) {
    if (cin <= 0 || cin % 32 || na < 1 || na > 4 || nc < 1 || nc > 11 || na * (nc + 5) > 32) return 0;
    for (const HeadKernel& k : kHead)
        if (k.ks == cin / 32) return 1;
    return 0;
}

// w: fp32 [cout][cin] (the 1x1 conv's KRSC weights), cout = na * (nc + 5) <= 32.  Image: A fragments [M block 0..1][k-step][lane] x 8
// bf16 (lane: row = lane & 15, input channels 32 ks + 8 (lane >> 4) .. + 7; rows >= cout are zero), then bias[32] (zero padded).
extern "C" int aq_pack_head_weights(const float* w_host, const float* bias_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_host && bytes && cin > 0 && cin % 32 == 0 && cout > 0 && cout <= 32, "pack_head_weights: unsupported %d -> %d", cin, cout);
    const int KS = cin / 32;
    *bytes = (size_t)2 * KS * 1024 + 32 * sizeof(float);
    if (!packed_dev) return AQ_OK;
    AQ_REQUIRE(bias_host, "pack_head_weights: null bias");
    std::vector<unsigned char> host(*bytes, 0);
    bf16_t* dst = (bf16_t*)host.data();
    for (int mb = 0; mb < 2; ++mb)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = 16 * mb + (lane & 15), ci = 32 * ks + 8 * (lane >> 4) + e;
                    *dst++ = co < cout ? aq_f2bf(w_host[(size_t)co * cin + ci]) : (bf16_t)0;
                }
    float* b = (float*)(host.data() + (size_t)2 * KS * 1024);
    for (int c = 0; c < cout; ++c) b[c] = bias_host[c];
    AQ_CHECK_HIP(hipMemcpyAsync(packed_dev, host.data(), host.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
    AQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return AQ_OK;
}

// One Detect level: in = bf16 NHWC feature map slice (`cin` channels from in_choff of rows of in_ld elements), ny x nx pixels per image.
// Appends to cand / cand_rows exactly as aq_detect_decode does; image b's counter is cand_count_dev[b * count_stride] (zeroed by the
// caller before the first level; aq_head_counts_gather copies strided counters into the compact array aq_nms reads).
extern "C" int aq_head_decode(const void* in_dev, int in_ld, int in_choff, int cin, const void* packed_dev, int B, int ny, int nx,
                              int cand_off, float stride, const float* anchors_px, int nc, int na, float conf_thres,
                              int32_t* cand_dev, float* cand_rows_dev, int32_t* cand_count_dev, int count_stride, int cand_cap, void* stream) {
    AQ_REQUIRE(count_stride >= 1, "head_decode: counter stride");
    AQ_REQUIRE(in_dev && packed_dev && cand_dev && cand_rows_dev && cand_count_dev && anchors_px, "head_decode: null pointer");
    AQ_REQUIRE(aq_head_decode_supported(cin, na, nc), "head_decode: unsupported cin=%d na=%d nc=%d", cin, na, nc);
    AQ_REQUIRE(B > 0 && ny > 0 && nx > 0 && (long long)B * ny * nx < (1LL << 30) && in_ld % 8 == 0 && in_choff % 8 == 0 && in_choff + cin <= in_ld && cand_cap > 0,
               "head_decode: bad geometry");
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "head_decode: device ordinal %d", dev);
    if (g_head_cus[dev] == 0) {
        int cus = 256;
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_head_cus[dev] = cus;
    }
    HeadDecParams p{};
    p.in = (const char*)in_dev + (size_t)in_choff * 2; p.in_ld_b = in_ld * 2;
    p.w = (const char*)packed_dev;
    p.npix = B * ny * nx; p.ny = ny; p.nx = nx; p.off = cand_off; p.nc = nc; p.na = na; p.no = nc + 5;
    p.stride = stride;
    for (int a = 0; a < na; ++a) { p.anchor[a][0] = anchors_px[2 * a]; p.anchor[a][1] = anchors_px[2 * a + 1]; }
    p.conf_thres = conf_thres; p.cand = cand_dev; p.cand_rows = cand_rows_dev; p.cand_count = cand_count_dev; p.count_stride = count_stride; p.cap = cand_cap;
    const HeadKernel* k = nullptr;
    for (const HeadKernel& c : kHead)
        if (c.ks == cin / 32) k = &c;
    const int pxi = 64 / k->split;
    const int nit = (p.npix + pxi - 1) / pxi;
    long long grid = (long long)g_head_cus[dev] * 4;                      // persistent (four workgroups per CU): the weights are loaded once per wave
    if (grid > nit) grid = nit;
    hipLaunchKernelGGL(k->fn, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}


extern "C" int aq_head_counts_gather(const int32_t* wide_dev, int count_stride, int32_t* compact_dev, int B, void* stream) {
    AQ_REQUIRE(wide_dev && compact_dev && B > 0 && count_stride >= 1, "head_counts_gather: bad arguments");
    hipLaunchKernelGGL(head_counts_gather_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, wide_dev, count_stride, compact_dev, B);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
