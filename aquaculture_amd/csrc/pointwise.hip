// Memory-bound NHWC helpers: preprocess (u8 -> space-to-depth /255), SPPF max pools, nearest 2x upsample.
// All are HBM/L2-bound byte movers: one lane moves one 16-byte group, lanes of a wave cover consecutive
// groups of a pixel row so every access is coalesced.
#include "aq_common.h"

namespace {

// [UPSTREAM detect.py run()]: im = torch.from_numpy(im).to(device).float(); im /= 255
// fused with the 2x2 space-to-depth the stem conv (6x6/s2/p2 == 3x3/s1/p1 on s2d) reads:
// out[b][Y][X][(dy*2+dx)*3 + c] = in[b][2Y+dy][2X+dx][c] / 255, channels 12..15 = 0.
template <bool F32>
__global__ __launch_bounds__(256) void preprocess_s2d_kernel(const uint8_t* __restrict__ in, char* __restrict__ out,
                                                            int B, int H, int W) {
    const int H2 = H >> 1, W2 = W >> 1;
    const unsigned n = (unsigned)B * H2 * W2;   // < 2^31 (checked on the host): 32-bit index math, no 64-bit division
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned t = i / (unsigned)W2;
        const int X = (int)(i - t * W2);
        const int b = (int)(t / (unsigned)H2), Y = (int)(t - (unsigned)b * H2);
        float v[16];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const uint8_t* src = in + (((long long)b * H + 2 * Y + dy) * W + 2 * X) * 3;
#pragma unroll
            for (int e = 0; e < 6; ++e) v[dy * 6 + e] = (float)src[e] / 255.0f;
        }
        v[12] = v[13] = v[14] = v[15] = 0.0f;
        if (F32) {
            f32x4* o = (f32x4*)(out + (size_t)i * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) { f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; o[q] = x; }
        } else {
            uint4* o = (uint4*)(out + (size_t)i * 32);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint4 x;
                x.x = aq_f2bf(v[8 * q + 0]) | ((uint32_t)aq_f2bf(v[8 * q + 1]) << 16);
                x.y = aq_f2bf(v[8 * q + 2]) | ((uint32_t)aq_f2bf(v[8 * q + 3]) << 16);
                x.z = aq_f2bf(v[8 * q + 4]) | ((uint32_t)aq_f2bf(v[8 * q + 5]) << 16);
                x.w = aq_f2bf(v[8 * q + 6]) | ((uint32_t)aq_f2bf(v[8 * q + 7]) << 16);
                o[q] = x;
            }
        }
    }
}

__device__ __forceinline__ uint32_t max_bf16x2(uint32_t a, uint32_t b) {
    const float al = aq_bf2f((bf16_t)(a & 0xffff)), ah = aq_bf2f((bf16_t)(a >> 16));
    const float bl = aq_bf2f((bf16_t)(b & 0xffff)), bh = aq_bf2f((bf16_t)(b >> 16));
    const uint32_t lo = (bl > al) ? (b & 0xffff) : (a & 0xffff);
    const uint32_t hi = (bh > ah) ? (b >> 16) : (a >> 16);
    return lo | (hi << 16);
}

// nn.MaxPool2d(kernel_size=5, stride=1, padding=2): implicit -inf padding == clipped window.
// One lane = one 16-byte channel group of one output pixel.
template <bool F32>
__global__ __launch_bounds__(256) void maxpool5_kernel(const char* __restrict__ in, char* __restrict__ out,
                                                      int ld_b, int groups, int B, int H, int W) {
    const unsigned n = (unsigned)B * H * W * groups;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned t = i / (unsigned)groups;
        const int g = (int)(i - t * groups);
        unsigned t2 = t / (unsigned)W;
        const int x = (int)(t - t2 * W);
        const int b = (int)(t2 / (unsigned)H), y = (int)(t2 - (unsigned)b * H);
        const int y0 = max(y - 2, 0), y1 = min(y + 2, H - 1), x0 = max(x - 2, 0), x1 = min(x + 2, W - 1);
        uint4 m = *(const uint4*)(in + (((long long)b * H + y) * W + x) * ld_b + g * 16);
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
                const uint4 v = *(const uint4*)(in + (((long long)b * H + yy) * W + xx) * ld_b + g * 16);
                if (F32) {
                    m.x = __float_as_uint(fmaxf(__uint_as_float(m.x), __uint_as_float(v.x)));
                    m.y = __float_as_uint(fmaxf(__uint_as_float(m.y), __uint_as_float(v.y)));
                    m.z = __float_as_uint(fmaxf(__uint_as_float(m.z), __uint_as_float(v.z)));
                    m.w = __float_as_uint(fmaxf(__uint_as_float(m.w), __uint_as_float(v.w)));
                } else {
                    m.x = max_bf16x2(m.x, v.x); m.y = max_bf16x2(m.y, v.y);
                    m.z = max_bf16x2(m.z, v.z); m.w = max_bf16x2(m.w, v.w);
                }
            }
        *(uint4*)(out + (((long long)b * H + y) * W + x) * ld_b + g * 16) = m;
    }
}

// SPPF's three chained pools in ONE launch: a workgroup owns one (image, 16-byte channel group) plane, keeps it in
// LDS (two ping-pong planes) and writes y1, y2, y3 to their channel slices; x is read from HBM once.
// bf16 <-> order-preserving int16 key (an involution): flip the magnitude bits of negative values, then bf16 order ==
// signed 16-bit integer order and a window maximum is one v_pk_max_i16 per two values.
typedef __attribute__((ext_vector_type(2))) short short2_t;
__device__ __forceinline__ uint32_t bf16x2_key(uint32_t x) {
    short2_t v; __builtin_memcpy(&v, &x, 4);
    const short2_t m = (v >> 15) & (short)0x7fff;
    v = v ^ m;
    uint32_t r; __builtin_memcpy(&r, &v, 4);
    return r;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    short2_t x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const short2_t m = __builtin_elementwise_max(x, y);
    uint32_t r; __builtin_memcpy(&r, &m, 4);
    return r;
}
__device__ __forceinline__ uint4 key16(uint4 v) { return make_uint4(bf16x2_key(v.x), bf16x2_key(v.y), bf16x2_key(v.z), bf16x2_key(v.w)); }

template <bool F32>
__device__ __forceinline__ uint4 max16(uint4 a, uint4 v) {
    if (F32) {
        a.x = __float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(v.x)));
        a.y = __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(v.y)));
        a.z = __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(v.z)));
        a.w = __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(v.w)));
    } else {
        a.x = max_bf16x2(a.x, v.x); a.y = max_bf16x2(a.y, v.y); a.z = max_bf16x2(a.z, v.z); a.w = max_bf16x2(a.w, v.w);
    }
    return a;
}

template <bool F32, int GPB>   // GPB = adjacent 16-byte channel groups per workgroup (GPB * 16 B contiguous per pixel)
__global__ __launch_bounds__(256) void sppf_pool3_kernel(char* __restrict__ base, int ld_b, int slice_b, int groups, int H, int W) {
    // Three chained 5x5 / stride 1 / pad 2 max pools of one image plane, kept in LDS.  Each pool is done separably -- a horizontal 5-tap
    // pass into a temporary plane, then a vertical 5-tap pass (max is exact, so the result is the 5x5 window maximum bit for bit) --
    // 10 LDS reads per output instead of 25: the kernel is bound by LDS bandwidth, not by HBM.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hw = H * W, n = hw * GPB;
    uint4* const src = (uint4*)smem;       // stage input, overwritten by the stage output (its rows are dead once tmp is complete)
    uint4* const tmp = (uint4*)smem + n;   // row maxima
    const int gblocks = groups / GPB;
    const int b = blockIdx.x / gblocks, g0 = (blockIdx.x - b * gblocks) * GPB;
    char* img = base + (long long)b * hw * ld_b + g0 * 16;
    auto mx = [](uint4 m, uint4 v) -> uint4 {
        if (F32) return max16<true>(m, v);
        m.x = pk_max_i16(m.x, v.x); m.y = pk_max_i16(m.y, v.y); m.z = pk_max_i16(m.z, v.z); m.w = pk_max_i16(m.w, v.w);
        return m;
    };
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int pix = i / GPB, g = i - pix * GPB;
        const uint4 v = *(const uint4*)(img + (long long)pix * ld_b + g * 16);
        src[i] = F32 ? v : key16(v);                       // bf16: keep the planes as order-preserving int16 keys
    }
    __syncthreads();
    for (int s = 0; s < 3; ++s) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int pix = i / GPB, g = i - pix * GPB;
            const int y = pix / W, x = pix - y * W;
            const int x0 = max(x - 2, 0), x1 = min(x + 2, W - 1);
            uint4 m = src[i];
            for (int xx = x0; xx <= x1; ++xx) m = mx(m, src[(y * W + xx) * GPB + g]);
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int pix = i / GPB, g = i - pix * GPB;
            const int y = pix / W, x = pix - y * W;
            const int y0 = max(y - 2, 0), y1 = min(y + 2, H - 1);
            uint4 m = tmp[i];
            for (int yy = y0; yy <= y1; ++yy) m = mx(m, tmp[(yy * W + x) * GPB + g]);
            src[i] = m;
            *(uint4*)(img + (long long)pix * ld_b + (long long)(s + 1) * slice_b + g * 16) = F32 ? m : key16(m);
        }
        __syncthreads();
    }
}

// nn.Upsample(scale_factor=2, mode='nearest'): out[b][y][x] = in[b][y/2][x/2]; H, W are the INPUT size.
// One lane per INPUT 16-byte group: read once, write the 2 x 2 output pixels.  blockIdx.y = input row (b * H + y); one integer
// division per lane (x from the flat (x, group) index).  (A variant whose blocks loop over rows was not faster.)
__global__ __launch_bounds__(256) void upsample2x_kernel(const char* __restrict__ in, int in_ld_b,
                                                        char* __restrict__ out, int out_ld_b,
                                                        int groups, int H, int W) {
    const unsigned xg = blockIdx.x * 256u + threadIdx.x;
    if (xg >= (unsigned)(W * groups)) return;
    const unsigned x = xg / (unsigned)groups, g = xg - x * (unsigned)groups;
    const unsigned row = blockIdx.y;                         // b * H + y; the output rows are 2 * row and 2 * row + 1
    const uint4 v = *(const uint4*)(in + ((long long)row * W + x) * in_ld_b + g * 16);
    char* o = out + ((long long)2 * row * (2 * W) + 2 * x) * out_ld_b + g * 16;
    const long long orow = (long long)2 * W * out_ld_b;
    *(uint4*)o = v;
    *(uint4*)(o + out_ld_b) = v;
    *(uint4*)(o + orow) = v;
    *(uint4*)(o + orow + out_ld_b) = v;
}

// [UPSTREAM utils/augmentations.py letterbox -> cv2.resize(INTER_LINEAR) + copyMakeBorder(114)] on uint8 RGB tiles.
// OpenCV's 8-bit bilinear kernel in fixed point (11-bit coefficients): horizontal pass in int (scale 2^11), vertical pass
// ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  The coefficient tables (source index pair + weights per
// destination column / row) are built on the host exactly as resize.cpp does (float32 fractions, cvRound) and passed in.
// Tile b starts at src + tile_off[b] (tile_off == nullptr: b * H0 * row_b, tiles stored one after the other) and its rows are row_b
// bytes apart: with offsets into one big scene raster the crop of reference src/load_data/tile_tifs.py:33-47 costs no pass of its own.
__global__ __launch_bounds__(256) void letterbox_u8_kernel(const uint8_t* __restrict__ src, long long row_b, const long long* __restrict__ tile_off,
                                                          uint8_t* __restrict__ dst, const int4* __restrict__ xtab, const int4* __restrict__ ytab,
                                                          int B, int H0, int W0, int H, int W, int new_w, int new_h, int top, int left) {
    const unsigned n = (unsigned)B * H * W;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned t = i / (unsigned)W;
        const int x = (int)(i - t * W);
        const int b = (int)(t / (unsigned)H), y = (int)(t - (unsigned)b * H);
        uint8_t* o = dst + (size_t)i * 3;
        const int yy = y - top, xx = x - left;
        if (yy < 0 || yy >= new_h || xx < 0 || xx >= new_w) { o[0] = o[1] = o[2] = 114; continue; }
        const int4 xt = xtab[xx], yt = ytab[yy];        // (i0, i1, w0, w1)
        const uint8_t* tile = src + (tile_off ? tile_off[b] : (long long)b * H0 * row_b);
        const uint8_t* r0 = tile + yt.x * row_b;
        const uint8_t* r1 = tile + yt.y * row_b;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[xt.x * 3 + c] * xt.z + r0[xt.y * 3 + c] * xt.w;
            const int h1 = r1[xt.x * 3 + c] * xt.z + r1[xt.y * 3 + c] * xt.w;
            int v = (((yt.z * (h0 >> 4)) >> 16) + ((yt.w * (h1 >> 4)) >> 16) + 2) >> 2;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

inline unsigned grid_for(long long n, int block) {
    long long g = (n + block - 1) / block;
    const long long cap = 256 * 16;   // 256 CUs x 16 blocks, grid-stride the rest
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int aq_preprocess_s2d(const uint8_t* tiles_dev, void* out_dev, int B, int H, int W, int precision, void* stream) {
    AQ_REQUIRE(tiles_dev && out_dev, "preprocess: null pointer");
    AQ_REQUIRE(B > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "preprocess: bad shape B=%d H=%d W=%d", B, H, W);
    const long long n = (long long)B * (H / 2) * (W / 2);
    AQ_REQUIRE(n < (1LL << 31), "preprocess: batch too large");
    if (precision == AQ_FP32)
        hipLaunchKernelGGL(preprocess_s2d_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, tiles_dev, (char*)out_dev, B, H, W);
    else
        hipLaunchKernelGGL(preprocess_s2d_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, tiles_dev, (char*)out_dev, B, H, W);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" int aq_sppf_pool(void* buf_dev, int ld, int ch_off, int c, int B, int H, int W, int precision, void* stream) {
    const int eb = aq_elem_bytes(precision);
    AQ_REQUIRE(buf_dev, "sppf_pool: null pointer");
    AQ_REQUIRE((c * eb) % 16 == 0 && (ch_off * eb) % 16 == 0 && (ld * eb) % 16 == 0, "sppf_pool: channels must be 16-byte groups");
    AQ_REQUIRE(ch_off + 4 * c <= ld, "sppf_pool: slices [x|y1|y2|y3] exceed the buffer width");
    const int groups = c * eb / 16;
    const long long n = (long long)B * H * W * groups;
    AQ_REQUIRE(n < (1LL << 31), "sppf_pool: batch too large");
    char* base = (char*)buf_dev + (long long)ch_off * eb;
    constexpr int GPB = 4;
    const size_t plane_lds = (size_t)2 * H * W * 16 * GPB;
    if (plane_lds <= 64 * 1024 && groups % GPB == 0 && (long long)B * groups < (1LL << 31)) {   // plane fits LDS: one fused launch
        const unsigned grid = (unsigned)(B * (groups / GPB));
        if (precision == AQ_FP32)
            hipLaunchKernelGGL((sppf_pool3_kernel<true, GPB>), dim3(grid), dim3(256), plane_lds, (hipStream_t)stream, base, ld * eb, c * eb, groups, H, W);
        else
            hipLaunchKernelGGL((sppf_pool3_kernel<false, GPB>), dim3(grid), dim3(256), plane_lds, (hipStream_t)stream, base, ld * eb, c * eb, groups, H, W);
        AQ_CHECK_HIP(hipGetLastError());
        return AQ_OK;
    }
    for (int s = 0; s < 3; ++s) {   // y1 = m(x), y2 = m(y1), y3 = m(y2)  [UPSTREAM SPPF.forward]
        const char* in = base + (long long)s * c * eb;
        char* out = base + (long long)(s + 1) * c * eb;
        if (precision == AQ_FP32)
            hipLaunchKernelGGL(maxpool5_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, ld * eb, groups, B, H, W);
        else
            hipLaunchKernelGGL(maxpool5_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, ld * eb, groups, B, H, W);
    }
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" int aq_upsample2x(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff,
                             int c, int B, int H, int W, int precision, void* stream) {
    const int eb = aq_elem_bytes(precision);
    AQ_REQUIRE(in_dev && out_dev, "upsample2x: null pointer");
    AQ_REQUIRE((c * eb) % 16 == 0 && (in_choff * eb) % 16 == 0 && (out_choff * eb) % 16 == 0 &&
               (in_ld * eb) % 16 == 0 && (out_ld * eb) % 16 == 0, "upsample2x: channels must be 16-byte groups");
    const int groups = c * eb / 16;
    const long long n = (long long)B * 4 * H * W * groups;
    AQ_REQUIRE(n < (1LL << 31), "upsample2x: batch too large");
    AQ_REQUIRE((long long)B * H < 65536, "upsample2x: more than 65535 input rows");
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((W * groups + 255) / 256), (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)in_dev + (long long)in_choff * eb, in_ld * eb,
                       (char*)out_dev + (long long)out_choff * eb, out_ld * eb, groups, H, W);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" int aq_letterbox_u8(const uint8_t* src_dev, int B, int H0, int W0, uint8_t* dst_dev, int H, int W, int new_w, int new_h,
                               int top, int left, const int32_t* xtab_dev, const int32_t* ytab_dev, void* stream) {
    AQ_REQUIRE(src_dev && dst_dev && xtab_dev && ytab_dev, "letterbox: null pointer");
    AQ_REQUIRE(B > 0 && H0 > 0 && W0 > 0 && H > 0 && W > 0 && new_w > 0 && new_h > 0 && top >= 0 && left >= 0 &&
               top + new_h <= H && left + new_w <= W, "letterbox: bad geometry");
    const long long n = (long long)B * H * W;
    AQ_REQUIRE(n < (1LL << 31), "letterbox: batch too large");
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, src_dev, (long long)W0 * 3, nullptr, dst_dev,
                       (const int4*)xtab_dev, (const int4*)ytab_dev, B, H0, W0, H, W, new_w, new_h, top, left);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" int aq_letterbox_tiles_u8(const uint8_t* scene_dev, long long scene_bytes, long long row_bytes, const long long* tile_off_dev,
                                     const long long* tile_off_host, int B, int H0, int W0, uint8_t* dst_dev, int H, int W, int new_w, int new_h,
                                     int top, int left, const int32_t* xtab_dev, const int32_t* ytab_dev, void* stream) {
    AQ_REQUIRE(scene_dev && tile_off_dev && tile_off_host && dst_dev && xtab_dev && ytab_dev, "letterbox_tiles: null pointer");
    AQ_REQUIRE(B > 0 && H0 > 0 && W0 > 0 && H > 0 && W > 0 && new_w > 0 && new_h > 0 && top >= 0 && left >= 0 &&
               top + new_h <= H && left + new_w <= W && row_bytes >= (long long)W0 * 3, "letterbox_tiles: bad geometry");
    for (int b = 0; b < B; ++b)      // every tile must lie inside the raster: the kernel trusts the offsets
        AQ_REQUIRE(tile_off_host[b] >= 0 && tile_off_host[b] + (long long)(H0 - 1) * row_bytes + (long long)W0 * 3 <= scene_bytes &&
                       tile_off_host[b] % row_bytes + (long long)W0 * 3 <= row_bytes,     /* no wrap into the next raster row */
                   "letterbox_tiles: tile %d (offset %lld) leaves the raster of %lld bytes", b, tile_off_host[b], scene_bytes);
    const long long n = (long long)B * H * W;
    AQ_REQUIRE(n < (1LL << 31), "letterbox_tiles: batch too large");
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, scene_dev, row_bytes, tile_off_dev, dst_dev,
                       (const int4*)xtab_dev, (const int4*)ytab_dev, B, H0, W0, H, W, new_w, new_h, top, left);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

// Diagnostics: register-only bf16 MFMA loop (no memory traffic) to read the chip's sustained matrix rate under load
// (tools/mfma_peak.py).  blocks x 256 threads, each wave issues iters x 8 independent v_mfma_f32_32x32x16_bf16.
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters, unsigned seed) {
    f32x16 acc[8];
    bf16x8 a, b;
    const unsigned t = threadIdx.x * 2654435761u + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((t >> i) & 0x3f)); b[i] = (short)(0xbf80 + ((t >> (i + 3)) & 0x3f)); }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k][threadIdx.x & 15];
    if (s == 123.456f) out[0] = s;   // keep the loop alive without a store in practice
}

// the same with v_mfma_f32_16x16x32_bf16 (the instruction of the fused / direct kernels): 16 independent accumulators per wave
__global__ __launch_bounds__(256) void mfma_peak16_kernel(float* out, int iters, unsigned seed) {
    f32x4 acc[16];
    bf16x8 a, b;
    const unsigned t = threadIdx.x * 2654435761u + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((t >> i) & 0x3f)); b[i] = (short)(0xbf80 + ((t >> (i + 3)) & 0x3f)); }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k][threadIdx.x & 3];
    if (s == 123.456f) out[0] = s;
}

// max |x| over a channel slice of an NHWC bf16 tensor -> *out_dev (float, the caller zeroes it first): calibration of the fp8 path's
// per-tensor activation scales (aq_engine_calibrate_amax).  |x| as an unsigned integer orders like the float, so atomicMax on the bits.
namespace {
__global__ void absmax_bf16_kernel(const unsigned short* x, int ld, int c8, long long ngroups, unsigned* out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups; i += (long long)gridDim.x * blockDim.x) {
        const long long px = i / c8;
        const int g = (int)(i - px * c8);
        const uint4 v = *(const uint4*)(x + px * ld + g * 8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = (w[e] << 16) & 0x7fffffffu, hi = w[e] & 0x7fff0000u;
            m = lo > m ? lo : m;
            m = hi > m ? hi : m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
}  // namespace

extern "C" int aq_absmax_bf16(const void* t_dev, int ld, int choff, int c, long long npix, float* out_dev, void* stream) {
    AQ_REQUIRE(t_dev && out_dev && c > 0 && c % 8 == 0 && ld % 8 == 0 && choff % 8 == 0 && choff + c <= ld && npix > 0, "absmax_bf16: bad slice");
    const long long ngroups = npix * (c / 8);
    long long grid = (ngroups + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)t_dev + choff, ld, c / 8, ngroups, (unsigned*)out_dev);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" int aq_debug_mfma_peak(int blocks, int iters, void* out_dev, void* stream) {
    AQ_REQUIRE(blocks > 0 && iters != 0 && out_dev, "mfma_peak: bad argument");
    if (iters < 0) {       // negative count: the 16x16x32 form, 16 MFMAs (the same FLOPs as 8 of the 32x32x16 form) per iteration
        hipLaunchKernelGGL(mfma_peak16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)out_dev, -iters, 12345u);
        AQ_CHECK_HIP(hipGetLastError());
        return AQ_OK;
    }
    hipLaunchKernelGGL(mfma_peak_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)out_dev, iters, 12345u);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
