// Shared declarations for the HIP translation units of libaqengine.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../../include/aq_engine.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

void aq_set_error(const char* fmt, ...);
#define AQ_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            aq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AQ_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)
#define AQ_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            aq_set_error(__VA_ARGS__);                                                  \
            return AQ_ERR_INVALID;                                                      \
        }                                                                               \
    } while (0)

static inline int aq_elem_bytes(int precision) { return (precision == AQ_FP32 || precision == AQ_F16X3) ? 4 : 2; }

// f32 -> bf16 round-to-nearest-even (finite inputs; NaN stays NaN via the quiet bit)
__host__ __device__ static inline bf16_t aq_f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);
    return (bf16_t)((v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16);
}
__host__ __device__ static inline float aq_bf2f(bf16_t h) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)h) << 16; return v.f;
}

// ---- conv kernel parameter block (conv_igemm.hip) ----
struct ConvParams {
    const char* in;      // input tensor base + first-channel offset (bytes)
    char* out;
    const char* res;     // nullptr: none
    const char* w;       // packed [cout_pad][kgroups_pad] x 16 B
    const float* bias;   // [cout_pad]
    const char* zero;    // >= 16 zero bytes
    int in_ld_b, out_ld_b, res_ld_b;   // pixel strides in BYTES
    int B, H, W, Ho, Wo;
    int cout;            // real cout (multiple of 8)
    int k, stride, pad, taps;
    int G;               // 16-byte groups per tap = cin * sizeof(T) / 16
    int kgroups;         // taps * G
    int kgroups_pad;     // multiple of 8
    int nchunks;         // kgroups_pad / 8
    int npix;            // B * Ho * Wo
    int act;
    int n_tiles_m, n_tiles_n;
    int bias_n;          // bias entries staged into LDS (filled by aq_launch_conv)
    int x3_off;          // AQ_F16X3: the per-channel 2^-s follow the (scaled) bias at this float offset of `bias`
    unsigned long long* debug;   // diagnostic (stamped) builds only: per-wave phase cycle sums
    int halo, xrows, nixr, xper;   // conv_halo.hip: W + 1, region rows in LDS, region rows / 8, loads per step part
    float inv_hw, inv_wo;        // reciprocals for division-free pixel decode (filled by aq_launch_conv)
    unsigned magic_G, magic_k, magic_ntm;   // floor(2^32 / d) + 1
};

int aq_launch_conv(const ConvParams& p, int precision, int out_f32, int cfg, hipStream_t stream);
int aq_conv_pick_config(int cout, int npix, int precision);
int aq_launch_conv_halo(const ConvParams& p, int precision, int out_f32, int hcfg, bool one_tile_per_wg, hipStream_t stream);
int aq_conv_halo_num_configs();
unsigned long long* aq_stamp_buffer(size_t* bytes);
// 256 zero bytes on the CURRENT device (allocated on first use, one per device, never freed): the LDS-DMA source for pixels
// outside the image in the standalone kernel entry points (the engine passes its own zero page to the conv kernels).
const char* aq_zero_page();
// Compute units the persistent grids are sized for: the device's count, or AQ_NUM_CUS when the caller runs this process's streams on a
// subset of the CUs (hipExtStreamCreateWithCUMask: bench.py --cu-split gives each of the two batches in flight half of the chip).
// Read once per kernel family (their grid caches are process-global), so set the variable before the first launch.
inline hipError_t aq_query_cus(int* cus, int dev) {
    const char* v = getenv("AQ_NUM_CUS");
    if (v && atoi(v) > 0) { *cus = atoi(v); return hipSuccess; }
    return hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev);
}
int aq_conv_halo_tiles(int hcfg, int* bm, int* bn);
extern "C" int aq_conv_config_tiles(int cfg, int* bm, int* bn);
extern "C" int aq_conv_num_configs(void);
