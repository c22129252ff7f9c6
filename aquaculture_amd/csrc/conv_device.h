// Device helpers shared by the convolution kernels (gfx950).
#pragma once
#include "aq_common.h"

namespace aqdev {

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    // LDS-DMA: 16 B per lane from a PER-LANE global address to wave-uniform LDS base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool F32>
__device__ __forceinline__ float silu(float v) {
    // [UPSTREAM nn.SiLU]: v * sigmoid(v) = v / (1 + exp(-v))
    if (F32) return v / (1.0f + expf(-v));                       // parity mode: IEEE divide, accurate exp
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));         // bf16 mode: v_exp + v_rcp (error << bf16 ulp)
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (RNE)
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // all but the wave's N youngest vector-memory operations are done (loads, LDS-DMA, stores: in issue order)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS store the compiler cannot see as one.  LLVM's waitcnt pass makes every visible LDS store wait for ALL outstanding
// LDS-DMA (s_waitcnt vmcnt(0)): in the epilogue that would drain the next tile's prefetch, which is exactly what the
// epilogue is meant to overlap.  The "memory" clobber keeps the compiler from moving other LDS accesses across it; the
// hardware executes one wave's LDS operations in order, and readers wait with an explicit s_waitcnt lgkmcnt(0).
__device__ __forceinline__ void lds_write_b128(char* dst, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(uintptr_t)dst), "v"(v) : "memory");
}

constexpr int kStgRow = 32 * 4 + 16;          // padded fp32 row of the 32x32 epilogue staging block
constexpr int kStgBytes = 32 * kStgRow;       // per wave

// Epilogue of one wave: acc[TM][TN] (32x32 MFMA C/D blocks; column = pixel = lane & 31,
// row = cout = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) -> +bias, SiLU -> LDS transpose (one block at a time,
// in the wave's own staging region) -> (+residual) -> 16-byte coalesced NHWC stores, 8 channels per lane.
template <bool F32, bool OUT4, int TM, int TN>
__device__ __forceinline__ void epilogue_store(const ConvParams& p, f32x16 (&acc)[TM][TN], char* stg, const float* sbias,
                                               int cwave /*first cout of this wave*/, int pwave /*first pixel of this wave*/,
                                               int lane) {
    const int h = lane >> 5, l31 = lane & 31;
    const int pix = lane >> 2, ch = lane & 3;
    // bf16 shortcut layers: fetch the whole residual tile of this wave up front (TM*TN*2 x 16 B per lane), so the
    // per-block loop below never waits on a global load
    uint4 rpre[TM][TN][2];
    const bool pre = !F32 && p.res != nullptr;
    if (pre) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int P = pwave + j * 32 + pix + 16 * it, c0 = cwave + i * 32 + ch * 8;
                    rpre[i][j][it] = (P < p.npix && c0 < p.cout) ? *(const uint4*)(p.res + (long long)P * p.res_ld_b + c0 * 2)
                                                                  : make_uint4(0, 0, 0, 0);
                }
    }
    // LDS reads of this function are inline asm with their own lgkmcnt wait.  For a VISIBLE LDS read the compiler, which cannot tell
    // the staging / bias area from the buffers an LDS-DMA may still be filling, emits s_waitcnt vmcnt(0) -- and vmcnt being in order,
    // that waited for the output stores of the previous block as well: the stores of an epilogue went out one block at a time.
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cblk = cwave + i * 32;
        f32x4 bias4[4];                                        // the block row's bias, rows 8 g + 4 h .. + 3 (read once per i, not per (i, j))
        {
            const uint32_t ba = (uint32_t)(uintptr_t)(sbias + cblk + 4 * h);
            asm volatile("ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:32\nds_read_b128 %2, %4 offset:64\nds_read_b128 %3, %4 offset:96\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(bias4[0]), "=&v"(bias4[1]), "=&v"(bias4[2]), "=&v"(bias4[3]) : "v"(ba) : "memory");
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pbase = pwave + j * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = 8 * g + 4 * h;
                if (cblk + 8 * g >= p.cout) continue;          // padding rows of the last M block: nothing to compute
                const f32x4 bv = bias4[g];
                f32x4 v;
                if constexpr (F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[i][j][4 * g + e] + bv[e];
                        if (p.act) t = silu<true>(t);
                        v[e] = t;
                    }
                } else {
                    // bf16 mode: bias add, exp argument, 1 + e and the final product as packed 2 x fp32 ops
                    const f32x4 a4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    v = a4 + bv;
                    if (p.act) {
                        const f32x4 t = v * -1.44269504f;
                        f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]),
                                   __builtin_amdgcn_exp2f(t[3])};
                        d = d + 1.0f;
                        const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]),
                                         __builtin_amdgcn_rcpf(d[3])};
                        v = v * r;
                    }
                }
                lds_write_b128(stg + l31 * kStgRow + cl * 4, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            f32x4 lohi[2][2];
            {
                const uint32_t sa = (uint32_t)(uintptr_t)(stg + pix * kStgRow + ch * 32);
                asm volatile("ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:16\nds_read_b128 %2, %4 offset:%5\nds_read_b128 %3, %4 offset:%6\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(lohi[0][0]), "=&v"(lohi[0][1]), "=&v"(lohi[1][0]), "=&v"(lohi[1][1])
                             : "v"(sa), "n"(16 * kStgRow), "n"(16 * kStgRow + 16) : "memory");
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int px = pix + 16 * it;
                const int P = pbase + px, c0 = cblk + ch * 8;
                const f32x4 lo = lohi[it][0], hi = lohi[it][1];
                if (P < p.npix && c0 < p.cout) {
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    char* orow = p.out + (long long)P * p.out_ld_b;
                    if (F32) {
                        if (p.res) {
                            const char* rrow = p.res + (long long)P * p.res_ld_b + c0 * 4;
                            const f32x4 r0 = *(const f32x4*)rrow, r1 = *(const f32x4*)(rrow + 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                        }
                    } else if (p.res) {
                        const uint4 rv = rpre[i][j][it];
                        const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += __uint_as_float(rw[e] << 16);
                            v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                        }
                    }
                    if (OUT4) {
                        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                        *(f32x4*)(orow + c0 * 4) = o0;
                        *(f32x4*)(orow + c0 * 4 + 16) = o1;
                    } else {
                        uint4 o;
                        o.x = pack_bf16x2(v[0], v[1]);
                        o.y = pack_bf16x2(v[2], v[3]);
                        o.z = pack_bf16x2(v[4], v[5]);
                        o.w = pack_bf16x2(v[6], v[7]);
                        *(uint4*)(orow + c0 * 2) = o;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// XCD-aware bijective block -> first-tile map of a persistent grid of G blocks (blocks sharing an XCD get consecutive ids)
__device__ __forceinline__ int first_tile(int G, int bid) {
    const int q = G >> 3, r = G & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

}  // namespace aqdev
