// What does a vector instruction cost a SIMD -- alone, and beside a second wave of the same SIMD?  (round 4: the fused Bottleneck kernels
// spend half their VALU issue on v_exp_f32 + v_rcp_f32, and VERDICT r03 item 5 asks whether SiLU in packed fp16 would be cheaper.)
//
// One workgroup, 256 threads (one wave per SIMD) or 512 (two per SIMD: waves w and w + 4 share a SIMD).  Every wave runs `iters` x 32
// independent instructions of ONE kind (kind A for waves 0-3, kind B for waves 4-7) between two s_memtime stamps and reports cycles per
// instruction.  Kinds: 0 v_mul_f32, 1 v_pk_mul_f32, 2 v_exp_f32, 3 v_rcp_f32, 4 v_exp_f16, 5 v_rcp_f16, 6 v_pk_mul_f16, 7 v_pk_fma_f16,
// 8 v_mfma_f32_16x16x32_bf16, 9 v_cvt_pk_bf16_f32, 10 exp/mul alternating (exp, mul, exp, mul ...), 11 idle (s_sleep), 12 v_cvt_pkrtz_f16_f32,
// 13 mfma / exp alternating
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

template <int KIND> __device__ __forceinline__ void body() {
    if constexpr (KIND == 0) asm volatile(REP32("v_mul_f32 v10, v11, v12\n v_mul_f32 v13, v11, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 1) asm volatile(REP32("v_pk_mul_f32 v[10:11], v[12:13], v[14:15]\n v_pk_mul_f32 v[16:17], v[12:13], v[14:15]\n") ::: "v10", "v11", "v16", "v17");
    if constexpr (KIND == 2) asm volatile(REP32("v_exp_f32 v10, v11\n v_exp_f32 v13, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 3) asm volatile(REP32("v_rcp_f32 v10, v11\n v_rcp_f32 v13, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 4) asm volatile(REP32("v_exp_f16 v10, v11\n v_exp_f16 v13, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 5) asm volatile(REP32("v_rcp_f16 v10, v11\n v_rcp_f16 v13, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 6) asm volatile(REP32("v_pk_mul_f16 v10, v11, v12\n v_pk_mul_f16 v13, v11, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 7) asm volatile(REP32("v_pk_fma_f16 v10, v11, v12, v14\n v_pk_fma_f16 v13, v11, v12, v14\n") ::: "v10", "v13");
    if constexpr (KIND == 8) asm volatile(REP32("v_mfma_f32_16x16x32_bf16 v[20:23], v[10:13], v[14:17], v[20:23]\n v_mfma_f32_16x16x32_bf16 v[24:27], v[10:13], v[14:17], v[24:27]\n")
                                          ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if constexpr (KIND == 9) asm volatile(REP32("v_cvt_pk_bf16_f32 v10, v11, v12\n v_cvt_pk_bf16_f32 v13, v11, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 10) asm volatile(REP32("v_exp_f32 v10, v11\n v_mul_f32 v13, v11, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 11) asm volatile(REP32("s_sleep 1\n s_sleep 1\n"));
    if constexpr (KIND == 12) asm volatile(REP32("v_cvt_pkrtz_f16_f32 v10, v11, v12\n v_cvt_pkrtz_f16_f32 v13, v11, v12\n") ::: "v10", "v13");
    if constexpr (KIND == 13) asm volatile(REP32("v_mfma_f32_16x16x32_bf16 v[20:23], v[10:13], v[14:17], v[20:23]\n v_exp_f32 v18, v19\n") ::: "v20", "v21", "v22", "v23", "v18");
}

template <int KA, int KB> __global__ __launch_bounds__(512) void bench(unsigned long long* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("v_mov_b32 v11, 1.0\n v_mov_b32 v12, 0.5\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v10, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n"
                 "v_mov_b32 v19, 0.5\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
                 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) { for (int i = 0; i < iters; ++i) body<KA>(); }
    else { for (int i = 0; i < iters; ++i) body<KB>(); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

typedef void (*Fn)(unsigned long long*, int);
template <int KA> Fn pickB(int kb) {
    switch (kb) {
        case 0: return bench<KA, 0>; case 1: return bench<KA, 1>; case 2: return bench<KA, 2>; case 3: return bench<KA, 3>; case 8: return bench<KA, 8>;
        case 10: return bench<KA, 10>; case 11: return bench<KA, 11>; default: return bench<KA, KA>;
    }
}
Fn pick(int ka, int kb) {
    switch (ka) {
        case 0: return pickB<0>(kb); case 1: return pickB<1>(kb); case 2: return pickB<2>(kb); case 3: return pickB<3>(kb); case 4: return pickB<4>(kb);
        case 5: return pickB<5>(kb); case 6: return pickB<6>(kb); case 7: return pickB<7>(kb); case 8: return pickB<8>(kb); case 9: return pickB<9>(kb);
        case 10: return pickB<10>(kb); case 12: return pickB<12>(kb); case 13: return pickB<13>(kb); default: return pickB<11>(kb);
    }
}

int main() {
    const char* names[] = {"v_mul_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_mfma_16x16x32_bf16",
                           "v_cvt_pk_bf16_f32", "exp/mul alternating", "idle (s_sleep)", "v_cvt_pkrtz_f16_f32", "mfma/exp alternating"};
    unsigned long long* d;
    hipMalloc(&d, 64);
    const int iters = 200;
    auto run = [&](int ka, int kb, int threads) {
        unsigned long long h[8] = {0};
        hipMemset(d, 0, 64);
        Fn f = pick(ka, kb);
        hipLaunchKernelGGL(f, dim3(1), dim3(threads), 0, 0, d, 10);
        hipLaunchKernelGGL(f, dim3(1), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        const double n = 64.0 * iters;
        if (threads == 256) printf("%-22s alone (1 wave / SIMD):            %6.2f cycles per instruction\n", names[ka], h[0] / n);
        else printf("%-22s beside %-22s: waves 0-3 %6.2f   waves 4-7 %6.2f cycles per instruction (each wave runs its own %d instructions)\n",
                    names[ka], names[kb], h[0] / n, h[4] / n, (int)n);
    };
    for (int k : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13}) run(k, k, 256);
    printf("-- two waves per SIMD --\n");
    for (int k : {0, 1, 2, 3, 8}) run(k, k, 512);
    int pairs[][2] = {{8, 2}, {2, 8}, {8, 0}, {0, 8}, {8, 1}, {1, 8}, {2, 0}, {0, 2}, {2, 3}, {8, 10}, {10, 8}, {2, 11}, {8, 11}, {13, 13}, {10, 10}};
    for (auto& p : pairs) run(p[0], p[1], 512);
    return 0;
}
