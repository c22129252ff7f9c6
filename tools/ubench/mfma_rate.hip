// Micro-benchmark (runs ON the GPU box; built here with hipcc --offload-arch=gfx950, the binary travels with the tree):
// cycles per MFMA for the accumulator / operand patterns of the planar 3x3 kernel, one wave per SIMD, no memory traffic.
// The MFMAs are inline asm on fixed accumulator registers so that the instruction stream is exactly what is written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;

#define ACC_CLOBBER                                                                                                        \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", \
        "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", \
        "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", \
        "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", \
        "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", \
        "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", \
        "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129",     \
        "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145",     \
        "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"

// 39 independent 16x16x32 MFMAs (3 A x 13 B), accumulators a[4n : 4n+3]
#define M16(n, A, B) "v_mfma_f32_16x16x32_bf16 a[(" #n ")*4:(" #n ")*4+3], %" #A ", %" #B ", a[(" #n ")*4:(" #n ")*4+3]\n\t"
#define ROW16(j, B) M16(j * 3 + 0, 0, B) M16(j * 3 + 1, 1, B) M16(j * 3 + 2, 2, B)
#define BODY16 ROW16(0, 3) ROW16(1, 4) ROW16(2, 5) ROW16(3, 6) ROW16(4, 7) ROW16(5, 8) ROW16(6, 9) ROW16(7, 10) ROW16(8, 11) ROW16(9, 12) ROW16(10, 13) ROW16(11, 14) ROW16(12, 15)
// same with a plain VALU instruction after every MFMA / after every third MFMA
#define M16V(n, A, B) M16(n, A, B) "v_add_f32 %16, %16, %16\n\t"
#define ROW16V(j, B) M16V(j * 3 + 0, 0, B) M16V(j * 3 + 1, 1, B) M16V(j * 3 + 2, 2, B)
#define BODY16V ROW16V(0, 3) ROW16V(1, 4) ROW16V(2, 5) ROW16V(3, 6) ROW16V(4, 7) ROW16V(5, 8) ROW16V(6, 9) ROW16V(7, 10) ROW16V(8, 11) ROW16V(9, 12) ROW16V(10, 13) ROW16V(11, 14) ROW16V(12, 15)
// 10 independent 32x32x16 MFMAs, accumulators a[16n : 16n+15]
#define M32(n, A, B) "v_mfma_f32_32x32x16_bf16 a[(" #n ")*16:(" #n ")*16+15], %" #A ", %" #B ", a[(" #n ")*16:(" #n ")*16+15]\n\t"
#define BODY32 M32(0, 0, 3) M32(1, 1, 4) M32(2, 2, 5) M32(3, 0, 6) M32(4, 1, 7) M32(5, 2, 8) M32(6, 0, 9) M32(7, 1, 10) M32(8, 2, 11) M32(9, 0, 12)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* in, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 A0 = in[lane], A1 = in[64 + lane], A2 = in[128 + lane];
    bf16x8 B[13];
    for (int j = 0; j < 13; ++j) B[j] = in[((j + 3) * 64 + lane) & 1023];
    float dummy = 1.0f;
    const unsigned long long t0 = clock64(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0)
            asm volatile(BODY16 BODY16 : : "v"(A0), "v"(A1), "v"(A2), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(B[6]), "v"(B[7]),
                         "v"(B[8]), "v"(B[9]), "v"(B[10]), "v"(B[11]), "v"(B[12]) : ACC_CLOBBER);
        if constexpr (MODE == 1)
            asm volatile(BODY32 BODY32 : : "v"(A0), "v"(A1), "v"(A2), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(B[6]), "v"(B[7]),
                         "v"(B[8]), "v"(B[9]), "v"(B[10]), "v"(B[11]), "v"(B[12]) : ACC_CLOBBER);
        if constexpr (MODE == 3)   // the chunk body of the planar kernel: 702 MFMAs of straight-line code (5.6 KB) per loop iteration
            asm volatile(BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16 BODY16
                         : : "v"(A0), "v"(A1), "v"(A2), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(B[6]), "v"(B[7]),
                         "v"(B[8]), "v"(B[9]), "v"(B[10]), "v"(B[11]), "v"(B[12]) : ACC_CLOBBER);
        if constexpr (MODE == 2)
            asm volatile(BODY16V BODY16V : : "v"(A0), "v"(A1), "v"(A2), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(B[6]), "v"(B[7]),
                         "v"(B[8]), "v"(B[9]), "v"(B[10]), "v"(B[11]), "v"(B[12]), "v"(dummy) : ACC_CLOBBER);
    }
    const unsigned long long t1 = clock64(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(s));
    out[blockIdx.x * 256 + threadIdx.x] = s + dummy;
    if (lane == 0) { cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(const char* name, int per_iter, double flop_per, const bf16x8* in, float* out, unsigned long long* cyc, int grid, int iters = 2000) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < grid * 4; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    c /= grid * 4; r /= grid * 4;
    printf("%-44s grid %3d: %6.2f cycles/MFMA  clock %4.0f MHz  %7.1f TFLOP/s (whole launch %.3f ms)\n", name, grid, c / ((double)iters * per_iter), c / r * 100.0,
           flop_per * per_iter * iters * grid * 4 / (ms * 1e-3) / 1e12, ms);
}

int main() {
    bf16x8* in; float* out; unsigned long long* cyc;
    (void)hipMalloc(&in, 1024 * 16); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<unsigned short> h(1024 * 8);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) + ((x >> 31) << 15)); }   // random bf16 around +-1
    (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int grid : {1, 256}) {
        run<0>("16x16x32 x78, 39 accumulators", 78, 2.0 * 16 * 16 * 32, in, out, cyc, grid);
        run<2>("16x16x32 x78 + one v_add_f32 after each", 78, 2.0 * 16 * 16 * 32, in, out, cyc, grid);
        run<1>("32x32x16 x20, 10 accumulators", 20, 2.0 * 32 * 32 * 16, in, out, cyc, grid);
        run<3>("16x16x32 x702 straight-line, 300 iterations", 702, 2.0 * 16 * 16 * 32, in, out, cyc, grid, 300);
        run<3>("16x16x32 x702 straight-line, 6 iterations", 702, 2.0 * 16 * 16 * 32, in, out, cyc, grid, 6);
    }
    return 0;
}
