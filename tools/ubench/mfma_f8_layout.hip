// v_mfma_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, unscaled form): which k does byte j of lane l's 32-byte operand hold, and what does
// one instruction cost?  A[i][k] = 1 for one chosen (i, k), B[k][n] = code of small integers: D[i][n] picks B[k][n].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const unsigned char* A, const unsigned char* B, float* D) {
    // A, B: [64 lanes][32 bytes] raw operand registers
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = ((const int*)A)[l * 8 + r]; b[r] = ((const int*)B)[l * 8 + r]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
__global__ void rate(float* out, int iters) {
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = 0x38383838 + threadIdx.x; b[r] = 0x38383838 ^ (threadIdx.x * 7); }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (8.0f * iters); out[1] = s; }
}
int main() {
    unsigned char hA[2048], hB[2048];
    float hD[256];
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    // e4m3 codes of 1, 2, 3, ...: use exact small integers 0..15 -> code table
    auto code = [](int v) -> unsigned char {   // e4m3fn of small non-negative integers (exact up to 16)
        if (v == 0) return 0;
        int e = 0; while ((1 << (e + 1)) <= v) ++e;              // v = m * 2^e, m in [1, 2)
        int mant = ((v << 3) >> e) & 7;
        return (unsigned char)(((e + 7) << 3) | mant);
    };
    // test: for lane group g (0..3) and byte j (0..31): put A = 1 at row 0 for (lane = 16 g, byte j) only; B[lane][byte] = code((lane >> 4) * 4 + (byte >> 3)) ... too coarse;
    // instead B byte value encodes (lane>>4) in 0..3 and j/2 in 0..15 separately in two runs.
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 32; j += 5) {
            memset(hA, 0, 2048); memset(hB, 0, 2048);
            hA[(16 * g + 0) * 32 + j] = code(1);                     // A row 0 (lane & 15 == 0), k-group g, byte j
            for (int l = 0; l < 64; ++l) for (int b = 0; b < 32; ++b) hB[l * 32 + b] = code(((l >> 4) == g && b == j) ? 2 : ((l >> 4) == g ? 1 : 0));
            hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, dD);
            hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
            // D row 0 lives in lanes 0..15, reg 0 (row = 4 (lane >> 4) + reg): expect 2 everywhere in row 0 if k of A(g, j) == k of B(g, j)
            printf("A(row0, lanegroup %d, byte %2d) x B: D[0][0..3] = %g %g %g %g  (2 = same k on both sides, 1 = another k of the group, 0 = another group)\n",
                   g, j, hD[0], hD[4], hD[8], hD[12]);
        }
    float* dr; hipMalloc(&dr, 8);
    rate<<<1, 64>>>(dr, 1000);
    rate<<<1, 64>>>(dr, 100000);
    float hr[2]; hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost);
    printf("cycles per v_mfma_scale_f32_16x16x128_f8f6f4 (8 independent accumulators, one wave): %.1f\n", hr[0]);
    return 0;
}
