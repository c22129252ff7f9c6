// What v_cvt_pk_f32_fp8 decodes on this GPU: all 256 codes -> float, printed for comparison with OCP e4m3fn (quant.py).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int c = threadIdx.x;                         // 256 threads: code c
    const int packed = c | (c << 8) | (c << 16) | (c << 24);
    auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(packed, false);
    auto hi = __builtin_amdgcn_cvt_pk_f32_fp8(packed, true);
    out[c] = lo[0];
    out[256 + c] = hi[1];
}
int main() {
    float* d; hipMalloc(&d, 2048);
    k<<<1, 256>>>(d);
    float h[512]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
    for (int c = 0; c < 256; ++c) printf("%d %.10g %.10g\n", c, h[c], h[256 + c]);
    return 0;
}
