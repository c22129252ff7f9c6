// v_permlane16_swap_b32 vdst, src (gfx950): which lanes exchange?  Used by the wide-store epilogue of the planar 3x3 kernels
// (gen_conv3x3_pl_asm.py, W16): expected = lanes 16-31 / 48-63 of vdst swap with lanes 0-15 / 32-47 of src, the rest stay put.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
}
int main() {
    unsigned *o, h[128];
    hipMalloc(&o, 512);
    k<<<1, 64>>>(o);
    hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int L = 0; L < 64; ++L) {
        const unsigned ea = (L & 16) ? 100 + (L - 16) : L, eb = (L & 16) ? 100 + L : L + 16;
        if (h[L] != ea || h[64 + L] != eb) ok = 0;
    }
    for (int L = 0; L < 64; L += 8) printf("lane %2d: vdst %3u src %3u\n", L, h[L], h[64 + L]);
    printf("v_permlane16_swap_b32: %s\n", ok ? "odd rows of vdst <-> even rows of src (as expected)" : "UNEXPECTED lane map");
    return ok ? 0 : 1;
}
