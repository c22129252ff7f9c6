// Where does the immediate offset of global_load_lds_dwordx4 go: to the global address only, or to the LDS address as well?
// (gen_conv3x3_pl_asm.py, 32-channel-chunk families: one source address per region-row group, the planes by immediate offset.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const char* p = (const char*)src + threadIdx.x * 64;      // lane L: bytes 64 L ...
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:16\n\ts_waitcnt vmcnt(0)" ::"v"(p), "s"(1024) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *s, *o, hs[2048], ho[2048];
    for (int i = 0; i < 2048; ++i) hs[i] = i;
    hipMalloc(&s, 8192); hipMalloc(&o, 8192);
    hipMemcpy(s, hs, 8192, hipMemcpyHostToDevice);
    k<<<1, 64, 8192>>>(s, o);
    hipMemcpy(ho, o, 8192, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 2048; ++i) if (ho[i] != 0xdeadbeefu) { first = i; break; }
    printf("first written dword %d (byte %d), value %u (source dword; lane 0 + offset 16 = dword 4)\n", first, first * 4, first >= 0 ? ho[first] : 0);
    printf("=> the immediate offset %s added to the LDS address\n", first * 4 == 1024 ? "is NOT" : first * 4 == 1040 ? "IS" : "??");
    for (int i = first; i < first + 12 && i >= 0; ++i) printf("%u ", ho[i]);
    printf("\n");
    return 0;
}
