// buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor): what lands in LDS for a lane whose offset is out of range --
// zeros (usable as the padding of a region) or nothing?  And does the immediate offset move the LDS address as for global_load_lds?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void k(const unsigned* src, unsigned bytes, unsigned* out) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    unsigned long long p = (unsigned long long)src;
    u32x4 srd = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, bytes, 0x00020000u};
    srd.x = __builtin_amdgcn_readfirstlane(srd.x); srd.y = __builtin_amdgcn_readfirstlane(srd.y);
    srd.z = __builtin_amdgcn_readfirstlane(srd.z); srd.w = __builtin_amdgcn_readfirstlane(srd.w);
    // lane L reads 16 bytes at 64 * (L / 4) + 16 * (L % 4): four lanes = one 64-byte piece; lanes 8..11 and 60..63 out of range
    unsigned off = 64u * (threadIdx.x >> 2) + 16u * (threadIdx.x & 3);
    if ((threadIdx.x >= 8 && threadIdx.x < 12) || threadIdx.x >= 60) off = 0x80000000u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:0 lds\n\ts_waitcnt vmcnt(0)" ::"v"(off), "s"(srd), "s"(1024) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *s, *o, hs[2048], ho[2048];
    for (int i = 0; i < 2048; ++i) hs[i] = 1000 + i;
    hipMalloc(&s, 8192); hipMalloc(&o, 8192);
    hipMemcpy(s, hs, 8192, hipMemcpyHostToDevice);
    k<<<1, 64, 8192>>>(s, 8192, o);
    hipMemcpy(ho, o, 8192, hipMemcpyDeviceToHost);
    for (int L = 0; L < 64; ++L) {
        const unsigned* d = ho + 256 + 4 * L;
        printf("lane %2d -> LDS dwords %u %u %u %u%s\n", L, d[0], d[1], d[2], d[3], d[0] == 0xdeadbeefu ? "  (untouched)" : d[0] == 0 ? "  (zeros)" : "");
    }
    return 0;
}
