// buffer_load_dwordx4 ... lds with some lanes switched off in EXEC: do the inactive lanes leave their 16 bytes of LDS alone, and do the active
// lanes still land at M0 + 16 * lane (not compacted)?  Needed before an LDS image may end in the middle of an LDS-DMA instruction (the tail
// of a patch next to another buffer: round 4's C = 48 Bottleneck re-reads its shortcut from memory for want of a third x buffer).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_exec_mask tools/ubench/lds_dma_exec_mask.hip && /tmp/lds_dma_exec_mask
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
    unsigned off = 16u * threadIdx.x;
    // lanes 0-23 and 40-47 active (EXEC = 0x0000ff0000ffffff)
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 s22, 0x00ffffff\n\ts_mov_b32 s23, 0x0000ff00\n\ts_mov_b64 exec, s[22:23]\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:0 lds\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, s[20:21]"
                 ::"v"(off), "s"(srd), "s"(1024) : "memory", "s20", "s21", "s22", "s23");
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
    int ok = 1;
    for (int L = 0; L < 64; ++L) {
        const unsigned* d = ho + 256 + 4 * L;
        const bool active = L < 24 || (L >= 40 && L < 48);
        const bool good = active ? d[0] == 1000u + 4 * L : d[0] == 0xdeadbeefu;
        ok &= good;
        printf("lane %2d (%s) -> LDS dwords %u %u %u %u%s\n", L, active ? "on " : "off", d[0], d[1], d[2], d[3], d[0] == 0xdeadbeefu ? "  (untouched)" : "");
    }
    printf("%s\n", ok ? "RESULT: inactive lanes leave LDS alone; active lanes land at M0 + 16 * lane" : "RESULT: NOT as assumed");
    return 0;
}
