// How fast do the store shapes of the conv epilogues drain?  (tools/ubench: a measurement, not product code)
// Every kernel writes the same bytes (a bf16 tensor, 16-pixel blocks, every byte once); only the lane -> address map differs:
//   0  contiguous: a wave's block is 1.5 KB written front to back, 16 bytes per lane (what an LDS transpose would give the stem)
//   1  stem / down-block shape: 96-byte pixel rows, lane (p = lane & 15, g = lane >> 4) writes 16 + 8 bytes at 96 p + 24 g
//   2  unpermuted MFMA layout: three 8-byte pieces per lane at 96 p + 32 m + 8 g
//   3  wide 1x1 (conv1x1_asm): 1536-byte pixel rows, wave w of 8 writes 16 bytes at 1536 p + 96 w + 16 g and 8 bytes at 1536 p + 96 w + 64 + 8 g
//   4  the same tensor as 3 written as whole rows: a wave's 1.5 KB = two consecutive 768-byte half rows, 16 bytes per lane
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_patterns tools/ubench/store_patterns.hip && /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* out, long long nblocks) {      // nblocks: 16-pixel blocks of 1.5 KB (PAT 0-2) / 16-pixel x 768-byte groups (PAT 3-4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, g = lane >> 4;
    const u32x4 v4 = {0x3f803f80u, 0x3f803f80u, (unsigned)lane, (unsigned)blockIdx.x};
    const u32x2 v2 = {0x3f803f80u, (unsigned)lane};
    if (PAT <= 2) {
        for (long long b = (long long)blockIdx.x * 8 + wave; b < nblocks; b += (long long)gridDim.x * 8) {
            char* base = out + b * 1536;
            if (PAT == 0) {
                *(u32x4*)(base + 16 * lane) = v4;
                if (lane < 32) *(u32x4*)(base + 1024 + 16 * lane) = v4;
            } else if (PAT == 1) {
                *(u32x4*)(base + 96 * p + 24 * g) = v4;
                *(u32x2*)(base + 96 * p + 24 * g + 16) = v2;
            } else {
                for (int m = 0; m < 3; ++m) *(u32x2*)(base + 96 * p + 32 * m + 8 * g) = v2;
            }
        }
    } else {
        // a workgroup's eight waves cover 16 pixels x 1536-byte rows... of which this tensor is the first 768 bytes x 2 channel tiles: keep it simple -- rows of 768 bytes
        for (long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
            char* base = out + b * 16 * 768;
            if (PAT == 3) {
                *(u32x4*)(base + 768 * p + 96 * wave + 16 * g) = v4;
                *(u32x2*)(base + 768 * p + 96 * wave + 64 + 8 * g) = v2;
            } else {
                *(u32x4*)(base + 1536 * wave + 16 * lane) = v4;
                if (lane < 32) *(u32x4*)(base + 1536 * wave + 1024 + 16 * lane) = v4;
            }
        }
    }
}

template <int PAT> void run(const char* name, char* buf, size_t bytes, int reps) {
    const long long nblocks = PAT <= 2 ? bytes / 1536 : bytes / (16 * 768);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int cus : {256, 512}) {
        store_kernel<PAT><<<cus, 512>>>(buf, nblocks);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) store_kernel<PAT><<<cus, 512>>>(buf, nblocks);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %4d workgroups x 8 waves: %7.1f us per %.0f MB = %5.2f TB/s\n", name, cus, ms * 1e3 / reps, bytes / 1e6, bytes / (ms * 1e-3 / reps) / 1e12);
    }
}

int main() {
    const size_t bytes = (size_t)64 * 320 * 320 * 96;        // the stem's output at batch 64: 629 MB
    char* buf; CK(hipMalloc(&buf, bytes));
    run<0>("0 contiguous 16 B per lane", buf, bytes, 10);
    run<1>("1 96-B rows, 16 + 8 B at 24 g (stem)", buf, bytes, 10);
    run<2>("2 96-B rows, three 8-B pieces (raw MFMA)", buf, bytes, 10);
    run<3>("3 768-B rows, 64 + 32 B per wave (1x1 asm)", buf, bytes, 10);
    run<4>("4 768-B rows written whole", buf, bytes, 10);
    const size_t small = (size_t)25600 * 768 * 2;            // the 1x1 layers' output at 20x20: 39 MB (cache-resident between launches)
    run<3>("3 the same on 39 MB", buf, small, 30);
    run<4>("4 the same on 39 MB", buf, small, 30);
    return 0;
}
