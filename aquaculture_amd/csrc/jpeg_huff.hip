// GPU entropy decode of baseline JPEG scans (round 4): the Huffman stage that csrc/jpeg_coef.c runs
// on the host, one LANE per restart segment -- per image for the tiles of reference src/load_data/tile_tifs.py:66-74 (GDAL writes no
// restart markers).  The stage is serial per segment, so the parallelism is ACROSS images: a wave decodes 64 segments, a launch decodes a
// whole "super-batch" (hundreds of tiles), and the caller keeps a few launches in flight beside the detector's own kernels (a wave per 64
// tiles is nothing to the chip).  Input: what aq_jpeg_prepare (include/aq_jpeg.h) wrote on the host -- the scan's bytes without byte
// stuffing, cut at restart markers, every segment 4-byte aligned and zero-padded -- so the bit reader here is 32-bit loads and shifts.
// Output: the same quantised int16 coefficient blocks, natural order, that aq_jpeg_decode_coeffs writes (Y [2 rows][2 cols] blocks of the
// padded image, then Cb, then Cr; the buffer must be zero on entry), bit for bit (tests/test_jpeg.py), for aq_jpeg_idct_rgb.
//
// SIMT shape: ONE flat loop in which every active lane decodes one Huffman symbol per iteration (DC and AC symbols go through the same
// code: a DC symbol is "run 0, size s" at k = 0); lanes sit at different blocks and MCUs, but there is no per-block reconvergence point
// to wait at.  A launch ends when its longest segment ends.  Tables: the six tables of one table set (all tiles of a sweep share one:
// GDAL / Pillow write the standard tables) are staged in LDS; lanes whose set differs read theirs from global memory.
#include "aq_common.h"

namespace {

struct JpegGpuTab {                  // = aq_jpeg_gpu_tab (include/aq_jpeg.h)
    unsigned short look[512];
    int maxcode[18];
    int valoff[18];
    unsigned char vals[256];
};
static_assert(sizeof(JpegGpuTab) == 1424, "table layout");

struct JpegSeg {                     // one restart segment (host-built; 32 bytes)
    unsigned stream_off;             // bytes from the stream base (multiple of 4)
    unsigned stream_len;             // bytes of entropy-coded data (without the zero padding)
    unsigned long long coef_off;     // int16 index of the IMAGE's first coefficient
    unsigned mcu0, n_mcu;            // MCUs of the scan this segment covers
    unsigned short mcu_cols, mcu_rows;
    unsigned tabset;                 // index into the table sets
};
static_assert(sizeof(JpegSeg) == 32, "segment descriptor");

__constant__ unsigned char kZig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// status per segment: 0 ok, 2 corrupt (bad code, coefficient index past 63, DC category > 11, or bits consumed past the segment's end)
//
// The bit stream of a lane goes through a 2 x 64-byte window in LDS ([window][word][lane]: lanes on consecutive banks) that is refilled one
// window at a time by four 16-byte loads, with the window after next already in flight in registers.  First build: one 4-byte global load
// per refill -- the compiler's wait for it (vmcnt is in order and counts stores too) also drained the scattered 2-byte coefficient stores
// issued since, once per ~5 symbols: 1.9-2.4 k cycles per symbol (65 ms per 512 smooth 1024-px tiles).  Now the wave meets vector memory
// latency once per 64 bytes of stream, and what it waits for was issued a window (~90 symbols) earlier.
__global__ __launch_bounds__(64) void jpeg_huff_kernel(const unsigned char* __restrict__ streams, const JpegSeg* __restrict__ segs, int nseg,
                                                       const JpegGpuTab* __restrict__ sets, short* __restrict__ coef, int* __restrict__ status) {
    __shared__ unsigned short s_look[6][512];
    __shared__ unsigned s_win[2][16][64];
    __shared__ unsigned char s_zig[64];
    const int lane = threadIdx.x;
    const int si = blockIdx.x * 64 + lane;
    const bool have = si < nseg;
    const JpegSeg sg = segs[have ? si : nseg - 1];
    const unsigned set0 = __builtin_amdgcn_readfirstlane(sg.tabset);
    for (int i = lane; i < 6 * 512; i += 64) s_look[i >> 9][i & 511] = sets[(size_t)set0 * 6 + (i >> 9)].look[i & 511];
    s_zig[lane] = kZig[lane];
    const bool lds_tabs = sg.tabset == set0;
    const JpegGpuTab* my = sets + (size_t)sg.tabset * 6;

    // chunk c = bytes 64 c .. 64 c + 63 of the segment (16-byte aligned; behind the last segment the upload buffer has >= 256 spare bytes,
    // so whole chunks are always readable -- what lies past stream_len is never counted as data: the overrun check works on bit counts)
    const uint4* src = (const uint4*)(streams + sg.stream_off);
    auto put_window = [&](int wb, const uint4 (&q)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_win[wb][4 * j + 0][lane] = q[j].x; s_win[wb][4 * j + 1][lane] = q[j].y;
            s_win[wb][4 * j + 2][lane] = q[j].z; s_win[wb][4 * j + 3][lane] = q[j].w;
        }
    };
    uint4 pf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pf[j] = src[j];
    put_window(0, pf);
#pragma unroll
    for (int j = 0; j < 4; ++j) pf[j] = src[4 + j];
    put_window(1, pf);
#pragma unroll
    for (int j = 0; j < 4; ++j) pf[j] = src[8 + j];              // chunk 2: in flight until window 0 is used up
    unsigned next_chunk = 3;
    __syncthreads();

    int wb = 0, widx = 0;
    unsigned words = 0;                                          // 32-bit words taken so far
    unsigned long long acc = 0;
    int nbits = 0;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    unsigned mcu = sg.mcu0;
    const unsigned mcu_end = sg.mcu0 + sg.n_mcu;
    int blk = 0;                                                 // block of the MCU: 0-3 Y, 4 Cb, 5 Cr
    int k = 0;                                                   // next coefficient (zigzag index); 0 = the DC symbol comes next
    // offsets (int16 units, relative to the image's first coefficient; < 2^31 for any image the host accepts) of the MCU's blocks, advanced
    // by additions: a division per block end would be executed by the whole wave whenever ANY lane finishes a block -- almost every iteration
    const unsigned cols = sg.mcu_cols;
    const unsigned ny64 = 4u * cols * sg.mcu_rows * 64u, nc64 = cols * sg.mcu_rows * 64u, yrow = 2u * cols * 64u;
    short* const base = coef + sg.coef_off;
    unsigned mx = sg.mcu0 % cols;
    unsigned yoff = ((2u * (sg.mcu0 / cols)) * (2u * cols) + 2u * mx) * 64u;      // Y block (2 my, 2 mx)
    unsigned coff = sg.mcu0 * 64u;                                               // Cb / Cr block of the MCU, relative to its plane
    auto block_off = [&](int b) -> unsigned {
        return b < 4 ? yoff + (b & 1) * 64u + (b >> 1) * yrow : ny64 + (b == 5 ? nc64 : 0u) + coff;
    };
    short* out = base + block_off(0);
    bool active = have && sg.n_mcu > 0;
    int st = 0;

    while (__any(active)) {
        if (active) {
            if (nbits <= 32) {                                   // refill: 32 bits from the window
                acc = (acc << 32) | __builtin_bswap32(s_win[wb][widx][lane]);
                nbits += 32;
                ++words;
                if (++widx == 16) {                              // window used up: the other one is ready; this one takes the chunk in flight
                    put_window(wb, pf);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pf[j] = src[4 * next_chunk + j];
                    ++next_chunk;
                    wb ^= 1;
                    widx = 0;
                }
            }
            const int comp = blk < 4 ? 0 : blk - 3;
            const int ti = 2 * comp + (k ? 1 : 0);
            // ONE 64-bit shift per symbol: the next 32 bits of the stream, left-aligned; the 9-bit table index, the long codes and the value
            // bits all come out of this word with 32-bit shifts (a symbol <= 16 bits and its <= 15 value bits fit: nbits >= 33 here)
            const unsigned top = (unsigned)(acc >> (nbits - 32));
            const unsigned idx = top >> 23;
            unsigned e = s_look[ti][idx];
            asm volatile("" : "+v"(e));                          // (opaque: keeps the two loads two instructions, see the header)
            if (!lds_tabs) e = my[ti].look[idx];
            int len, sym;
            if (e) {
                len = (int)(e >> 8);
                sym = (int)(e & 0xff);
            } else {                                             // code longer than 9 bits (rare): the canonical-code search of the host decoder
                const JpegGpuTab* t = my + ti;
                len = 10;
                int code = (int)(top >> 22);
                while (len <= 16 && code > t->maxcode[len]) { ++len; code = (int)(top >> (32 - len)); }
                if (len > 16) { st = 2; active = false; continue; }
                sym = t->vals[(code + t->valoff[len]) & 0xff];
            }
            const int r = k ? sym >> 4 : 0;
            const int s = k ? sym & 15 : sym;
            if (k == 0 && s > 11) { st = 2; active = false; continue; }
            nbits -= len + s;                                    // (s = 0 for EOB / ZRL / a zero DC difference)
            if (k != 0 && s == 0) {
                if (r == 15) k += 16;                            // ZRL
                else k = 64;                                     // EOB
            } else {
                k += r;
                if (k > 63) { st = 2; active = false; continue; }
                int v = 0;
                if (s) {
                    v = (int)((top << len) >> (32 - s));
                    if (v < (1 << (s - 1))) v += 1 - (1 << s);
                }
                if (k == 0) {
                    int p = comp == 0 ? pred0 : comp == 1 ? pred1 : pred2;
                    p += v;
                    if (comp == 0) pred0 = p; else if (comp == 1) pred1 = p; else pred2 = p;
                    v = p;
                }
                if (v) out[s_zig[k]] = (short)v;                 // (the buffer is zero on entry)
                ++k;
            }
            if (k >= 64) {                                       // next block
                k = 0;
                if (++blk == 6) {
                    blk = 0; ++mcu;
                    yoff += 128u; coff += 64u;
                    if (++mx == cols) { mx = 0; yoff += yrow; }       // next MCU row: skip the second Y block row of this one
                }
                if (mcu >= mcu_end) {
                    active = false;
                    // bits consumed = 32 x words taken - bits left; more than the segment holds: a truncated / interrupted scan
                    if ((long long)words * 32 - nbits > (long long)sg.stream_len * 8) st = 2;
                } else {
                    out = base + block_off(blk);
                }
            }
        }
    }
    if (have) status[si] = st;
}

}  // namespace

// Decodes `nseg` restart segments (aq_jpeg_prepare's output, uploaded as it is) into zeroed coefficient buffers.  streams_dev: the upload
// buffer; segs_dev: nseg descriptors of 32 bytes {u32 stream_off, u32 stream_len, u64 coef_off (int16 index of the image's first
// coefficient in coef_dev), u32 mcu0, u32 n_mcu, u16 mcu_cols, u16 mcu_rows, u32 tabset}; tabsets_dev: n table sets of six aq_jpeg_gpu_tab;
// status_dev: int32 per segment (0 ok, 2 corrupt).  The caller has validated every offset (the kernel trusts them) and zeroed coef_dev.
extern "C" int aq_jpeg_huffman_decode(const void* streams_dev, const void* segs_dev, int nseg, const void* tabsets_dev, void* coef_dev,
                                      void* status_dev, void* stream) {
    AQ_REQUIRE(streams_dev && segs_dev && tabsets_dev && coef_dev && status_dev && nseg > 0, "jpeg_huffman_decode: bad argument");
    const int grid = (nseg + 63) / 64;
    hipLaunchKernelGGL(jpeg_huff_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (const unsigned char*)streams_dev, (const JpegSeg*)segs_dev, nseg,
                       (const JpegGpuTab*)tabsets_dev, (short*)coef_dev, (int*)status_dev);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
