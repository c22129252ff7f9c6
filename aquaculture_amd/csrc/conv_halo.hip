// 3x3 / stride 1 / pad 1 convolution with LDS-staged input tiles (halo reuse), gfx950.
//
// Same contract as conv_igemm.hip (out = (res +) SiLU(conv(x, W') + b'), NHWC, implicit GEMM on MFMA, persistent
// workgroups, coalesced epilogue) but a different data flow for the 22 stride-1 3x3 layers of YOLOv5
// ([UPSTREAM models/common.py Bottleneck.cv2]; reached through reference README.md:77):
//
//   * the K loop runs channel-chunk outer, tap inner.  For a 64-channel chunk the input REGION of the pixel tile --
//     the linear pixel range [n0 - (W+1), n0 + BN + (W+1)) -- is brought into LDS ONCE by LDS-DMA and all nine taps read
//     it at shifted rows (row = pixel + dy*W + dx).  A tap that falls outside the image reads a 128-byte zero row
//     instead: each lane carries a 9-bit validity mask per output pixel.  Input bytes through L2 drop from 9x to
//     (1 + 2(W+1)/BN)x.
//   * only the weight slice of (tap, chunk) streams per step (double buffered); the NEXT region's loads are spread over
//     taps 0..7 of the current chunk, so every step puts the same small amount of DMA in flight and waits with a
//     counted vmcnt for exactly the weight slice it needs next.
//   * rows stay 128 B with the source-side XOR swizzle (slot ^= (row >> 1) & 7): 32 lanes reading 32 consecutive
//     region rows are ds_read_b128 conflict-free for every tap shift.
#include "conv_device.h"
#include <type_traits>

using namespace aqdev;

namespace {

template <bool F32, int BM, int BN, int WM, int WN, bool STAMP = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) >= 4 ? (WM * WN) / 4 : 1) void conv3x3_halo_kernel(const ConvParams p) {
    constexpr int NW = WM * WN;
    constexpr int ROWB = 128;
    constexpr int NIW = BM / 8, JW = NIW / NW;
    static_assert(NIW % NW == 0, "weight rows must split evenly over the waves");
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave tiles are 32x32 MFMA blocks");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int WBUF = BM * ROWB;
    constexpr bool STAGE_IN_W = WBUF >= NW * kStgBytes;   // else the epilogue stages in the just-consumed region buffer
    constexpr int NSTORE = TM * TN * 2 * (F32 ? 2 : 1);
    static_assert(NSTORE <= 63, "vmcnt immediate");
    constexpr int KMAX = 3;                               // region-load instructions a wave may issue per step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long ph_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic build only (STAMP), see conv_igemm.hip
    unsigned long long ph_t = 0;
    if constexpr (STAMP) ph_t = clock64();
    auto stamp = [&](int ph) {
        if constexpr (STAMP) {
            const unsigned long long t = clock64();
            ph_sum[ph] += t - ph_t;
            ph_t = t;
        }
    };
    const int G = gridDim.x;
    int tile = first_tile(G, blockIdx.x);
    const int ntiles = p.n_tiles_m * p.n_tiles_n;
    if (tile >= ntiles) return;

    const int XB = p.xrows * ROWB;
    auto wptr = [&](int sel) -> char* { return smem + sel * WBUF; };
    auto xptr = [&](int sel) -> char* { return smem + 2 * WBUF + sel * XB; };
    char* zrow = smem + 2 * WBUF + 2 * XB;
    float* sbias = (float*)(zrow + ROWB);
    for (int i = tid; i < p.bias_n; i += NW * 64) sbias[i] = p.bias[i];
    if (tid < 8) *(uint4*)(zrow + tid * 16) = make_uint4(0, 0, 0, 0);

    const int lrow = lane >> 3, lslot = lane & 7;
    const int gsw = lslot ^ (((wave & 1) << 2) | (lrow >> 1));   // weight tile: instruction index parity == wave parity
    const long long wstep = (long long)8 * NW * p.kgroups_pad * 16;
    const int hw = p.H * p.W;
    const int CC = (p.G + 7) >> 3;

    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int tn = p.n_tiles_m == 1 ? t : (int)__umulhi((unsigned)t, p.magic_ntm);
        m0 = (t - tn * p.n_tiles_m) * BM;
        n0 = tn * BN;
    };
    auto w_src = [&](int m0s, int tap, int cc) -> const char* {
        return p.w + ((long long)(m0s + 8 * wave + lrow) * p.kgroups_pad + (tap * p.G + 8 * cc + gsw)) * 16;
    };
    auto stage_w = [&](int m0s, int tap, int cc, char* wb) {
        const char* src = w_src(m0s, tap, cc);
#pragma unroll
        for (int j = 0; j < JW; ++j) glds16(src + j * wstep, wb + (wave + NW * j) * 1024);
    };
    // part `part` (0..7) of the region of (tile origin n0s, chunk cc): returns how many loads THIS wave issued
    auto stage_x_part = [&](int n0s, int cc, int part, char* xb) -> int {
        int issued = 0;
        const int q0 = part * p.xper, q1 = min(q0 + p.xper, p.nixr);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int q = q0 + wave + NW * k;
            if (q < q1) {
                const int r = 8 * q + lrow;
                const int Q = n0s - p.halo + r;
                const int grp = 8 * cc + (lslot ^ (((q & 1) << 2) | (lrow >> 1)));
                const bool ok = (unsigned)Q < (unsigned)p.npix && grp < p.G;
                const char* src = ok ? p.in + (long long)Q * p.in_ld_b + grp * 16 : p.zero;
                glds16(src, xb + q * 1024);
                ++issued;
            }
        }
        return issued;
    };
    auto wait_n = [&](int n) {      // n is wave-uniform
        if (n == 0) wait_vmcnt<0>();
        else if (n == 1) wait_vmcnt<1>();
        else if (n == 2) wait_vmcnt<2>();
        else wait_vmcnt<3>();
    };

    // ---------------- MFMA state ----------------
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;
    const int swa = (l31 >> 1) & 7;
    const int a_off = (wm * (BM / WM) + l31) * ROWB;
    f32x16 acc[TM][TN];
    int rbase[TN];          // region row of this lane's output pixel (tap 0,0)
    unsigned mask[TN];      // 9-bit tap validity of this lane's output pixels
#pragma unroll
    for (int j = 0; j < TN; ++j) rbase[j] = wn * (BN / WN) + j * 32 + l31 + p.halo;

    auto tile_masks = [&](int n0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int P = n0 + wn * (BN / WN) + j * 32 + l31;
            unsigned m = 0;
            if (P < p.npix) {
                int b = (int)((float)P * p.inv_hw);
                int rem = P - b * hw;
                if (rem < 0) { rem += hw; } else if (rem >= hw) { rem -= hw; }
                int y = (int)((float)rem * p.inv_wo);
                int x = rem - y * p.W;
                if (x < 0) { --y; x += p.W; } else if (x >= p.W) { ++y; x -= p.W; }
                const unsigned colbits = (x > 0 ? 1u : 0u) | 2u | (x + 1 < p.W ? 4u : 0u);
                m = (y > 0 ? colbits : 0u) | (colbits << 3) | (y + 1 < p.H ? colbits << 6 : 0u);
            }
            mask[j] = m;
        }
    };

    // next-step DMA is issued in slices between the k-steps' MFMA bursts: weight-slice instructions after k-steps 0..2,
    // the region part after k-step 3 (so the region loads are the youngest and the counted wait can leave them in flight)
    int nx_issued = 0;
    auto compute = [&](const char* wb, const char* xb, int tap, int cc, const char* nw_src, char* nw_buf,
                       bool x_on, int x_n0, int x_cc, int x_part, char* x_buf) {
        nx_issued = 0;
        const int ty = (tap * 11) >> 5;                    // tap / 3 for tap < 9
        const int dy = ty - 1, dx = tap - ty * 3 - 1;
        const int off = dy * p.W + dx;
        const char* rowp[TN];
        int swb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rr = rbase[j] + off;
            const bool valid = (mask[j] >> tap) & 1u;
            rowp[j] = valid ? xb + rr * ROWB : zrow;
            swb[j] = valid ? ((rr >> 1) & 7) : 0;
        }
        const int ksn = (min(8, p.G - 8 * cc) + 1) >> 1;    // k16 steps that hold real channels in this chunk
        using frag_t = typename std::conditional<F32, f32x4, bf16x8>::type;
        auto load_frags = [&](int ks, frag_t (&fa)[TM], frag_t (&fb)[TN]) {
            const int soa = (((2 * ks + h) ^ swa) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const frag_t*)(wb + a_off + i * 32 * ROWB + soa);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const frag_t*)(rowp[j] + (((2 * ks + h) ^ swb[j]) << 4));
        };
        auto mfmas = [&](const frag_t (&fa)[TM], const frag_t (&fb)[TN]) {
            if constexpr (!F32) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
            }
        };
        // One wave per SIMD (4-wave shapes, 512 registers): nothing else hides LDS latency, so the fragment reads of
        // k-step ks+1 are issued before the MFMAs of k-step ks.  With two waves per SIMD the partner wave covers it.
        constexpr bool PIPE = (NW == 4);
        frag_t a[2][TM], b[2][TN];
        if (ksn == 4 && nw_src != nullptr) {
            // Fast path (full chunk, a weight slice to stage): NO data-dependent branch inside, so the whole step is one
            // basic block and the scheduler can hoist fragment reads over MFMAs of earlier k-steps.
            if constexpr (PIPE) load_frags(0, a[0], b[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (PIPE) {
                    if (ks + 1 < 4) load_frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
                    mfmas(a[ks & 1], b[ks & 1]);
                } else {
                    load_frags(ks, a[ks & 1], b[ks & 1]);
                    mfmas(a[ks & 1], b[ks & 1]);
                }
#pragma unroll
                for (int j = 0; j < JW; ++j)
                    if ((j * 3) / JW == ks) glds16(nw_src + j * wstep, nw_buf + (wave + NW * j) * 1024);
            }
            if (x_on) nx_issued = stage_x_part(x_n0, x_cc, x_part, x_buf);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < ksn) {
                load_frags(ks, a[0], b[0]);
                mfmas(a[0], b[0]);
            }
            if (nw_src) {
#pragma unroll
                for (int j = 0; j < JW; ++j)
                    if ((j * 3) / JW == ks) glds16(nw_src + j * wstep, nw_buf + (wave + NW * j) * 1024);
            }
            if (ks == 3 && x_on) nx_issued = stage_x_part(x_n0, x_cc, x_part, x_buf);
        }
    };

    // ---------------- prologue: whole first region + first weight slice ----------------
    int m0, n0;
    tile_origin(tile, m0, n0);
    for (int part = 0; part < 8; ++part) stage_x_part(n0, 0, part, xptr(0));
    stage_w(m0, 0, 0, wptr(0));
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(0);

    int wsel = 0, xsel = 0;
    while (true) {
        tile_masks(n0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        const int next_tile = tile + G;
        const bool has_next = next_tile < ntiles;
        int m0n = 0, n0n = 0;
        if (has_next) tile_origin(next_tile, m0n, n0n);
        stamp(1);

        for (int cc = 0; cc < CC; ++cc) {
            const bool last_chunk = (cc + 1 == CC);
            const bool next_chunk_exists = !last_chunk || has_next;
            const int nc_n0 = last_chunk ? n0n : n0, nc_cc = last_chunk ? 0 : cc + 1;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                // weight slice of the next step + one eighth of the next region, issued inside compute()
                const char* nws = nullptr;
                if (tap < 8) nws = w_src(m0, tap + 1, cc);
                else if (!last_chunk) nws = w_src(m0, 0, cc + 1);
                else if (has_next) nws = w_src(m0n, 0, 0);
                compute(wptr(wsel), xptr(xsel), tap, cc, nws, wptr(wsel ^ 1), tap < 8 && next_chunk_exists, nc_n0, nc_cc, tap,
                        xptr(xsel ^ 1));
                const int nx = nx_issued;
                if constexpr (STAMP) asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
                stamp(2);
                if (!(last_chunk && tap == 8)) {
                    wait_n(nx);
                    stamp(3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    stamp(4);
                }
                wsel ^= 1;
            }
            xsel ^= 1;
        }
        // every wave is done with the last slices -> the just-consumed buffer becomes the epilogue staging area
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(5);
        char* region = STAGE_IN_W ? wptr(wsel ^ 1) : xptr(xsel ^ 1);
        epilogue_store<F32, F32, TM, TN>(p, acc, region + wave * kStgBytes, sbias, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
        stamp(6);
        if (!has_next) break;
        wait_vmcnt<NSTORE>();                 // next tile's first weight slice (and all of its first region) has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(7);
        tile = next_tile; m0 = m0n; n0 = n0n;
    }
    if constexpr (STAMP) {
        if (lane == 0 && p.debug)
            for (int i = 0; i < 8; ++i) p.debug[((long long)blockIdx.x * NW + wave) * 8 + i] = ph_sum[i];
    }
}

struct HaloConfig {
    int bm, bn, threads;
    void (*bf16)(const ConvParams);
    void (*f32)(const ConvParams);
};
#define HCFG(BM, BN, WM, WN) { BM, BN, WM * WN * 64, conv3x3_halo_kernel<false, BM, BN, WM, WN>, conv3x3_halo_kernel<true, BM, BN, WM, WN> }
const HaloConfig kHalo[] = {
    HCFG(192, 256, 2, 4),   // per-wave 96x64
    HCFG(96, 256, 1, 4),    // per-wave 96x64, 4 waves
    HCFG(64, 256, 1, 4),    // per-wave 64x64, 4 waves
    HCFG(192, 128, 2, 2),   // per-wave 96x64, 4 waves
    HCFG(384, 128, 2, 4),   // per-wave 192x32
    HCFG(128, 256, 2, 4),   // per-wave 64x64
    HCFG(96, 128, 1, 4),    // per-wave 96x32, 4 waves
    HCFG(64, 128, 1, 4),    // per-wave 64x32, 4 waves
    HCFG(256, 128, 4, 2),   // per-wave 64x64
    HCFG(192, 256, 2, 2),   // per-wave 96x128, 4 waves (one per SIMD, 512 registers): fewest LDS reads per MFMA
    HCFG(384, 128, 2, 2),   // per-wave 192x64, 4 waves
    HCFG(256, 256, 2, 4),   // per-wave 128x64
    HCFG(128, 256, 2, 2),   // per-wave 64x128, 4 waves
};
constexpr int kNumHalo = sizeof(kHalo) / sizeof(kHalo[0]);
bool g_halo_attr[kNumHalo][2];
int g_halo_cus = 0;

}  // namespace

int aq_conv_halo_num_configs() { return kNumHalo; }

int aq_conv_halo_tiles(int hcfg, int* bm, int* bn) {
    if (hcfg < 0 || hcfg >= kNumHalo) return AQ_ERR_INVALID;
    *bm = kHalo[hcfg].bm; *bn = kHalo[hcfg].bn;
    return AQ_OK;
}

int aq_launch_conv_halo(const ConvParams& p_in, int precision, int out_f32, int hcfg, bool one_tile_per_wg, hipStream_t stream) {
    if (hcfg < 0 || hcfg >= kNumHalo) { aq_set_error("halo conv: bad config %d", hcfg); return AQ_ERR_INVALID; }
    const HaloConfig& k = kHalo[hcfg];
    ConvParams p = p_in;
    if (precision == AQ_F16X3) { aq_set_error("halo conv: no split-mode (AQ_F16X3) build"); return AQ_ERR_INVALID; }
    if (p.k != 3 || p.stride != 1 || p.pad != 1 || p.H != p.Ho || p.W != p.Wo || (out_f32 && precision != AQ_FP32)) {
        aq_set_error("halo conv: only 3x3 / stride 1 / pad 1 layers with same-precision output");
        return AQ_ERR_INVALID;
    }
    if (p.npix >= (1 << 24) || p.G <= 0 || p.kgroups_pad >= (1 << 15)) { aq_set_error("halo conv: shape outside the fast-index range"); return AQ_ERR_INVALID; }
    const int nw = k.threads / 64;
    p.halo = p.W + 1;
    p.xrows = (k.bn + 2 * p.halo + 7) / 8 * 8;
    p.nixr = p.xrows / 8;
    p.xper = (p.nixr + 7) / 8;
    if (p.xper > 3 * nw) { aq_set_error("halo conv: region too large for config %d", hcfg); return AQ_ERR_INVALID; }
    p.inv_hw = 1.0f / (float)(p.H * p.W);
    p.inv_wo = 1.0f / (float)p.W;
    p.n_tiles_m = (p.cout + k.bm - 1) / k.bm;
    p.n_tiles_n = (p.npix + k.bn - 1) / k.bn;
    p.magic_ntm = (unsigned)(0x100000000ull / (unsigned)p.n_tiles_m) + 1u;
    p.bias_n = p.n_tiles_m * k.bm;
    const long long ntiles = (long long)p.n_tiles_m * p.n_tiles_n;
    if (ntiles <= 0 || ntiles * p.n_tiles_m >= (1LL << 31)) { aq_set_error("halo conv: bad tile count"); return AQ_ERR_INVALID; }
    const size_t wbuf = (size_t)k.bm * 128, xb = (size_t)p.xrows * 128;
    const size_t lds = 2 * wbuf + 2 * xb + 128 + (size_t)p.bias_n * 4;
    const size_t stg = (size_t)nw * aqdev::kStgBytes;
    if (lds > 160 * 1024 || (wbuf < stg && xb < stg)) { aq_set_error("halo conv: config %d does not fit LDS for W=%d", hcfg, p.W); return AQ_ERR_INVALID; }
    const int variant = precision == AQ_FP32 ? 1 : 0;
    auto fn = variant ? k.f32 : k.bf16;
    if (!g_halo_attr[hcfg][variant]) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_halo_attr[hcfg][variant] = true;
    }
    if (g_halo_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_halo_cus = cus;
    }
    long long grid = g_halo_cus;          // > 80 KiB of LDS per workgroup: one resident workgroup per CU
    if (lds <= 80 * 1024) grid *= 2;
    if (grid > ntiles) grid = ntiles;
    if (one_tile_per_wg) grid = ntiles;     // see aq_launch_conv
    size_t sbytes = 0;
    unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
    if (sbuf && variant == 0 && hcfg == 0 && (size_t)grid * nw * 64 <= sbytes) {
        auto sfn = conv3x3_halo_kernel<false, 192, 256, 2, 4, true>;
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)sfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p.debug = sbuf;
        hipLaunchKernelGGL(sfn, dim3((unsigned)grid), dim3(k.threads), lds, stream, p);
        AQ_CHECK_HIP(hipGetLastError());
        return AQ_OK;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(k.threads), lds, stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
