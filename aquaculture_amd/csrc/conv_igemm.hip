// Implicit-GEMM convolution for gfx950 (CDNA4), NHWC, im2col-free.
//
// Replaces Conv.forward_fuse / Bottleneck.forward of the reference's yolov5 dependency
// ([UPSTREAM models/common.py]; invoked by reference README.md:77):  out = (res +) SiLU(conv(x, W') + b').
//
//   D[cout][pixel] = sum_{tap, cin} W[cout][tap][cin] * X[pixel + tap][cin]
//
// GEMM view: M = Cout (MFMA A operand = weights), N = B*Ho*Wo output pixels (MFMA B operand = activations),
// K = taps*Cin.  Both operands are K-contiguous in HBM (weights packed [Cout][tap][Cin], activations NHWC),
// so one 16-byte "group" = 8 bf16 (4 f32) consecutive input channels is the unit of every transfer.
//
// Data movement per K chunk of 8 groups (128 B per row): every lane issues LDS-DMA loads
// (global_load_lds_dwordx4, 1 KiB per wave-instruction) with a PER-LANE SOURCE address, which is what makes
// the im2col gather free: lane (row, slot) fetches group (slot ^ swz(row)) of its pixel's tap, or 16 zero
// bytes from a zero page when the tap falls in the padding.  The LDS image is therefore lane-linear
// [row][8 slots x 16 B] with the XOR swizzle applied on the SOURCE side; fragment reads apply the same XOR
// and are ds_read_b128 bank-conflict-free (rows are 128 B; slot ^= (row >> 1) & 7).
// Two LDS buffers: chunk c+1 streams in while chunk c feeds the MFMAs.
//
// fp32 parity mode uses the same byte geometry (a 16-byte group = 4 floats) on v_mfma_f32_32x32x2_f32,
// which is bit-for-bit an fmaf chain (exact fp32 products, fp32 accumulate).
//
// AQ_F16X3 ("split" mode, round 3): fp32 activations in HBM and LDS exactly as in fp32 mode, but every product is THREE fp16 MFMAs on
// hi / lo halves -- x = xh + xl with xh = fp16(x), xl = fp16(x - xh) (22 significant bits), the weights split the same way on the
// host after a per-output-channel power-of-two scale that keeps both halves in fp16's normal range, and
// acc += wh xh + wh xl + wl xh (the dropped wl xl is 2^-22 of the product) into the fp32 accumulator.  v_mfma_f32_32x32x16_f16 runs at
// 16x the f32-input MFMA's rate, so a 32-channel chunk costs 2 x 3 x 32 cycles per 32x32 block instead of 16 x 64; the activations are
// split in registers between the LDS read and the MFMAs (five VALU instructions per pair of values).  A weight row holds, per 8
// channels, 16 bytes of hi halves then 16 bytes of lo halves -- the same bytes per channel as fp32, so the loader, the K walk and
// the LDS image are those of fp32 mode; a lane reads slots (4 ks + 2 h, + 1) of both operands: hi / lo of its 8 channels (weights),
// 2 x 4 floats of the same 8 channels (activations).  Epilogue: (acc + bias 2^s) 2^-s, which is exact, then fp32 mode's.
#include "aq_common.h"
#include <type_traits>

namespace {

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool F32>
__device__ __forceinline__ float silu(float v) {
    // [UPSTREAM nn.SiLU]: v * sigmoid(v) = v / (1 + exp(-v))
    if (F32) return v / (1.0f + expf(-v));                       // parity mode: IEEE divide, accurate exp
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));         // bf16 mode: v_exp + v_rcp (error << bf16 ulp)
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16 (round-to-nearest-even) in one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // s_waitcnt vmcnt(N): all but the wave's N youngest vector-memory operations are done (loads, LDS-DMA and
    // stores count together, in issue order).
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// workgroups per CU the LDS footprint allows (<= 4), as waves per SIMD for __launch_bounds__: keeps the register
// allocator from trading occupancy away on the small, memory-bound tile shapes.
constexpr int conv_min_waves(int bm, int bn, int nw, int nstage) {
    const int lds = nstage * (bm + bn) * 128;
    int wgs = (160 * 1024) / lds;
    wgs = wgs < 1 ? 1 : (wgs > 4 ? 4 : wgs);
    const int w = (nw / 4) * wgs;
    return w > 8 ? 8 : w;
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// x[0..7] (fp32) -> hi = fp16(x) and lo = fp16(x - hi), element-wise (v_cvt_pk_f16_f32, v_cvt_f32_f16 x 2, v_pk_add_f32, v_cvt_pk_f16_f32
// per pair).  x - float(hi) is exact in fp32 (hi keeps x's leading 11 bits), so hi + lo carries 22 bits of x.
__device__ __forceinline__ void split_f16(const f32x4& x0, const f32x4& x1, f16x8& hi, f16x8& lo) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

template <bool F32, int BM, int BN, int WM, int WN, bool OUT_F32, int NSTAGE, bool BIAS_LDS, bool STAMP = false, bool X3 = false>
__global__ __launch_bounds__(WM * WN * 64, conv_min_waves(BM, BN, WM * WN, NSTAGE)) void conv_igemm_kernel(const ConvParams p) {
    static_assert(!X3 || (F32 && OUT_F32), "split mode keeps fp32 activations");
    constexpr int NW = WM * WN;
    static_assert(NSTAGE == 2 || NSTAGE == 3, "pipeline depth");
    constexpr int ROWB = 128;                 // LDS bytes per row = one K chunk
    constexpr int NIW = BM / 8, NIX = BN / 8; // 1-KiB LDS-DMA instructions per chunk (weights, activations)
    constexpr int JW = (NIW + NW - 1) / NW, JX = NIX / NW;
    static_assert(BN % (8 * NW) == 0, "BN must split evenly over the waves");
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave tiles are 32x32 MFMA blocks");
    static_assert(NW % 2 == 0, "swizzle phase assumes an even wave count");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int BUF = (BM + BN) * ROWB;
    // epilogue staging: one 32-pixel x 32-cout fp32 block per wave at a time, rows padded by 16 B
    constexpr int SROW = 32 * 4 + 16;
    constexpr int STG = 32 * SROW;
    static_assert(NW * STG <= BUF, "epilogue staging must fit in one pipeline buffer");
    constexpr bool OUT4 = F32 || OUT_F32;                    // 4-byte outputs: two 16-B stores per item
    constexpr int NSTORE = TM * TN * 2 * (OUT4 ? 2 : 1);     // store instructions a wave issues per tile
    constexpr int LSTAGE = JW + JX;                          // LDS-DMA instructions a wave issues per stage
    static_assert(NSTAGE == 2 || NIW % NW == 0, "counted vmcnt needs the same load count in every wave");
    static_assert(NSTORE + LSTAGE <= 63, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // diagnostic build only (STAMP): per-wave cycle sums of 8 phases -> p.debug (tools/stamp_conv.py); the shipped
    // kernels contain no stamp
    unsigned long long ph_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = 0;
    if constexpr (STAMP) ph_t = clock64();
    auto stamp = [&](int ph) {
        if constexpr (STAMP) {
            const unsigned long long t = clock64();
            ph_sum[ph] += t - ph_t;
            ph_t = t;
        }
    };

    // Persistent workgroups: block w handles tiles w, w + G, w + 2G, ...  The XCD-aware, bijective remap gives the
    // blocks that share an XCD (same blockIdx % 8) consecutive w, so at any moment one L2 serves neighbouring pixel
    // tiles (shared 3x3 halo rows) and one weight panel.
    const int G = gridDim.x;
    int tile;
    {
        const int bid = blockIdx.x;
        const int q = G >> 3, r = G & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ntiles = p.n_tiles_m * p.n_tiles_n;
    if (tile >= ntiles) return;

    // bias vector -> LDS once (behind the pipeline buffers): the epilogue then issues no global loads of its own,
    // so nothing in it makes the compiler drain the in-flight LDS-DMA prefetches (vmcnt is in-order).
    // (BIAS_LDS = false for the one shape whose two resident workgroups use all 160 KiB already.)
    const float* sbias = p.bias;
    const float* sscale = p.bias + p.x3_off;                  // split mode: 2^-s per output channel, behind the (scaled) bias
    if constexpr (BIAS_LDS) {
        float* sb = (float*)(smem + NSTAGE * BUF);
        for (int i = tid; i < p.bias_n; i += NW * 64) sb[i] = p.bias[i];
        sbias = sb;
        if constexpr (X3) {
            for (int i = tid; i < p.bias_n; i += NW * 64) sb[p.bias_n + i] = p.bias[p.x3_off + i];
            sscale = sb + p.bias_n;
        }
    }

    // ---------------- loader state (for the tile being staged) ----------------
    const int lrow = lane >> 3, lslot = lane & 7;
    const int gs = lslot ^ (((wave & 1) << 2) | (lrow >> 1));   // source group of this lane's LDS slot
    long long xbase[JX];
    unsigned xmask[JX];
    const char* wsrc;
    const long long wstep = (long long)8 * NW * p.kgroups_pad * 16;   // bytes between this lane's weight rows
    const int hw = p.Ho * p.Wo;

    auto decode_tile = [&](int t, int& m0, int& n0) {
        const int tile_n = p.n_tiles_m == 1 ? t : (int)__umulhi((unsigned)t, p.magic_ntm);   // t / n_tiles_m
        const int tile_m = t - tile_n * p.n_tiles_m;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
        wsrc = p.w + ((long long)(m0 + 8 * wave + lrow) * p.kgroups_pad + gs) * 16;
        if (p.k == 1 && p.stride == 1 && p.pad == 0) {
            // 1x1 / stride 1: the input pixel IS the output pixel and no tap can fall outside the image
#pragma unroll
            for (int j = 0; j < JX; ++j) {
                const int P = n0 + 8 * (wave + NW * j) + lrow;
                xbase[j] = (long long)P * p.in_ld_b;
                xmask[j] = P < p.npix ? 1u : 0u;
            }
            return;
        }
        // General case.  The 8 lanes that load one pixel row (same lrow) need the same (base, mask); each of them
        // decodes ONE of the lane's JX pixels (lane with slot j decodes pixel j) and the results are exchanged inside
        // the 8-lane group with ds_bpermute: one decode per lane per tile instead of JX.
        // pixel -> (b, y, x) without integer division: npix < 2^24 (checked on the host) so P is exact in fp32 and a
        // reciprocal multiply is off by at most one, which the remainder test repairs.
        static_assert(JX <= 8, "one decode per lane needs JX <= 8");
        unsigned my_mask = 0;
        long long my_base = 0;
        {
            const int j = lslot < JX ? lslot : 0;
            const int P = n0 + 8 * (wave + NW * j) + lrow;
            if (P < p.npix) {
                int b = (int)((float)P * p.inv_hw);
                int rem = P - b * hw;
                if (rem < 0) { --b; rem += hw; } else if (rem >= hw) { ++b; rem -= hw; }
                int y = (int)((float)rem * p.inv_wo);
                int x = rem - y * p.Wo;
                if (x < 0) { --y; x += p.Wo; } else if (x >= p.Wo) { ++y; x -= p.Wo; }
                const int iy0 = y * p.stride - p.pad, ix0 = x * p.stride - p.pad;
                my_base = ((long long)(b * p.H + iy0) * p.W + ix0) * p.in_ld_b;
                // tap mask = (valid rows) x (valid columns), k bits each
                unsigned colbits = 0;
                for (int kx = 0; kx < p.k; ++kx) colbits |= (unsigned)((unsigned)(ix0 + kx) < (unsigned)p.W) << kx;
                for (int ky = 0; ky < p.k; ++ky)
                    if ((unsigned)(iy0 + ky) < (unsigned)p.H) my_mask |= colbits << (ky * p.k);
            }
        }
#pragma unroll
        for (int j = 0; j < JX; ++j) {
            const int srcl = (lane & ~7) | j;
            const unsigned lo = (unsigned)__shfl((int)(unsigned)(my_base & 0xffffffffll), srcl);
            const unsigned hi = (unsigned)__shfl((int)(my_base >> 32), srcl);
            xbase[j] = (long long)(((unsigned long long)hi << 32) | lo);
            xmask[j] = (unsigned)__shfl((int)my_mask, srcl);
        }
    };

    // Staging of one (tile, chunk) step is split into NL = JW + JX single-instruction slots so that the slots can be
    // issued BETWEEN the MFMA bursts of the step being computed (LDS-DMA issue is ~100 cycles per instruction; issued
    // up front by every wave at once it would leave the matrix pipe idle after each barrier).
    struct StageCtx { bool active; char* buf; int wofs; int tap; int tapoff; bool kvalid; };
    constexpr int NL = JW + JX;
    auto stage_slot = [&](const StageCtx& c, int i) {
        if (!c.active) return;
        if (i < JW) {                                          // weights: rows [0, BM)
            const int q = wave + NW * i;
            if (NIW % NW == 0 || q < NIW) glds16(wsrc + i * wstep + c.wofs, c.buf + q * 1024);
        } else {                                               // activations: rows [BM, BM+BN)
            const int j = i - JW;
            const bool ok = c.kvalid && ((xmask[j] >> c.tap) & 1u);
            const char* src = ok ? p.in + xbase[j] + c.tapoff : p.zero;
            glds16(src, c.buf + BM * ROWB + (wave + NW * j) * 1024);
        }
    };
    // This lane's source group walks the flattened K axis 8 groups per chunk; (tap, ky, kx, channel group) advance by
    // add/compare only (no division per step).  Offsets fit 32 bits: |tap offset| <= (k*W) * pixel stride.
    int k_tap = 0, k_cg = 0, k_ky = 0, k_kx = 0;
    auto stage_ctx = [&](int chunk, char* buf) -> StageCtx {
        if (chunk == 0) {
            k_tap = p.G == 1 ? gs : (int)__umulhi((unsigned)gs, p.magic_G);
            k_cg = gs - k_tap * p.G;
            k_ky = p.k == 1 ? k_tap : (int)__umulhi((unsigned)k_tap, p.magic_k);
            k_kx = k_tap - k_ky * p.k;
        } else {
            k_cg += 8;
            while (k_cg >= p.G) {
                k_cg -= p.G; ++k_tap;
                if (++k_kx == p.k) { k_kx = 0; ++k_ky; }
            }
        }
        StageCtx c;
        c.active = true; c.buf = buf; c.wofs = chunk * 128;
        c.tap = k_tap;
        c.tapoff = (k_ky * p.W + k_kx) * p.in_ld_b + k_cg * 16;
        c.kvalid = chunk * 8 + gs < p.kgroups;
        return c;
    };

    // ---------------- MFMA state ----------------
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;
    const int sw = (l31 >> 1) & 7;
    const int a_off = (wm * (BM / WM) + l31) * ROWB;
    const int b_off = BM * ROWB + (wn * (BN / WN) + l31) * ROWB;
    f32x16 acc[TM][TN];

    // Fragment reads are software-pipelined: the ds_read_b128s of k-step ks+1 are issued before the MFMAs of k-step ks
    // (second register set), so the matrix pipe does not wait on LDS latency between k-steps.
    auto compute = [&](const char* buf, const StageCtx& nx) {
        if constexpr (X3) {
            // two k-steps of 16 channels; this lane's 8 channels of a step sit in slots 4 ks + 2 h and 4 ks + 2 h + 1 of both operands
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int s0 = (((4 * ks + 2 * h) ^ sw) << 4), s1 = (((4 * ks + 2 * h + 1) ^ sw) << 4);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *(const f16x8*)(buf + a_off + i * 32 * ROWB + s0);
                    al[i] = *(const f16x8*)(buf + a_off + i * 32 * ROWB + s1);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    split_f16(*(const f32x4*)(buf + b_off + j * 32 * ROWB + s0), *(const f32x4*)(buf + b_off + j * 32 * ROWB + s1), bh[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);   // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < NL; ++i)
                    if ((i * 2) / NL == ks) stage_slot(nx, i);     // this k-step's share of the next stage's LDS-DMA
            }
            return;
        }
        using frag_t = typename std::conditional<F32, f32x4, bf16x8>::type;
        frag_t a[2][TM], b[2][TN];
        auto load_frags = [&](int ks, frag_t (&fa)[TM], frag_t (&fb)[TN]) {
            const int so = (((2 * ks + h) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const frag_t*)(buf + a_off + i * 32 * ROWB + so);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const frag_t*)(buf + b_off + j * 32 * ROWB + so);
        };
        load_frags(0, a[0], b[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
            if constexpr (!F32) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks & 1][i][e], b[ks & 1][j][e], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i)
                if ((i * 4) / NL == ks) stage_slot(nx, i);     // this k-step's share of the next stage's LDS-DMA
        }
    };

    // ---------------- epilogue: +bias, SiLU -> LDS transpose -> (+residual) -> coalesced NHWC stores -------------
    // MFMA C/D map (32x32): column (pixel) = lane & 31, row (cout) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5):
    // a lane holds 4-channel slivers of one pixel, so a direct store would touch 32 cache lines per instruction.
    // Each wave transposes one 32-cout x 32-pixel MFMA block at a time through its own LDS region (fp32, padded
    // rows) and writes 8 consecutive channels (16 B of bf16) per lane, 4 lanes per pixel.
    auto epilogue = [&](char* region, int m0, int n0) {
        char* stg = region + wave * STG;
        const int cwave = m0 + wm * (BM / WM);
        const int pix = lane >> 2, ch = lane & 3;            // item = lane (+64): pixel, 8-channel chunk
        // bf16 shortcut layers: fetch the whole residual tile of this wave up front (TM*TN*2 x 16 B per lane)
        uint4 rpre[TM][TN][2];
        if (!F32 && p.res) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int P = n0 + wn * (BN / WN) + j * 32 + pix + 16 * it, c0 = cwave + i * 32 + ch * 8;
                        rpre[i][j][it] = (P < p.npix && c0 < p.cout) ? *(const uint4*)(p.res + (long long)P * p.res_ld_b + c0 * 2)
                                                                      : make_uint4(0, 0, 0, 0);
                    }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pbase = n0 + wn * (BN / WN) + j * 32;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int cblk = cwave + i * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = 8 * g + 4 * h;
                    if (cblk + 8 * g >= p.cout) continue;      // padding rows of the last M block: nothing to compute
                    const f32x4 bv = *(const f32x4*)(sbias + cblk + cl);
                    f32x4 v;
                    if constexpr (F32) {
                        f32x4 sv = {1.0f, 1.0f, 1.0f, 1.0f};
                        if constexpr (X3) sv = *(const f32x4*)(sscale + cblk + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = acc[i][j][4 * g + e] + bv[e];
                            if constexpr (X3) t *= sv[e];          // (acc + b 2^s) 2^-s: exact
                            if (p.act) t = silu<true>(t);
                            v[e] = t;
                        }
                    } else {
                        // bf16 mode: bias add, exp argument, 1 + e and the final product as packed 2 x fp32 ops
                        const f32x4 a4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        v = a4 + bv;
                        if (p.act) {
                            const f32x4 t = v * -1.44269504f;
                            f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]),
                                       __builtin_amdgcn_exp2f(t[3])};
                            d = d + 1.0f;
                            const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]),
                                             __builtin_amdgcn_rcpf(d[3])};
                            v = v * r;
                        }
                    }
                    *(f32x4*)(stg + l31 * SROW + cl * 4) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int px = pix + 16 * it;
                    const int P = pbase + px, c0 = cblk + ch * 8;
                    const f32x4 lo = *(const f32x4*)(stg + px * SROW + ch * 32);
                    const f32x4 hi = *(const f32x4*)(stg + px * SROW + ch * 32 + 16);
                    if (P < p.npix && c0 < p.cout) {
                        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        char* orow = p.out + (long long)P * p.out_ld_b;
                        if (F32) {
                            if (p.res) {
                                const char* rrow = p.res + (long long)P * p.res_ld_b + c0 * 4;
                                const f32x4 r0 = *(const f32x4*)rrow, r1 = *(const f32x4*)(rrow + 16);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                            }
                        } else if (p.res) {
                            const uint4 rv = rpre[i][j][it];
                            const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += __uint_as_float(rw[e] << 16);
                                v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                            }
                        }
                        if (OUT4) {
                            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                            *(f32x4*)(orow + c0 * 4) = o0;
                            *(f32x4*)(orow + c0 * 4 + 16) = o1;
                        } else {
                            uint4 o;
                            o.x = pack_bf16x2(v[0], v[1]);
                            o.y = pack_bf16x2(v[2], v[3]);
                            o.z = pack_bf16x2(v[4], v[5]);
                            o.w = pack_bf16x2(v[6], v[7]);
                            *(uint4*)(orow + c0 * 2) = o;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    // ---------------- flattened (tile, chunk) pipeline over NSTAGE LDS buffers ----------------
    // Step s = (tile, chunk) in this workgroup's order.  While step s feeds the MFMAs, steps s+1 .. s+NSTAGE-1 stream
    // in by LDS-DMA (a staging cursor runs NSTAGE-1 steps ahead, straight across tile boundaries), so a tile's
    // epilogue overlaps the next tile's first chunks.  Waits are counted: vmcnt leaves the younger stage (and the
    // epilogue's NSTORE stores) outstanding, never draining the queue inside the loop.
    int st_tile = tile, st_chunk = 0, st_step = 0;   // staging cursor
    int sm0, sn0;
    auto stage_begin = [&]() -> StageCtx {
        StageCtx c; c.active = false;
        if (st_tile >= ntiles) return c;
        if (st_chunk == 0) decode_tile(st_tile, sm0, sn0);
        c = stage_ctx(st_chunk, smem + (st_step % NSTAGE) * BUF);
        ++st_step;
        if (++st_chunk == p.nchunks) { st_chunk = 0; st_tile += G; }
        return c;
    };
    // wait until step `need` has landed in this wave's view: at most (st_step - need - 1) younger stages (+ extra
    // younger non-stage operations) may stay in flight
    auto wait_step = [&](int need, auto extra_tag) {
        constexpr int EXTRA = decltype(extra_tag)::value;
        const int ahead = st_step - need - 1;
        if (NSTAGE == 3 && ahead >= 1) wait_vmcnt<LSTAGE + EXTRA>();
        else wait_vmcnt<EXTRA>();
    };
    using Tag0 = std::integral_constant<int, 0>;
    using TagS = std::integral_constant<int, NSTORE>;

#pragma unroll
    for (int d = 0; d < NSTAGE - 1; ++d) {
        const StageCtx c = stage_begin();
#pragma unroll
        for (int i = 0; i < NL; ++i) stage_slot(c, i);
    }
    wait_step(0, Tag0{});
    __builtin_amdgcn_s_barrier();
    stamp(0);
    int step = 0;
    for (; tile < ntiles; tile += G) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        char* cur = smem;
        for (int c = 0; c < p.nchunks; ++c, ++step) {
            cur = smem + (step % NSTAGE) * BUF;
            const StageCtx nx = stage_begin();             // step + NSTAGE - 1 (its buffer was freed by the last barrier)
            stamp(1);
            compute(cur, nx);                              // ... loaded in slices between this step's MFMA bursts
            if constexpr (STAMP) asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
            stamp(2);
            if (c + 1 < p.nchunks) {
                wait_step(step + 1, Tag0{});
                stamp(3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stamp(4);
            }
        }
        // every wave is done reading `cur` -> it becomes the epilogue staging area (raw barrier: no vmcnt drain)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(5);
        const int tile_n = p.n_tiles_m == 1 ? tile : (int)__umulhi((unsigned)tile, p.magic_ntm);
        epilogue(cur, (tile - tile_n * p.n_tiles_m) * BM, tile_n * BN);
        stamp(6);
        if (tile + G >= ntiles) break;
        wait_step(step, TagS{});                           // the next tile's first chunk has landed (this wave's part)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... and everyone's part; staging reads are finished too
        stamp(7);
    }
    if constexpr (STAMP) {
        if (lane == 0 && p.debug)
            for (int i = 0; i < 8; ++i) p.debug[((long long)blockIdx.x * NW + wave) * 8 + i] = ph_sum[i];
    }
}

struct ConvConfig {
    int bm, bn, threads, tm, nstage, bias_lds;
    void (*bf16)(const ConvParams);
    void (*bf16_f32out)(const ConvParams);
    void (*f32)(const ConvParams);
    void (*x3)(const ConvParams);      // AQ_F16X3: fp32 activations, products as three fp16 MFMAs on hi / lo halves
};

#define CFG(BM, BN, WM, WN, NS, BL)                                                                    \
    { BM, BN, WM * WN * 64, BM / WM / 32, NS, BL, conv_igemm_kernel<false, BM, BN, WM, WN, false, NS, BL>, \
      conv_igemm_kernel<false, BM, BN, WM, WN, true, NS, BL>, conv_igemm_kernel<true, BM, BN, WM, WN, true, NS, BL>,      \
      conv_igemm_kernel<true, BM, BN, WM, WN, true, NS, BL, false, true> }

const ConvConfig kConfigs[] = {
    CFG(256, 256, 2, 4, 2, true),   // 0: per-wave 128x64
    CFG(192, 256, 2, 4, 2, true),   // 1: per-wave  96x64
    CFG(128, 256, 2, 4, 2, true),   // 2: per-wave  64x64
    CFG(96, 512, 1, 8, 2, true),    // 3: per-wave  96x64
    CFG(64, 512, 1, 8, 2, true),    // 4: per-wave  64x64
    CFG(32, 512, 1, 8, 2, true),    // 5: per-wave  32x64  (detect heads, cout padded to 32)
    CFG(192, 128, 2, 4, 2, false),   // 6: per-wave  96x32  (small-M layers: more tiles)
    CFG(128, 128, 2, 2, 2, true),   // 7: per-wave  64x64, 4 waves
    CFG(64, 256, 1, 4, 2, true),    // 8: per-wave  64x64, 4 waves
    CFG(96, 256, 1, 4, 2, true),    // 9: per-wave  96x64, 4 waves
    CFG(64, 128, 1, 4, 2, true),    // 10: per-wave 64x32, 4 waves
    // 3-stage pipelines (two K chunks in flight)
    CFG(192, 128, 2, 4, 3, true),   // 11
    CFG(128, 256, 2, 4, 3, true),   // 12
    CFG(128, 128, 2, 2, 3, true),   // 13
    CFG(64, 256, 1, 4, 3, true),    // 14
    CFG(96, 256, 1, 4, 3, true),    // 15
    CFG(64, 128, 1, 4, 3, true),    // 16
    CFG(256, 128, 4, 2, 3, true),   // 17: per-wave 64x64
    CFG(192, 192, 2, 2, 3, true),   // 18: per-wave 96x96, 4 waves
    CFG(192, 256, 2, 2, 2, true),   // 19: per-wave 96x128, 4 waves (one per SIMD, 512 registers)
    CFG(128, 256, 2, 2, 3, true),   // 20: per-wave 64x128, 4 waves, 3 stages
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

// diagnostic (stamped) builds of a few shapes, bf16 only: used when aq_debug_conv_stamp() armed a buffer
struct StampedKernel { int cfg; void (*fn)(const ConvParams); };
const StampedKernel kStamped[] = {
    {1, conv_igemm_kernel<false, 192, 256, 2, 4, false, 2, true, true>},
    {6, conv_igemm_kernel<false, 192, 128, 2, 4, false, 2, false, true>},
    {7, conv_igemm_kernel<false, 128, 128, 2, 2, false, 2, true, true>},
    {10, conv_igemm_kernel<false, 64, 128, 1, 4, false, 2, true, true>},
};
unsigned long long* g_stamp_buf = nullptr;
size_t g_stamp_bytes = 0;

// dynamic LDS: NSTAGE K-chunk buffers (the epilogue staging lives inside the just-consumed one)
size_t conv_lds_bytes(const ConvConfig& k, int bias_n) { return (size_t)k.nstage * (k.bm + k.bn) * 128 + (k.bias_lds ? (size_t)bias_n * 4 : 0); }
bool g_attr_set[kNumConfigs][4];
struct OccEntry { size_t lds; int blocks; };
OccEntry g_occ[kNumConfigs][4][8];   // resident blocks per CU by dynamic-LDS size (a handful of sizes per kernel)
int g_num_cus = 0;

}  // namespace

// one index space for tuning: [0, kNumConfigs) = implicit-GEMM tiles, then the halo-reuse 3x3 kernel's tiles
unsigned long long* aq_stamp_buffer(size_t* bytes) { if (bytes) *bytes = g_stamp_bytes; return g_stamp_buf; }
const char* aq_zero_page() {
    static void* pages[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        void* q = nullptr;
        if (hipMalloc(&q, 256) != hipSuccess || hipMemset(q, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = q;
    }
    return (const char*)pages[dev];
}
extern "C" int aq_debug_conv_stamp(void* buf_dev, size_t bytes) {
    g_stamp_buf = (unsigned long long*)buf_dev;
    g_stamp_bytes = buf_dev ? bytes : 0;
    return AQ_OK;
}

extern "C" int aq_conv_num_configs(void) { return kNumConfigs + aq_conv_halo_num_configs(); }

extern "C" int aq_conv_config_tiles(int cfg, int* bm, int* bn) {
    if (cfg >= 0) cfg &= ~AQ_CONV_CFG_ONE_TILE_PER_WG;
    if (cfg >= kNumConfigs) return aq_conv_halo_tiles(cfg - kNumConfigs, bm, bn);
    if (cfg < 0) return AQ_ERR_INVALID;
    *bm = kConfigs[cfg].bm;
    *bn = kConfigs[cfg].bn;
    return AQ_OK;
}

// Heuristic tile choice: fit Cout without waste, then prefer the tile that fills 256 CUs with the least tail.
int aq_conv_pick_config(int cout, int npix, int precision) {
    (void)precision;
    int best = -1;
    double best_cost = 1e30;
    for (int c = 0; c < kNumConfigs; ++c) {
        const ConvConfig& k = kConfigs[c];
        const int tm = (cout + k.bm - 1) / k.bm, tn = (npix + k.bn - 1) / k.bn;
        const double waste = (double)(tm * k.bm) / cout;              // padded MFMA rows
        const int lds = (int)conv_lds_bytes(k, tm * k.bm);
        if (lds > 160 * 1024) continue;
        const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
        const double slots = 256.0 * wg_per_cu;
        const double tiles = (double)tm * tn;
        const double rounds = (double)((long long)((tiles + slots - 1) / slots));
        const double tail = rounds * slots / tiles;                    // >= 1
        // bigger tiles amortise LDS-DMA issue; score = relative time estimate
        const double eff = (k.bm * k.bn >= 192 * 256) ? 1.0 : (k.bm * k.bn >= 128 * 256 ? 1.1 : (k.bm * k.bn >= 64 * 256 ? 1.3 : 1.6));
        const double cost = waste * tail * eff;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

int aq_launch_conv(const ConvParams& p_in, int precision, int out_f32, int cfg_in, hipStream_t stream) {
    // AQ_CONV_CFG_ONE_TILE_PER_WG: grid = tile count, every workgroup computes one tile and exits, so tiles go to CUs in the order
    // CUs become free (within the launch and, with several batches in flight, across launches of different streams).  The
    // default grid is persistent: one or two resident workgroups per CU walk a static stride of tiles and prefetch the next tile
    // under the current epilogue -- better alone when the tile count divides evenly, worse when it leaves a ragged last round.
    const bool one_tile_per_wg = cfg_in >= 0 && (cfg_in & AQ_CONV_CFG_ONE_TILE_PER_WG) != 0;
    const int cfg = cfg_in >= 0 ? (cfg_in & ~AQ_CONV_CFG_ONE_TILE_PER_WG) : cfg_in;
    if (cfg >= kNumConfigs) return aq_launch_conv_halo(p_in, precision, out_f32, cfg - kNumConfigs, one_tile_per_wg, stream);
    if (cfg < 0) { aq_set_error("conv: bad config %d", cfg); return AQ_ERR_INVALID; }
    const ConvConfig& k = kConfigs[cfg];
    ConvParams p = p_in;
    if (p.npix >= (1 << 24) || p.kgroups_pad >= (1 << 15) || p.G <= 0 || p.G >= (1 << 15) || p.k <= 0) {
        aq_set_error("conv: shape outside the fast-index range (npix=%d kgroups=%d)", p.npix, p.kgroups_pad);
        return AQ_ERR_INVALID;
    }
    p.inv_hw = 1.0f / (float)(p.Ho * p.Wo);
    p.inv_wo = 1.0f / (float)p.Wo;
    p.magic_G = (unsigned)(0x100000000ull / (unsigned)p.G) + 1u;   // floor(n * magic / 2^32) == n / G for n < 2^15
    p.magic_k = (unsigned)(0x100000000ull / (unsigned)p.k) + 1u;
    p.n_tiles_m = (p.cout + k.bm - 1) / k.bm;
    p.n_tiles_n = (p.npix + k.bn - 1) / k.bn;
    const int variant = precision == AQ_F16X3 ? 3 : precision == AQ_FP32 ? 2 : (out_f32 ? 1 : 0);
    auto fn = variant == 3 ? k.x3 : variant == 2 ? k.f32 : (variant == 1 ? k.bf16_f32out : k.bf16);
    p.bias_n = ((p.cout + k.bm - 1) / k.bm) * k.bm;   // every tile row has a bias slot (bias buffer is zero padded)
    if (variant == 3 && (p.x3_off < p.bias_n || (p.G % 2) != 0)) {
        aq_set_error("conv: split mode needs the packer's bias / scale buffer and Cin %% 8 == 0");
        return AQ_ERR_INVALID;
    }
    const size_t lds = conv_lds_bytes(k, variant == 3 ? 2 * p.bias_n : p.bias_n);
    if (lds > 160 * 1024) { aq_set_error("conv: config %d needs %zu B of LDS", cfg, lds); return AQ_ERR_INVALID; }
    if (!g_attr_set[cfg][variant]) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_attr_set[cfg][variant] = true;
    }
    const long long ntiles = (long long)p.n_tiles_m * p.n_tiles_n;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) { aq_set_error("conv: bad tile count %lld", ntiles); return AQ_ERR_INVALID; }
    p.magic_ntm = (unsigned)(0x100000000ull / (unsigned)p.n_tiles_m) + 1u;
    if (ntiles * p.n_tiles_m >= (1LL << 31)) { aq_set_error("conv: too many tiles"); return AQ_ERR_INVALID; }
    // persistent grid: as many workgroups as stay resident (occupancy query once per kernel), capped by the tile count
    if (g_num_cus == 0) {
        int dev = 0, cus = 256;
        AQ_CHECK_HIP(hipGetDevice(&dev));
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_num_cus = cus;
    }
    int blocks = 0;
    OccEntry* occ = g_occ[cfg][variant];
    for (int i = 0; i < 8; ++i) {
        if (occ[i].blocks && occ[i].lds == lds) { blocks = occ[i].blocks; break; }
        if (!occ[i].blocks || i == 7) {
            int nb = 0;
            AQ_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, k.threads, lds));
            occ[i].lds = lds; occ[i].blocks = blocks = nb > 0 ? nb : 1;
            break;
        }
    }
    long long grid = (long long)g_num_cus * blocks;
    if (grid > ntiles) grid = ntiles;
    if (one_tile_per_wg) grid = ntiles;     // hardware dispatch order instead of the static persistent split
    if (g_stamp_buf && variant == 0) {
        for (const StampedKernel& sk : kStamped)
            if (sk.cfg == cfg && (size_t)grid * (k.threads / 64) * 64 <= g_stamp_bytes) {
                AQ_CHECK_HIP(hipFuncSetAttribute((const void*)sk.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                p.debug = g_stamp_buf;
                hipLaunchKernelGGL(sk.fn, dim3((unsigned)grid), dim3(k.threads), lds, stream, p);
                AQ_CHECK_HIP(hipGetLastError());
                return AQ_OK;
            }
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(k.threads), lds, stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
