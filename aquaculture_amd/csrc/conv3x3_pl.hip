// 3x3 / stride 1 / pad 1 convolution, bf16, for the wide Bottleneck layers (Cin = Cout = 192, 384 in yolov5m), gfx950.
//
// Same operation as conv3x3_halo_kernel (out = (res +) SiLU(conv(x, W') + b'), NHWC in and out;
// [UPSTREAM models/common.py Bottleneck.cv2 / Conv.forward_fuse], reached through reference README.md:77), rebuilt around
// what round 1's ablation found: the per-(tap, chunk) workgroup barrier, the per-step address arithmetic and the epilogue
// did not overlap with the MFMAs.  Structure:
//
//   * 4 waves, one per SIMD, the whole 512-register file each.  The waves split M: wave w owns output channels
//     [48 w, 48 w + 48) of a 192-row M tile for ALL pixels of the tile (NB blocks of 16 pixels), on
//     v_mfma_f32_16x16x32_bf16 (3 x NB accumulators).
//   * WEIGHTS NEVER TOUCH LDS: they are packed on the host in MFMA A-fragment order, one contiguous 6 KB stream per
//     (wave, tap-step), and each wave loads its own fragments straight from L2 into registers two tap-steps ahead
//     (no sharing between waves, so nothing to stage).
//   * LDS holds only the INPUT REGION of the pixel tile, slot-major: for a 64-channel chunk, 8 planes (one per 16-byte
//     channel group) of 384 region rows x 16 B.  16 lanes reading 16 consecutive pixels of one plane hit 256 contiguous
//     bytes, so every ds_read_b128 is conflict-free for every tap shift WITHOUT a swizzle, and tap, k-step and pixel
//     block are plain address offsets.  Region rows are PADDED pixel coordinates: one zero pixel after every image row
//     and one zero row after every image, so a tap that falls outside the image reads zeros with no mask.  The padding
//     exists only in LDS: the LDS-DMA's per-lane source address is the NHWC pixel (or a zero page).
//   * three chunk buffers form a ring: while chunk g is computed, chunk g + 2 (possibly of the NEXT tile) is loading.
//     ONE workgroup barrier per chunk (9 taps x 2 k-steps x 3 NB MFMAs per wave), not per (tap, chunk).
//   * every vector-memory operation of the loop (LDS-DMA, weight loads, residual loads, output stores) is inline asm
//     with hand-counted s_waitcnt vmcnt(N): the compiler never sees them, so it never drains them (round 1's vmcnt(0)
//     findings); fragment reads and MFMAs are ordinary code that it schedules and pads for hazards.
//
// Round 3: the generated assembly (gen_conv3x3_pl_asm.py) has a second region layout for the same kernel -- families pm13w20 / pm13w40,
// chosen by pl_conv() for 20- and 40-wide images: two half-planes of 64-byte PIXEL-MAJOR rows per chunk, loaded 16 rows per instruction
// through a buffer descriptor (out-of-range lanes write the zero padding), channel group q at position q ^ 2 b with b = bit 2 of the
// pixel's column.  The slot-major planes above cost one cache-line look-up per LANE of every LDS-DMA instruction, and that -- not the
// bytes -- was what the stream waited for (point 7).  This HIP-source kernel keeps the slot-major layout.
#include "conv_device.h"
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

using namespace aqdev;

namespace {

constexpr int PL_ROWS = 384;                 // region rows per plane: 6 LDS-DMA groups of 64 rows
constexpr int PL_PS = PL_ROWS * 16;          // plane stride (a multiple of 256 B: keeps the 16-lane groups on distinct banks)
constexpr int PL_CHUNK = 8 * PL_PS;          // one 64-channel chunk of the region
constexpr int PL_BIAS = 3 * PL_CHUNK;        // bias follows the three ring buffers (4 KB: 256 floats per wave)
constexpr int PL_BM = 192;                   // M tile: 4 waves x 3 blocks of 16 rows
constexpr int PL_STEP_B = 6 * 1024;          // weight bytes per (wave, tap-step): 2 k-steps x 3 M blocks x 1 KB

struct PlParams {
    const char* in;            // first channel of the input slice
    long long in_sp, in_ss;    // byte strides: per pixel, per 16-byte channel group (NHWC: row length, 16)
    char* out;
    const char* res;
    const char* w;             // aq_pack_conv3x3_pl image
    const float* bias;
    const char* zero;          // >= 16 zero bytes
    int out_ld_b, res_ld_b;
    int B, H, W, npix, cout, act;
    int CC;                    // 64-channel chunks (Cin / 64)
    int n_mt, ntiles;
    float inv_hw, inv_w, inv_hpwp, inv_wp;
    unsigned long long* debug;   // stamped diagnostic build (ABL & 16) only: 8 x uint64 per wave
};

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1, so every register-array index is a constant
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

__device__ __forceinline__ void pl_dma16(const char* gsrc, unsigned lds_dst) {
    // LDS-DMA the compiler does not see: 16 B per lane from a per-lane address to lds_dst + lane * 16
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst));
}
template <int OFF>
__device__ __forceinline__ void pl_ldw(bf16x8& d, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pl_ld8(u32x2& d, unsigned voff, const char* sbase) {
    // destination in the accumulator half of the register file ("a"): 3 NB of these are in flight through a tile's last chunk, and
    // the arch VGPRs are full of fragments there
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pl_st8(unsigned voff, char* sbase, u32x2 v, unsigned long long mask) {
    // always issued (vmcnt bookkeeping is by count); lanes outside `mask` are switched off
    unsigned long long save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %3, %4 offset:%5\n\ts_nop 0\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}
// All but the wave's N youngest vector-memory operations are done.  The wait itself names no register, so that a wait chosen at run
// time (two immediates behind a wave-uniform branch) does not make the compiler copy registers where the paths merge; pl_tie_a after
// the merge is what orders the users of the six A fragments behind it (volatile asm statements keep their order).
template <int N>
__device__ __forceinline__ void pl_wait() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ void pl_tie_a(bf16x8 (&a)[6]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
}
template <int I, int CNT>
__device__ __forceinline__ void pl_tie(u32x2 (&r)[CNT]) {
    if constexpr (I < CNT) {
        asm volatile("" : "+v"(r[I]));
        pl_tie<I + 1, CNT>(r);
    }
}
template <int N, int CNT>
__device__ __forceinline__ void pl_wait_r(u32x2 (&r)[CNT]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
    pl_tie<0, CNT>(r);
}

// ABL: timing-only ablation builds (results are wrong): 1 = no weight loads in the chunk body, 2 = no LDS-DMA in the chunk body,
// 4 = no B fragment reads, 8 = no epilogue arithmetic (tools/time_conv3x3.py --abl); 16 = stamped build (correct results): per-wave
// cycle sums of 0 prologue, 1 chunk barrier, 2 element stream, 3 tile set-up, 4 epilogue, 5 chunk top; 6 = wave lifetime in cycles, 7 = in 100 MHz ticks
template <int NB, bool RES, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_pl_kernel(const PlParams p) {
    constexpr int BN = NB * 16;
    constexpr int PD = 8;                        // B fragments in flight ahead of their MFMAs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    float* sbias = (float*)(smem + PL_BIAS);
    const int G = gridDim.x;
    int tile = first_tile(G, blockIdx.x);
    if (tile >= p.ntiles) return;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = 0, ph_t0 = 0, ph_r0 = 0;
    if constexpr (ABL & 16) { ph_t0 = ph_t = clock64(); ph_r0 = __builtin_amdgcn_s_memrealtime(); }
    auto stamp = [&](int k) {
        if constexpr (ABL & 16) {
            const unsigned long long t = clock64();
            ph[k] += t - ph_t;
            ph_t = t;
        }
    };

    const int HW = p.H * p.W, Wp = p.W + 1, HpWp = (p.H + 1) * Wp, lead = p.W + 2;
    const unsigned smem_base = (unsigned)(uintptr_t)smem;
    const long long zoff = p.zero - p.in;        // the zero page as an offset from the input (a select between offsets, not a branch)

    // pixel index -> padded coordinate (one zero pixel after each image row, one zero row after each image, `lead` zeros in front)
    auto pp_of = [&](int P) -> int {
        // reciprocal multiply + one correction step, branch-free (exact for P < 2^23)
        int b = (int)((float)P * p.inv_hw);
        int rem = P - b * HW;
        const int lo = rem < 0, hi = rem >= HW;
        b += hi - lo; rem += (lo - hi) * HW;
        int y = (int)((float)rem * p.inv_w);
        const int x = rem - y * p.W;
        y += (int)(x >= p.W) - (int)(x < 0);
        return P + lead + b * (p.H + p.W + 1) + y;
    };
    // padded coordinate -> pixel index, or -1 for padding / outside the batch
    auto unpad = [&](int Pp) -> int {
        const int Pq = Pp - lead;
        int b = (int)((float)Pq * p.inv_hpwp);
        int rem = Pq - b * HpWp;
        const int lo = rem < 0, hi = rem >= HpWp;
        b += hi - lo; rem += (lo - hi) * HpWp;
        int y = (int)((float)rem * p.inv_wp);
        int x = rem - y * Wp;
        const int xl = x < 0, xh = x >= Wp;
        y += xh - xl; x += (xl - xh) * Wp;
        const bool ok = (Pq >= 0) & (b < p.B) & (x < p.W) & (y < p.H);
        return ok ? b * HW + y * p.W + x : -1;
    };

    int prow[6];                  // source pixel of region row 64 k + lane, for the tile the LDS-DMA is currently loading
    auto region_rows = [&](int t_) {
        const int rs = pp_of((t_ / p.n_mt) * BN) - lead;
        static_for<6>([&](auto K) { prow[K] = unpad(rs + 64 * K + lane); });
    };
    auto no_rows = [&]() { static_for<6>([&](auto K) { prow[K] = -1; }); };
    // the two LDS-DMA instructions of slot k (0..5) of a chunk: planes 2 w and 2 w + 1, rows 64 k .. 64 k + 63
    auto dma_one = [&](auto K, auto S2, int cd, int bd) {
        const int P = prow[K];
        const int s = 2 * wave + S2;
        const long long goff = (long long)(8 * cd + s) * p.in_ss + (long long)max(P, 0) * p.in_sp;
        const long long m = (long long)(P >> 31);                 // all ones for a padding row: take the zero page
        pl_dma16(p.in + ((goff & ~m) | (zoff & m)), smem_base + bd * PL_CHUNK + s * PL_PS + K * 1024);
    };
    auto dma_pair = [&](auto K, int cd, int bd) {
        dma_one(K, std::integral_constant<int, 0>(), cd, bd);
        dma_one(K, std::integral_constant<int, 1>(), cd, bd);
    };
    const unsigned aoff = lane * 16;
    auto a_base = [&](int t_, int c) -> const char* {     // weight stream of (this wave, M tile of tile t_, chunk c, tap 0), biased by 3 KB
        const int mt = t_ % p.n_mt;
        return p.w + ((long long)((mt * 4 + wave) * p.CC + c) * 9) * PL_STEP_B + 3072;
    };
    auto load_a = [&](bf16x8 (&a)[6], const char* sb) {
        pl_ldw<-3072>(a[0], aoff, sb); pl_ldw<-2048>(a[1], aoff, sb); pl_ldw<-1024>(a[2], aoff, sb);
        pl_ldw<0>(a[3], aoff, sb); pl_ldw<1024>(a[4], aoff, sb); pl_ldw<2048>(a[5], aoff, sb);
    };

    bf16x8 A0[6], A1[6], A2[6];   // weight fragments of tap-steps T, T + 1, T + 2 (set = tap % 3)
    f32x4 acc[3][NB];
    int addr[NB];                 // byte offset of this lane's B fragment for (pixel block j, tap (-1,-1), k-step 0) in ring buffer 0
    u32x2 rA[NB], rB[NB];         // residual batches (one M block x NB pixel blocks each)

    // ---------------- prologue: chunks 0 and 1 of the first tile, weights of tap-steps 0 and 1 ----------------
    load_a(A0, a_base(tile, 0));              // first: they depend on nothing
    load_a(A1, a_base(tile, 0) + PL_STEP_B);
    region_rows(tile);
    static_for<6>([&](auto K) { dma_pair(K, 0, 0); });
    {   // the bias, 256 floats per wave, by LDS-DMA as well (one instruction per wave: the counts below stay per-wave constants)
        const int f0 = wave * 256 + lane * 4;
        pl_dma16(f0 + 4 <= p.cout ? (const char*)(p.bias + f0) : p.zero, smem_base + PL_BIAS + wave * 1024);
    }
    static_for<6>([&](auto K) { dma_pair(K, 1, 1); });
    // all but chunk 1's LDS-DMA: chunk 0, the bias and the weights of taps 0 and 1 are in; the accumulators start from the bias
    wait_vmcnt<12>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bool first = true;
    int buf = 0;                  // ring buffer of the chunk being computed

    while (true) {
        const int nt = tile / p.n_mt, mt = tile - nt * p.n_mt;
        const int n0 = nt * BN;
        const int next_tile = tile + G;
        const bool has_next = next_tile < p.ntiles;
        const int cbase = mt * PL_BM + wave * 48;
        {
            const int rs = pp_of(n0) - lead;
            static_for<NB>([&](auto J) {
                const int P = min(n0 + 16 * J + l15, p.npix - 1);
                addr[J] = q * PL_PS + (pp_of(P) - rs - Wp - 1) * 16;
            });
        }
        if (!first) stamp(4);
        static_for<3>([&](auto I) {         // the accumulators start from the bias (one add per output less in the epilogue)
            const f32x4 bv = *(const f32x4*)(sbias + cbase + 16 * I + 4 * q);
            static_for<NB>([&](auto J) { acc[I][J] = bv; });
        });

        for (int c = 0; c < p.CC; ++c) {
            const bool lastc = c + 1 == p.CC;
            // chunk c has landed in every wave's view (own DMA: covered by the counted waits of the previous chunk, or the
            // explicit one below for the very first chunk; other waves': the barrier), and every wave is done with the
            // buffer that chunk c + 2 will overwrite
            if (c == 0 && !first) stamp(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(first && c == 0 ? 0 : 1);
            // target of this chunk's LDS-DMA: chunk c + 2 of this tile, or chunk c + 2 - CC of the next one
            int cd = c + 2, bd = buf + 2;
            if (bd >= 3) bd -= 3;
            if (cd >= p.CC) {
                cd -= p.CC;
                if (cd == 0) { if (has_next) region_rows(next_tile); else no_rows(); }
            }
            const char* a_cur = a_base(tile, c);
            const char* a_nxt = !lastc ? a_base(tile, c + 1) : a_base(has_next ? next_tile : tile, 0);
            const int bufoff = buf * PL_CHUNK;
            if constexpr (RES) {
                // residual batch 0, fetched at the top of EVERY chunk so that the wait counts of the chunk body are constants (the
                // last chunk's copy is the one the epilogue reads; pl_tie below holds the registers until every copy has landed:
                // the destination of an asm load whose value is never read is free to the compiler from the next instruction on)
                static_for<NB>([&](auto J) {
                    const int P = min(n0 + 16 * J + l15, p.npix - 1);
                    pl_ld8<0>(rA[J], (unsigned)(P * p.res_ld_b + (cbase + 4 * q) * 2), p.res);
                });
            }
            // One stream of 18 NB elements per chunk: element n = (tap t, k-step ks, pixel block j) is ONE B fragment read from the
            // region and three MFMAs (the wave's three M blocks).  The read runs PD elements ahead of its MFMAs through a small
            // rotating set of registers; all indices are compile-time constants (the chunk body is fully unrolled).
            constexpr int NE = 18 * NB;
            bf16x8 bq[PD + 1];
            int so[9];                  // ring buffer + tap shift; opaque to the optimiser so that addr[j] + so[t] is not pre-added per tap and tile (117 registers)
            static_for<9>([&](auto T) {
                int v = bufoff + ((T / 3) * Wp + (T % 3)) * 16;
                asm volatile("" : "+s"(v));
                so[T] = v;
            });
            auto b_read = [&](auto N_) -> bf16x8 {
                constexpr int n = N_, h = n / NB, j = n - h * NB, t = h >> 1, ks = h & 1;
                return *(const bf16x8*)(smem + (addr[j] + so[t]) + ks * 4 * PL_PS);
            };
            static_for<PD>([&](auto N_) { bq[N_] = b_read(N_); });
            stamp(5);
            static_for<NE>([&](auto N_) {
                constexpr int n = N_, h = n / NB, j = n - h * NB, t = h >> 1, ks = h & 1;
                if constexpr (n + PD < NE && !(ABL & 4)) bq[(n + PD) % (PD + 1)] = b_read(std::integral_constant<int, n + PD>());
                bf16x8 (&Acur)[6] = t % 3 == 0 ? A0 : (t % 3 == 1 ? A1 : A2);
                // The tap-step's vector-memory instructions go out ONE PER ELEMENT among the MFMAs (a burst of eight stalls the wave
                // at the texture addresser's queue with the matrix pipe idle): element 0 waits for this tap's weights, elements
                // 1, 3, .. 11 load one fragment each of tap-step T + 2, two later elements issue the LDS-DMA pair (taps 0..5).
                constexpr int e = ks * NB + j;                       // element index inside the tap-step, 0 .. 2 NB - 1
                constexpr int kD0 = 12, kD1 = 12 + (2 * NB - 12) / 2;
                static_assert(2 * NB >= 14 && kD1 < 2 * NB && kD1 > kD0, "issue positions");
                if constexpr (e == 0) {
                    // younger than the fragments of tap T (issued in tap T - 2): that tap's LDS-DMA pair, tap T - 1's six loads and
                    // pair, and for taps 0 and 1 the residual batch issued at the top of the chunk.  No count depends on a run-time
                    // condition; after a tile seam the epilogue's loads and stores are younger too and ignoring them only waits longer.
                    constexpr int kN = 6 + (t >= 2 && t <= 7 ? 2 : 0) + (t >= 1 && t <= 6 ? 2 : 0) + (RES && t < 2 ? NB : 0);
                    if constexpr (ABL & 3) pl_wait<0>(); else pl_wait<kN>();
                    pl_tie_a(Acur);
                }
                if constexpr (e >= 1 && e <= 11 && (e & 1) && !(ABL & 1)) {
                    constexpr int k = e >> 1;                        // fragment 0..5 of tap-step T + 2
                    bf16x8 (&Anew)[6] = (t + 2) % 3 == 0 ? A0 : ((t + 2) % 3 == 1 ? A1 : A2);
                    const char* sb = t + 2 < 9 ? a_cur + (t + 2) * PL_STEP_B : a_nxt + (t + 2 - 9) * PL_STEP_B;
                    pl_ldw<1024 * k - 3072>(Anew[k], aoff, sb);
                }
                if constexpr (t < 6 && (e == kD0 || e == kD1) && !(ABL & 2)) dma_one(std::integral_constant<int, (t < 6 ? t : 0)>(), std::integral_constant<int, (e == kD0 ? 0 : 1)>(), cd, bd);
                static_for<3>([&](auto I) {
                    acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Acur[3 * ks + I], bq[n % (PD + 1)], acc[I][j], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);      // keep this order: the read of element n + PD, the issue slot, three MFMAs
            });
            if constexpr (ABL & 16) asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[2][NB - 1][3]));
            stamp(2);
            if constexpr (RES) pl_tie<0, NB>(rA);      // landed by now (older than the last tap-steps' counted waits), and never free in between
            buf = buf == 2 ? 0 : buf + 1;
            first = false;
        }

        // ---------------- epilogue: SiLU, (+residual), bf16, 8-byte stores; one M block (NB pixel blocks) at a time ----------------
        // (the bias is already in the accumulators).  Addresses are a scalar base + a 32-bit per-lane offset + an immediate per block.
        unsigned long long omask[NB];
        unsigned oo[NB], ro[NB];
        static_for<NB>([&](auto J) {
            omask[J] = __ballot(n0 + 16 * J + l15 < p.npix);
            const int P = min(n0 + 16 * J + l15, p.npix - 1);
            oo[J] = (unsigned)(P * p.out_ld_b + (cbase + 4 * q) * 2);
            if constexpr (RES) ro[J] = (unsigned)(P * p.res_ld_b + (cbase + 4 * q) * 2);
        });
        auto epilogue = [&](auto ACT) {
            static_for<3>([&](auto I) {
                constexpr int i = I;
                u32x2 (&rcur)[NB] = (i & 1) ? rB : rA;
                u32x2 (&rnxt)[NB] = (i & 1) ? rA : rB;
                if constexpr (RES) {
                    if constexpr (i < 2) static_for<NB>([&](auto J) { pl_ld8<32 * (i + 1)>(rnxt[J], ro[J], p.res); });
                    // batch 0 landed during the last chunk; younger than batch 1: the stores of block 0 + batch 2; than batch 2: the stores of block 1
                    if constexpr (i == 1) pl_wait_r<2 * NB, NB>(rcur);
                    else if constexpr (i == 2) pl_wait_r<NB, NB>(rcur);
                }
                static_for<NB>([&](auto J) {
                    f32x4 v = acc[i][J];
                    if constexpr (decltype(ACT)::value && !(ABL & 8)) {
                        const f32x4 t = v * -1.44269504f;
                        f32x4 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
                        d = d + 1.0f;
                        const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
                        v = v * r;
                    }
                    if constexpr (RES) {
                        const u32x2 rv = rcur[J];
                        v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
                        v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
                    }
                    const u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    pl_st8<32 * i>(oo[J], p.out, o, omask[J]);
                });
                __builtin_amdgcn_sched_barrier(0);      // one M block at a time: do not pull all 3 NB accumulators into VGPRs up front
            });
        };
        if (p.act) epilogue(std::true_type()); else epilogue(std::false_type());
        if (!has_next) break;
        tile = next_tile;
    }
    // the weight loads issued for a tile that does not exist must not outlive their registers: wait, THEN let the registers go
    wait_vmcnt<0>();
    if constexpr (ABL & 16) {
        stamp(4);
        if (lane == 0 && p.debug) {
            unsigned long long* d = p.debug + ((long long)blockIdx.x * 4 + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = ph[k];
            d[6] = clock64() - ph_t0;
            d[7] = __builtin_amdgcn_s_memrealtime() - ph_r0;
        }
    }
    pl_tie_a(A0); pl_tie_a(A1); pl_tie_a(A2);
}

// ---- the hand-scheduled assembly build of the same kernel (gen_conv3x3_pl_asm.py; NB = 13): code object embedded at build time ----
struct PlAsmArgs {                 // must match ARG in gen_conv3x3_pl_asm.py
    const char* in; long long in_sp, in_ss;
    char* out; const char* res; const char* w; const float* bias; const char* zero;
    int out_ld_b, res_ld_b, B, H, W, npix, cout, act;
    int CC, mt_log2, ntiles, G;
    float inv_hw, inv_w, inv_hpwp, inv_wp;
    unsigned long long* debug;
    long long in_row;                 // stride-2 family: bytes per INPUT image row (H, W, npix are the OUTPUT's there)
};
static_assert(sizeof(PlAsmArgs) == 144, "kernel argument block");
const unsigned char kPlAsmCode[] = {
#include "conv3x3_pl_asm_hsaco.inc"
};
// families of the assembly build: pixel blocks per tile, region rows of its LDS planes, workgroups per CU (gen_conv3x3_pl_asm.py CONFIGS)
struct PlAsmFamily { int nb, rows, occ; };
const PlAsmFamily kPlAsm[] = {{13, 384, 1}, {7, 256, 2}, {8, 256, 2}};
constexpr int kNumPlAsm = sizeof(kPlAsm) / sizeof(kPlAsm[0]);
hipModule_t g_pl_asm_mod[64];
hipFunction_t g_pl_asm_fn[64][kNumPlAsm][5];     // res0, res1, res1 stamped, and (NB = 13 only) the fp8-weight res0, res1
hipFunction_t g_pl_s2_fn[64][2];                 // stride-2 family (s2nb13): plain, stamped
hipFunction_t g_pl_f8_fn[64][3];                 // fp8 family (f8nb13): res0, res1, res1 stamped
hipFunction_t g_pl_pm_fn[64][3][3];              // pixel-major builds of the stride-1 bf16 kernel (pm13: any width, pm13w20, pm13w40) x (res0, res1, res1 stamped)
constexpr int PL_S2_NB = 13, PL_S2_ROWS = 304;   // gen_conv3x3_pl_asm.py CONFIGS["s2nb13"]

struct PlKernel { int nb; void (*plain)(const PlParams); void (*res)(const PlParams); };
struct PlAblation { int abl; void (*fn)(const PlParams); };
#define PLA(A) { A, conv3x3_pl_kernel<13, true, A> }
const PlAblation kPlAbl[] = { PLA(16) };
#define PLK(NB) { NB, conv3x3_pl_kernel<NB, false>, conv3x3_pl_kernel<NB, true> }
const PlKernel kPl[] = { PLK(13), PLK(10), PLK(7) };
constexpr int kNumPl = sizeof(kPl) / sizeof(kPl[0]);
bool g_pl_attr[64][kNumPl][2];
int g_pl_cus[64];

// widest run of padded coordinates any tile of bn pixels needs, halo included.  O(npix / bn) divisions: remembered per geometry, because
// every launch asks several times (~1 ms of launch-thread time per step otherwise, invisible to the event-timed tuner)
int pl_region_rows_uncached(int B, int H, int W, int bn);
int pl_region_rows(int B, int H, int W, int bn) {          // (bn < 0: the one-sided halo of the stride-2 family, -bn pixels per tile)
    struct Entry { int B, H, W, bn, rows; };
    static Entry cache[64];
    static int used = 0, next = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used; ++i)
        if (cache[i].B == B && cache[i].H == H && cache[i].W == W && cache[i].bn == bn) return cache[i].rows;
    const int rows = pl_region_rows_uncached(B, H, W, bn);
    cache[next] = Entry{B, H, W, bn, rows};
    next = (next + 1) % 64;
    if (used < 64) ++used;
    return rows;
}
int pl_region_rows_uncached(int B, int H, int W, int bn) {
    const int halos = bn < 0 ? 1 : 2;
    if (bn < 0) bn = -bn;
    const long long npix = (long long)B * H * W;
    auto pp = [&](long long P) { const long long b = P / (H * W), y = (P % (H * W)) / W; return P + b * (H + W + 1) + y; };
    long long worst = 0;
    for (long long n0 = 0; n0 < npix; n0 += bn) {
        const long long n1 = (n0 + bn - 1 < npix - 1) ? n0 + bn - 1 : npix - 1;
        const long long span = pp(n1) - pp(n0) + 1 + halos * (W + 2);
        if (span > worst) worst = span;
    }
    return (int)worst;
}

int pl_load_module(int dev) {
    if (g_pl_asm_mod[dev]) return AQ_OK;
    hipModule_t mod = nullptr;
    AQ_CHECK_HIP(hipModuleLoadData(&mod, kPlAsmCode));
    for (int i = 0; i < kNumPlAsm; ++i)
        for (int v = 0; v < (kPlAsm[i].nb == 13 ? 5 : 3); ++v) {
            char name[64];
            snprintf(name, sizeof name, "conv3x3_pl_asm_nb%d_res%d%s", kPlAsm[i].nb, v == 0 || v == 3 ? 0 : 1,
                     v == 2 ? "_stamped" : v >= 3 ? "_w8" : "");
            const hipError_t e = hipModuleGetFunction(&g_pl_asm_fn[dev][i][v], mod, name);
            if (e != hipSuccess && kPlAsm[i].nb != 13) {    // the two-workgroup families are built only with AQ_GEN_EXPERIMENTAL=1
                (void)hipGetLastError();
                g_pl_asm_fn[dev][i][v] = nullptr;
                continue;
            }
            AQ_CHECK_HIP(e);
        }
    AQ_CHECK_HIP(hipModuleGetFunction(&g_pl_s2_fn[dev][0], mod, "conv3x3_pl_asm_s2nb13_res0"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_pl_s2_fn[dev][1], mod, "conv3x3_pl_asm_s2nb13_res0_stamped"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_pl_f8_fn[dev][0], mod, "conv3x3_pl_asm_f8nb13_res0"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_pl_f8_fn[dev][1], mod, "conv3x3_pl_asm_f8nb13_res1"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_pl_f8_fn[dev][2], mod, "conv3x3_pl_asm_f8nb13_res1_stamped"));
    for (int f = 0; f < 3; ++f)
        for (int v = 0; v < 3; ++v) {
            char name[64];
            snprintf(name, sizeof name, "conv3x3_pl_asm_pm13%s_res%d%s", f == 0 ? "" : f == 1 ? "w20" : "w40", v ? 1 : 0, v == 2 ? "_stamped" : "");
            AQ_CHECK_HIP(hipModuleGetFunction(&g_pl_pm_fn[dev][f][v], mod, name));
        }
    g_pl_asm_mod[dev] = mod;
    return AQ_OK;
}

}  // namespace

// 1 when the assembly family with `nb` pixel blocks per tile is in this build (13 always; 7 and 8 -- two workgroups per CU, measured no
// faster anywhere on the BASELINE geometries -- only when the library was built with AQ_GEN_EXPERIMENTAL=1), 0 when not, < 0 on error.
extern "C" int aq_conv3x3_pl_asm_family(int nb) {
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "conv3x3_pl_asm_family: device ordinal %d", dev);
    { const int rc = pl_load_module(dev); if (rc) return rc; }
    for (int i = 0; i < kNumPlAsm; ++i)
        if (kPlAsm[i].nb == nb) return g_pl_asm_fn[dev][i][0] ? 1 : 0;
    return 0;
}

extern "C" int aq_conv3x3_pl_supported(int cin, int cout) {
    return cin >= 128 && cin % 64 == 0 && cout % PL_BM == 0 && cout <= 960;
}

// Packs fused fp32 weights KRSC (cout, 3, 3, cin) into per-wave A-fragment streams:
// [M tile][wave][chunk][tap][k-step][M block i][lane] x 8 bf16, where lane (r = lane & 15, g = lane >> 4) holds output channel
// 192 mt + 48 wave + 16 i + r and input channels 64 chunk + 32 kstep + 8 g .. + 7 of that tap.
extern "C" int aq_pack_conv3x3_pl(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_host && bytes && aq_conv3x3_pl_supported(cin, cout), "pack_conv3x3_pl: unsupported %d -> %d", cin, cout);
    const int n_mt = cout / PL_BM, CC = cin / 64;
    *bytes = (size_t)n_mt * 4 * CC * 9 * PL_STEP_B;
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_conv3x3_pl: out of host memory");
    bf16_t* dst = host;
    for (int mt = 0; mt < n_mt; ++mt)
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 0; c < CC; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    for (int ks = 0; ks < 2; ++ks)
                        for (int i = 0; i < 3; ++i)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int co = mt * PL_BM + wv * 48 + i * 16 + (lane & 15);
                                const int ci = 64 * c + 32 * ks + 8 * (lane >> 4);
                                const float* src = w_host + ((size_t)co * 9 + tap) * cin + ci;
                                for (int e = 0; e < 8; ++e) *dst++ = aq_f2bf(src[e]);
                            }
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

// Pixel blocks per tile for this geometry: the fewest (rounds x tile size) over the kernel's instantiations whose region fits
static int pl_pick(int B, int H, int W, int n_mt, int cus, int* rows_out) {
    const long long npix = (long long)B * H * W;
    int best = -1;
    double best_cost = 1e30;
    for (int k = 0; k < kNumPl; ++k) {
        const int bn = kPl[k].nb * 16;
        const int rows = pl_region_rows(B, H, W, bn);
        if (rows > PL_ROWS) continue;
        const long long tiles = (npix + bn - 1) / bn * n_mt;
        const long long rounds = (tiles + cus - 1) / cus;
        const double cost = (double)rounds * (bn + 24);      // + a tile's fixed cost (epilogue, barriers) in pixel units
        if (cost < best_cost) { best_cost = cost; best = k; if (rows_out) *rows_out = rows; }
    }
    return best;
}

// in: bf16 pixels, `cin` channels; element (pixel P, channel group g) at in + P * in_sp + g * in_ss bytes (NHWC: in_sp = row bytes,
// in_ss = 16).  out / res: NHWC bf16 with row lengths out_ld / res_ld (elements), channel slices at *_choff.
static int pl_conv(const void* in_dev, long long in_sp, long long in_ss, int cin, void* out_dev, int out_ld, int out_choff,
                   int cout, const void* res_dev, int res_ld, int res_choff, const void* packed_w_dev, const float* bias_dev,
                   int B, int H, int W, int act, void* stream, bool w8) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "conv3x3_pl: null pointer");
    AQ_REQUIRE(aq_conv3x3_pl_supported(cin, cout), "conv3x3_pl: unsupported %d -> %d", cin, cout);
    AQ_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)B * (H + 1) * (W + 1) + W + 2 < (1LL << 23), "conv3x3_pl: shape outside the fast-index range");
    AQ_REQUIRE(in_sp % 16 == 0 && in_ss % 16 == 0 && out_ld % 4 == 0 && out_choff % 4 == 0 && out_choff + cout <= out_ld,
               "conv3x3_pl: slices must be 8-byte aligned and inside their rows");
    AQ_REQUIRE(!res_dev || (res_ld % 4 == 0 && res_choff % 4 == 0 && res_choff + cout <= res_ld), "conv3x3_pl: bad residual slice");
    AQ_REQUIRE((long long)B * H * W * out_ld * 2 < (1LL << 31) && (long long)B * H * W * res_ld * 2 < (1LL << 31),
               "conv3x3_pl: output / residual tensors beyond the 32-bit offset range");
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "conv3x3_pl: device ordinal %d", dev);
    if (g_pl_cus[dev] == 0) {
        int cus = 256;
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_pl_cus[dev] = cus;
    }
    PlParams p{};
    p.in = (const char*)in_dev; p.in_sp = in_sp; p.in_ss = in_ss;
    p.out = (char*)out_dev + (size_t)out_choff * 2; p.out_ld_b = out_ld * 2;
    if (res_dev) { p.res = (const char*)res_dev + (size_t)res_choff * 2; p.res_ld_b = res_ld * 2; }
    p.w = (const char*)packed_w_dev; p.bias = bias_dev;
    p.zero = aq_zero_page();
    AQ_REQUIRE(p.zero, "conv3x3_pl: zero page allocation failed");
    p.B = B; p.H = H; p.W = W; p.npix = B * H * W; p.cout = cout; p.act = act;
    p.CC = cin / 64; p.n_mt = cout / PL_BM;
    p.inv_hw = 1.0f / (float)(H * W); p.inv_w = 1.0f / (float)W;
    p.inv_hpwp = 1.0f / (float)((H + 1) * (W + 1)); p.inv_wp = 1.0f / (float)(W + 1);
    // tile shape: AQ_PL_NB forces the pixel-block count; the assembly build of that count is used when it exists and fits
    // (AQ_PL_ASM=0: HIP-source kernels only -- A/B and fallback; =2: the stamped assembly build when a stamp buffer is armed)
    int rows = 0, k = pl_pick(B, H, W, p.n_mt, g_pl_cus[dev], &rows);
    const char* forced = getenv("AQ_PL_NB");
    const char* use_asm = getenv("AQ_PL_ASM");
    const char* abl = getenv("AQ_PL_ABL");                  // timing-only diagnostic builds (wrong results), NB = 13 with shortcut only
    int nb = k >= 0 ? kPl[k].nb : 0;
    if (forced && *forced) {
        nb = atoi(forced);
        k = -1;
        for (int i = 0; i < kNumPl; ++i)
            if (kPl[i].nb == nb && pl_region_rows(B, H, W, nb * 16) <= PL_ROWS) k = i;
    }
    { const int rc = pl_load_module(dev); if (rc) return rc; }
    int fam = -1;
    if (w8) nb = 13;                                        // the fp8-weight stream exists in the NB = 13 assembly family only
    if ((w8 || (!(use_asm && *use_asm == '0') && !(abl && *abl))) && (p.n_mt & (p.n_mt - 1)) == 0 && in_sp < (1LL << 32) && in_ss < (1LL << 31))
        for (int i = 0; i < kNumPlAsm; ++i)
            if (kPlAsm[i].nb == nb && g_pl_asm_fn[dev][i][0] && pl_region_rows(B, H, W, nb * 16) <= kPlAsm[i].rows) fam = i;
    AQ_REQUIRE(!w8 || fam >= 0, "conv3x3_pl_w8: shape outside the fp8-weight kernel (aq_conv3x3_pl_w8_supported)");
    AQ_REQUIRE(k >= 0 || fam >= 0, "conv3x3_pl: no tile of this kernel fits a %d-wide image in its region rows", W);
    const int bn = nb * 16;
    const long long ntiles = ((long long)p.npix + bn - 1) / bn * p.n_mt;
    AQ_REQUIRE(ntiles > 0 && ntiles < (1LL << 30), "conv3x3_pl: bad tile count");
    p.ntiles = (int)ntiles;
    if (fam >= 0) {
        long long grid = (long long)g_pl_cus[dev] * kPlAsm[fam].occ;
        if (grid > ntiles) grid = ntiles;
        PlAsmArgs a{};
        a.in = p.in; a.in_sp = p.in_sp; a.in_ss = p.in_ss; a.out = p.out; a.res = p.res; a.w = p.w; a.bias = p.bias; a.zero = p.zero;
        a.out_ld_b = p.out_ld_b; a.res_ld_b = p.res_ld_b; a.B = p.B; a.H = p.H; a.W = p.W; a.npix = p.npix; a.cout = p.cout; a.act = p.act;
        a.CC = p.CC; a.ntiles = p.ntiles; a.G = (int)grid;
        a.mt_log2 = 0;
        while ((1 << a.mt_log2) < p.n_mt) ++a.mt_log2;
        a.inv_hw = p.inv_hw; a.inv_w = p.inv_w; a.inv_hpwp = p.inv_hpwp; a.inv_wp = p.inv_wp; a.debug = nullptr;
        int which = (res_dev ? 1 : 0) + (w8 ? 3 : 0);
        if (!w8 && use_asm && *use_asm == '2' && res_dev) {        // stamped diagnostic build (tools/time_conv3x3.py --stamp): per-wave phase cycle sums
            size_t sbytes = 0;
            unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
            if (sbuf && sbytes >= (size_t)grid * 4 * 64) { a.debug = sbuf; which = 2; }
        }
        hipFunction_t fn_asm = g_pl_asm_fn[dev][fam][which];
        // The pixel-major build of the same kernel (family pm13: same tiles, weights, arguments and results; 64-byte pixel-major region rows
        // loaded through a buffer descriptor -- a quarter of the cache-line look-ups per LDS-DMA instruction) wherever its addressing holds:
        // contiguous channel groups, 24-bit pixel stride and pixel count, the tensor inside 2 GiB.  It pays where the image width is known at
        // build time (the kernel-row offset is then an immediate: builds for yolov5m's 20 and 40); the any-width build moves its addresses
        // with VALU adds in the stream and is slower than the slot-major kernel, so it runs only on request.
        // AQ_PL_PM=0: slot-major everywhere (A/B); AQ_PL_PM=2: pixel-major, any-width build, everywhere (tests).
        const char* pm_env = getenv("AQ_PL_PM");
        const int pm_mode = pm_env && *pm_env ? atoi(pm_env) : 1;
        const int pmf = pm_mode == 2 ? 0 : W == 20 ? 1 : W == 40 ? 2 : 0;
        const bool pm = nb == 13 && !w8 && pm_mode != 0 && (pmf != 0 || pm_mode == 2) && in_ss == 16 && in_sp < (1 << 24) && in_sp % 16 == 0 &&
                        p.npix < (1 << 24) && (long long)p.npix * in_sp < (1LL << 31);
        if (pm) fn_asm = g_pl_pm_fn[dev][pmf][which];
        const char* asm_abl = getenv("AQ_PL_ASM_ABL");      // timing-only ablations of the stamped build (wrong results)
        if (which == 2 && asm_abl && *asm_abl) {
            char name[80];
            if (pm) snprintf(name, sizeof name, "conv3x3_pl_asm_pm13w40_res1_stamped_abl%d", atoi(asm_abl));
            else snprintf(name, sizeof name, "conv3x3_pl_asm_nb%d_res1_stamped_abl%d", nb, atoi(asm_abl));
            AQ_CHECK_HIP(hipModuleGetFunction(&fn_asm, g_pl_asm_mod[dev], name));
        }
        size_t asz = sizeof(a);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        AQ_CHECK_HIP(hipModuleLaunchKernel(fn_asm, (unsigned)grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
        return AQ_OK;
    }
    auto fn = res_dev ? kPl[k].res : kPl[k].plain;
    bool ablated = false;
    if (abl && *abl && kPl[k].nb == 13 && res_dev)
        for (const PlAblation& a : kPlAbl)
            if (a.abl == atoi(abl)) { fn = a.fn; ablated = true; }
    if (ablated) {
        size_t sbytes = 0;
        unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
        if (sbuf && sbytes >= (size_t)g_pl_cus[dev] * 4 * 64) p.debug = sbuf;
    }
    const size_t lds = (size_t)PL_BIAS + 4096;
    AQ_REQUIRE(lds <= 160 * 1024, "conv3x3_pl: LDS");
    if (ablated) AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (!g_pl_attr[dev][k][res_dev ? 1 : 0]) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_pl_attr[dev][k][res_dev ? 1 : 0] = true;
    }
    long long grid = g_pl_cus[dev];
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" int aq_conv3x3_pl(const void* in_dev, long long in_sp, long long in_ss, int cin, void* out_dev, int out_ld, int out_choff,
                             int cout, const void* res_dev, int res_ld, int res_choff, const void* packed_w_dev, const float* bias_dev,
                             int B, int H, int W, int act, void* stream) {
    return pl_conv(in_dev, in_sp, in_ss, cin, out_dev, out_ld, out_choff, cout, res_dev, res_ld, res_choff, packed_w_dev, bias_dev, B, H, W, act,
                   stream, false);
}

// ---- fp8-weight stream (precision AQ_BF16_W8): the same kernel loading OCP e4m3fn codes -- half the L2 -> register weight traffic, the
// bottleneck of the bf16 stream -- and converting them to bf16 fragments in the shadow of the MFMAs; the per-output-channel scale 2^e is
// applied to the accumulators in the epilogue (and its inverse to the bias they start from), which is exact, so the outputs are bit for
// bit those of the bf16 stream on the dequantised weights.
extern "C" int aq_conv3x3_pl_w8_supported(int cin, int cout, int B, int H, int W) {
    if (!aq_conv3x3_pl_supported(cin, cout) || B <= 0 || H <= 0 || W <= 0) return 0;
    const int n_mt = cout / PL_BM;
    return (n_mt & (n_mt - 1)) == 0 && pl_region_rows(B, H, W, 13 * 16) <= PL_ROWS;
}

// e4m3fn code of v if v is exactly representable (|v| <= 448, 3 mantissa bits, subnormal step 2^-9), else -1
static int pl_e4m3_code(double v) {
    const int sign = std::signbit(v) ? 0x80 : 0;
    const double a = std::fabs(v);
    if (a == 0.0) return sign;
    if (a > 448.0) return -1;
    if (a < 0.015625) {                                     // subnormals: multiples of 2^-9 below 2^-6
        const double m = a * 512.0;
        return (m == std::floor(m) && m >= 1.0 && m <= 7.0) ? sign | (int)m : -1;
    }
    int ex = 0;
    std::frexp(a, &ex);                                     // a = f * 2^ex, f in [0.5, 1)
    ex -= 1;
    const double m = std::ldexp(a, 3 - ex);                 // in [8, 16)
    if (m != std::floor(m) || ex < -6 || ex > 8) return -1;
    return sign | ((ex + 7) << 3) | ((int)m - 8);
}

// Weights must already lie on a per-output-channel grid code x 2^e (aquaculture_amd/quant.py quantize_rows; AQ_ERR_INVALID otherwise).
// Image: [M tile][wave][chunk][tap][fragment pair][lane] x 16 codes (fragment 2 p then 2 p + 1, aq_pack_conv3x3_pl's fragments);
// scale_bias_dev: float[2048] = bias x 2^-e (1024, zero padded), then 2^e (1024).
extern "C" int aq_pack_conv3x3_pl_w8(const float* w_host, const float* bias_host, int cin, int cout, void* packed_dev, size_t* bytes,
                                     float* scale_bias_dev, void* stream) {
    AQ_REQUIRE(w_host && bytes && aq_conv3x3_pl_supported(cin, cout) && cout <= 1024, "pack_conv3x3_pl_w8: unsupported %d -> %d", cin, cout);
    const int CC = cin / 64, n_mt = cout / PL_BM;
    *bytes = (size_t)n_mt * 4 * CC * 9 * (PL_STEP_B / 2);
    if (!packed_dev) return AQ_OK;
    AQ_REQUIRE(bias_host && scale_bias_dev, "pack_conv3x3_pl_w8: null pointer");
    std::vector<int> ex(cout, 0);
    std::vector<float> sb(2048, 0.0f);
    const size_t kk = (size_t)9 * cin;
    for (int co = 0; co < cout; ++co) {
        const float* row = w_host + (size_t)co * kk;
        double amax = 0.0;
        for (size_t i = 0; i < kk; ++i) amax = std::fmax(amax, std::fabs((double)row[i]));
        int e = amax > 0.0 ? (int)std::ceil(std::log2(amax / 448.0)) : 0;
        bool ok = false;
        for (int attempt = 0; attempt < 3 && !ok; ++attempt, ++e) {      // a dequantised maximum of 224 x 2^e reads as one binade lower
            ok = true;
            for (size_t i = 0; i < kk && ok; ++i) ok = pl_e4m3_code(std::ldexp((double)row[i], -e)) >= 0;
            if (ok) break;
        }
        AQ_REQUIRE(ok, "pack_conv3x3_pl_w8: output channel %d is not on an e4m3 x 2^e grid", co);
        ex[co] = e;
        sb[co] = std::ldexp(bias_host[co], -e);
        sb[1024 + co] = std::ldexp(1.0f, e);
    }
    unsigned char* host = (unsigned char*)malloc(*bytes);
    AQ_REQUIRE(host, "pack_conv3x3_pl_w8: host allocation failed");
    unsigned char* dst = host;
    for (int mt = 0; mt < n_mt; ++mt)
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 0; c < CC; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    for (int pr = 0; pr < 3; ++pr)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int h = 0; h < 2; ++h) {
                                const int k = 2 * pr + h, ks = k / 3, i = k % 3;
                                const int co = mt * PL_BM + wv * 48 + i * 16 + (lane & 15);
                                const int ci = 64 * c + 32 * ks + 8 * (lane >> 4);
                                const float* src = w_host + ((size_t)co * 9 + tap) * cin + ci;
                                for (int e = 0; e < 8; ++e) *dst++ = (unsigned char)pl_e4m3_code(std::ldexp((double)src[e], -ex[co]));
                            }
    hipError_t err = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err == hipSuccess) err = hipMemcpyAsync(scale_bias_dev, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(err);
    return AQ_OK;
}

extern "C" int aq_conv3x3_pl_w8(const void* in_dev, long long in_sp, long long in_ss, int cin, void* out_dev, int out_ld, int out_choff,
                                int cout, const void* res_dev, int res_ld, int res_choff, const void* packed_w8_dev,
                                const float* scale_bias_dev, int B, int H, int W, int act, void* stream) {
    return pl_conv(in_dev, in_sp, in_ss, cin, out_dev, out_ld, out_choff, cout, res_dev, res_ld, res_choff, packed_w8_dev, scale_bias_dev, B, H, W,
                   act, stream, true);
}


// ---- stride 2 (round 3): the same planar scheme on the four PARITY planes of the input (gen_conv3x3_pl_asm.py, family s2nb13) --------------
// 3x3 / stride 2 / pad 1, bf16, Cin a multiple of 32 (32-channel chunks, one k-step per tap), Cout a multiple of 192 with Cout / 192 a power
// of two, even H and W: yolov5m's model.5 / 7 / 18 / 21 ([UPSTREAM models/common.py Conv.forward_fuse], reference README.md:77).  Assembly
// only: the implicit-GEMM kernels remain the fallback and the comparand (tests/test_gpu_conv.py).
extern "C" int aq_conv3x3_pl_s2_supported(int cin, int cout, int B, int H, int W) {
    if (cin < 64 || cin % 32 || cout % PL_BM || cout > 960 || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    const int n_mt = cout / PL_BM;
    if (n_mt & (n_mt - 1)) return 0;
    if ((long long)B * (H / 2 + 1) * (W / 2 + 1) + W / 2 + 2 >= (1LL << 23)) return 0;
    return pl_region_rows(B, H / 2, W / 2, -PL_S2_NB * 16) <= PL_S2_ROWS;
}

// [M tile][wave][32-channel chunk][tap][M block i][lane] x 8 bf16: lane (r = lane & 15, g = lane >> 4) holds output channel
// 192 mt + 48 wave + 16 i + r and input channels 32 chunk + 8 g .. + 7 of that tap (KRSC source, as aq_pack_conv3x3_pl)
extern "C" int aq_pack_conv3x3_pl_s2(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_host && bytes && cin % 32 == 0 && cout % PL_BM == 0, "pack_conv3x3_pl_s2: unsupported %d -> %d", cin, cout);
    const int n_mt = cout / PL_BM, CC = cin / 32;
    *bytes = (size_t)n_mt * 4 * CC * 9 * 3072;
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_conv3x3_pl_s2: out of host memory");
    bf16_t* dst = host;
    for (int mt = 0; mt < n_mt; ++mt)
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 0; c < CC; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    for (int i = 0; i < 3; ++i)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = mt * PL_BM + wv * 48 + i * 16 + (lane & 15);
                            const int ci = 32 * c + 8 * (lane >> 4);
                            const float* src = w_host + ((size_t)co * 9 + tap) * cin + ci;
                            for (int e = 0; e < 8; ++e) *dst++ = aq_f2bf(src[e]);
                        }
    hipError_t e = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(e);
    return AQ_OK;
}

// in: NHWC bf16, B x H x W pixels of in_ld channels, the conv's cin channels from in_choff (a multiple of 8); out: NHWC bf16
// B x H/2 x W/2 x out_ld, cout channels from out_choff.
extern "C" int aq_conv3x3_pl_s2(const void* in_dev, int in_ld, int in_choff, int cin, void* out_dev, int out_ld, int out_choff, int cout,
                                const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int act, void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "conv3x3_pl_s2: null pointer");
    AQ_REQUIRE(aq_conv3x3_pl_s2_supported(cin, cout, B, H, W), "conv3x3_pl_s2: unsupported %d -> %d on %d x %d x %d", cin, cout, B, H, W);
    AQ_REQUIRE(in_choff % 8 == 0 && in_choff + cin <= in_ld && in_ld % 8 == 0 && out_ld % 4 == 0 && out_choff % 4 == 0 && out_choff + cout <= out_ld,
               "conv3x3_pl_s2: slices must be 16-byte (input) / 8-byte (output) aligned and inside their rows");
    const int Ho = H / 2, Wo = W / 2;
    // (the input goes through a buffer descriptor with num_records = 2^31: valid offsets must stay below it, the padding rows' 2^31 is beyond)
    AQ_REQUIRE((long long)B * Ho * Wo * out_ld * 2 < (1LL << 31) && (long long)B * H * W * in_ld * 2 < (1LL << 31) && in_ld * 2 < (1 << 24),
               "conv3x3_pl_s2: tensors beyond the 31-bit offset range");
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "conv3x3_pl_s2: device ordinal %d", dev);
    if (g_pl_cus[dev] == 0) {
        int cus = 256;
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_pl_cus[dev] = cus;
    }
    { const int rc = pl_load_module(dev); if (rc) return rc; }
    PlAsmArgs a{};
    a.in = (const char*)in_dev + (size_t)in_choff * 2; a.in_sp = (long long)in_ld * 2; a.in_ss = 16;
    a.in_row = (long long)W * in_ld * 2;
    a.out = (char*)out_dev + (size_t)out_choff * 2; a.out_ld_b = out_ld * 2;
    a.w = (const char*)packed_w_dev; a.bias = bias_dev;
    a.zero = aq_zero_page();
    AQ_REQUIRE(a.zero, "conv3x3_pl_s2: zero page allocation failed");
    a.B = B; a.H = Ho; a.W = Wo; a.npix = B * Ho * Wo; a.cout = cout; a.act = act;
    a.CC = cin / 32;
    const int n_mt = cout / PL_BM;
    while ((1 << a.mt_log2) < n_mt) ++a.mt_log2;
    const int bn = PL_S2_NB * 16;
    const long long ntiles = ((long long)a.npix + bn - 1) / bn * n_mt;
    AQ_REQUIRE(ntiles > 0 && ntiles < (1LL << 30), "conv3x3_pl_s2: bad tile count");
    a.ntiles = (int)ntiles;
    long long grid = g_pl_cus[dev];
    if (grid > ntiles) grid = ntiles;
    a.G = (int)grid;
    a.inv_hw = 1.0f / (float)(Ho * Wo); a.inv_w = 1.0f / (float)Wo;
    a.inv_hpwp = 1.0f / (float)((Ho + 1) * (Wo + 1)); a.inv_wp = 1.0f / (float)(Wo + 1);
    hipFunction_t fn = g_pl_s2_fn[dev][0];
    const char* use_asm = getenv("AQ_PL_ASM");
    if (use_asm && *use_asm == '2') {                          // stamped diagnostic build (tools/time_conv3x3.py --stamp): per-wave phase cycle sums
        size_t sbytes = 0;
        unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
        if (sbuf && sbytes >= (size_t)grid * 4 * 64) {
            a.debug = sbuf;
            fn = g_pl_s2_fn[dev][1];
            const char* asm_abl = getenv("AQ_PL_ASM_ABL");    // timing-only ablations of the stamped build (wrong results)
            if (asm_abl && *asm_abl) {
                char name[80];
                snprintf(name, sizeof name, "conv3x3_pl_asm_s2nb13_res0_stamped_abl%d", atoi(asm_abl));
                AQ_CHECK_HIP(hipModuleGetFunction(&fn, g_pl_asm_mod[dev], name));
            }
        }
    }
    size_t asz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    AQ_CHECK_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
    return AQ_OK;
}


// ---- fp8 on both MFMA operands (round 3; BASELINE.json configs[3] "fp8 weights (CDNA4 fp8 MFMA)") -- gen_conv3x3_pl_asm.py, family f8nb13 ----
// 3x3 / stride 1 / pad 1 with OCP e4m3fn activations AND weights on v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 MFMA's rate):
// y = (res +) SiLU(act_scale * w_scale[co] * sum(qx * qw) + bias), bf16 out.  The input tensor holds e4m3 codes (one byte per channel,
// NHWC) written by the producing kernel with one scale per tensor; the weights are quantised here with one scale per output channel.

// round-to-nearest-even fp32 -> e4m3fn code, saturating at +-448 (what v_cvt_pk_fp8_f32 and torch.float8_e4m3fn do inside the range)
extern "C" unsigned char aq_f32_to_e4m3(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const unsigned sign = (v.u >> 24) & 0x80u;
    const float a = std::fabs(f);
    if (!(a == a)) return (unsigned char)(sign | 0x7f);
    if (a >= 448.0f) return (unsigned char)(sign | 0x7e);
    if (a < 0.015625f) {                                    // subnormals: multiples of 2^-9
        const float m = std::nearbyint(a * 512.0f);         // default rounding mode: to nearest even
        return (unsigned char)(sign | (unsigned)m);         // m == 8 is the smallest normal, code 0x08: the same formula
    }
    uint32_t u = v.u & 0x7fffffffu;
    const uint32_t lsb = (u >> 20) & 1u;
    u += 0x7ffffu + lsb;                                    // round the 23-bit mantissa to 3 bits, ties to even
    u &= ~0xfffffu;
    const int ex = (int)(u >> 23) - 127;                    // -6 .. 8
    const unsigned mant = (u >> 20) & 7u;
    return (unsigned char)(sign | ((unsigned)(ex + 7) << 3) | mant);
}

extern "C" int aq_conv3x3_pl_f8_supported(int cin, int cout, int B, int H, int W) {
    if (cin < 64 || cin % 64 || cout % PL_BM || cout > 960 || B <= 0 || H <= 0 || W <= 0) return 0;
    const int n_mt = cout / PL_BM;
    if (n_mt & (n_mt - 1)) return 0;
    if ((long long)B * (H + 1) * (W + 1) + W + 2 >= (1LL << 23)) return 0;
    return pl_region_rows(B, H, W, 13 * 16) <= PL_ROWS;
}

// Weight codes in MFMA A-operand order: [M tile][wave][64-channel chunk][step p = 0..4][M block i][half][lane] x 16 codes; lane (r = lane & 15,
// g = lane >> 4) holds output channel 192 mt + 48 wave + 16 i + r, tap 2 p + (g >> 1) (tap 9 of step 4: zeros) and input channels
// 64 chunk + 32 (g & 1) + 16 half .. + 15.  scale_bias_dev: float[2048] = bias / (act_scale w_scale) (1024), then act_scale w_scale (1024),
// w_scale[co] = max |w[co]| / 448.
extern "C" int aq_pack_conv3x3_pl_f8(const float* w_host, const float* bias_host, int cin, int cout, float act_scale, void* packed_dev,
                                     size_t* bytes, float* scale_bias_dev, void* stream) {
    AQ_REQUIRE(w_host && bytes && cin % 64 == 0 && cout % PL_BM == 0 && cout <= 1024, "pack_conv3x3_pl_f8: unsupported %d -> %d", cin, cout);
    const int CC = cin / 64, n_mt = cout / PL_BM;
    *bytes = (size_t)n_mt * 4 * CC * 5 * 6144;
    if (!packed_dev) return AQ_OK;
    AQ_REQUIRE(bias_host && scale_bias_dev && act_scale > 0.0f && std::isfinite(act_scale), "pack_conv3x3_pl_f8: null pointer or bad activation scale");
    std::vector<float> sb(2048, 0.0f), ws(cout, 1.0f);
    const size_t kk = (size_t)9 * cin;
    for (int co = 0; co < cout; ++co) {
        const float* row = w_host + (size_t)co * kk;
        float amax = 0.0f;
        for (size_t i = 0; i < kk; ++i) amax = std::fmax(amax, std::fabs(row[i]));
        ws[co] = amax > 0.0f ? amax / 448.0f : 1.0f;
        const float sc = act_scale * ws[co];
        sb[co] = bias_host[co] / sc;
        sb[1024 + co] = sc;
    }
    unsigned char* host = (unsigned char*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_conv3x3_pl_f8: host allocation failed");
    unsigned char* dst = host;
    for (int mt = 0; mt < n_mt; ++mt)
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 0; c < CC; ++c)
                for (int p = 0; p < 5; ++p)
                    for (int i = 0; i < 3; ++i)
                        for (int half = 0; half < 2; ++half)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int co = mt * PL_BM + wv * 48 + i * 16 + (lane & 15), g = lane >> 4;
                                const int tap = 2 * p + (g >> 1);
                                const int ci = 64 * c + 32 * (g & 1) + 16 * half;
                                for (int e = 0; e < 16; ++e)
                                    *dst++ = tap < 9 ? aq_f32_to_e4m3(w_host[((size_t)co * 9 + tap) * cin + ci + e] / ws[co]) : (unsigned char)0;
                            }
    hipError_t err = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err == hipSuccess) err = hipMemcpyAsync(scale_bias_dev, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(err);
    return AQ_OK;
}

// in: e4m3 codes, NHWC, in_ld BYTES per pixel, the conv's cin channels from byte in_choff (a multiple of 16); out / res: NHWC bf16 slices.
extern "C" int aq_conv3x3_pl_f8(const void* in_dev, int in_ld, int in_choff, int cin, void* out_dev, int out_ld, int out_choff, int cout,
                                const void* res_dev, int res_ld, int res_choff, const void* packed_w_dev, const float* scale_bias_dev,
                                int B, int H, int W, int act, void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && scale_bias_dev, "conv3x3_pl_f8: null pointer");
    AQ_REQUIRE(aq_conv3x3_pl_f8_supported(cin, cout, B, H, W), "conv3x3_pl_f8: unsupported %d -> %d on %d x %d x %d", cin, cout, B, H, W);
    AQ_REQUIRE(in_choff % 16 == 0 && in_ld % 16 == 0 && in_choff + cin <= in_ld && out_ld % 4 == 0 && out_choff % 4 == 0 && out_choff + cout <= out_ld,
               "conv3x3_pl_f8: slices must be 16-byte (input) / 8-byte (output) aligned and inside their rows");
    AQ_REQUIRE(!res_dev || (res_ld % 4 == 0 && res_choff % 4 == 0 && res_choff + cout <= res_ld), "conv3x3_pl_f8: bad residual slice");
    AQ_REQUIRE((long long)B * H * W * out_ld * 2 < (1LL << 31) && (long long)B * H * W * res_ld * 2 < (1LL << 31) && (long long)B * H * W * in_ld < (1LL << 31) &&
                   in_ld < (1 << 24), "conv3x3_pl_f8: tensors beyond the 31-bit offset range");
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "conv3x3_pl_f8: device ordinal %d", dev);
    if (g_pl_cus[dev] == 0) {
        int cus = 256;
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_pl_cus[dev] = cus;
    }
    { const int rc = pl_load_module(dev); if (rc) return rc; }
    PlAsmArgs a{};
    a.in = (const char*)in_dev + in_choff; a.in_sp = in_ld; a.in_ss = 16;
    a.out = (char*)out_dev + (size_t)out_choff * 2; a.out_ld_b = out_ld * 2;
    if (res_dev) { a.res = (const char*)res_dev + (size_t)res_choff * 2; a.res_ld_b = res_ld * 2; }
    a.w = (const char*)packed_w_dev; a.bias = scale_bias_dev;
    a.zero = aq_zero_page();
    AQ_REQUIRE(a.zero, "conv3x3_pl_f8: zero page allocation failed");
    a.B = B; a.H = H; a.W = W; a.npix = B * H * W; a.cout = cout; a.act = act;
    a.CC = cin / 64;
    const int n_mt = cout / PL_BM;
    while ((1 << a.mt_log2) < n_mt) ++a.mt_log2;
    const long long ntiles = ((long long)a.npix + 207) / 208 * n_mt;
    AQ_REQUIRE(ntiles > 0 && ntiles < (1LL << 30), "conv3x3_pl_f8: bad tile count");
    a.ntiles = (int)ntiles;
    long long grid = g_pl_cus[dev];
    if (grid > ntiles) grid = ntiles;
    a.G = (int)grid;
    a.inv_hw = 1.0f / (float)(H * W); a.inv_w = 1.0f / (float)W;
    a.inv_hpwp = 1.0f / (float)((H + 1) * (W + 1)); a.inv_wp = 1.0f / (float)(W + 1);
    hipFunction_t fn = g_pl_f8_fn[dev][res_dev ? 1 : 0];
    const char* use_asm = getenv("AQ_PL_ASM");
    if (use_asm && *use_asm == '2' && res_dev) {              // stamped diagnostic build
        size_t sbytes = 0;
        unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
        if (sbuf && sbytes >= (size_t)grid * 4 * 64) {
            a.debug = sbuf;
            fn = g_pl_f8_fn[dev][2];
            const char* asm_abl = getenv("AQ_PL_ASM_ABL");
            if (asm_abl && *asm_abl) {
                char name[80];
                snprintf(name, sizeof name, "conv3x3_pl_asm_f8nb13_res1_stamped_abl%d", atoi(asm_abl));
                AQ_CHECK_HIP(hipModuleGetFunction(&fn, g_pl_asm_mod[dev], name));
            }
        }
    }
    size_t asz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    AQ_CHECK_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
    return AQ_OK;
}
