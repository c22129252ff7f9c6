// Detect-head decode and per-tile NMS.  Compiled with -ffp-contract=off: every fp32 operation below is
// written in the order the reference's dependencies evaluate it so that results can be compared bit-level
// against the oracle (no fused multiply-add may merge two of their roundings).
//
//   decode  = [UPSTREAM models/yolo.py Detect.forward, inference branch + _make_grid]
//   nms     = [UPSTREAM utils/general.py non_max_suppression(conf, iou, classes=None, agnostic=False,
//              multi_label=False, max_det)] over torchvision.ops.nms (greedy, fp32, strict '>')
// Both are invoked through reference README.md:77 (yolov5/detect.py).
#include "aq_common.h"

namespace {

constexpr float kMaxWH = 7680.0f;   // class offset [UPSTREAM non_max_suppression: max_wh]
constexpr int kMaxNms = 30000;      // [UPSTREAM non_max_suppression: max_nms]
constexpr int kNmsThreads = 1024;
constexpr int kSortLds = 4096;      // keys sorted in LDS up to this many (32 KiB)

struct DecodeParams {
    const float* head[3];
    int head_ld;           // floats per pixel
    int B, nc, na, no;
    int ny[3], nx[3];
    int off[4];            // candidate index offsets per level, off[3] = N
    float stride[3];
    float anchor[3][8][2]; // pixels
    float* pred;           // [B][N][no] or null
    float conf_thres;
    int32_t* cand;         // [B][cap] candidate indices or null
    float* cand_rows;      // [B][cap][no] decoded rows of the candidates (same order as cand) or null
    int32_t* cand_count;   // [B]
    int cap;
};

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void decode_row(const DecodeParams& p, const float* src, int lvl, int a, int x, int y,
                                           float obj, float* dst) {
    // xy = (xy * 2 + grid) * stride, grid = index - 0.5 ; wh = (wh * 2) ** 2 * anchor_grid
    const float s0 = sigmoidf_ref(src[0]), s1 = sigmoidf_ref(src[1]);
    const float s2 = sigmoidf_ref(src[2]), s3 = sigmoidf_ref(src[3]);
    const float gx = (float)x - 0.5f, gy = (float)y - 0.5f;
    dst[0] = (s0 * 2.0f + gx) * p.stride[lvl];
    dst[1] = (s1 * 2.0f + gy) * p.stride[lvl];
    const float tw = s2 * 2.0f, th = s3 * 2.0f;
    dst[2] = (tw * tw) * p.anchor[lvl][a][0];
    dst[3] = (th * th) * p.anchor[lvl][a][1];
    dst[4] = obj;
    for (int c = 0; c < p.nc; ++c) dst[5 + c] = sigmoidf_ref(src[5 + c]);
}

// One lane per (pixel, anchor), anchors of a pixel on adjacent lanes (they share the pixel's 128-byte head row).
// Objectness is evaluated for every candidate; the other 9 sigmoids only where a row is actually written:
// every row when the full pred tensor is requested (S1 alone), else only rows with obj > conf_thres, which go
// to the compact candidate list that NMS reads.
constexpr int kDecPerLane = 4;                               // candidates per lane: their head-map loads are issued together

__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams p) {
    // grid = (ceil(N / (256 * kDecPerLane)), B).  Passing candidates take a slot from an LDS counter; ONE global atomic per block
    // reserves the block's range in the image's compact list (same-address global atomics serialise).
    __shared__ int s_cnt, s_base;
    const int N = p.off[3];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    bool pass[kDecPerLane];
    int local[kDecPerLane], n[kDecPerLane], lvl[kDecPerLane], a[kDecPerLane], x[kDecPerLane], y[kDecPerLane];
    const float* src[kDecPerLane];
    float raw[kDecPerLane];
#pragma unroll
    for (int k = 0; k < kDecPerLane; ++k) {                  // index math + the objectness loads (independent: all in flight together)
        const int w = (blockIdx.x * kDecPerLane + k) * blockDim.x + threadIdx.x;
        pass[k] = false; local[k] = 0; n[k] = 0; lvl[k] = 0; a[k] = 0; x[k] = 0; y[k] = 0; src[k] = nullptr; raw[k] = 0.f;
        if (w < N) {
            int m = w;                                     // position in the pixel-major walk of this image
            const int na = p.na;
            lvl[k] = m >= p.off[2] ? 2 : (m >= p.off[1] ? 1 : 0);
            m -= p.off[lvl[k]];
            const int ny = p.ny[lvl[k]], nx = p.nx[lvl[k]];
            const int pix = (int)((unsigned)m / (unsigned)na);
            a[k] = m - pix * na;
            y[k] = (int)((unsigned)pix / (unsigned)nx);
            x[k] = pix - y[k] * nx;
            n[k] = p.off[lvl[k]] + a[k] * ny * nx + pix;     // upstream candidate index: a * ny * nx + y * nx + x
            src[k] = p.head[lvl[k]] + ((long long)(b * ny + y[k]) * nx + x[k]) * p.head_ld + a[k] * p.no;
            raw[k] = src[k][4];
        }
    }
    float obj[kDecPerLane];
#pragma unroll
    for (int k = 0; k < kDecPerLane; ++k) {
        obj[k] = 0.f;
        if (src[k]) {
            obj[k] = sigmoidf_ref(raw[k]);
            if (p.pred) decode_row(p, src[k], lvl[k], a[k], x[k], y[k], obj[k], p.pred + ((long long)b * N + n[k]) * p.no);
            pass[k] = p.cand && obj[k] > p.conf_thres;
            if (pass[k]) local[k] = atomicAdd(&s_cnt, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt > 0) s_base = atomicAdd(p.cand_count + b, s_cnt);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kDecPerLane; ++k) {
        if (pass[k]) {
            const int pos = s_base + local[k];
            if (pos < p.cap) {
                p.cand[(long long)b * p.cap + pos] = n[k];
                if (p.cand_rows)
                    decode_row(p, src[k], lvl[k], a[k], x[k], y[k], obj[k], p.cand_rows + ((long long)b * p.cap + pos) * p.no);
            }
        }
    }
}

struct NmsParams {
    const float* rows;       // decoded rows: full pred [B][N][no] (rows_per_tile = N) or compact [B][cap][no]
    int rows_per_tile;
    int B, N, nc, no;
    float conf_thres, iou_thres;
    int max_det;
    const int32_t* cand;     // [B][cap] candidate index of each compact row, or null (row t IS candidate t)
    const int32_t* cand_count;
    int cap;
    unsigned long long* keys;   // [B][npow2]
    float4* sbox;               // [B][N]        (slow path)
    unsigned long long* mask;   // [B][kFast][kFast/64] suppression bit matrix (fast path)
    int npow2;
    aq_det* dets;               // [B][max_det]
    int32_t* counts;            // [B]
    unsigned long long* dbg;    // diagnostics (armed stamp buffer): 8 phase timestamps per tile, else null
    float class_step;           // kMaxWH, or 0 for class-agnostic NMS [UPSTREAM non_max_suppression: agnostic]
    unsigned long long cls_lo, cls_hi;   // classes kept (bit c of lo / bit c - 64 of hi) [UPSTREAM: classes]; all ones = no filter
};

// Candidates handled by the bit-matrix (fast) path: 2048 for 640-px tiles (25,200 rows, a few hundred candidates), 4096 for larger tiles
// (round 3: at 1280 px -- BASELINE.json configs[4] -- the synthetic head passes ~2,000 candidates per tile, and the tiles beyond 2,048 fell
// to the bitonic-sort path: 1.9 ms of a 21.6 ms step).  nms_kernel<KF>; aq_nms picks KF from the tile's row count.
constexpr int kFast = 2048;
constexpr int kFastBig = 4096;
constexpr int kFastRowsLimit = 40000;    // rows per tile up to which the 2048 kernel is launched
__host__ __device__ constexpr int nms_fast(int N) { return N > kFastRowsLimit ? kFastBig : kFast; }
constexpr int kIdxBits = 17;             // candidate index / row slot < 2^17 (1280x1280 tiles: 100,800)
constexpr int nms_fast_lds(int kf) { return kf * 8 + kf * 16 + kf * 4 + kf * 2; }   // keys + boxes + slots + kept ranks = 60 KiB at 2048
constexpr int nms_mask_lds(int kf) { return kf == kFast ? 64 * 1024 : 32 * 1024; }  // LDS kept for the suppression bit matrix (n <= ~700 / ~500 candidates)
constexpr int kSlowLds = kSortLds * 8 + kMaxNms + 16;
constexpr int nms_lds(int kf) { return nms_fast_lds(kf) + nms_mask_lds(kf) > kSlowLds ? nms_fast_lds(kf) + nms_mask_lds(kf) : kSlowLds; }
static_assert(nms_lds(kFastBig) <= 160 * 1024, "LDS");

// conf = obj * cls_conf, best class = first maximum; box = xywh2xyxy (x -/+ w/2)
__device__ __forceinline__ bool candidate_row(const float* row, int nc, float thr, float4& box, float& conf, int& cls,
                                              unsigned long long cls_lo = ~0ULL, unsigned long long cls_hi = ~0ULL) {
    const float obj = row[4];
    if (!(obj > thr)) return false;
    float best = row[5] * obj;
    int bj = 0;
    for (int c = 1; c < nc; ++c) {
        const float v = row[5 + c] * obj;
        if (v > best) { best = v; bj = c; }
    }
    if (!(best > thr)) return false;
    if (!(((bj < 64 ? cls_lo >> bj : cls_hi >> (bj - 64)) & 1ULL))) return false;      // x = x[(x[:, 5:6] == classes).any(1)]
    const float hw = row[2] / 2.0f, hh = row[3] / 2.0f;
    box.x = row[0] - hw; box.y = row[1] - hh; box.z = row[0] + hw; box.w = row[1] + hh;
    conf = best; cls = bj;
    return true;
}

// key = conf bits (30: 0 <= conf <= 1) | inverted candidate index (17) | row slot (17).  Descending key order is
// descending confidence with ties broken by ascending candidate index (the slot never decides: indices are unique).
__device__ __forceinline__ unsigned long long make_key(float conf, int idx, int slot) {
    return ((unsigned long long)__float_as_uint(conf) << (2 * kIdxBits)) |
           ((unsigned long long)(((1u << kIdxBits) - 1u) - (unsigned)idx) << kIdxBits) | (unsigned long long)slot;
}
__device__ __forceinline__ int key_slot(unsigned long long k) { return (int)(k & ((1u << kIdxBits) - 1u)); }

__device__ __forceinline__ void bitonic_desc(unsigned long long* k, int n, int tid, int nthreads) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int lo = 2 * t - (t & (stride - 1));   // index with bit `stride` clear
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = k[lo], b = k[hi];
                if ((a < b) == desc) { k[lo] = b; k[hi] = a; }
            }
            __syncthreads();
        }
    }
}

// torchvision nms_kernel_impl's test, fp32, evaluated in its order: inter / (area_i + area_j - inter) > thr
__device__ __forceinline__ bool iou_gt(const float4 bi, float iarea, const float4 bj, float thr) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float jarea = (bj.z - bj.x) * (bj.w - bj.y);
    const float ovr = inter / (iarea + jarea - inter);
    return ovr > thr;
}

// The same predicate, cheaper on average and bit-identical: an empty intersection gives 0 (or NaN) > thr = false without touching
// the areas; otherwise inter * rcp(union) (relative error < 3 ulp) decides unless it lies within 4e-7 of the threshold, where the
// exact division is evaluated.
__device__ __forceinline__ bool iou_gt_fast(const float4 bi, float iarea, const float4 bj, float thr, float thr_lo, float thr_hi) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    if (!(inter > 0.0f)) return false;                       // also the NaN case
    const float jarea = (bj.z - bj.x) * (bj.w - bj.y);
    const float uni = iarea + jarea - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    if (q > thr_hi) return true;
    if (q < thr_lo) return false;
    return inter / uni > thr;                                // ambiguous band (and non-finite q): the reference's own expression
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

template <int KF>
__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const NmsParams p) {
    constexpr int kFast = KF, kFastWords = KF / 64, kFastLds = nms_fast_lds(KF), kMaskLds = nms_mask_lds(KF);
    __shared__ int s_n;
    // one dynamic LDS block, two uses: fast path = unsorted keys [kFast] + sorted offset boxes [kFast] + slots + kept
    //                                  list + (when it fits) the suppression bit matrix;
    //                                  slow path = keys for the in-LDS bitonic sort [kSortLds] + suppression flags
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ int s_kept;
    const int b = blockIdx.x, tid = threadIdx.x;
    auto stamp = [&](int i) { if (p.dbg && tid == 0) p.dbg[(long long)b * 8 + i] = clock64(); };
    stamp(0);
    const float* rows = p.rows + (long long)b * p.rows_per_tile * p.no;
    unsigned long long* keys = p.keys + (long long)b * p.npow2;

    if (tid == 0) s_n = 0;
    __syncthreads();
    // A: threshold twice (obj, then obj*cls) and build the sort keys
    const int n0 = p.cand ? min(p.cand_count[b], p.cap) : p.N;
    for (int t = tid; t < n0; t += kNmsThreads) {
        const int idx = p.cand ? p.cand[(long long)b * p.cap + t] : t;
        float4 box; float conf; int cls;
        if (candidate_row(rows + (long long)t * p.no, p.nc, p.conf_thres, box, conf, cls, p.cls_lo, p.cls_hi)) {
            const int pos = atomicAdd(&s_n, 1);
            keys[pos] = make_key(conf, idx, t);
        }
    }
    __syncthreads();
    stamp(1);
    int n = s_n;
    if (n == 0) {
        if (tid == 0) p.counts[b] = 0;
        return;
    }

    auto emit = [&](int slot, int k) {   // one output row [x1, y1, x2, y2, conf, cls], boxes WITHOUT the class offset
        float4 box; float conf; int cls;
        candidate_row(rows + (long long)slot * p.no, p.nc, p.conf_thres, box, conf, cls, p.cls_lo, p.cls_hi);
        aq_det d; d.x1 = box.x; d.y1 = box.y; d.x2 = box.z; d.y2 = box.w; d.conf = conf; d.cls = (float)cls;
        p.dets[(long long)b * p.max_det + k] = d;
    };

    if (n <= kFast) {
        // ---------------- fast path ----------------
        unsigned long long* kin = (unsigned long long*)s_raw;
        float4* sb = (float4*)(s_raw + kFast * 8);
        int* sslot = (int*)(s_raw + kFast * 8 + kFast * 16);          // row slot of the r-th most confident box
        unsigned short* skept = (unsigned short*)(s_raw + kFast * 8 + kFast * 16 + kFast * 4);   // kept ranks
        for (int t = tid; t < n; t += kNmsThreads) kin[t] = keys[t];
        __syncthreads();
        // B: rank sort (keys are unique): rank = number of larger keys; all lanes read the same kin[u] (broadcast)
        for (int t = tid; t < n; t += kNmsThreads) {
            const unsigned long long k = kin[t];
            int r = 0;
            for (int u = 0; u < n; ++u) r += (kin[u] > k) ? 1 : 0;
            sslot[r] = key_slot(k);
            // C: sorted boxes + class offset (boxes + cls * max_wh, fp32)
            float4 box; float conf; int cls;
            candidate_row(rows + (long long)key_slot(k) * p.no, p.nc, p.conf_thres, box, conf, cls, p.cls_lo, p.cls_hi);
            const float c = (float)cls * p.class_step;
            box.x = box.x + c; box.y = box.y + c; box.z = box.z + c; box.w = box.w + c;
            sb[r] = box;
        }
        __syncthreads();
        stamp(2);
        // D1: suppression bit matrix, upper triangle: bit j of mask[i][w] = IoU(i, 64w + j) > thr, 64w + j > i
        const int nw = (n + 63) >> 6;
        // the bit matrix lives in LDS when n * nw words fit behind the fast-path arrays, else in the global scratch.  The two
        // cases run separate copies of the code below so that the LDS case compiles to ds_read/ds_write instead of the flat
        // loads a run-time selected pointer would need (the greedy scan's critical path is a chain of row loads).
        const float thr_lo = p.iou_thres * (1.0f - 4e-7f), thr_hi = p.iou_thres * (1.0f + 4e-7f);
        auto matrix_and_scan = [&](unsigned long long* mask) __attribute__((always_inline)) {
            // Only words on or right of the diagonal are needed (and read): row block bi = i >> 6 has nw - bi of them.  Items are
            // enumerated over exactly those words so that every thread gets the same amount of work (a plain (i, w) grid gives
            // the threads that own the last word column four times the pairs of those that own the first).
            int total_items = 0;
            for (int bi = 0; bi < nw; ++bi) total_items += 64 * (nw - bi);
            for (int item = tid; item < total_items; item += kNmsThreads) {
                int bi = 0, rem = item;
                while (rem >= 64 * (nw - bi)) { rem -= 64 * (nw - bi); ++bi; }      // <= nw iterations
                const int wpr = nw - bi;                     // words per row in this row block
                const int i = 64 * bi + rem / wpr, w = bi + rem % wpr;
                if (i >= n) continue;
                const float4 bi4 = sb[i];
                const float iarea = (bi4.z - bi4.x) * (bi4.w - bi4.y);
                const int j0 = w << 6;
                const int jbeg = max(j0, i + 1), jend = min(j0 + 64, n);
                unsigned long long bits = 0ull;
#pragma unroll 4
                for (int j = jbeg; j < jend; ++j)
                    if (iou_gt_fast(bi4, iarea, sb[j], p.iou_thres, thr_lo, thr_hi)) bits |= 1ull << (j - j0);
                mask[(long long)i * nw + w] = bits;
            }
            __syncthreads();
            stamp(3);
            // D2: greedy scan by one wave, one 64-candidate word at a time.  Lane t keeps word t of the removed set.  For word w,
            //   1. lane i loads the DIAGONAL word of candidate 64 w + i (which later candidates of the same word it suppresses);
            //   2. the intra-word greedy runs on scalar bit operations: take the lowest alive bit, keep it, clear the bits its
            //      diagonal word (v_readlane) suppresses -- one iteration per KEPT candidate, no memory access in the chain;
            //   3. the kept rows are OR-ed into the removed words of later blocks with independent (pipelined) loads.
            if (tid < 64) {
                unsigned long long removed = 0ull;
                int kept = 0;
                bool done = false;
                for (int w = 0; w < nw && !done; ++w) {
                    const int i_l = 64 * w + tid;
                    const unsigned long long diag = i_l < n ? mask[(long long)i_l * nw + w] : 0ull;
                    const int left = n - 64 * w;
                    const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
                    unsigned long long alive = valid & ~readlane64(removed, w);
                    unsigned long long keptmask = 0ull;
                    while (alive != 0ull) {
                        const int bit = __builtin_ctzll(alive);
                        keptmask |= 1ull << bit;
                        alive &= ~(1ull << bit);
                        alive &= ~readlane64(diag, bit);
                        if (kept + __builtin_popcountll(keptmask) >= p.max_det) { done = true; break; }
                    }
                    if ((keptmask >> tid) & 1ull)
                        skept[kept + __builtin_popcountll(keptmask & ((1ull << tid) - 1ull))] = (unsigned short)(64 * w + tid);
                    kept += __builtin_popcountll(keptmask);
                    unsigned long long km = keptmask, acc = 0ull;
                    const bool mine = tid > w && tid < nw;   // words of later blocks
                    while (km != 0ull) {                     // four independent row loads per round
                        int bits4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            bits4[u] = km != 0ull ? __builtin_ctzll(km) : -1;
                            km &= km - 1ull;                 // 0 stays 0
                        }
                        unsigned long long r4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) r4[u] = (mine && bits4[u] >= 0) ? mask[(long long)(64 * w + bits4[u]) * nw + tid] : 0ull;
                        acc |= (r4[0] | r4[1]) | (r4[2] | r4[3]);
                    }
                    removed |= acc;
                }
                if (tid == 0) { p.counts[b] = kept; s_kept = kept; }
            }
        };
        if ((size_t)n * nw * 8 <= kMaskLds) matrix_and_scan((unsigned long long*)(s_raw + kFastLds));
        else matrix_and_scan(p.mask + (long long)b * kFast * kFastWords);
        __syncthreads();
        stamp(4);
        // E: all threads write the kept rows (descending confidence = ascending rank)
        for (int k = tid; k < s_kept; k += kNmsThreads) emit(sslot[skept[k]], k);
        __syncthreads();
        stamp(5);
        return;
    }

    // ---------------- slow path (n > kFast): bitonic sort + barrier-per-kept greedy loop ----------------
    unsigned long long* s_keys = (unsigned long long*)s_raw;
    unsigned char* s_supp = s_raw + kSortLds * 8;
    float4* sbox = p.sbox + (long long)b * p.N;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int t = n + tid; t < np2; t += kNmsThreads) keys[t] = 0ull;
    __syncthreads();
    if (np2 <= kSortLds) {
        for (int t = tid; t < np2; t += kNmsThreads) s_keys[t] = keys[t];
        __syncthreads();
        bitonic_desc(s_keys, np2, tid, kNmsThreads);
        for (int t = tid; t < np2; t += kNmsThreads) keys[t] = s_keys[t];
        __syncthreads();
    } else {
        bitonic_desc(keys, np2, tid, kNmsThreads);
    }
    n = min(n, kMaxNms);
    for (int t = tid; t < n; t += kNmsThreads) {
        float4 box; float conf; int cls;
        candidate_row(rows + (long long)key_slot(keys[t]) * p.no, p.nc, p.conf_thres, box, conf, cls, p.cls_lo, p.cls_hi);
        const float c = (float)cls * p.class_step;
        box.x = box.x + c; box.y = box.y + c; box.z = box.z + c; box.w = box.w + c;
        sbox[t] = box;
        s_supp[t] = 0;
    }
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        if (s_supp[i]) continue;   // uniform: flags only change between barriers
        if (tid == 0) emit(key_slot(keys[i]), kept);
        ++kept;
        if (kept >= p.max_det) break;
        const float4 bi = sbox[i];
        const float iarea = (bi.z - bi.x) * (bi.w - bi.y);
        for (int j = i + 1 + tid; j < n; j += kNmsThreads) {
            if (s_supp[j]) continue;
            if (iou_gt(bi, iarea, sbox[j], p.iou_thres)) s_supp[j] = 1;
        }
        __syncthreads();
    }
    if (tid == 0) p.counts[b] = kept;
}

inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" int aq_detect_decode(const float* const head_dev[3], int head_ld, int B, int H, int W, int nc, int na,
                                const float* anchors_px, const float* stride, float* pred_dev, float conf_thres,
                                int32_t* cand_dev, float* cand_rows_dev, int32_t* cand_count_dev, int cand_cap, void* stream) {
    AQ_REQUIRE(head_dev && head_dev[0] && head_dev[1] && head_dev[2], "decode: null head pointer");
    AQ_REQUIRE(na >= 1 && na <= 8 && nc >= 1 && head_ld >= na * (nc + 5), "decode: bad na=%d nc=%d head_ld=%d", na, nc, head_ld);
    AQ_REQUIRE(pred_dev || cand_dev, "decode: nothing to write");
    AQ_REQUIRE(!cand_dev || (cand_count_dev && cand_cap > 0), "decode: candidate list needs a counter and a capacity");
    AQ_REQUIRE(!cand_rows_dev || cand_dev, "decode: candidate rows need the candidate index list");
    DecodeParams p;
    p.head_ld = head_ld; p.B = B; p.nc = nc; p.na = na; p.no = nc + 5;
    int off = 0;
    for (int l = 0; l < 3; ++l) {
        const int s = (int)stride[l];
        AQ_REQUIRE(s > 0 && H % s == 0 && W % s == 0, "decode: H=%d W=%d not a multiple of stride %d", H, W, s);
        p.head[l] = head_dev[l];
        p.ny[l] = H / s; p.nx[l] = W / s;
        p.off[l] = off;
        off += na * p.ny[l] * p.nx[l];
        p.stride[l] = stride[l];
        for (int a = 0; a < na; ++a) { p.anchor[l][a][0] = anchors_px[(l * na + a) * 2]; p.anchor[l][a][1] = anchors_px[(l * na + a) * 2 + 1]; }
    }
    p.off[3] = off;
    p.pred = pred_dev; p.conf_thres = conf_thres; p.cand = cand_dev; p.cand_rows = cand_rows_dev;
    p.cand_count = cand_count_dev; p.cap = cand_cap;
    if (cand_dev) AQ_CHECK_HIP(hipMemsetAsync(cand_count_dev, 0, sizeof(int32_t) * B, (hipStream_t)stream));
    AQ_REQUIRE(B <= 65535, "decode: batch too large for the grid");
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((off + 256 * kDecPerLane - 1) / (256 * kDecPerLane)), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" size_t aq_nms_scratch_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return align_up((size_t)B * next_pow2(N) * sizeof(unsigned long long), 256) + align_up((size_t)B * N * sizeof(float4), 256) +
           align_up((size_t)B * nms_fast(N) * (nms_fast(N) / 64) * sizeof(unsigned long long), 256);
}

extern "C" int aq_nms(const float* rows_dev, int rows_per_tile, int B, int N, int nc, float conf_thres, float iou_thres,
                      int max_det, const int32_t* cand_dev, const int32_t* cand_count_dev, int cand_cap,
                      void* scratch_dev, aq_det* dets_dev, int32_t* counts_dev, void* stream) {
    return aq_nms_opts(rows_dev, rows_per_tile, B, N, nc, conf_thres, iou_thres, max_det, cand_dev, cand_count_dev, cand_cap, scratch_dev, dets_dev,
                       counts_dev, 0, ~0ULL, ~0ULL, stream);
}

// The same with upstream's two remaining options: agnostic (boxes of different classes suppress each other) and classes (keep only the
// listed classes: bit c of classes_lo, bit c - 64 of classes_hi; all ones = every class).
extern "C" int aq_nms_opts(const float* rows_dev, int rows_per_tile, int B, int N, int nc, float conf_thres, float iou_thres,
                           int max_det, const int32_t* cand_dev, const int32_t* cand_count_dev, int cand_cap,
                           void* scratch_dev, aq_det* dets_dev, int32_t* counts_dev, int agnostic, unsigned long long classes_lo,
                           unsigned long long classes_hi, void* stream) {
    AQ_REQUIRE(nc <= 128, "nms: the class filter covers 128 classes (nc = %d)", nc);
    AQ_REQUIRE(rows_dev && scratch_dev && dets_dev && counts_dev, "nms: null pointer");
    AQ_REQUIRE(B > 0 && N > 0 && nc >= 1 && max_det > 0, "nms: bad shape B=%d N=%d nc=%d max_det=%d", B, N, nc, max_det);
    AQ_REQUIRE(N < (1 << kIdxBits), "nms: at most %d candidates per tile are supported (got %d)", (1 << kIdxBits) - 1, N);
    AQ_REQUIRE(!cand_dev || (cand_count_dev && cand_cap > 0 && cand_cap <= N && rows_per_tile == cand_cap),
               "nms: a compact candidate list needs a counter, 0 < cap <= N and rows_per_tile == cap");
    AQ_REQUIRE(cand_dev || rows_per_tile == N, "nms: without a candidate list the rows are the full pred (rows_per_tile == N)");
    NmsParams p;
    p.rows = rows_dev; p.rows_per_tile = rows_per_tile; p.B = B; p.N = N; p.nc = nc; p.no = nc + 5;
    p.conf_thres = conf_thres; p.iou_thres = iou_thres; p.max_det = max_det;
    p.cand = cand_dev; p.cand_count = cand_count_dev; p.cap = cand_cap;
    p.class_step = agnostic ? 0.0f : kMaxWH; p.cls_lo = classes_lo; p.cls_hi = classes_hi;
    p.npow2 = next_pow2(N);
    char* s = (char*)scratch_dev;
    p.keys = (unsigned long long*)s;
    s += align_up((size_t)B * p.npow2 * sizeof(unsigned long long), 256);
    p.sbox = (float4*)s;
    s += align_up((size_t)B * N * sizeof(float4), 256);
    p.mask = (unsigned long long*)s;
    p.dets = dets_dev; p.counts = counts_dev;
    {   // diagnostics: an armed stamp buffer (aq_debug_conv_stamp) receives 8 phase timestamps per tile
        size_t sbytes = 0;
        unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
        p.dbg = (sbuf && (size_t)B * 64 <= sbytes) ? sbuf : nullptr;
    }
    static bool attr_set[2] = {false, false};
    const bool big = nms_fast(N) == kFastBig;
    auto fn = big ? nms_kernel<kFastBig> : nms_kernel<kFast>;
    const int lds = nms_lds(big ? kFastBig : kFast);
    if (!attr_set[big]) {
        AQ_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set[big] = true;
    }
    hipLaunchKernelGGL(fn, dim3(B), dim3(kNmsThreads), lds, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
