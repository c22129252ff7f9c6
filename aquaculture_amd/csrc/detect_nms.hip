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
    int32_t* cand;         // [B][cap] or null
    int32_t* cand_count;   // [B]
    int cap;
};

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams p) {
    const int N = p.off[3];
    const long long total = (long long)p.B * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), n = (int)(i - (long long)b * N);
        const int lvl = n >= p.off[2] ? 2 : (n >= p.off[1] ? 1 : 0);
        const int m = n - p.off[lvl];
        const int ny = p.ny[lvl], nx = p.nx[lvl];
        const int a = m / (ny * nx), rem = m - a * ny * nx;
        const int y = rem / nx, x = rem - y * nx;
        const float* src = p.head[lvl] + (((long long)b * ny + y) * nx + x) * p.head_ld + a * p.no;
        float* dst = p.pred ? p.pred + i * p.no : nullptr;
        const float s0 = sigmoidf_ref(src[0]), s1 = sigmoidf_ref(src[1]);
        const float s2 = sigmoidf_ref(src[2]), s3 = sigmoidf_ref(src[3]);
        const float obj = sigmoidf_ref(src[4]);
        if (dst) {
            // xy = (xy * 2 + grid) * stride, grid = index - 0.5 ; wh = (wh * 2) ** 2 * anchor_grid
            const float gx = (float)x - 0.5f, gy = (float)y - 0.5f;
            dst[0] = (s0 * 2.0f + gx) * p.stride[lvl];
            dst[1] = (s1 * 2.0f + gy) * p.stride[lvl];
            const float tw = s2 * 2.0f, th = s3 * 2.0f;
            dst[2] = (tw * tw) * p.anchor[lvl][a][0];
            dst[3] = (th * th) * p.anchor[lvl][a][1];
            dst[4] = obj;
            for (int c = 0; c < p.nc; ++c) dst[5 + c] = sigmoidf_ref(src[5 + c]);
        }
        if (p.cand && obj > p.conf_thres) {
            const int pos = atomicAdd(p.cand_count + b, 1);
            if (pos < p.cap) p.cand[(long long)b * p.cap + pos] = n;
        }
    }
}

struct NmsParams {
    const float* pred;       // [B][N][no]
    int B, N, nc, no;
    float conf_thres, iou_thres;
    int max_det;
    const int32_t* cand;     // [B][cap] or null (then every index is examined)
    const int32_t* cand_count;
    int cap;
    unsigned long long* keys;   // [B][npow2]
    float4* sbox;               // [B][N]
    int npow2;
    aq_det* dets;               // [B][max_det]
    int32_t* counts;            // [B]
};

// conf = obj * cls_conf, best class = first maximum; box = xywh2xyxy (x -/+ w/2)
__device__ __forceinline__ bool candidate_row(const float* row, int nc, float thr, float4& box, float& conf, int& cls) {
    const float obj = row[4];
    if (!(obj > thr)) return false;
    float best = row[5] * obj;
    int bj = 0;
    for (int c = 1; c < nc; ++c) {
        const float v = row[5 + c] * obj;
        if (v > best) { best = v; bj = c; }
    }
    if (!(best > thr)) return false;
    const float hw = row[2] / 2.0f, hh = row[3] / 2.0f;
    box.x = row[0] - hw; box.y = row[1] - hh; box.z = row[0] + hw; box.w = row[1] + hh;
    conf = best; cls = bj;
    return true;
}

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

__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const NmsParams p) {
    __shared__ int s_n;
    __shared__ unsigned long long s_keys[kSortLds];
    __shared__ unsigned char s_supp[kMaxNms + 16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pred = p.pred + (long long)b * p.N * p.no;
    unsigned long long* keys = p.keys + (long long)b * p.npow2;
    float4* sbox = p.sbox + (long long)b * p.N;

    if (tid == 0) s_n = 0;
    __syncthreads();
    // A: threshold twice (obj, then obj*cls), key = (conf bits, ~index): descending key order is
    //    descending confidence, ties by ascending candidate index.
    const int n0 = p.cand ? min(p.cand_count[b], p.cap) : p.N;
    for (int t = tid; t < n0; t += kNmsThreads) {
        const int idx = p.cand ? p.cand[(long long)b * p.cap + t] : t;
        float4 box; float conf; int cls;
        if (candidate_row(pred + (long long)idx * p.no, p.nc, p.conf_thres, box, conf, cls)) {
            const int pos = atomicAdd(&s_n, 1);
            keys[pos] = ((unsigned long long)__float_as_uint(conf) << 32) | (unsigned long long)(0xffffffffu - (unsigned)idx);
        }
    }
    __syncthreads();
    int n = s_n;
    if (n == 0) {
        if (tid == 0) p.counts[b] = 0;
        return;
    }
    // B: sort
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
    // C: sorted boxes + class offset (boxes + cls * max_wh, fp32)
    for (int t = tid; t < n; t += kNmsThreads) {
        const int idx = (int)(0xffffffffu - (unsigned)(keys[t] & 0xffffffffull));
        float4 box; float conf; int cls;
        candidate_row(pred + (long long)idx * p.no, p.nc, p.conf_thres, box, conf, cls);
        const float c = (float)cls * kMaxWH;
        box.x = box.x + c; box.y = box.y + c; box.z = box.z + c; box.w = box.w + c;
        sbox[t] = box;
        s_supp[t] = 0;
    }
    __syncthreads();
    // D: greedy pass (torchvision nms_kernel_impl): i kept unless suppressed; suppress j > i with IoU > thr
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        if (s_supp[i]) continue;   // uniform: flags only change between barriers
        if (tid == 0) {
            const int idx = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
            float4 box; float conf; int cls;
            candidate_row(pred + (long long)idx * p.no, p.nc, p.conf_thres, box, conf, cls);
            aq_det d; d.x1 = box.x; d.y1 = box.y; d.x2 = box.z; d.y2 = box.w; d.conf = conf; d.cls = (float)cls;
            p.dets[(long long)b * p.max_det + kept] = d;
        }
        ++kept;
        if (kept >= p.max_det) break;
        const float4 bi = sbox[i];
        const float iarea = (bi.z - bi.x) * (bi.w - bi.y);
        for (int j = i + 1 + tid; j < n; j += kNmsThreads) {
            if (s_supp[j]) continue;
            const float4 bj = sbox[j];
            const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
            const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float jarea = (bj.z - bj.x) * (bj.w - bj.y);
            const float ovr = inter / (iarea + jarea - inter);
            if (ovr > p.iou_thres) s_supp[j] = 1;
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
                                int32_t* cand_dev, int32_t* cand_count_dev, int cand_cap, void* stream) {
    AQ_REQUIRE(head_dev && head_dev[0] && head_dev[1] && head_dev[2], "decode: null head pointer");
    AQ_REQUIRE(na >= 1 && na <= 8 && nc >= 1 && head_ld >= na * (nc + 5), "decode: bad na=%d nc=%d head_ld=%d", na, nc, head_ld);
    AQ_REQUIRE(pred_dev || cand_dev, "decode: nothing to write");
    AQ_REQUIRE(!cand_dev || (cand_count_dev && cand_cap > 0), "decode: candidate list needs a counter and a capacity");
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
    p.pred = pred_dev; p.conf_thres = conf_thres; p.cand = cand_dev; p.cand_count = cand_count_dev; p.cap = cand_cap;
    if (cand_dev) AQ_CHECK_HIP(hipMemsetAsync(cand_count_dev, 0, sizeof(int32_t) * B, (hipStream_t)stream));
    const long long total = (long long)B * off;
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}

extern "C" size_t aq_nms_scratch_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return align_up((size_t)B * next_pow2(N) * sizeof(unsigned long long), 256) + align_up((size_t)B * N * sizeof(float4), 256);
}

extern "C" int aq_nms(const float* pred_dev, int B, int N, int nc, float conf_thres, float iou_thres, int max_det,
                      const int32_t* cand_dev, const int32_t* cand_count_dev, int cand_cap,
                      void* scratch_dev, aq_det* dets_dev, int32_t* counts_dev, void* stream) {
    AQ_REQUIRE(pred_dev && scratch_dev && dets_dev && counts_dev, "nms: null pointer");
    AQ_REQUIRE(B > 0 && N > 0 && nc >= 1 && max_det > 0, "nms: bad shape B=%d N=%d nc=%d max_det=%d", B, N, nc, max_det);
    AQ_REQUIRE(!cand_dev || (cand_count_dev && cand_cap > 0), "nms: candidate list needs a counter and a capacity");
    NmsParams p;
    p.pred = pred_dev; p.B = B; p.N = N; p.nc = nc; p.no = nc + 5;
    p.conf_thres = conf_thres; p.iou_thres = iou_thres; p.max_det = max_det;
    p.cand = cand_dev; p.cand_count = cand_count_dev; p.cap = cand_cap;
    p.npow2 = next_pow2(N);
    p.keys = (unsigned long long*)scratch_dev;
    p.sbox = (float4*)((char*)scratch_dev + align_up((size_t)B * p.npow2 * sizeof(unsigned long long), 256));
    p.dets = dets_dev; p.counts = counts_dev;
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(kNmsThreads), 0, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
