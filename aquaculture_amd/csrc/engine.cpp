// Host side of libaqengine.so: C ABI (include/aq_engine.h), plan executor, weight packing, profiling.
// The engine allocates device memory only in aq_engine_create (packed weights, zero page, events);
// every launch-path function only enqueues work on the caller's stream.
#include "aq_common.h"
#include <unistd.h>
#include <string>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <dlfcn.h>
#include <vector>

namespace {
thread_local char g_err[512] = "";
}

void aq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* aq_last_error(void) { return g_err; }
extern "C" int aq_version(void) { return 7; }   // 7: generated-assembly wide 1x1 (AQ_CONV_CFG_ASM1X1), assembly C = 48 Bottleneck; 6: planar 3x3/s2 and fp8 families, AQ_F16X3, fp8 pairs; 5: AQ_BF16_W8, aq_conv3x3_pl_w8; 2: fused stem / Bottleneck / down-block ops, direct 1x1 and 3x3/s2 candidates; 3: one-tile-per-workgroup grids

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kAlign = 256;
constexpr int kCoutSlack = 256;   // packed weight rows beyond cout so any BM tile may over-read zeros

// Optional roctx ranges around every plan op (AQ_ROCTX=1; `rocprofv3 --kernel-trace --marker-trace` then shows which kernels belong to
// which op).  The library is looked up at run time so that libaqengine.so does not depend on the profiler SDK.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
    bool on() {
        if (!tried) {
            tried = true;
            const char* v = getenv("AQ_ROCTX");
            if (v && *v == '1') {
                void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                    pop = (int (*)())dlsym(h, "roctxRangePop");
                }
            }
        }
        return push && pop;
    }
} g_roctx;

constexpr int kCountStride = 1024;   // ints between the per-image candidate counters of the fused head kernels (4 KB)

struct PackedW {
    void* w = nullptr;      // [cout_rows][kgroups_pad] x 16 B
    float* bias = nullptr;  // [cout_rows]
    int kgroups = 0, kgroups_pad = 0, G = 0, cout_rows = 0;
    bool x3 = false;            // AQ_F16X3 engine: this conv runs in split mode (weights as fp16 hi / lo halves; bias = [bias 2^s | 2^-s])
    void* w_direct = nullptr;   // layers a direct kernel supports: its A-fragment image (csrc/conv1x1_direct.hip, csrc/downblock.hip)
    int direct_cfg = -1;        // AQ_CONV_CFG_DIRECT1X1 / AQ_CONV_CFG_DIRECT3X3S2 / AQ_CONV_CFG_PL3X3 / AQ_CONV_CFG_PL3X3S2 / AQ_CONV_CFG_ASM1X1
    void* w_head = nullptr;     // bf16 engines, Detect head convs with a small head: aq_pack_head_weights image (csrc/head_decode.hip)
    void* w_pl8 = nullptr;      // AQ_BF16_W8 engines, planar 3x3 layers: the e4m3 code stream and its float[2048] bias x 2^-e | 2^e
    float* sb_pl8 = nullptr;
    // fp8 path (aq_engine_set_fp8_scales): a 3x3 consumer on the fp8 planar kernel, fed with e4m3 codes by its 1x1 producer
    void* w_f8 = nullptr;       // consumer: aq_pack_conv3x3_pl_f8 image ...
    float* sb_f8 = nullptr;     // ... and its bias / scale vectors
    float act_scale = 0.0f;     // consumer: the scale of its input tensor; producer: of its output tensor
    int f8_consumer = -1;       // producer: the op that reads its codes (the pair runs in fp8 only when that op's geometry fits)
    std::vector<float> w_host, b_host;   // layers with an fp8 form keep their fused fp32 weights: the fp8 packer needs the activation scale, known later
};

// Host-side packing: KRSC fp32 -> [cout_rows][kgroups_pad*16 B] of bf16 / fp32, zero padded.
void pack_host(const float* w, int cout, int k, int cin, int precision, std::vector<unsigned char>& out,
               int* kgroups, int* kgroups_pad, int* G, int* cout_rows) {
    const int eb = aq_elem_bytes(precision);
    const int g = cin * eb / 16;
    const int kg = k * k * g;
    const int kgp = (kg + 7) / 8 * 8;
    const int rows = (cout + kCoutSlack + 31) / 32 * 32;
    out.assign((size_t)rows * kgp * 16, 0);
    const int kelems = k * k * cin;
    for (int co = 0; co < cout; ++co) {
        unsigned char* dst = out.data() + (size_t)co * kgp * 16;
        const float* src = w + (size_t)co * kelems;
        if (precision == AQ_FP32) {
            memcpy(dst, src, (size_t)kelems * 4);
        } else {
            bf16_t* d = (bf16_t*)dst;
            for (int e = 0; e < kelems; ++e) d[e] = aq_f2bf(src[e]);
        }
    }
    *kgroups = kg; *kgroups_pad = kgp; *G = g; *cout_rows = rows;
}

// fp32 -> fp16 bits, round to nearest even (finite inputs inside fp16's range; subnormal results included)
uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

// AQ_F16X3: row co of the packed image holds, per 8 input channels of a tap, 8 fp16 hi halves then 8 fp16 lo halves of w * 2^s[co]
// (the same bytes per channel as fp32, so kgroups / G / the K walk are fp32 mode's); bias_scale = [bias * 2^s | 2^-s], `rows` each.
void pack_host_x3(const float* w, const float* bias, int cout, int k, int cin, std::vector<unsigned char>& out, std::vector<float>& bias_scale,
                  int* kgroups, int* kgroups_pad, int* G, int* cout_rows) {
    const int g = cin * 4 / 16;
    const int kg = k * k * g;
    const int kgp = (kg + 7) / 8 * 8;
    const int rows = (cout + kCoutSlack + 31) / 32 * 32;
    out.assign((size_t)rows * kgp * 16, 0);
    bias_scale.assign((size_t)2 * rows, 0.0f);
    for (int r = 0; r < rows; ++r) bias_scale[rows + r] = 1.0f;
    const int kelems = k * k * cin;
    for (int co = 0; co < cout; ++co) {
        const float* src = w + (size_t)co * kelems;
        float amax = 0.0f;
        for (int e = 0; e < kelems; ++e) amax = fmaxf(amax, fabsf(src[e]));
        int sexp = 0;
        if (amax > 0.0f && std::isfinite(amax)) {
            int ex = 0;
            frexpf(amax, &ex);                    // amax = f * 2^ex, f in [0.5, 1)
            sexp = 15 - ex;                       // amax * 2^s in [2^14, 2^15)
            sexp = sexp > 60 ? 60 : (sexp < -60 ? -60 : sexp);
        }
        const float sc = ldexpf(1.0f, sexp);
        bias_scale[co] = bias ? bias[co] * sc : 0.0f;
        bias_scale[rows + co] = ldexpf(1.0f, -sexp);
        uint16_t* dst = (uint16_t*)(out.data() + (size_t)co * kgp * 16);
        for (int e0 = 0; e0 < kelems; e0 += 8) {  // (cin % 8 == 0: a group of 8 never straddles taps)
            for (int e = 0; e < 8; ++e) {
                const float v = src[e0 + e] * sc;
                const uint16_t hi = f2h(v);
                dst[2 * e0 + e] = hi;
                dst[2 * e0 + 8 + e] = f2h(v - h2f(hi));
            }
        }
    }
    *kgroups = kg; *kgroups_pad = kgp; *G = g; *cout_rows = rows;
}

struct TensorPlace { size_t offset = 0; size_t bytes = 0; int h = 0, w = 0, elem = 0; };

}  // namespace

struct aq_engine {
    int device = 0;
    aq_model_desc desc{};
    std::vector<aq_tensor_desc> tensors;
    std::vector<aq_op_desc> ops;
    std::vector<PackedW> packed;        // per op (conv ops only)
    std::vector<int> conv_cfg;          // per op override, -1 = heuristic
    int nms_agnostic = 0;               // aq_engine_set_nms_options
    unsigned long long nms_cls_lo = ~0ULL, nms_cls_hi = ~0ULL;
    std::vector<int> tuned_cfg;         // per op result of aq_engine_autotune for (tuned_B, tuned_H, tuned_W)
    int tuned_B = 0, tuned_H = 0, tuned_W = 0;
    void* zero_page = nullptr;
    // workspace layout of the last sizing / call
    int lay_B = 0, lay_H = 0, lay_W = 0;
    std::vector<TensorPlace> place;
    size_t off_pred = 0, off_cand = 0, off_cand_rows = 0, off_cand_count = 0, off_cand_wide = 0, off_nms = 0, total_bytes = 0;
    int N = 0;
    void* last_ws = nullptr;
    const uint8_t* last_tiles = nullptr;
    float* calib_amax = nullptr;        // device float[n_ops] while aq_engine_calibrate_amax runs: max |output| of every bf16 conv op
    std::vector<int> last_family, last_cfg;   // per op: which kernel family (AQ_FAM_*) its most recent launch went to (aq_engine_last_launch)
    // profiling
    bool prof = false;
    int ring = 0;
    long long prof_calls = 0;
    std::vector<hipEvent_t> ev;         // [ring][n_ops + 1]
};

namespace {

int layout(aq_engine* e, int B, int H, int W) {
    AQ_REQUIRE(B > 0 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0,
               "tile size must be a positive multiple of the max stride 32 (got B=%d H=%d W=%d)", B, H, W);
    if (e->lay_B == B && e->lay_H == H && e->lay_W == W) return AQ_OK;
    const int eb = aq_elem_bytes(e->desc.precision);
    size_t off = 0;
    e->place.assign(e->tensors.size(), TensorPlace());
    for (size_t t = 0; t < e->tensors.size(); ++t) {
        const aq_tensor_desc& td = e->tensors[t];
        TensorPlace& pl = e->place[t];
        pl.h = H / td.down; pl.w = W / td.down;
        pl.elem = td.dtype == AQ_T_F32 ? 4 : (td.dtype == AQ_T_U8 ? 1 : eb);
        pl.bytes = (size_t)B * pl.h * pl.w * td.channels * pl.elem;
        if ((int)t == e->desc.input_tensor) { pl.offset = (size_t)-1; continue; }   // caller's tiles
        pl.offset = off;
        off += align_up(pl.bytes, kAlign);
    }
    int N = 0;
    for (int l = 0; l < e->desc.nl; ++l) {
        const int s = (int)e->desc.stride[l];
        N += e->desc.na * (H / s) * (W / s);
    }
    e->N = N;
    e->off_pred = off; off += align_up((size_t)B * N * (e->desc.nc + 5) * sizeof(float), kAlign);
    e->off_cand = off; off += align_up((size_t)B * N * sizeof(int32_t), kAlign);
    e->off_cand_rows = off; off += align_up((size_t)B * N * (e->desc.nc + 5) * sizeof(float), kAlign);
    e->off_cand_count = off; off += align_up((size_t)B * sizeof(int32_t), kAlign);
    e->off_cand_wide = off; off += align_up((size_t)B * kCountStride * sizeof(int32_t), kAlign);   // fused heads: counters 4 KB apart
    e->off_nms = off; off += align_up(aq_nms_scratch_bytes(B, N), kAlign);
    e->total_bytes = off;
    e->lay_B = B; e->lay_H = H; e->lay_W = W;
    return AQ_OK;
}

inline char* tptr(aq_engine* e, void* ws, const uint8_t* tiles, int t) {
    if (t == e->desc.input_tensor) return (char*)tiles;
    return (char*)ws + e->place[t].offset;
}

// Does the fp8 pair (1x1 producer -> 3x3 consumer `ci`) run in fp8 for this batch geometry?  Decided by the consumer's kernel alone, so
// that both ops of the pair always agree.
bool f8_pair_active(const aq_engine* e, int ci, int B) {
    if (ci < 0 || !e->packed[ci].w_f8) return false;
    const aq_op_desc& c = e->ops[ci];
    const TensorPlace& ps = e->place[c.src.tensor];
    return aq_conv3x3_pl_f8_supported(c.src.channels, c.dst.channels, B, ps.h, ps.w) != 0;
}

inline void note_launch(aq_engine* e, int oi, int family, int cfg = -1) {
    if (e->last_family.size() != e->ops.size()) { e->last_family.assign(e->ops.size(), AQ_FAM_NONE); e->last_cfg.assign(e->ops.size(), -1); }
    e->last_family[oi] = family; e->last_cfg[oi] = cfg;
}

int run_conv(aq_engine* e, int oi, void* ws, const uint8_t* tiles, int B, hipStream_t stream, int force_cfg = -1, bool no_table = false) {
    if (force_cfg < 0 && !e->calib_amax) {              // (the tuner and the calibration pass run every op in bf16)
        const aq_op_desc& op = e->ops[oi];
        const PackedW& pw = e->packed[oi];
        if (pw.f8_consumer >= 0 && f8_pair_active(e, pw.f8_consumer, B)) {
            // producer: e4m3 codes of y / scale into the first bytes of each pixel's bf16 slot of the temporary (pitch = the bf16 row)
            const TensorPlace& pd = e->place[op.dst.tensor];
            note_launch(e, oi, AQ_FAM_DIRECT1X1_F8OUT);
            return aq_conv1x1_direct_f8out(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                                           tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels * 2, op.dst.ch_off * 2,
                                           op.src.channels, op.dst.channels, pw.w_direct, pw.bias, (long long)B * pd.h * pd.w, op.act,
                                           pw.act_scale, stream);
        }
        if (pw.w_f8 && f8_pair_active(e, oi, B)) {
            const TensorPlace& ps = e->place[op.src.tensor];
            note_launch(e, oi, AQ_FAM_PL3X3_F8);
            return aq_conv3x3_pl_f8(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels * 2, op.src.ch_off * 2, op.src.channels,
                                    tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off, op.dst.channels,
                                    op.res.tensor >= 0 ? tptr(e, ws, tiles, op.res.tensor) : nullptr,
                                    op.res.tensor >= 0 ? e->tensors[op.res.tensor].channels : 0, op.res.ch_off,
                                    pw.w_f8, pw.sb_f8, B, ps.h, ps.w, op.act, stream);
        }
    }
    if (force_cfg < 0 && !no_table && e->conv_cfg[oi] < 0 && e->tuned_B > 0 && B != e->tuned_B && e->tuned_H == e->lay_H && e->tuned_W == e->lay_W &&
        e->tuned_cfg[oi] >= 0) {
        // A table entry was validated (and timed) at tuned_B only.  Should it reject another batch size -- the ragged last batch of a
        // sweep, a rank with fewer tiles -- that batch runs on the heuristic shape instead of ending the sweep.
        const int rc = run_conv(e, oi, ws, tiles, B, stream, e->tuned_cfg[oi]);
        if (rc != AQ_ERR_INVALID) return rc;
        return run_conv(e, oi, ws, tiles, B, stream, -1, true);
    }
    const aq_op_desc& op = e->ops[oi];
    const PackedW& pw = e->packed[oi];
    const int prec = e->desc.precision;
    const TensorPlace& ps = e->place[op.src.tensor];
    const TensorPlace& pd = e->place[op.dst.tensor];
    const int eb = aq_elem_bytes(prec);
    const int out_f32 = e->tensors[op.dst.tensor].dtype == AQ_T_F32;
    ConvParams p{};
    p.in = tptr(e, ws, tiles, op.src.tensor) + (size_t)op.src.ch_off * eb;
    p.in_ld_b = e->tensors[op.src.tensor].channels * eb;
    p.out = tptr(e, ws, tiles, op.dst.tensor) + (size_t)op.dst.ch_off * pd.elem;
    p.out_ld_b = e->tensors[op.dst.tensor].channels * pd.elem;
    if (op.res.tensor >= 0) {
        p.res = tptr(e, ws, tiles, op.res.tensor) + (size_t)op.res.ch_off * eb;
        p.res_ld_b = e->tensors[op.res.tensor].channels * eb;
    }
    p.w = (const char*)pw.w; p.bias = pw.bias; p.zero = (const char*)e->zero_page;
    p.B = B; p.H = ps.h; p.W = ps.w; p.Ho = pd.h; p.Wo = pd.w;
    p.cout = op.dst.channels;
    p.k = op.k; p.stride = op.stride; p.pad = op.pad; p.taps = op.k * op.k;
    p.G = pw.G; p.kgroups = pw.kgroups; p.kgroups_pad = pw.kgroups_pad; p.nchunks = pw.kgroups_pad / 8;
    p.npix = B * pd.h * pd.w;
    p.act = op.act;
    int cfg = force_cfg >= 0 ? force_cfg : e->conv_cfg[oi];
    // The tuned table serves EVERY batch size of its tile geometry (a sweep's ragged last batch, a rank with fewer tiles): a kernel's
    // accumulation order per output element does not depend on the batch, so a tile's result no longer depends on which batch it
    // landed in (before, batches of another size fell back to the heuristic kernels and could round differently in bf16).
    if (cfg < 0 && !no_table && e->tuned_B > 0 && e->tuned_H == e->lay_H && e->tuned_W == e->lay_W) cfg = e->tuned_cfg[oi];
    if (cfg == AQ_CONV_CFG_DIRECT3X3S2) {
        if (pw.direct_cfg != cfg) { aq_set_error("conv op %d has no direct 3x3/s2 form", oi); return AQ_ERR_INVALID; }
        note_launch(e, oi, AQ_FAM_DIRECT3X3S2, cfg);
        return aq_conv3x3s2_direct(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                                   tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                                   op.src.channels, op.dst.channels, pw.w_direct, pw.bias, B, ps.h, ps.w, op.act, stream);
    }
    if (cfg == AQ_CONV_CFG_PL3X3) {
        if (pw.direct_cfg != cfg) { aq_set_error("conv op %d has no planar 3x3 form", oi); return AQ_ERR_INVALID; }
        const int ld = e->tensors[op.src.tensor].channels;
        note_launch(e, oi, pw.w_pl8 && aq_conv3x3_pl_w8_supported(op.src.channels, op.dst.channels, B, ps.h, ps.w) ? AQ_FAM_PL3X3_W8 : AQ_FAM_PL3X3, cfg);
        if (pw.w_pl8 && aq_conv3x3_pl_w8_supported(op.src.channels, op.dst.channels, B, ps.h, ps.w))
            return aq_conv3x3_pl_w8(tptr(e, ws, tiles, op.src.tensor) + (size_t)op.src.ch_off * 2, (long long)ld * 2, 16, op.src.channels,
                                    tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off, op.dst.channels,
                                    op.res.tensor >= 0 ? tptr(e, ws, tiles, op.res.tensor) : nullptr,
                                    op.res.tensor >= 0 ? e->tensors[op.res.tensor].channels : 0, op.res.ch_off,
                                    pw.w_pl8, pw.sb_pl8, B, ps.h, ps.w, op.act, stream);
        return aq_conv3x3_pl(tptr(e, ws, tiles, op.src.tensor) + (size_t)op.src.ch_off * 2, (long long)ld * 2, 16, op.src.channels,
                             tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off, op.dst.channels,
                             op.res.tensor >= 0 ? tptr(e, ws, tiles, op.res.tensor) : nullptr,
                             op.res.tensor >= 0 ? e->tensors[op.res.tensor].channels : 0, op.res.ch_off,
                             pw.w_direct, pw.bias, B, ps.h, ps.w, op.act, stream);
    }
    if (cfg == AQ_CONV_CFG_PL3X3S2) {
        if (pw.direct_cfg != cfg) { aq_set_error("conv op %d has no planar 3x3/s2 form", oi); return AQ_ERR_INVALID; }
        if (!aq_conv3x3_pl_s2_supported(op.src.channels, op.dst.channels, B, ps.h, ps.w)) {      // (the tile's region depends on the image width)
            aq_set_error("conv op %d: planar 3x3/s2 does not fit %d x %d x %d", oi, B, ps.h, ps.w);
            return AQ_ERR_INVALID;
        }
        note_launch(e, oi, AQ_FAM_PL3X3S2, cfg);
        return aq_conv3x3_pl_s2(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off, op.src.channels,
                                tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off, op.dst.channels,
                                pw.w_direct, pw.bias, B, ps.h, ps.w, op.act, stream);
    }
    if (cfg == AQ_CONV_CFG_DIRECT1X1) {
        if (pw.direct_cfg != cfg) { aq_set_error("conv op %d has no direct 1x1 form", oi); return AQ_ERR_INVALID; }
        note_launch(e, oi, AQ_FAM_DIRECT1X1, cfg);
        return aq_conv1x1_direct(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                                 tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                                 op.src.channels, op.dst.channels, pw.w_direct, pw.bias, (long long)B * pd.h * pd.w, op.act, stream);
    }
    if (cfg == AQ_CONV_CFG_ASM1X1) {
        if (pw.direct_cfg != cfg) { aq_set_error("conv op %d has no assembly 1x1 form", oi); return AQ_ERR_INVALID; }
        note_launch(e, oi, AQ_FAM_ASM1X1, cfg);
        return aq_conv1x1_asm(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                              tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                              op.src.channels, op.dst.channels, pw.w_direct, pw.bias, (long long)B * pd.h * pd.w, op.act, stream);
    }
    if (cfg < 0) cfg = aq_conv_pick_config(p.cout, p.npix, prec);
    if (pw.x3) p.x3_off = pw.cout_rows;
    note_launch(e, oi, AQ_FAM_IGEMM_OR_HALO, cfg);
    return aq_launch_conv(p, pw.x3 ? (int)AQ_F16X3 : prec, out_f32, cfg, stream);
}

int run_plan(aq_engine* e, const uint8_t* tiles, int B, int H, int W, void* ws, size_t ws_bytes,
             float* pred_out, aq_det* dets, int32_t* counts, float conf, float iou, int max_det, hipStream_t stream) {
    AQ_REQUIRE(e && tiles && ws, "infer: null pointer");
    int rc = layout(e, B, H, W);
    if (rc) return rc;
    if (ws_bytes < e->total_bytes) {
        aq_set_error("workspace too small: need %zu bytes, got %zu", e->total_bytes, ws_bytes);
        return AQ_ERR_WORKSPACE;
    }
    e->last_ws = ws; e->last_tiles = tiles;
    const int prec = e->desc.precision;
    const int n_ops = (int)e->ops.size();
    hipEvent_t* ev = nullptr;
    if (e->prof && e->ring > 0) ev = e->ev.data() + (size_t)(e->prof_calls % e->ring) * (n_ops + 1);
    float* pred = pred_out ? pred_out : (float*)((char*)ws + e->off_pred);
    int32_t* cand = (int32_t*)((char*)ws + e->off_cand);
    int32_t* cand_count = (int32_t*)((char*)ws + e->off_cand_count);
    float* cand_rows = (float*)((char*)ws + e->off_cand_rows);
    // model.0 + model.1 + model.2.cv1|cv2 as ONE launch (csrc/downblock.hip, STEM form; AQ_STEMDOWN=1).  Opt-in: it removes the stem
    // output's 629 MB write and 613 MB read per 64-tile batch (10 % of the step's HBM traffic) and is bit-identical, but the stem's MFMAs,
    // SiLU and LDS traffic then sit between the same barriers as the down-block's instead of overlapping another kernel's memory time:
    // 480 us against 211 + 253, and 14.88 k vs 14.83 k tiles/s in bench.py (three runs each) -- a wash.  The stem's output tensor is not
    // written in this mode (tensor taps read it).
    const char* use_sd = getenv("AQ_STEMDOWN");
    const bool stemdown = use_sd && *use_sd == '1' && prec == AQ_BF16 && n_ops > 2 && e->ops[0].kind == AQ_OP_STEM && e->ops[1].kind == AQ_OP_DOWNBLOCK &&
                          e->ops[0].dst.channels == 48 && e->ops[1].src.tensor == e->ops[0].dst.tensor && e->ops[1].src.ch_off == e->ops[0].dst.ch_off &&
                          aq_stemdown_supported(H, W) && ((uintptr_t)tiles & 3) == 0;
    // the NMS path of a bf16 engine whose three head convs all have a fused form skips the fp32 head maps
    bool fuse_heads = dets != nullptr, heads_started = false;
    int n_heads = 0, heads_done = 0;
    for (int oi = 0; oi < n_ops; ++oi)
        if (e->ops[oi].kind == AQ_OP_CONV && e->ops[oi].level >= 0) {
            ++n_heads;
            if (!e->packed[oi].w_head) fuse_heads = false;
        }
    if (n_heads != 3) fuse_heads = false;
    for (int oi = 0; oi < n_ops; ++oi) {
        const aq_op_desc& op = e->ops[oi];
        if (ev) AQ_CHECK_HIP(hipEventRecord(ev[oi], stream));
        const bool mark = g_roctx.on();
        if (mark) {
            char name[96];
            snprintf(name, sizeof name, "aq op %02d kind %d: %d -> %d ch, k%d s%d%s", oi, op.kind, op.src.channels, op.dst.channels, op.k, op.stride,
                     op.level >= 0 ? " (detect head)" : "");
            g_roctx.push(name);
        }
        switch (op.kind) {
        case AQ_OP_PREPROCESS:
            rc = aq_preprocess_s2d(tiles, tptr(e, ws, tiles, op.dst.tensor), B, H, W, prec, stream);
            break;
        case AQ_OP_CONV:
            if (fuse_heads && op.level >= 0) {
                // Detect level fused with its decode (csrc/head_decode.hip): candidates go straight to the compact list NMS reads
                int32_t* wide = (int32_t*)((char*)ws + e->off_cand_wide);
                if (!heads_started) {
                    AQ_CHECK_HIP(hipMemsetAsync(wide, 0, sizeof(int32_t) * B * kCountStride, stream));
                    heads_started = true;
                }
                const TensorPlace& pl = e->place[op.src.tensor];
                int off = 0;
                for (int l = 0; l < op.level; ++l) off += e->desc.na * (H / (int)e->desc.stride[l]) * (W / (int)e->desc.stride[l]);
                float anchors[8 * 2];
                for (int a = 0; a < e->desc.na; ++a) {
                    anchors[2 * a] = e->desc.anchors_px[op.level][a][0];
                    anchors[2 * a + 1] = e->desc.anchors_px[op.level][a][1];
                }
                note_launch(e, oi, AQ_FAM_HEAD_DECODE);
                rc = aq_head_decode(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off, op.src.channels,
                                    e->packed[oi].w_head, B, pl.h, pl.w, off, e->desc.stride[op.level], anchors, e->desc.nc, e->desc.na, conf,
                                    cand, cand_rows, wide, kCountStride, e->N, stream);
                if (!rc && ++heads_done == 3) rc = aq_head_counts_gather(wide, kCountStride, cand_count, B, stream);
                break;
            }
            rc = run_conv(e, oi, ws, tiles, B, stream);
            if (!rc && e->calib_amax && e->tensors[op.dst.tensor].dtype == AQ_T_ACT && prec == AQ_BF16 && op.dst.channels % 8 == 0) {
                const TensorPlace& pdc = e->place[op.dst.tensor];
                rc = aq_absmax_bf16(tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off, op.dst.channels,
                                    (long long)B * pdc.h * pdc.w, e->calib_amax + oi, stream);
            }
            break;
        case AQ_OP_STEM:
            note_launch(e, oi, AQ_FAM_STEM);
            if (stemdown) break;                         // computed inside the down-block launch below
            rc = aq_stem_conv(tiles, tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                              op.dst.channels, e->packed[oi].w, e->packed[oi].bias, B, H, W, op.act, prec, stream);
            break;
        case AQ_OP_BOTTLENECK: {
            note_launch(e, oi, AQ_FAM_BOTTLENECK);
            const TensorPlace& pl = e->place[op.src.tensor];
            rc = aq_bottleneck(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                               tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                               op.src.channels, e->packed[oi].w, e->packed[oi].bias, B, pl.h, pl.w, op.res.tensor >= 0, stream);
            break;
        }
        case AQ_OP_DOWNBLOCK: {
            note_launch(e, oi, AQ_FAM_DOWNBLOCK);
            const TensorPlace& pl = e->place[op.src.tensor];
            if (stemdown && oi == 1) {
                rc = aq_stemdown(tiles, tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                                 e->packed[0].w, e->packed[0].bias, e->packed[oi].w, e->packed[oi].bias, B, H, W, stream);
                break;
            }
            rc = aq_downblock(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                              tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                              e->packed[oi].w, e->packed[oi].bias, B, pl.h, pl.w, stream);
            break;
        }
        case AQ_OP_SPPF_POOL: {
            const TensorPlace& pl = e->place[op.src.tensor];
            rc = aq_sppf_pool(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                              op.src.channels, B, pl.h, pl.w, prec, stream);
            break;
        }
        case AQ_OP_UPSAMPLE2X: {
            const TensorPlace& pl = e->place[op.src.tensor];
            rc = aq_upsample2x(tptr(e, ws, tiles, op.src.tensor), e->tensors[op.src.tensor].channels, op.src.ch_off,
                               tptr(e, ws, tiles, op.dst.tensor), e->tensors[op.dst.tensor].channels, op.dst.ch_off,
                               op.src.channels, B, pl.h, pl.w, prec, stream);
            break;
        }
        case AQ_OP_DECODE: {
            if (fuse_heads) break;                       // done level by level behind the head convs
            const float* heads[3];
            for (int l = 0; l < 3; ++l) heads[l] = (const float*)tptr(e, ws, tiles, e->desc.head_tensor[l]);
            float anchors[3 * 8 * 2];
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < e->desc.na; ++a) {
                    anchors[(l * e->desc.na + a) * 2] = e->desc.anchors_px[l][a][0];
                    anchors[(l * e->desc.na + a) * 2 + 1] = e->desc.anchors_px[l][a][1];
                }
            const bool want_nms = dets != nullptr;
            rc = aq_detect_decode(heads, e->tensors[e->desc.head_tensor[0]].channels, B, H, W, e->desc.nc, e->desc.na,
                                  anchors, e->desc.stride, want_nms ? nullptr : pred, conf, want_nms ? cand : nullptr,
                                  want_nms ? cand_rows : nullptr, want_nms ? cand_count : nullptr, e->N, stream);
            break;
        }
        case AQ_OP_NMS:
            if (dets)
                rc = aq_nms_opts(cand_rows, e->N, B, e->N, e->desc.nc, conf, iou, max_det, cand, cand_count, e->N,
                                 (char*)ws + e->off_nms, dets, counts, e->nms_agnostic, e->nms_cls_lo, e->nms_cls_hi, stream);
            break;
        default:
            aq_set_error("unknown op kind %d", op.kind);
            rc = AQ_ERR_INVALID;
        }
        if (mark) g_roctx.pop();
        if (rc) return rc;
    }
    if (ev) {
        AQ_CHECK_HIP(hipEventRecord(ev[n_ops], stream));
        e->prof_calls++;
    }
    return AQ_OK;
}

}  // namespace

extern "C" int aq_pack_conv_weights(const float* w, int cout, int k, int cin, int precision, void* packed_dev,
                                    size_t* bytes, void* stream) {
    AQ_REQUIRE(w && bytes, "pack: null pointer");
    AQ_REQUIRE(cout > 0 && k > 0 && cin > 0 && (cin * aq_elem_bytes(precision)) % 16 == 0,
               "pack: cin=%d must make whole 16-byte groups", cin);
    std::vector<unsigned char> host;
    int kg, kgp, g, rows;
    pack_host(w, cout, k, cin, precision, host, &kg, &kgp, &g, &rows);
    *bytes = host.size();
    if (packed_dev) {
        AQ_CHECK_HIP(hipMemcpyAsync(packed_dev, host.data(), host.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
        AQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));   // host staging buffer dies here
    }
    return AQ_OK;
}

extern "C" int aq_pack_conv_weights_x3(const float* w, const float* bias, int cout, int k, int cin, void* packed_dev, size_t* bytes,
                                       float* bias_scale_dev, size_t* bias_floats, void* stream) {
    AQ_REQUIRE(w && bytes && bias_floats, "pack_conv_weights_x3: null pointer");
    AQ_REQUIRE(cout > 0 && k > 0 && cin > 0 && cin % 8 == 0, "pack_conv_weights_x3: cin %d is not a multiple of 8", cin);
    std::vector<unsigned char> host;
    std::vector<float> bs;
    int kg, kgp, g, rows;
    pack_host_x3(w, bias, cout, k, cin, host, bs, &kg, &kgp, &g, &rows);
    *bytes = host.size();
    *bias_floats = bs.size();
    if (!packed_dev) return AQ_OK;
    AQ_REQUIRE(bias && bias_scale_dev, "pack_conv_weights_x3: null pointer");
    AQ_CHECK_HIP(hipMemcpyAsync(packed_dev, host.data(), host.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
    AQ_CHECK_HIP(hipMemcpyAsync(bias_scale_dev, bs.data(), bs.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    AQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return AQ_OK;
}

extern "C" int aq_conv2d(const void* in_dev, int in_ld, int in_choff, int cin, void* out_dev, int out_ld, int out_choff,
                         int cout, const void* res_dev, int res_ld, int res_choff, const void* packed_w_dev,
                         const float* bias_dev, int B, int H, int W, int k, int stride, int pad, int act, int precision,
                         int out_f32, const void* zero_page_dev, void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev && zero_page_dev, "conv2d: null pointer");
    const int eb = aq_elem_bytes(precision);
    const int oeb = (eb == 4 || out_f32) ? 4 : 2;
    AQ_REQUIRE((cin * eb) % 16 == 0 && (in_ld * eb) % 16 == 0 && (in_choff * eb) % 16 == 0,
               "conv2d: input channels must be whole 16-byte groups (cin=%d ld=%d off=%d)", cin, in_ld, in_choff);
    AQ_REQUIRE(cout % 8 == 0 && out_choff % 8 == 0 && out_ld % 8 == 0, "conv2d: cout/out_choff/out_ld must be multiples of 8");
    AQ_REQUIRE(!res_dev || (res_choff % 8 == 0 && res_ld % 8 == 0), "conv2d: residual slice must be 8-channel aligned");
    AQ_REQUIRE(k >= 1 && k * k <= 25 && stride >= 1 && pad >= 0, "conv2d: unsupported k=%d stride=%d pad=%d", k, stride, pad);
    AQ_REQUIRE(precision == AQ_BF16 || precision == AQ_FP32 || precision == AQ_F16X3, "conv2d: bad precision %d", precision);
    AQ_REQUIRE(precision != AQ_F16X3 || cin % 8 == 0, "conv2d: split mode needs cin %% 8 == 0");
    ConvParams p{};
    p.x3_off = precision == AQ_F16X3 ? (cout + kCoutSlack + 31) / 32 * 32 : 0;      // bias_dev = aq_pack_conv_weights_x3's bias / scale buffer
    p.in = (const char*)in_dev + (size_t)in_choff * eb; p.in_ld_b = in_ld * eb;
    p.out = (char*)out_dev + (size_t)out_choff * oeb; p.out_ld_b = out_ld * oeb;
    if (res_dev) { p.res = (const char*)res_dev + (size_t)res_choff * eb; p.res_ld_b = res_ld * eb; }
    p.w = (const char*)packed_w_dev; p.bias = bias_dev; p.zero = (const char*)zero_page_dev;
    p.B = B; p.H = H; p.W = W;
    p.Ho = (H + 2 * pad - k) / stride + 1; p.Wo = (W + 2 * pad - k) / stride + 1;
    AQ_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d: empty output");
    p.cout = cout; p.k = k; p.stride = stride; p.pad = pad; p.taps = k * k;
    p.G = cin * eb / 16; p.kgroups = p.taps * p.G; p.kgroups_pad = (p.kgroups + 7) / 8 * 8; p.nchunks = p.kgroups_pad / 8;
    p.npix = B * p.Ho * p.Wo; p.act = act;
    int cfg = aq_conv_pick_config(cout, p.npix, precision);
    const char* forced = getenv("AQ_CONV_CFG");
    if (forced && *forced) cfg = atoi(forced);
    return aq_launch_conv(p, precision, out_f32, cfg, (hipStream_t)stream);
}

extern "C" int aq_engine_create(const aq_model_desc* d, int device, aq_engine** out) {
    AQ_REQUIRE(d && out, "engine_create: null pointer");
    AQ_REQUIRE(d->n_ops > 0 && d->n_tensors > 0 && d->ops && d->tensors, "engine_create: empty plan");
    AQ_REQUIRE(d->nl == 3 && d->na >= 1 && d->na <= 8 && d->nc >= 1, "engine_create: unsupported head nl=%d na=%d nc=%d", d->nl, d->na, d->nc);
    AQ_REQUIRE(d->precision == AQ_BF16 || d->precision == AQ_FP32 || d->precision == AQ_BF16_W8 || d->precision == AQ_F16X3,
               "engine_create: bad precision %d", d->precision);
    {   // One device per process (one process per GPU is how every entry point of this package runs): the kernels' launch-attribute,
        // CU-count and occupancy caches in conv_igemm / conv_halo / downblock / conv1x1_direct / detect_nms are process-global, and
        // hipFuncSetAttribute is per device -- a second device would skip it and fail to launch anything above 64 KiB of LDS.
        static std::atomic<int> first_device{-1};
        int expected = -1;
        if (!first_device.compare_exchange_strong(expected, device))
            AQ_REQUIRE(expected == device, "engine_create: this process already runs engines on device %d; device %d needs its own process", expected, device);
    }
    AQ_CHECK_HIP(hipSetDevice(device));
    aq_engine* e = new aq_engine();
    e->device = device;
    aq_model_desc compute_desc = *d;                   // AQ_BF16_W8 computes exactly as AQ_BF16; only the planar 3x3 weight stream differs
    const bool w8 = d->precision == AQ_BF16_W8;
    if (w8) compute_desc.precision = AQ_BF16;
    // AQ_F16X3 is AQ_FP32 everywhere (tensors, stem, pools, head, decode) except inside the implicit-GEMM convs, which take their
    // products as three fp16 MFMAs on split operands (conv_igemm.hip); a conv whose Cin is not a multiple of 8 stays on the fp32 kernel
    const bool x3 = d->precision == AQ_F16X3;
    if (x3) compute_desc.precision = AQ_FP32;
    d = &compute_desc;
    e->desc = *d;
    e->tensors.assign(d->tensors, d->tensors + d->n_tensors);
    e->ops.assign(d->ops, d->ops + d->n_ops);
    e->desc.tensors = e->tensors.data();
    e->desc.ops = e->ops.data();
    e->packed.resize(e->ops.size());
    e->conv_cfg.assign(e->ops.size(), -1);
    e->tuned_cfg.assign(e->ops.size(), -1);
    const int eb = aq_elem_bytes(d->precision);
    auto fail = [&](int rc) { aq_engine_destroy(e); return rc; };
    if (hipMalloc(&e->zero_page, 4096) != hipSuccess || hipMemset(e->zero_page, 0, 4096) != hipSuccess) {
        aq_set_error("engine_create: zero page allocation failed");
        return fail(AQ_ERR_NOMEM);
    }
    for (size_t oi = 0; oi < e->ops.size(); ++oi) {
        aq_op_desc& op = e->ops[oi];
        for (const aq_slice* s : {&op.src, &op.dst}) {
            if (op.kind == AQ_OP_DECODE || op.kind == AQ_OP_NMS) continue;
            if (s->tensor < 0 || s->tensor >= d->n_tensors || s->ch_off < 0 ||
                s->ch_off + s->channels > e->tensors[s->tensor].channels) {
                aq_set_error("engine_create: op %zu has a slice outside its tensor", oi);
                return fail(AQ_ERR_INVALID);
            }
        }
        if (op.kind == AQ_OP_STEM) {
            PackedW& pw = e->packed[oi];
            size_t nb = 0;
            std::vector<float> bias(64, 0.0f);
            if (!op.weight || !op.bias || op.k != 6 || op.stride != 2 || op.pad != 2 || op.src.channels != 3 || op.dst.channels > 64 ||
                aq_pack_stem_weights(op.weight, op.dst.channels, d->precision, nullptr, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: stem op %zu unsupported (needs k=6 s=2 p=2, 3 -> <=64 channels)", oi);
                return fail(AQ_ERR_INVALID);
            }
            memcpy(bias.data(), op.bias, sizeof(float) * op.dst.channels);
            if (hipMalloc(&pw.w, nb) != hipSuccess || hipMalloc((void**)&pw.bias, bias.size() * sizeof(float)) != hipSuccess) {
                aq_set_error("engine_create: stem weight allocation failed");
                return fail(AQ_ERR_NOMEM);
            }
            if (aq_pack_stem_weights(op.weight, op.dst.channels, d->precision, pw.w, &nb, nullptr) != AQ_OK ||
                hipMemcpy(pw.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                aq_set_error("engine_create: stem weight upload failed");
                return fail(AQ_ERR_HIP);
            }
            op.weight = nullptr; op.bias = nullptr;
            continue;
        }
        if (op.kind == AQ_OP_BOTTLENECK) {
            PackedW& pw = e->packed[oi];
            const int C = op.src.channels;
            size_t nb = 0;
            if (!op.weight || !op.bias || d->precision != AQ_BF16 || op.k != 3 || op.stride != 1 || op.pad != 1 || op.dst.channels != C ||
                op.src.tensor == op.dst.tensor ||
                aq_pack_bottleneck_weights(op.weight, op.weight + (size_t)C * C, C, nullptr, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: bottleneck op %zu unsupported (bf16, C in {16,32,48,64,96}, distinct src/dst tensors)", oi);
                return fail(AQ_ERR_INVALID);
            }
            if (hipMalloc(&pw.w, nb) != hipSuccess || hipMalloc((void**)&pw.bias, 2 * C * sizeof(float)) != hipSuccess) {
                aq_set_error("engine_create: bottleneck weight allocation failed");
                return fail(AQ_ERR_NOMEM);
            }
            if (aq_pack_bottleneck_weights(op.weight, op.weight + (size_t)C * C, C, pw.w, &nb, nullptr) != AQ_OK ||
                hipMemcpy(pw.bias, op.bias, 2 * C * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                aq_set_error("engine_create: bottleneck weight upload failed");
                return fail(AQ_ERR_HIP);
            }
            op.weight = nullptr; op.bias = nullptr;
            continue;
        }
        if (op.kind == AQ_OP_DOWNBLOCK) {
            PackedW& pw = e->packed[oi];
            size_t nb = 0;
            if (!op.weight || !op.bias || d->precision != AQ_BF16 || op.k != 3 || op.stride != 2 || op.pad != 1 || op.src.channels != 48 ||
                op.dst.channels != 96 || aq_pack_downblock_weights(op.weight, op.weight + (size_t)96 * 9 * 48, nullptr, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: downblock op %zu unsupported (bf16, 3x3/s2 48 -> 96 then 1x1 96 -> 96)", oi);
                return fail(AQ_ERR_INVALID);
            }
            if (hipMalloc(&pw.w, nb) != hipSuccess || hipMalloc((void**)&pw.bias, 192 * sizeof(float)) != hipSuccess) {
                aq_set_error("engine_create: downblock weight allocation failed");
                return fail(AQ_ERR_NOMEM);
            }
            if (aq_pack_downblock_weights(op.weight, op.weight + (size_t)96 * 9 * 48, pw.w, &nb, nullptr) != AQ_OK ||
                hipMemcpy(pw.bias, op.bias, 192 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                aq_set_error("engine_create: downblock weight upload failed");
                return fail(AQ_ERR_HIP);
            }
            op.weight = nullptr; op.bias = nullptr;
            continue;
        }
        if (op.kind != AQ_OP_CONV) continue;
        if (!op.weight || !op.bias || (op.src.channels * eb) % 16 != 0 || op.dst.channels % 8 != 0 || op.dst.ch_off % 8 != 0 || op.k * op.k > 25) {
            aq_set_error("engine_create: conv op %zu unsupported (cin=%d cout=%d k=%d)", oi, op.src.channels, op.dst.channels, op.k);
            return fail(AQ_ERR_INVALID);
        }
        std::vector<unsigned char> host;
        PackedW& pw = e->packed[oi];
        std::vector<float> bias;
        if (x3 && op.src.channels % 8 == 0) {
            pack_host_x3(op.weight, op.bias, op.dst.channels, op.k, op.src.channels, host, bias, &pw.kgroups, &pw.kgroups_pad, &pw.G, &pw.cout_rows);
            pw.x3 = true;
        } else {
            pack_host(op.weight, op.dst.channels, op.k, op.src.channels, d->precision, host, &pw.kgroups, &pw.kgroups_pad, &pw.G, &pw.cout_rows);
            bias.assign(pw.cout_rows, 0.0f);
            memcpy(bias.data(), op.bias, sizeof(float) * op.dst.channels);
        }
        if (hipMalloc(&pw.w, host.size()) != hipSuccess || hipMalloc((void**)&pw.bias, bias.size() * sizeof(float)) != hipSuccess) {
            aq_set_error("engine_create: weight allocation failed (op %zu, %zu bytes)", oi, host.size());
            return fail(AQ_ERR_NOMEM);
        }
        if (hipMemcpy(pw.w, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(pw.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            aq_set_error("engine_create: weight upload failed (op %zu)", oi);
            return fail(AQ_ERR_HIP);
        }
        if (d->precision == AQ_BF16 && op.k == 1 && op.stride == 1 && op.res.tensor < 0 && e->tensors[op.dst.tensor].dtype == AQ_T_ACT &&
            aq_conv1x1_direct_supported(op.src.channels, op.dst.channels)) {
            size_t nb = 0;
            if (aq_pack_conv1x1_direct(op.weight, op.src.channels, op.dst.channels, nullptr, &nb, nullptr) != AQ_OK ||
                hipMalloc(&pw.w_direct, nb) != hipSuccess ||
                aq_pack_conv1x1_direct(op.weight, op.src.channels, op.dst.channels, pw.w_direct, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: direct 1x1 weight upload failed (op %zu)", oi);
                return fail(AQ_ERR_HIP);
            }
            pw.direct_cfg = AQ_CONV_CFG_DIRECT1X1;
        }
        if (d->precision == AQ_BF16 && op.k == 1 && op.stride == 1 && op.res.tensor < 0 && op.act == 1 && e->tensors[op.dst.tensor].dtype == AQ_T_ACT &&
            pw.direct_cfg < 0 && op.src.ch_off % 8 == 0 && e->tensors[op.src.tensor].channels % 8 == 0 && op.dst.ch_off % 4 == 0 &&
            e->tensors[op.dst.tensor].channels % 4 == 0 && aq_conv1x1_asm_supported(op.src.channels, op.dst.channels)) {
            size_t nb = 0;
            if (aq_pack_conv1x1_asm(op.weight, op.src.channels, op.dst.channels, nullptr, &nb, nullptr) != AQ_OK ||
                hipMalloc(&pw.w_direct, nb) != hipSuccess ||
                aq_pack_conv1x1_asm(op.weight, op.src.channels, op.dst.channels, pw.w_direct, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: assembly 1x1 weight upload failed (op %zu)", oi);
                return fail(AQ_ERR_HIP);
            }
            pw.direct_cfg = AQ_CONV_CFG_ASM1X1;
        }
        if (d->precision == AQ_BF16 && op.k == 3 && op.stride == 2 && op.pad == 1 && op.res.tensor < 0 && e->tensors[op.dst.tensor].dtype == AQ_T_ACT &&
            aq_conv3x3s2_direct_supported(op.src.channels, op.dst.channels)) {
            size_t nb = 0;
            if (aq_pack_conv3x3s2_direct(op.weight, op.src.channels, op.dst.channels, nullptr, &nb, nullptr) != AQ_OK ||
                hipMalloc(&pw.w_direct, nb) != hipSuccess ||
                aq_pack_conv3x3s2_direct(op.weight, op.src.channels, op.dst.channels, pw.w_direct, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: direct 3x3/s2 weight upload failed (op %zu)", oi);
                return fail(AQ_ERR_HIP);
            }
            pw.direct_cfg = AQ_CONV_CFG_DIRECT3X3S2;
        }
        const char* no_pl2 = getenv("AQ_DISABLE_PL3X3S2");   // A/B switch: keep the planar 3x3/s2 kernel out of the candidate list
        if (d->precision == AQ_BF16 && op.k == 3 && op.stride == 2 && op.pad == 1 && op.res.tensor < 0 && e->tensors[op.dst.tensor].dtype == AQ_T_ACT &&
            pw.direct_cfg < 0 && op.src.ch_off % 8 == 0 && e->tensors[op.src.tensor].channels % 8 == 0 && !(no_pl2 && *no_pl2 == '1') &&
            aq_conv3x3_pl_s2_supported(op.src.channels, op.dst.channels, 1, 2, 2)) {       // channel counts only: the geometry is checked per launch
            size_t nb = 0;
            if (aq_pack_conv3x3_pl_s2(op.weight, op.src.channels, op.dst.channels, nullptr, &nb, nullptr) != AQ_OK ||
                hipMalloc(&pw.w_direct, nb) != hipSuccess ||
                aq_pack_conv3x3_pl_s2(op.weight, op.src.channels, op.dst.channels, pw.w_direct, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: planar 3x3/s2 weight upload failed (op %zu)", oi);
                return fail(AQ_ERR_HIP);
            }
            pw.direct_cfg = AQ_CONV_CFG_PL3X3S2;
        }
        const char* no_pl = getenv("AQ_DISABLE_PL3X3");      // A/B switch: keep the planar 3x3 kernel out of the candidate list
        if (d->precision == AQ_BF16 && op.k == 3 && op.stride == 1 && op.pad == 1 && e->tensors[op.dst.tensor].dtype == AQ_T_ACT &&
            op.src.ch_off % 8 == 0 && aq_conv3x3_pl_supported(op.src.channels, op.dst.channels) && !(no_pl && *no_pl == '1')) {
            size_t nb = 0;
            if (aq_pack_conv3x3_pl(op.weight, op.src.channels, op.dst.channels, nullptr, &nb, nullptr) != AQ_OK ||
                hipMalloc(&pw.w_direct, nb) != hipSuccess ||
                aq_pack_conv3x3_pl(op.weight, op.src.channels, op.dst.channels, pw.w_direct, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: planar 3x3 weight upload failed (op %zu)", oi);
                return fail(AQ_ERR_HIP);
            }
            pw.direct_cfg = AQ_CONV_CFG_PL3X3;
            // AQ_BF16_W8 engines: the e4m3 code stream is built on request only (AQ_PL_W8=1).  Measured (tools/time_conv3x3.py, batch 64):
            // 85.0 vs 82.2 us at 192 ch / 40^2 and 76.6 vs 68.3 us at 384 ch / 20^2 -- the stream is bound by what one wave per SIMD can
            // ISSUE between MFMAs, not by L2 -> register bytes, so halving the bytes at the price of 48 conversion instructions per tap
            // loses.  It is bit-identical to the bf16 stream and halves the planar layers' weight footprint.
            const char* use_w8 = getenv("AQ_PL_W8");
            if (w8 && use_w8 && *use_w8 == '1') {
                // a layer whose weights are not on an fp8 grid (the caller did not quantise it) keeps the bf16 stream
                if (aq_pack_conv3x3_pl_w8(op.weight, op.bias, op.src.channels, op.dst.channels, nullptr, &nb, nullptr, nullptr) == AQ_OK &&
                    hipMalloc(&pw.w_pl8, nb) == hipSuccess && hipMalloc((void**)&pw.sb_pl8, 2048 * sizeof(float)) == hipSuccess) {
                    if (aq_pack_conv3x3_pl_w8(op.weight, op.bias, op.src.channels, op.dst.channels, pw.w_pl8, &nb, pw.sb_pl8, nullptr) != AQ_OK) {
                        (void)hipFree(pw.w_pl8); (void)hipFree(pw.sb_pl8);
                        pw.w_pl8 = nullptr; pw.sb_pl8 = nullptr;
                    }
                }
            }
        }
        const char* no_hf = getenv("AQ_DISABLE_HEAD_FUSION");   // A/B switch: Detect heads as conv + aq_detect_decode even in `infer`
        if (d->precision == AQ_BF16 && op.level >= 0 && op.k == 1 && op.stride == 1 && !op.act && op.res.tensor < 0 &&
            op.dst.channels >= d->na * (d->nc + 5) && op.dst.channels <= 32 && aq_head_decode_supported(op.src.channels, d->na, d->nc) && !(no_hf && *no_hf == '1')) {
            size_t nb = 0;
            if (aq_pack_head_weights(op.weight, op.bias, op.src.channels, op.dst.channels, nullptr, &nb, nullptr) != AQ_OK ||
                hipMalloc(&pw.w_head, nb) != hipSuccess ||
                aq_pack_head_weights(op.weight, op.bias, op.src.channels, op.dst.channels, pw.w_head, &nb, nullptr) != AQ_OK) {
                aq_set_error("engine_create: fused head weight upload failed (op %zu)", oi);
                return fail(AQ_ERR_HIP);
            }
        }
        if (d->precision == AQ_BF16 && op.kind == AQ_OP_CONV && op.k == 3 && op.stride == 1 && op.pad == 1 && e->packed[oi].direct_cfg == AQ_CONV_CFG_PL3X3 &&
            aq_conv3x3_pl_f8_supported(op.src.channels, op.dst.channels, 1, 8, 8)) {
            e->packed[oi].w_host.assign(op.weight, op.weight + (size_t)op.dst.channels * 9 * op.src.channels);
            e->packed[oi].b_host.assign(op.bias, op.bias + op.dst.channels);
        }
        op.weight = nullptr; op.bias = nullptr;   // host pointers are not kept
    }
    *out = e;
    return AQ_OK;
}

extern "C" void aq_engine_destroy(aq_engine* e) {
    if (!e) return;
    for (PackedW& pw : e->packed) {
        if (pw.w) (void)hipFree(pw.w);
        if (pw.w_direct) (void)hipFree(pw.w_direct);
        if (pw.w_head) (void)hipFree(pw.w_head);
        if (pw.w_pl8) (void)hipFree(pw.w_pl8);
        if (pw.w_f8) (void)hipFree(pw.w_f8);
        if (pw.sb_f8) (void)hipFree(pw.sb_f8);
        if (pw.sb_pl8) (void)hipFree(pw.sb_pl8);
        if (pw.bias) (void)hipFree(pw.bias);
    }
    if (e->zero_page) (void)hipFree(e->zero_page);
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    delete e;
}

extern "C" int aq_engine_workspace_bytes(aq_engine* e, int max_batch, int H, int W, size_t* bytes) {
    AQ_REQUIRE(e && bytes, "workspace_bytes: null pointer");
    int rc = layout(e, max_batch, H, W);
    if (rc) return rc;
    *bytes = e->total_bytes;
    return AQ_OK;
}

extern "C" int aq_engine_infer(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W, void* ws, size_t ws_bytes,
                               aq_det* dets_dev, int32_t* counts_dev, float conf, float iou, int max_det, void* stream) {
    AQ_REQUIRE(dets_dev && counts_dev, "infer: null output pointer");
    AQ_REQUIRE(max_det > 0 && conf >= 0.f && conf <= 1.f && iou >= 0.f && iou <= 1.f,
               "infer: bad thresholds conf=%g iou=%g max_det=%d", conf, iou, max_det);
    return run_plan(e, tiles_dev, B, H, W, ws, ws_bytes, nullptr, dets_dev, counts_dev, conf, iou, max_det, (hipStream_t)stream);
}

extern "C" int aq_engine_forward_raw(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W, void* ws, size_t ws_bytes,
                                     float* pred_dev, void* stream) {
    AQ_REQUIRE(pred_dev, "forward_raw: null output pointer");
    return run_plan(e, tiles_dev, B, H, W, ws, ws_bytes, pred_dev, nullptr, nullptr, 0.f, 0.f, 1, (hipStream_t)stream);
}

// fp8 path, step 1: one bf16 forward pass over `tiles_dev` recording max |y| of every conv op's output (0 for the others) -- the caller turns
// the Bottleneck cv1 outputs' maxima into per-tensor e4m3 scales (max / 448, with whatever margin it wants) for aq_engine_set_fp8_scales.
extern "C" int aq_engine_calibrate_amax(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W, void* ws, size_t ws_bytes,
                                        float* amax_host, int n_ops, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ_REQUIRE(e && tiles_dev && ws && amax_host && n_ops == (int)e->ops.size(), "calibrate_amax: expected %d entries", e ? (int)e->ops.size() : 0);
    AQ_REQUIRE(e->desc.precision == AQ_BF16, "calibrate_amax: bf16 engines only");
    int rc = layout(e, B, H, W);
    if (rc) return rc;
    if (ws_bytes < e->total_bytes) { aq_set_error("calibrate_amax: workspace too small"); return AQ_ERR_WORKSPACE; }
    float* dev = nullptr;
    AQ_CHECK_HIP(hipMalloc((void**)&dev, sizeof(float) * n_ops));
    hipError_t he = hipMemsetAsync(dev, 0, sizeof(float) * n_ops, stream);
    e->calib_amax = dev;
    if (he == hipSuccess) rc = run_plan(e, tiles_dev, B, H, W, ws, ws_bytes, (float*)((char*)ws + e->off_pred), nullptr, nullptr, 0.f, 0.f, 1, stream);
    e->calib_amax = nullptr;
    if (he == hipSuccess && !rc) he = hipMemcpyAsync(amax_host, dev, sizeof(float) * n_ops, hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess && !rc) he = hipStreamSynchronize(stream);
    (void)hipFree(dev);
    if (rc) return rc;
    AQ_CHECK_HIP(he);
    return AQ_OK;
}

// fp8 path, step 2: act_scale[i] > 0 marks conv op i -- a 3x3 / stride-1 layer with a planar form -- as an fp8 consumer whose INPUT tensor
// carries e4m3 codes of x / act_scale[i]; its producer (the 1x1 conv whose output slice is exactly that input, read by nothing else) is
// switched to the code-writing epilogue.  Ops that do not qualify are refused (AQ_ERR_INVALID); act_scale <= 0 returns an op to bf16.
extern "C" int aq_engine_set_fp8_scales(aq_engine* e, const float* act_scale, int n_ops) {
    AQ_REQUIRE(e && act_scale && n_ops == (int)e->ops.size(), "set_fp8_scales: expected %d entries", e ? (int)e->ops.size() : 0);
    AQ_REQUIRE(e->desc.precision == AQ_BF16, "set_fp8_scales: bf16 engines only");
    for (int i = 0; i < n_ops; ++i) {                     // back to bf16 first (a second call replaces the first)
        PackedW& pw = e->packed[i];
        if (pw.w_f8) { (void)hipFree(pw.w_f8); pw.w_f8 = nullptr; }
        if (pw.sb_f8) { (void)hipFree(pw.sb_f8); pw.sb_f8 = nullptr; }
        pw.f8_consumer = -1;
        pw.act_scale = 0.0f;
    }
    for (int i = 0; i < n_ops; ++i) {
        if (!(act_scale[i] > 0.0f)) continue;
        const aq_op_desc& op = e->ops[i];
        PackedW& pw = e->packed[i];
        AQ_REQUIRE(op.kind == AQ_OP_CONV && op.k == 3 && op.stride == 1 && op.pad == 1 && pw.direct_cfg == AQ_CONV_CFG_PL3X3 && !pw.w_host.empty() &&
                       aq_conv3x3_pl_f8_supported(op.src.channels, op.dst.channels, 1, 8, 8), "set_fp8_scales: op %d has no fp8 form", i);
        int prod = -1, readers = 0;
        for (int j = 0; j < n_ops; ++j) {
            const aq_op_desc& o = e->ops[j];
            if (o.src.tensor == op.src.tensor && o.src.ch_off < op.src.ch_off + op.src.channels && op.src.ch_off < o.src.ch_off + o.src.channels) ++readers;
            if (o.res.tensor == op.src.tensor && o.res.tensor >= 0) ++readers;
            if (j < i && o.dst.tensor == op.src.tensor && o.dst.ch_off == op.src.ch_off && o.dst.channels == op.src.channels) prod = j;
        }
        AQ_REQUIRE(prod >= 0 && e->ops[prod].kind == AQ_OP_CONV && e->ops[prod].k == 1 && e->packed[prod].direct_cfg == AQ_CONV_CFG_DIRECT1X1 &&
                       e->ops[prod].res.tensor < 0 && e->ops[prod].src.channels == e->ops[prod].dst.channels &&
                       (e->ops[prod].dst.channels == 192 || e->ops[prod].dst.channels == 384),
                   "set_fp8_scales: the input of op %d is not written by a 1x1 layer with a code-writing form", i);
        // the temporary is shared by the Bottlenecks of a C3 stage: every reader of that slice must be an fp8 consumer as well
        for (int j = 0; j < n_ops; ++j) {
            const aq_op_desc& o = e->ops[j];
            const bool reads = o.src.tensor == op.src.tensor && o.src.ch_off == op.src.ch_off && o.src.channels == op.src.channels;
            AQ_REQUIRE(!reads || act_scale[j] > 0.0f, "set_fp8_scales: op %d reads the same tensor as op %d but stays bf16", j, i);
        }
        (void)readers;
        size_t nb = 0;
        AQ_REQUIRE(aq_pack_conv3x3_pl_f8(pw.w_host.data(), pw.b_host.data(), op.src.channels, op.dst.channels, act_scale[i], nullptr, &nb, nullptr, nullptr) == AQ_OK,
                   "set_fp8_scales: op %d cannot be packed", i);
        AQ_CHECK_HIP(hipMalloc(&pw.w_f8, nb));
        AQ_CHECK_HIP(hipMalloc((void**)&pw.sb_f8, 2048 * sizeof(float)));
        const int rc = aq_pack_conv3x3_pl_f8(pw.w_host.data(), pw.b_host.data(), op.src.channels, op.dst.channels, act_scale[i], pw.w_f8, &nb, pw.sb_f8, nullptr);
        if (rc) return rc;
        pw.act_scale = act_scale[i];
        e->packed[prod].f8_consumer = i;
        e->packed[prod].act_scale = act_scale[i];
    }
    return AQ_OK;
}

extern "C" int aq_engine_tensor_ptr(aq_engine* e, int tensor, void** ptr, int* channels, int* h, int* w, int* elem_bytes) {
    AQ_REQUIRE(e && ptr && tensor >= 0 && tensor < (int)e->tensors.size(), "tensor_ptr: bad tensor id %d", tensor);
    AQ_REQUIRE(e->last_ws, "tensor_ptr: no call has run yet");
    *ptr = tptr(e, e->last_ws, e->last_tiles, tensor);
    if (channels) *channels = e->tensors[tensor].channels;
    if (h) *h = e->place[tensor].h;
    if (w) *w = e->place[tensor].w;
    if (elem_bytes) *elem_bytes = e->place[tensor].elem;
    return AQ_OK;
}

extern "C" int aq_engine_num_ops(aq_engine* e) { return e ? (int)e->ops.size() : 0; }

extern "C" int aq_engine_set_nms_options(aq_engine* e, int agnostic, unsigned long long classes_lo, unsigned long long classes_hi) {
    AQ_REQUIRE(e, "set_nms_options: null engine");
    e->nms_agnostic = agnostic != 0; e->nms_cls_lo = classes_lo; e->nms_cls_hi = classes_hi;
    return AQ_OK;
}

extern "C" int aq_engine_set_conv_config(aq_engine* e, int op, int cfg) {
    AQ_REQUIRE(e && op >= 0 && op < (int)e->ops.size() && e->ops[op].kind == AQ_OP_CONV, "set_conv_config: op %d is not a conv", op);
    AQ_REQUIRE((cfg >= -1 && (cfg < 0 || (cfg & ~AQ_CONV_CFG_ONE_TILE_PER_WG) < aq_conv_num_configs())) ||
                   (cfg >= AQ_CONV_CFG_DIRECT1X1 && cfg == e->packed[op].direct_cfg),
               "set_conv_config: bad config %d for op %d", cfg, op);
    e->conv_cfg[op] = cfg;
    return AQ_OK;
}

// Times every tile configuration of every conv op on the caller's buffers (after one real forward pass so
// the activations are real data) and keeps the fastest per op for this (B, H, W).  Not a launch-path call:
// it synchronises the stream.
extern "C" int aq_engine_autotune(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W, void* ws, size_t ws_bytes,
                                  int reps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AQ_REQUIRE(e && tiles_dev && ws && reps > 0, "autotune: bad argument");
    const size_t npred = 0;
    (void)npred;
    int rc = layout(e, B, H, W);
    if (rc) return rc;
    if (ws_bytes < e->total_bytes) { aq_set_error("autotune: workspace too small"); return AQ_ERR_WORKSPACE; }
    std::vector<int> saved = e->tuned_cfg;
    e->tuned_B = 0;
    rc = run_plan(e, tiles_dev, B, H, W, ws, ws_bytes, (float*)((char*)ws + e->off_pred), nullptr, nullptr, 0.f, 0.f, 1, stream);
    if (rc) return rc;
    hipEvent_t ev0, ev1;
    AQ_CHECK_HIP(hipEventCreate(&ev0));
    AQ_CHECK_HIP(hipEventCreate(&ev1));
    const int ncfg = aq_conv_num_configs();
    for (size_t oi = 0; oi < e->ops.size() && rc == AQ_OK; ++oi) {
        if (e->ops[oi].kind != AQ_OP_CONV) continue;
        float best = 1e30f;
        int best_cfg = -1;
        // candidates: every tile shape as a persistent grid and as one workgroup per tile, then the direct kernel where one applies
        for (int ci = 0; ci <= 2 * ncfg && rc == AQ_OK; ++ci) {
            const int c = ci < ncfg ? ci : (ci < 2 * ncfg ? ((ci - ncfg) | AQ_CONV_CFG_ONE_TILE_PER_WG) : e->packed[oi].direct_cfg);
            if (c < 0) continue;
            rc = run_conv(e, (int)oi, ws, tiles_dev, B, stream, c);   // warm-up (also sets the LDS attribute)
            if (rc == AQ_ERR_INVALID) { rc = AQ_OK; continue; }       // this tile shape does not apply to this layer
            if (rc) break;
            (void)hipEventRecord(ev0, stream);
            for (int r = 0; r < reps && rc == AQ_OK; ++r) rc = run_conv(e, (int)oi, ws, tiles_dev, B, stream, c);
            (void)hipEventRecord(ev1, stream);
            if (hipEventSynchronize(ev1) != hipSuccess) { aq_set_error("autotune: sync failed"); rc = AQ_ERR_HIP; break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev0, ev1);
            if (ms < best) { best = ms; best_cfg = c; }
        }
        saved[oi] = best_cfg;
    }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    if (rc) return rc;
    e->tuned_cfg = saved;
    e->tuned_B = B; e->tuned_H = H; e->tuned_W = W;
    return AQ_OK;
}

extern "C" int aq_engine_set_tuned_table(aq_engine* e, int B, int H, int W, const int* cfgs, int n_ops) {
    AQ_REQUIRE(e && cfgs && n_ops == (int)e->ops.size() && B > 0 && H > 0 && W > 0, "set_tuned_table: expected %d entries", e ? (int)e->ops.size() : 0);
    for (int i = 0; i < n_ops; ++i) {
        const int c = cfgs[i];
        if (e->ops[i].kind != AQ_OP_CONV) continue;
        AQ_REQUIRE(c == -1 || (c >= 0 && (c & ~AQ_CONV_CFG_ONE_TILE_PER_WG) < aq_conv_num_configs()) ||
                       (c >= AQ_CONV_CFG_DIRECT1X1 && c < AQ_CONV_CFG_ONE_TILE_PER_WG && c == e->packed[i].direct_cfg),
                   "set_tuned_table: bad config %d for op %d", c, i);
    }
    int rc = layout(e, B, H, W);      // (H, W) as the engine lays them out: the table is matched against those
    if (rc) return rc;
    e->tuned_cfg.assign(cfgs, cfgs + n_ops);
    for (int i = 0; i < n_ops; ++i)
        if (e->ops[i].kind != AQ_OP_CONV) e->tuned_cfg[i] = -1;
    e->tuned_B = B; e->tuned_H = e->lay_H; e->tuned_W = e->lay_W;
    return AQ_OK;
}

// Which kernel family op `op` went to in its most recent launch (AQ_FAM_*), and with which tile-configuration id (-1 where none applies).
// Tests use it to assert that a layer did not silently fall back (an fp8 pair to bf16 at another batch size, a planar layer to the
// implicit-GEMM kernel).  AQ_FAM_NONE before the op's first launch.
extern "C" int aq_engine_last_launch(aq_engine* e, int op, int* family, int* cfg) {
    AQ_REQUIRE(e && family && op >= 0 && op < (int)e->ops.size(), "last_launch: bad argument");
    const bool have = e->last_family.size() == e->ops.size();
    *family = have ? e->last_family[op] : AQ_FAM_NONE;
    if (cfg) *cfg = have ? e->last_cfg[op] : -1;
    return AQ_OK;
}

extern "C" int aq_engine_get_conv_config(aq_engine* e, int op) {
    if (!e || op < 0 || op >= (int)e->ops.size()) return -1;
    if (e->conv_cfg[op] >= 0) return e->conv_cfg[op];
    return e->tuned_B ? e->tuned_cfg[op] : -1;
}

extern "C" int aq_engine_profile(aq_engine* e, int enable, int ring) {
    AQ_REQUIRE(e, "profile: null engine");
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    e->ev.clear();
    e->prof = false; e->ring = 0; e->prof_calls = 0;
    if (!enable) return AQ_OK;
    AQ_REQUIRE(ring > 0 && ring <= 4096, "profile: ring must be in [1, 4096]");
    const size_t n = (size_t)ring * (e->ops.size() + 1);
    e->ev.resize(n);
    for (size_t i = 0; i < n; ++i) AQ_CHECK_HIP(hipEventCreate(&e->ev[i]));
    e->prof = true; e->ring = ring;
    return AQ_OK;
}

extern "C" int aq_engine_op_times(aq_engine* e, float* ms_out, int n_ops, int* calls_recorded) {
    AQ_REQUIRE(e && ms_out && n_ops == (int)e->ops.size(), "op_times: expected %d slots", e ? (int)e->ops.size() : 0);
    AQ_REQUIRE(e->prof, "op_times: profiling is off");
    const int calls = (int)(e->prof_calls < e->ring ? e->prof_calls : e->ring);
    for (int i = 0; i < n_ops; ++i) ms_out[i] = 0.f;
    for (int c = 0; c < calls; ++c) {
        hipEvent_t* ev = e->ev.data() + (size_t)c * (n_ops + 1);
        AQ_CHECK_HIP(hipEventSynchronize(ev[n_ops]));
        for (int i = 0; i < n_ops; ++i) {
            float ms = 0.f;
            AQ_CHECK_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            ms_out[i] += ms;
        }
    }
    if (calls > 0) for (int i = 0; i < n_ops; ++i) ms_out[i] /= calls;
    if (calls_recorded) *calls_recorded = calls;
    return AQ_OK;
}

// Host helper of the label writer, a whole batch per call (round 4): tile t's rows are rows[offsets[t] .. offsets[t + 1]) (cls xc yc w h conf,
// fp32, already in file order); every tile with at least one row gets <dir>/<stem>.txt with exactly the bytes aq_format_label_rows produces,
// truncating what was there (a tile processed again after a crash leaves the same bytes); tiles without rows get no file (the consumer,
// reference src/process_yolo/geocode_results.py:123, relies on it).  One call per batch from a writer thread: the interpreter lock is
// released for the formatting (~2,000 %g conversions per tile with the synthetic checkpoint) AND the file system calls -- with a Python
// `open / write / close` per tile the four writer threads serialised on the lock at ~5 k tiles/s.  do_fsync: fsync every file before
// closing it (the fallback where syncfs is unavailable).  Returns the number of files written, or -1 - t when tile t could not be written.
extern "C" long aq_format_label_rows(const float* rows, int n, int save_conf, char* buf, size_t buflen);
extern "C" long aq_write_label_files(const char* dir, const char* const* stems, const float* rows, const long long* offsets, int n_tiles,
                                     int save_conf, int do_fsync) {
    if (!dir || !stems || !rows || !offsets || n_tiles < 0) return -1;
    std::vector<char> buf(1 << 16);
    std::string path;
    long written = 0;
    for (int t = 0; t < n_tiles; ++t) {
        const long long n = offsets[t + 1] - offsets[t];
        if (n <= 0) continue;
        long k = aq_format_label_rows(rows + (size_t)offsets[t] * 6, (int)n, save_conf, buf.data(), buf.size());
        if (k < 0) {
            buf.resize((size_t)(-k) + 64);
            k = aq_format_label_rows(rows + (size_t)offsets[t] * 6, (int)n, save_conf, buf.data(), buf.size());
        }
        path.assign(dir);
        path += '/';
        path += stems[t];
        path += ".txt";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return -1 - t;
        const bool ok = fwrite(buf.data(), 1, (size_t)k, f) == (size_t)k && fflush(f) == 0 && (!do_fsync || fsync(fileno(f)) == 0);
        if (fclose(f) != 0 || !ok) return -1 - t;
        ++written;
    }
    return written;
}

// Host helper of the label writer: n rows (cls xc yc w h conf, fp32) -> text exactly as `detect.py --save-txt --save-conf`
// prints them [UPSTREAM detect.py: ('%g ' * len(line)).rstrip() % line + '\n'], consumed by reference
// src/process_yolo/geocode_results.py:140.  Returns the number of bytes written, or -(bytes needed) if buf is too small.
extern "C" long aq_format_label_rows(const float* rows, int n, int save_conf, char* buf, size_t buflen) {
    if (!rows || !buf || n < 0) return 0;
    const int cols = save_conf ? 6 : 5;
    size_t off = 0;
    bool overflow = false;
    char tmp[32];
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < cols; ++c) {
            const int len = snprintf(tmp, sizeof(tmp), "%g", (double)rows[(size_t)i * 6 + c]);
            if (!overflow && off + (size_t)len + 1 <= buflen) {
                memcpy(buf + off, tmp, (size_t)len);
                buf[off + len] = (c + 1 == cols) ? '\n' : ' ';
            } else {
                overflow = true;
            }
            off += (size_t)len + 1;
        }
    }
    return overflow ? -(long)off : (long)off;
}
