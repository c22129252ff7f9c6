// Host side of the wide 1x1 convolution in generated gfx950 assembly (gen_conv1x1_asm.py -- read its header for the kernel): weight packing,
// argument block, launch.  y = SiLU(W x + b), bf16, Cin a multiple of 96, Cout in {384, 768, 1536}: yolov5m's K >= 768 1x1 layers
// ([UPSTREAM models/common.py C3.cv1 / cv2 / cv3, SPPF.cv1 / cv2; models/yolov5m.yaml model.8-10, 13, 23]).  The engine's autotuner times it per
// layer against the implicit-GEMM tile shapes under the config id AQ_CONV_CFG_ASM1X1.
#include "conv_device.h"

using namespace aqdev;

namespace {

constexpr int kNT = 384, kKC = 96;                      // output channels per tile, input channels per chunk (gen_conv1x1_asm.py NT, 32 KS)
constexpr int kTPX[2] = {208, 112};                     // pixels per tile of the two families (nb13, nb7)
constexpr int kStepB = 3 * 1024;                        // weight bytes per (wave, k-step): three 1 KB A fragments
constexpr int kMaxCout = 1536;

struct C1AsmArgs {                 // must match ARG in gen_conv1x1_asm.py
    const char* in; char* out; const char* w; const float* bias;
    int in_ld_b, out_ld_b, npix, nchunks, ntiles, nct_log2, G;
    unsigned in_bytes, out_bytes, w_bytes, stream_b;
    int cout;
    unsigned long long* debug;
    unsigned long long pad;
};
static_assert(sizeof(C1AsmArgs) == 96, "kernel argument block");

const unsigned char kC1AsmCode[] = {
#include "conv1x1_asm_hsaco.inc"
};
hipModule_t g_c1a_mod[64];
hipFunction_t g_c1a_fn[64][2][2];            // [family nb13 / nb7][plain, stamped]
int g_c1a_cus = 0;

int c1a_load(int dev) {
    if (g_c1a_mod[dev]) return AQ_OK;
    hipModule_t mod = nullptr;
    AQ_CHECK_HIP(hipModuleLoadData(&mod, kC1AsmCode));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_c1a_fn[dev][0][0], mod, "conv1x1_asm_nb13"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_c1a_fn[dev][0][1], mod, "conv1x1_asm_nb13_stamped"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_c1a_fn[dev][1][0], mod, "conv1x1_asm_nb7"));
    AQ_CHECK_HIP(hipModuleGetFunction(&g_c1a_fn[dev][1][1], mod, "conv1x1_asm_nb7_stamped"));
    g_c1a_mod[dev] = mod;
    return AQ_OK;
}

// Output channel of row r = 4 g + e of M block m of wave w in channel tile ct: lane group g's twelve outputs of a pixel are 8 consecutive
// channels (blocks 0 and 1) and 4 consecutive ones (block 2) of the wave's 48.
inline int c1a_channel(int ct, int w, int m, int r) {
    const int g = r >> 2, e = r & 3;
    return ct * kNT + 48 * w + (m < 2 ? 8 * g + 4 * m + e : 32 + 4 * g + e);
}

size_t c1a_weight_bytes(int cin, int cout) {
    // [channel tile][wave][k-step][M block][lane] x 8 bf16, + two k-steps of zeros: the weight loads run two k-steps ahead of the last tile's end
    return (size_t)(cout / kNT) * 8 * (cin / 32) * kStepB + 2 * kStepB;
}

}  // namespace

extern "C" int aq_conv1x1_asm_supported(int cin, int cout) {
    static const bool off = [] { const char* e = getenv("AQ_C1_ASM"); return e && *e == '0'; }();
    return !off && cin >= kKC && cin % kKC == 0 && cin <= 3072 && (cout == 384 || cout == 768 || cout == 1536);
}

// Packs fused fp32 weights KRSC (cout, 1, 1, cin) into the A-fragment streams the kernel's waves read: wave w of channel tile ct reads
// [k-step][M block m][lane] x 16 bytes front to back; lane (r = lane & 15, kg = lane >> 4) of block m holds input channels
// 32 kstep + 8 kg .. + 7 of output channel c1a_channel(ct, w, m, r).
extern "C" int aq_pack_conv1x1_asm(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream) {
    AQ_REQUIRE(w_host && bytes && aq_conv1x1_asm_supported(cin, cout), "pack_conv1x1_asm: unsupported %d -> %d", cin, cout);
    *bytes = c1a_weight_bytes(cin, cout);
    if (!packed_dev) return AQ_OK;
    bf16_t* host = (bf16_t*)calloc(1, *bytes);
    AQ_REQUIRE(host, "pack_conv1x1_asm: out of host memory");
    const int ks = cin / 32;
    for (int ct = 0; ct < cout / kNT; ++ct)
        for (int w = 0; w < 8; ++w)
            for (int s = 0; s < ks; ++s)
                for (int m = 0; m < 3; ++m)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = c1a_channel(ct, w, m, lane & 15), kg = lane >> 4;
                        bf16_t* dst = host + ((((size_t)(ct * 8 + w) * ks + s) * 3 + m) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) dst[e] = aq_f2bf(w_host[(size_t)co * cin + 32 * s + 8 * kg + e]);
                    }
    const hipError_t err = hipMemcpyAsync(packed_dev, host, *bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err == hipSuccess) (void)hipStreamSynchronize((hipStream_t)stream);
    free(host);
    AQ_CHECK_HIP(err);
    return AQ_OK;
}

// in_dev / out_dev: bf16 pixel-major tensors [npix][in_ld] / [npix][out_ld]; the layer reads channels in_choff .. + cin and writes
// out_choff .. + cout (channel slices of concat buffers).  act must be 1 (SiLU).
extern "C" int aq_conv1x1_asm(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int cin, int cout,
                              const void* packed_w_dev, const float* bias_dev, long long npix, int act, void* stream) {
    AQ_REQUIRE(in_dev && out_dev && packed_w_dev && bias_dev, "conv1x1_asm: null pointer");
    AQ_REQUIRE(aq_conv1x1_asm_supported(cin, cout), "conv1x1_asm: unsupported %d -> %d", cin, cout);
    AQ_REQUIRE(act == 1, "conv1x1_asm: SiLU only");
    AQ_REQUIRE(in_ld % 8 == 0 && out_ld % 4 == 0 && in_choff % 8 == 0 && out_choff % 4 == 0 && in_choff + cin <= in_ld && out_choff + cout <= out_ld,
               "conv1x1_asm: channel slices must be aligned and inside their rows");
    // 32-bit buffer offsets; "no tile left" fetches from in_bytes + 0x100 + lane parts, which must not wrap
    AQ_REQUIRE(npix > 0 && npix * (long long)in_ld * 2 < (1LL << 31) - (1LL << 22) && npix * (long long)out_ld * 2 < (1LL << 32) - (1LL << 22),
               "conv1x1_asm: tensor too large for 32-bit offsets");
    int dev = 0;
    AQ_CHECK_HIP(hipGetDevice(&dev));
    AQ_REQUIRE(dev >= 0 && dev < 64, "conv1x1_asm: device ordinal %d", dev);
    { const int rc = c1a_load(dev); if (rc) return rc; }
    if (g_c1a_cus == 0) {
        int cus = 256;
        AQ_CHECK_HIP(aq_query_cus(&cus, dev));
        g_c1a_cus = cus;
    }
    C1AsmArgs a{};
    a.in = (const char*)in_dev + (size_t)in_choff * 2;
    a.out = (char*)out_dev + (size_t)out_choff * 2;
    a.w = (const char*)packed_w_dev;
    a.bias = bias_dev;
    a.in_ld_b = in_ld * 2; a.out_ld_b = out_ld * 2;
    a.npix = (int)npix;
    a.nchunks = cin / kKC;
    const int nct = cout / kNT;
    a.nct_log2 = nct == 1 ? 0 : nct == 2 ? 1 : 2;
    // Family: 208-pixel tiles unless their grid leaves the CUs half idle and the 112-pixel tiles (twice the weight bytes per MFMA: about 15 % more
    // time per pixel) finish in fewer tile-heights: 768 -> 384 at 20x20, batch 64 = 124 tiles of 208 pixels, 229 of 112.  AQ_C1_ASM_NB=13 / 7 forces one.
    int fam = 0;
    {
        const long long t13 = (npix + kTPX[0] - 1) / kTPX[0] * nct, t7 = (npix + kTPX[1] - 1) / kTPX[1] * nct;
        const double c13 = (double)((t13 + g_c1a_cus - 1) / g_c1a_cus) * 13.0, c7 = (double)((t7 + g_c1a_cus - 1) / g_c1a_cus) * 7.0 * 1.15;
        if (c7 < c13) fam = 1;
        const char* fe = getenv("AQ_C1_ASM_NB");          // (read per call: tests and tools/time_conv1x1_asm.py switch it)
        const int forced = fe ? atoi(fe) : 0;
        if (forced == 13) fam = 0;
        if (forced == 7) fam = 1;
    }
    const long long ptiles = (npix + kTPX[fam] - 1) / kTPX[fam];
    a.ntiles = (int)(ptiles * nct);
    long long grid = g_c1a_cus;
    if (grid > a.ntiles) grid = a.ntiles;
    a.G = (int)grid;
    a.in_bytes = (unsigned)((npix - 1) * a.in_ld_b + (long long)cin * 2);
    a.out_bytes = (unsigned)((npix - 1) * a.out_ld_b + (long long)cout * 2);
    a.w_bytes = (unsigned)c1a_weight_bytes(cin, cout);
    a.stream_b = (unsigned)((cin / 32) * kStepB);
    a.cout = cout;
    int which = 0;
    size_t sbytes = 0;
    unsigned long long* sbuf = aq_stamp_buffer(&sbytes);
    if (sbuf && (size_t)grid * 8 * 64 <= sbytes) { a.debug = sbuf; which = 1; }
    hipFunction_t fn = g_c1a_fn[dev][fam][which];
    const char* exp_kernel = getenv("AQ_C1_ASM_KERNEL");       // timing experiments: another kernel of the code object, by name (tools/time_conv1x1.py)
    if (exp_kernel && *exp_kernel) {
        char name[96];
        snprintf(name, sizeof name, "%s%s", exp_kernel, which ? "_stamped" : "");
        AQ_CHECK_HIP(hipModuleGetFunction(&fn, g_c1a_mod[dev], name));
    }
    size_t asz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    AQ_CHECK_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 512, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
    return AQ_OK;
}
