// Device half of the split JPEG decode: quantised DCT coefficients -> RGB tiles in HBM.
//
// The host half (csrc/jpeg_coef.c, in the decode worker processes) undoes only the Huffman coding; what [UPSTREAM detect.py LoadImages ->
// cv2.imread] then gets from libjpeg(-turbo) is restated here bit for bit: dequantisation, jpeg_idct_islow (jidctint.c: 13-bit fixed
// point, columns then rows, DESCALE roundings, range limit), h2v2 "fancy" chroma upsampling (jdsample.c: 3/4 + 1/4 in each direction,
// the +8 / +7 rounding pair, the component's REAL edge replicated) and ycc_rgb_convert (jdcolor.c: 16-bit fixed-point tables).  All
// integer arithmetic, HBM-bound: 1.2 MB of coefficients in, 1.8 MB of planes out and back in, 1.2 MB of RGB out per 640-px tile --
// microseconds against the tile's 60 us in the network.  Oracle: oracle/jpeg_oracle.py, which tests/test_jpeg.py holds to Pillow's
// libjpeg-turbo byte for byte.
//
// Kernel 1: one thread per 8x8 block (128 contiguous bytes in, eight 8-byte rows out).  Kernel 2: one thread per 4 horizontal pixels.
#include "aq_common.h"

namespace {

constexpr int CB = 13, P1 = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
              F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
              F_2_562915447 = 20995, F_3_072711026 = 25172;

template <int SHIFT>
__device__ __forceinline__ void idct8(const int (&c)[8], int (&o)[8]) {
    // jidctint.c, one dimension; the products fit 32 bits for 8-bit JPEG data exactly as in the library (INT32 arithmetic there too)
    int z2 = c[2], z3 = c[6];
    int z1 = (z2 + z3) * F_0_541196100;
    int tmp2 = z1 + z3 * (-F_1_847759065);
    int tmp3 = z1 + z2 * F_0_765366865;
    z2 = c[0]; z3 = c[4];
    int tmp0 = (z2 + z3) * (1 << CB);
    int tmp1 = (z2 - z3) * (1 << CB);
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = c[7]; tmp1 = c[5]; tmp2 = c[3]; tmp3 = c[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    o[0] = (tmp10 + tmp3 + R) >> SHIFT; o[7] = (tmp10 - tmp3 + R) >> SHIFT;
    o[1] = (tmp11 + tmp2 + R) >> SHIFT; o[6] = (tmp11 - tmp2 + R) >> SHIFT;
    o[2] = (tmp12 + tmp1 + R) >> SHIFT; o[5] = (tmp12 - tmp1 + R) >> SHIFT;
    o[3] = (tmp13 + tmp0 + R) >> SHIFT; o[4] = (tmp13 - tmp0 + R) >> SHIFT;
}

struct JpegParams {
    const short* coef;            // all images' blocks, back to back
    const long long* off;         // [B] first int16 of image b
    const unsigned short* qt;     // [B][3][64], natural order
    unsigned char* planes;        // scratch: per image Y [Hp][Wp], Cb [Hp/2][Wp/2], Cr [Hp/2][Wp/2]
    unsigned char* out;           // [B][H][W][3]
    int B, H, W, mcu_cols, mcu_rows;
};

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegParams p) {
    const int ny = 4 * p.mcu_cols * p.mcu_rows, nc = p.mcu_cols * p.mcu_rows, per = ny + 2 * nc;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)p.B * per) return;
    const int b = (int)(t / per), k = (int)(t - (long long)b * per);
    const int comp = k < ny ? 0 : (k < ny + nc ? 1 : 2);
    const short* src = p.coef + p.off[b] + (long long)k * 64;
    const unsigned short* q = p.qt + ((long long)b * 3 + comp) * 64;
    const int Wp = 16 * p.mcu_cols, Hp = 16 * p.mcu_rows;
    unsigned char* plane = p.planes + (long long)b * (Hp * Wp + 2 * (Hp / 2) * (Wp / 2));
    int bw, bi;
    int pitch;
    if (comp == 0) { bw = 2 * p.mcu_cols; bi = k; pitch = Wp; }
    else { bw = p.mcu_cols; bi = k - ny - (comp - 1) * nc; pitch = Wp / 2; plane += Hp * Wp + (comp - 1) * (Hp / 2) * (Wp / 2); }
    const int by = bi / bw, bx = bi - by * bw;
    int ws[8][8];                                            // [row][col] after pass 1
#pragma unroll
    for (int r = 0; r < 8; ++r) {                            // load + dequantise a row of coefficients (16 bytes)
        const uint4 cv = *(const uint4*)(src + r * 8);
        const uint4 qv = *(const uint4*)(q + r * 8);
        const unsigned cw[4] = {cv.x, cv.y, cv.z, cv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ws[r][2 * e] = (int)(short)(cw[e] & 0xffff) * (int)(qw[e] & 0xffff);
            ws[r][2 * e + 1] = (int)(short)(cw[e] >> 16) * (int)(qw[e] >> 16);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {                            // pass 1: columns
        int in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = ws[r][c];
        idct8<CB - P1>(in, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][c] = o[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {                            // pass 2: rows, range limit, 8 bytes out
        int o[8];
        idct8<CB + P1 + 3>(ws[r], o);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo |= (unsigned)min(max(o[e] + 128, 0), 255) << (8 * e);
            hi |= (unsigned)min(max(o[4 + e] + 128, 0), 255) << (8 * e);
        }
        *(uint2*)(plane + (long long)(8 * by + r) * pitch + 8 * bx) = make_uint2(lo, hi);
    }
}

__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const JpegParams p) {
    const int W4 = (p.W + 3) / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)p.B * p.H * W4) return;
    const int b = (int)(t / ((long long)p.H * W4));
    const int rem = (int)(t - (long long)b * p.H * W4);
    const int y = rem / W4, x0 = (rem - y * W4) * 4;
    const int Wp = 16 * p.mcu_cols, Hp = 16 * p.mcu_rows, Wc = Wp / 2;
    const unsigned char* Y = p.planes + (long long)b * (Hp * Wp + 2 * (Hp / 2) * Wc);
    const unsigned char* Cb = Y + Hp * Wp;
    const unsigned char* Cr = Cb + (Hp / 2) * Wc;
    const int ch = (p.H + 1) / 2, cw = (p.W + 1) / 2;        // the chroma components' real extent: ITS edge is replicated
    const int cy = y >> 1, near = min(max(cy + ((y & 1) ? 1 : -1), 0), ch - 1);
    unsigned char* out = p.out + ((long long)b * p.H + y) * p.W * 3;
    unsigned char px[12];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int x = min(x0 + e, p.W - 1);
        const int cx = x >> 1;
        const int side = min(max(cx + ((x & 1) ? 1 : -1), 0), cw - 1);
        int c2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned char* C = k ? Cr : Cb;
            const int s_this = 3 * C[cy * Wc + cx] + C[near * Wc + cx];
            const int s_side = 3 * C[cy * Wc + side] + C[near * Wc + side];
            // first / last column of the component: (4 thiscolsum + 8 | 7) >> 4 -- which is what side == cx gives here (3 s + s)
            c2[k] = (3 * s_this + s_side + ((x & 1) ? 7 : 8)) >> 4;
        }
        const int yy = Y[y * Wp + x], cb = c2[0] - 128, cr = c2[1] - 128;
        const int r = yy + ((91881 * cr + 32768) >> 16);
        const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        const int bl = yy + ((116130 * cb + 32768) >> 16);
        px[3 * e] = (unsigned char)min(max(r, 0), 255);
        px[3 * e + 1] = (unsigned char)min(max(g, 0), 255);
        px[3 * e + 2] = (unsigned char)min(max(bl, 0), 255);
    }
    if (x0 + 3 < p.W && ((p.W * 3) % 4) == 0) {               // 12 bytes = three aligned dwords (row pitch a multiple of 4)
        unsigned* o32 = (unsigned*)(out + x0 * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) o32[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
    } else {
        for (int e = 0; e < 4 && x0 + e < p.W; ++e)
            for (int k = 0; k < 3; ++k) out[(x0 + e) * 3 + k] = px[3 * e + k];
    }
}

}  // namespace

// Scratch bytes the call needs for the intermediate sample planes of B images of H x W pixels.
extern "C" size_t aq_jpeg_scratch_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t Hp = (size_t)(H + 15) / 16 * 16, Wp = (size_t)(W + 15) / 16 * 16;
    return (size_t)B * (Hp * Wp + 2 * (Hp / 2) * (Wp / 2));
}

// coef_dev: the aq_jpeg_decode_coeffs images of B baseline 4:2:0 JPEGs of the SAME size (int16, image b from coef_off_dev[b]);
// qt_dev: uint16 [B][3][64]; out_dev: uint8 RGB [B][H][W][3] -- the pixels libjpeg(-turbo) would have produced.
extern "C" int aq_jpeg_idct_rgb(const int16_t* coef_dev, const long long* coef_off_dev, const uint16_t* qt_dev, int B, int H, int W,
                                void* scratch_dev, uint8_t* out_dev, void* stream) {
    AQ_REQUIRE(coef_dev && coef_off_dev && qt_dev && scratch_dev && out_dev, "jpeg_idct_rgb: null pointer");
    AQ_REQUIRE(B > 0 && H > 0 && W > 0 && H <= 65535 && W <= 65535 && (long long)B * H * W < (1LL << 40), "jpeg_idct_rgb: bad shape %d x %d x %d", B, H, W);
    AQ_REQUIRE(((uintptr_t)coef_dev & 15) == 0 && ((uintptr_t)qt_dev & 15) == 0 && ((uintptr_t)scratch_dev & 7) == 0, "jpeg_idct_rgb: unaligned buffer");
    JpegParams p;
    p.coef = coef_dev; p.off = coef_off_dev; p.qt = qt_dev; p.planes = (unsigned char*)scratch_dev; p.out = out_dev;
    p.B = B; p.H = H; p.W = W; p.mcu_cols = (W + 15) / 16; p.mcu_rows = (H + 15) / 16;
    const long long nblk = (long long)B * 6 * p.mcu_cols * p.mcu_rows;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    const long long nthr = (long long)B * H * ((W + 3) / 4);
    hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    AQ_CHECK_HIP(hipGetLastError());
    return AQ_OK;
}
