/* Plain-C restatement of the arithmetic kernels of the YOLOv5 detect path.  TEST INFRASTRUCTURE ONLY.
 *
 * *** PARITY UNPINNED *** (see oracle/yolov5_oracle.py: the reference's implementation of this path lives in the
 * un-vendored ultralytics/yolov5 submodule, /root/reference/yolov5/yolov5 is empty, and the reference has no tests
 * or golden vectors for it; anchor = the call site reference README.md:77 and the consumer's grammar
 * reference src/process_yolo/geocode_results.py:140-172.)
 *
 * Purpose: an implementation of the same formulas that shares no code with torch.nn.functional, so that the two
 * restatements (this file and oracle/yolov5_oracle.py) can be checked against each other in tests/.
 * Every function names the upstream function it restates ([UPSTREAM] = ultralytics/yolov5 v6.x/v7.0, torchvision).
 *
 * Build: make -C oracle   ->  oracle/_build/libref_kernels.so   (never linked into the product library)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* [UPSTREAM models/common.py Conv.forward_fuse]: out = act(conv2d(x, w) + b), NHWC in/out, weights KRSC
 * (cout, kh, kw, cin); act = SiLU (x * sigmoid(x)) when act != 0.  fp32 accumulate in (ky, kx, ci) order. */
void ref_conv2d_nhwc(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, int cin,
                     int cout, int k, int stride, int pad, int act) {
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x)
                for (int co = 0; co < cout; ++co) {
                    float acc = 0.0f;
                    for (int ky = 0; ky < k; ++ky) {
                        const int iy = y * stride - pad + ky;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < k; ++kx) {
                            const int ix = x * stride - pad + kx;
                            if (ix < 0 || ix >= W) continue;
                            const float* xp = in + (((size_t)b * H + iy) * W + ix) * cin;
                            const float* wp = w + (((size_t)co * k + ky) * k + kx) * cin;
                            for (int ci = 0; ci < cin; ++ci) acc += xp[ci] * wp[ci];
                        }
                    }
                    acc += bias[co];
                    if (act) acc = acc / (1.0f + expf(-acc));
                    out[(((size_t)b * Ho + y) * Wo + x) * cout + co] = acc;
                }
}

/* [UPSTREAM nn.MaxPool2d(5, 1, 2)] NHWC, implicit -inf padding */
void ref_maxpool5_nhwc(const float* in, float* out, int B, int H, int W, int C) {
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < C; ++c) {
                    float m = -INFINITY;
                    for (int yy = y - 2; yy <= y + 2; ++yy)
                        for (int xx = x - 2; xx <= x + 2; ++xx)
                            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                                const float v = in[(((size_t)b * H + yy) * W + xx) * C + c];
                                if (v > m) m = v;
                            }
                    out[(((size_t)b * H + y) * W + x) * C + c] = m;
                }
}

/* [UPSTREAM models/yolo.py Detect.forward inference branch] for one level: head NHWC (B, ny, nx, na*no) raw ->
 * rows (B, na*ny*nx, no) with candidate index a*ny*nx + y*nx + x. */
void ref_detect_decode(const float* head, float* pred, int B, int ny, int nx, int na, int no, float stride,
                       const float* anchor_px /* [na][2] */) {
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < na; ++a)
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) {
                    const float* src = head + (((size_t)b * ny + y) * nx + x) * (na * no) + a * no;
                    float* dst = pred + (((size_t)b * na + a) * ny * nx + (size_t)y * nx + x) * no;
                    float s[64];
                    for (int o = 0; o < no; ++o) s[o] = 1.0f / (1.0f + expf(-src[o]));
                    dst[0] = (s[0] * 2.0f + ((float)x - 0.5f)) * stride;
                    dst[1] = (s[1] * 2.0f + ((float)y - 0.5f)) * stride;
                    const float tw = s[2] * 2.0f, th = s[3] * 2.0f;
                    dst[2] = (tw * tw) * anchor_px[a * 2];
                    dst[3] = (th * th) * anchor_px[a * 2 + 1];
                    for (int o = 4; o < no; ++o) dst[o] = s[o];
                }
}

typedef struct { float conf; int idx; } ref_key;
static int ref_key_cmp(const void* pa, const void* pb) {
    const ref_key* a = (const ref_key*)pa; const ref_key* b = (const ref_key*)pb;
    if (a->conf > b->conf) return -1;
    if (a->conf < b->conf) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);   /* ties: ascending candidate index */
}

/* [UPSTREAM utils/general.py non_max_suppression(conf, iou, classes=None, agnostic=False, multi_label=False,
 * max_det)] + torchvision nms_kernel_impl for ONE image.  pred (n, 5+nc); out (max_det, 6) rows
 * x1 y1 x2 y2 conf cls.  Returns the number of rows.  No time limit; max_nms = 30000; max_wh = 7680. */
int ref_nms(const float* pred, int n, int nc, float conf_thres, float iou_thres, int max_det, float* out) {
    const int no = 5 + nc;
    ref_key* keys = (ref_key*)malloc(sizeof(ref_key) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float* r = pred + (size_t)i * no;
        if (!(r[4] > conf_thres)) continue;
        float best = r[5] * r[4];
        for (int c = 1; c < nc; ++c) { const float v = r[5 + c] * r[4]; if (v > best) best = v; }
        if (!(best > conf_thres)) continue;
        keys[m].conf = best; keys[m].idx = i; ++m;
    }
    qsort(keys, (size_t)m, sizeof(ref_key), ref_key_cmp);
    if (m > 30000) m = 30000;
    float* box = (float*)malloc(sizeof(float) * 4 * (size_t)(m > 0 ? m : 1));
    float* raw = (float*)malloc(sizeof(float) * 6 * (size_t)(m > 0 ? m : 1));
    unsigned char* sup = (unsigned char*)calloc((size_t)(m > 0 ? m : 1), 1);
    for (int t = 0; t < m; ++t) {
        const float* r = pred + (size_t)keys[t].idx * no;
        float best = r[5] * r[4]; int bj = 0;
        for (int c = 1; c < nc; ++c) { const float v = r[5 + c] * r[4]; if (v > best) { best = v; bj = c; } }
        const float hw = r[2] / 2.0f, hh = r[3] / 2.0f;
        raw[t * 6 + 0] = r[0] - hw; raw[t * 6 + 1] = r[1] - hh; raw[t * 6 + 2] = r[0] + hw; raw[t * 6 + 3] = r[1] + hh;
        raw[t * 6 + 4] = best; raw[t * 6 + 5] = (float)bj;
        const float c = (float)bj * 7680.0f;
        for (int e = 0; e < 4; ++e) box[t * 4 + e] = raw[t * 6 + e] + c;
    }
    int kept = 0;
    for (int i = 0; i < m && kept < max_det; ++i) {
        if (sup[i]) continue;
        memcpy(out + (size_t)kept * 6, raw + (size_t)i * 6, sizeof(float) * 6);
        ++kept;
        const float ix1 = box[i * 4], iy1 = box[i * 4 + 1], ix2 = box[i * 4 + 2], iy2 = box[i * 4 + 3];
        const float iarea = (ix2 - ix1) * (iy2 - iy1);
        for (int j = i + 1; j < m; ++j) {
            if (sup[j]) continue;
            const float xx1 = fmaxf(ix1, box[j * 4]), yy1 = fmaxf(iy1, box[j * 4 + 1]);
            const float xx2 = fminf(ix2, box[j * 4 + 2]), yy2 = fminf(iy2, box[j * 4 + 3]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float jarea = (box[j * 4 + 2] - box[j * 4]) * (box[j * 4 + 3] - box[j * 4 + 1]);
            const float ovr = inter / (iarea + jarea - inter);
            if (ovr > iou_thres) sup[j] = 1;
        }
    }
    free(keys); free(box); free(raw); free(sup);
    return kept;
}

/* [UPSTREAM detect.py --save-txt --save-conf]: one label line "cls xc yc w h conf", each value through %g
 * (the double that equals the fp32 value).  Returns the number of characters written (without the NUL). */
int ref_format_label(const float* row6, char* buf, int buflen) {
    return snprintf(buf, (size_t)buflen, "%g %g %g %g %g %g", (double)row6[0], (double)row6[1], (double)row6[2],
                    (double)row6[3], (double)row6[4], (double)row6[5]);
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Letterbox of the real 1024 x 1024 tiles (reference src/load_data/tile_tifs.py:13, src/utils.py:18-19) to the 640-px network input:
 * [UPSTREAM utils/augmentations.py letterbox] = cv2.resize(im, new_unpad, interpolation=cv2.INTER_LINEAR) + cv2.copyMakeBorder(114).
 * cv2 is not installed in this image, so what follows restates OpenCV's 8-bit bilinear resize from its published algorithm
 * (modules/imgproc/src/resize.cpp, resizeGeneric_ with HResizeLinear / VResizeLinear<uchar, int, short, FixedPtCast>):
 *   source coordinate  f = (d + 0.5) * (src / dst) - 0.5 computed in float, s = floor(f), f -= s;
 *                      s < 0 -> (s, f) = (0, 0);  s >= src - 1 -> (s, f) = (src - 1, 0)
 *   coefficients       short w1 = cvRound(f * 2048), w0 = cvRound((1 - f) * 2048)       (round half to even, fp32 products)
 *   horizontal pass    int   row[x] = S[s] * w0 + S[s + 1] * w1                          (scale 2^11)
 *   vertical pass      uchar dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
 * UNPINNED against OpenCV itself (none available); pinned against regressions by tests/golden/g9_letterbox.json. */
static void ref_axis_coeffs(int src, int dst, int* s0, int* s1, int* w0, int* w1) {
    const double scale = (double)src / (double)dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        s0[d] = s;
        s1[d] = s + 1 < src ? s + 1 : src - 1;
        w1[d] = (int)nearbyintf(f * 2048.f);
        w0[d] = (int)nearbyintf((1.f - f) * 2048.f);
    }
}

void ref_resize_linear_u8(const uint8_t* src, int h, int w, int c, uint8_t* dst, int nh, int nw) {
    int* xs0 = (int*)malloc(sizeof(int) * 4 * (size_t)nw);
    int* ys0 = (int*)malloc(sizeof(int) * 4 * (size_t)nh);
    int *xs1 = xs0 + nw, *xa0 = xs1 + nw, *xa1 = xa0 + nw;
    int *ys1 = ys0 + nh, *yb0 = ys1 + nh, *yb1 = yb0 + nh;
    ref_axis_coeffs(w, nw, xs0, xs1, xa0, xa1);
    ref_axis_coeffs(h, nh, ys0, ys1, yb0, yb1);
    for (int y = 0; y < nh; ++y) {
        const uint8_t* r0 = src + (size_t)ys0[y] * w * c;
        const uint8_t* r1 = src + (size_t)ys1[y] * w * c;
        for (int x = 0; x < nw; ++x)
            for (int k = 0; k < c; ++k) {
                const int h0 = r0[xs0[x] * c + k] * xa0[x] + r0[xs1[x] * c + k] * xa1[x];
                const int h1 = r1[xs0[x] * c + k] * xa0[x] + r1[xs1[x] * c + k] * xa1[x];
                int v = (((yb0[y] * (h0 >> 4)) >> 16) + ((yb1[y] * (h1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)y * nw + x) * c + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
    free(xs0);
    free(ys0);
}

/* [UPSTREAM letterbox] geometry: out = {new_w, new_h, top, bottom, left, right}.  Python's round() is half-to-even = nearbyint. */
void ref_letterbox_geometry(int h, int w, int new_h, int new_w, int auto_, int scaleup, int stride, int* out) {
    double r = fmin((double)new_h / h, (double)new_w / w);
    if (!scaleup) r = fmin(r, 1.0);
    const int uw = (int)nearbyint(w * r), uh = (int)nearbyint(h * r);
    double dw = new_w - uw, dh = new_h - uh;
    if (auto_) { dw = fmod(dw, stride); dh = fmod(dh, stride); }
    dw /= 2; dh /= 2;
    out[0] = uw; out[1] = uh;
    out[2] = (int)nearbyint(dh - 0.1); out[3] = (int)nearbyint(dh + 0.1);
    out[4] = (int)nearbyint(dw - 0.1); out[5] = (int)nearbyint(dw + 0.1);
}

/* letterbox(im, new_shape, auto, scaleup, stride): returns the output size through out_hw; dst may be NULL to query it. */
void ref_letterbox_u8(const uint8_t* src, int h, int w, int new_h, int new_w, int auto_, int scaleup, int stride, uint8_t* dst, int* out_hw) {
    int g[6];
    ref_letterbox_geometry(h, w, new_h, new_w, auto_, scaleup, stride, g);
    const int H = g[1] + g[2] + g[3], W = g[0] + g[4] + g[5];
    out_hw[0] = H; out_hw[1] = W;
    if (!dst) return;
    memset(dst, 114, (size_t)H * W * 3);
    uint8_t* tmp = (uint8_t*)malloc((size_t)g[0] * g[1] * 3);
    if (g[0] != w || g[1] != h) ref_resize_linear_u8(src, h, w, 3, tmp, g[1], g[0]);
    else memcpy(tmp, src, (size_t)h * w * 3);
    for (int y = 0; y < g[1]; ++y) memcpy(dst + ((size_t)(y + g[2]) * W + g[4]) * 3, tmp + (size_t)y * g[0] * 3, (size_t)g[0] * 3);
    free(tmp);
}
