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
