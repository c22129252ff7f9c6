// What does the vendor GEMM reach on the K >= 768 1x1 layers of yolov5m at batch 64?  (tools/ubench: a measurement, not product code)
//   D[N x px] (column-major, ldd = N) = SiLU(W[N x K] * X[K x px] + bias[N]):  W stored [N][K] (opA = T), X = the NHWC activation [px][K].
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hipblaslt_1x1 tools/ubench/hipblaslt_1x1.cpp -lhipblaslt
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

int main() {
    hipblasLtHandle_t h;
    CK(hipFree(0));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipblasLtCreate(&h));
    printf("hipblasLtCreate: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    struct Shape { const char* name; int N, K, px; } shapes[] = {
        {"model.8.cv1|cv2 / 8.cv3 / 23.*  768 -> 768 @ 20x20", 768, 768, 25600},
        {"model.9.cv1 / model.10          768 -> 384 @ 20x20", 384, 768, 25600},
        {"model.9.cv2                    1536 -> 768 @ 20x20", 768, 1536, 25600},
        {"model.13.cv1|cv2                768 -> 384 @ 40x40", 384, 768, 102400},
        {"model.6.cv1|cv2 / cv3 etc.      384 -> 384 @ 40x40", 384, 384, 102400},
        {"model.17.cv1|cv2                384 -> 192 @ 80x80", 192, 384, 409600},
    };
    const size_t ws_bytes = 0;
    void* ws = nullptr;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (auto& s : shapes) {
        const int R = 6;                       // rotate R buffer sets so that inputs are not L2 / MALL resident from the previous call
        size_t xb = (size_t)s.px * s.K * 2, db = (size_t)s.px * s.N * 2, wb = (size_t)s.N * s.K * 2;
        char *X, *D, *W; float* bias;
        CK(hipMalloc(&X, xb * R)); CK(hipMalloc(&D, db * R)); CK(hipMalloc(&W, wb)); CK(hipMalloc(&bias, s.N * 4));
        CK(hipMemset(X, 0x3c, xb * R)); CK(hipMemset(W, 0x3c, wb)); CK(hipMemset(bias, 0, s.N * 4));
        hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
        hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        int32_t bt = HIP_R_32F;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
        hipblasLtMatrixLayout_t la, lb, ld;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.K, s.N, s.K));      // stored K x N column-major = [N][K]
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.K, s.px, s.K));
        CK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_16BF, s.N, s.px, s.N));
        hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        const int NA = 16;
        float alpha0 = 1.f, beta0 = 0.f;
        hipblasLtMatmulHeuristicResult_t res[NA]; int got = 0;
        auto t1 = std::chrono::steady_clock::now();
        CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, NA, res, &got));
        printf("heuristic: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        t1 = std::chrono::steady_clock::now();
        if (got) { (void)hipblasLtMatmul(h, desc, &alpha0, W, la, X, lb, &beta0, D, ld, D, ld, &res[0].algo, ws, ws_bytes, st); CK(hipStreamSynchronize(st)); }
        printf("first matmul: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        printf("%s: %d algorithms\n", s.name, got);
        float alpha = 1.f, beta = 0.f;
        double best = 1e9; int besti = -1;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int a = 0; a < got; ++a) {
            bool ok = true;
            for (int it = 0; it < 3 && ok; ++it)
                ok = hipblasLtMatmul(h, desc, &alpha, W, la, X + xb * (it % R), lb, &beta, D + db * (it % R), ld, D + db * (it % R), ld, &res[a].algo, ws, ws_bytes, st) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) continue;
            CK(hipStreamSynchronize(st));
            const int IT = 24;
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < IT; ++it)
                (void)hipblasLtMatmul(h, desc, &alpha, W, la, X + xb * (it % R), lb, &beta, D + db * (it % R), ld, D + db * (it % R), ld, &res[a].algo, ws, ws_bytes, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double us = ms * 1e3 / IT;
            printf("   algo %2d: %.1f us (workspace %zu)\n", a, us, (size_t)res[a].workspaceSize);
            if (us < best) { best = us; besti = a; }
        }
        double fl = 2.0 * s.N * s.K * s.px;
        printf("   best algorithm %d: %.1f us = %.0f TFLOP/s (buffers rotated over %d sets)\n", besti, best, fl / best * 1e-6, R);
        fflush(stdout);
        CK(hipFree(X)); CK(hipFree(D)); CK(hipFree(W)); CK(hipFree(bias));
    }
    return 0;
}
