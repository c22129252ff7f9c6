/*
 * aq_engine.h -- C ABI of the MI355X-native YOLOv5 detect path (libaqengine.so).
 *
 * What it replaces.  The reference runs the path as a shell command,
 *     python3 yolov5/detect.py --weights W --source DIR --nosave --save-txt --save-conf
 * (reference README.md:77); there is no FFI in the reference for it, and the code it
 * runs is the un-vendored ultralytics/yolov5 submodule (/root/reference/yolov5/yolov5
 * is empty).  The two operator seams inside that script where a native engine plugs
 * in are [UPSTREAM detect.py run()]:
 *     S1  pred = model(im)                      float[B,3,H,W] -> float[B, N, 5+nc]
 *     S2  pred = non_max_suppression(pred, conf_thres, iou_thres, classes, agnostic, max_det)
 * aq_engine_infer() is S1+S2 fused (uint8 tiles in, per-tile detections out);
 * aq_engine_forward_raw() is S1 alone; aq_nms() is S2 alone; the remaining entry points
 * are the individual kernels, exported so each one can be parity-tested by itself.
 *
 * Conventions.
 *   - plain C, pointers and sizes only; every *_dev pointer is device (HBM) memory owned by
 *     the caller; nothing here allocates in a launch path (hipGraph-capture safe).
 *   - all work is enqueued on the hipStream_t passed in (as void*), no host sync.
 *   - return 0 (AQ_OK) or a negative aq_status; aq_last_error() gives a thread-local message.
 *   - activations are NHWC; a tensor argument is (base pointer, pixel stride in elements,
 *     first channel): channel slices of wider buffers are first-class (free Concat).
 */
#ifndef AQ_ENGINE_H
#define AQ_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum aq_status {
    AQ_OK = 0,
    AQ_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    AQ_ERR_HIP = -2,         /* a HIP runtime call failed */
    AQ_ERR_NOMEM = -3,
    AQ_ERR_WORKSPACE = -4    /* workspace too small */
} aq_status;

typedef enum aq_precision {
    AQ_BF16 = 0,   /* bf16 weights + activations, fp32 accumulate/epilogue/head (throughput mode) */
    AQ_FP32 = 1,   /* fp32 everywhere on f32-input MFMA (parity mode: detect.py without --half) */
    AQ_BF16_W8 = 2,/* AQ_BF16 whose conv weights lie on an OCP e4m3fn x 2^e[cout] grid (BASELINE.json configs[3], "fp8 weights"):
                    * kernels that have an fp8-weight stream (the planar 3x3) load the 1-byte codes, the others the same values as bf16 */
    AQ_F16X3 = 3   /* fp32 activations, epilogues and head as AQ_FP32; every conv product as THREE fp16 MFMAs on hi / lo halves of both
                    * operands (22 significant bits each, fp32 accumulate): fp32-grade results at several times the f32-input MFMA's
                    * rate -- the fast parity mode (north_star: boxes / conf within 1e-4 of detect.py's fp32 output).  Domain: activations
                    * within fp16's range (|x| < 65504), as upstream's own --half. */
} aq_precision;

typedef enum aq_op_kind {
    AQ_OP_PREPROCESS = 0, AQ_OP_CONV = 1, AQ_OP_SPPF_POOL = 2, AQ_OP_UPSAMPLE2X = 3,
    AQ_OP_DECODE = 4, AQ_OP_NMS = 5, AQ_OP_STEM = 6, AQ_OP_BOTTLENECK = 7, AQ_OP_DOWNBLOCK = 8
} aq_op_kind;

typedef enum aq_tensor_dtype { AQ_T_ACT = 0, AQ_T_F32 = 1, AQ_T_U8 = 2 } aq_tensor_dtype;

/* One HBM buffer of the plan: NHWC, `channels` contiguous, spatial = (H/down, W/down). */
typedef struct aq_tensor_desc {
    int32_t channels;
    int32_t down;
    int32_t dtype;            /* aq_tensor_dtype */
} aq_tensor_desc;

typedef struct aq_slice { int32_t tensor, ch_off, channels; } aq_slice;

/* One step of the plan (see aquaculture_amd/spec.py for how the YOLOv5 graph is flattened). */
typedef struct aq_op_desc {
    int32_t kind;             /* aq_op_kind */
    aq_slice src, dst, res;   /* res.tensor < 0: no residual */
    int32_t k, stride, pad;
    int32_t act;              /* 1 = SiLU, 0 = identity */
    int32_t level;            /* detect level of a head conv, else -1 */
    const float* weight;      /* host, fp32, KRSC [cout][k][k][cin], BN already folded */
    const float* bias;        /* host, fp32 [cout] */
    double flops_per_tile;    /* algorithmic FLOPs of this op for one tile (reporting only) */
} aq_op_desc;

typedef struct aq_model_desc {
    int32_t precision;        /* aq_precision */
    int32_t nc, na, nl;       /* classes, anchors per level, levels (3) */
    float anchors_px[3][8][2];/* anchor_grid = anchors * stride, pixels, [level][anchor][w,h] */
    float stride[3];
    int32_t head_tensor[3];   /* plan tensor ids of the raw fp32 head maps */
    int32_t input_tensor;     /* plan tensor id of the u8 tiles */
    int32_t n_tensors, n_ops;
    const aq_tensor_desc* tensors;
    const aq_op_desc* ops;
} aq_model_desc;

/* One output detection, in network-input pixels (before scale_boxes), as NMS returns it:
 * [UPSTREAM non_max_suppression] rows (x1, y1, x2, y2, conf, cls). */
typedef struct aq_det { float x1, y1, x2, y2, conf, cls; } aq_det;

typedef struct aq_engine aq_engine;

const char* aq_last_error(void);
int aq_version(void);

/* ---- engine (S1 + S2) ---------------------------------------------------------------- */
/* Copies + packs the weights to the device (the only allocation the engine ever does). */
int aq_engine_create(const aq_model_desc* desc, int device_ordinal, aq_engine** out);
void aq_engine_destroy(aq_engine* e);
/* Bytes of caller-provided workspace needed for batches up to max_batch of HxW tiles. */
int aq_engine_workspace_bytes(aq_engine* e, int max_batch, int H, int W, size_t* bytes);
/* tiles_dev: uint8 [B][H][W][3] RGB.  dets_dev: [B][max_det].  counts_dev: [B].
 * Replaces `pred = model(im); pred = non_max_suppression(pred, conf, iou, None, False, max_det)`. */
int aq_engine_infer(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W,
                    void* workspace_dev, size_t workspace_bytes,
                    aq_det* dets_dev, int32_t* counts_dev,
                    float conf_thres, float iou_thres, int max_det, void* stream);
/* S1 only: pred_dev float [B][N][5+nc] (xywh px, obj, cls) exactly as Detect.forward returns it. */
int aq_engine_forward_raw(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W,
                          void* workspace_dev, size_t workspace_bytes, float* pred_dev, void* stream);
/* Test hook: device address + geometry of plan tensor `tensor` inside the workspace of the last call. */
int aq_engine_tensor_ptr(aq_engine* e, int tensor, void** ptr, int* channels, int* h, int* w, int* elem_bytes);
/* Per-op device timing with HIP events on the launch stream (bench.py roofline).  ring = number of
 * infer calls whose events are kept; aq_engine_op_times returns mean ms per op over recorded calls. */
int aq_engine_profile(aq_engine* e, int enable, int ring);
int aq_engine_op_times(aq_engine* e, float* ms_out, int n_ops, int* calls_recorded);
int aq_engine_num_ops(aq_engine* e);
/* Tuning hook: force the tile configuration of one conv op (-1 = built-in heuristic). */
int aq_engine_set_conv_config(aq_engine* e, int op, int cfg);
/* Tuning: time every tile configuration of every conv op on real activations of a (B,H,W) batch and keep the
 * fastest per op for that geometry (cudnn.benchmark-style).  Synchronises the stream; call once before timing. */
int aq_engine_autotune(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W,
                       void* workspace_dev, size_t workspace_bytes, int reps, void* stream);
/* fp8 path of BASELINE.json configs[3] (a bf16 engine whose wide Bottleneck 3x3 layers run on the fp8 MFMA, both operands e4m3):
 * 1. aq_engine_calibrate_amax: one bf16 forward pass over calibration tiles; amax_host[i] = max |output| of conv op i (0 for other ops).
 * 2. aq_engine_set_fp8_scales: act_scale[i] > 0 switches conv op i (a 3x3 / stride-1 layer with a planar form, cin % 64 == 0) to
 *    aq_conv3x3_pl_f8 and the 1x1 layer that writes its input to the code-writing epilogue (aq_conv1x1_direct_f8out); act_scale[i] is the
 *    e4m3 scale of that input tensor (value = code x scale; typically amax of the producer / 448).  The pair falls back to bf16 for batch
 *    geometries the fp8 kernel cannot tile.  Weights are quantised per output channel (max |w| / 448) by the packer. */
int aq_engine_calibrate_amax(aq_engine* e, const uint8_t* tiles_dev, int B, int H, int W, void* workspace_dev, size_t workspace_bytes,
                             float* amax_host, int n_ops, void* stream);
int aq_engine_set_fp8_scales(aq_engine* e, const float* act_scale, int n_ops);
int aq_engine_get_conv_config(aq_engine* e, int op);
/* Which kernel family op `op` went to in its most recent launch, and with which tile-configuration id (-1 where none applies): how a
 * test asserts that a layer did not silently fall back -- an fp8 pair to bf16 at another batch size, a planar layer to the implicit-GEMM
 * kernel.  AQ_FAM_NONE before the op's first launch and for pointwise ops. */
enum {
    AQ_FAM_NONE = 0,
    AQ_FAM_IGEMM_OR_HALO = 1,      /* conv_igemm_kernel / conv3x3_halo_kernel; *cfg = tile-shape id (aq_conv_config_tiles) */
    AQ_FAM_PL3X3 = 2,              /* planar 3x3/s1 (generated assembly or the HIP-source build; aq_conv3x3_pl_asm_family names it) */
    AQ_FAM_PL3X3_W8 = 3,           /* the same with the e4m3 weight stream */
    AQ_FAM_PL3X3S2 = 4,            /* planar 3x3/s2 */
    AQ_FAM_PL3X3_F8 = 5,           /* fp8 x fp8 planar 3x3/s1 (consumer of an fp8 pair) */
    AQ_FAM_DIRECT1X1 = 6,
    AQ_FAM_DIRECT1X1_F8OUT = 7,    /* producer of an fp8 pair: writes e4m3 codes */
    AQ_FAM_DIRECT3X3S2 = 8,
    AQ_FAM_BOTTLENECK = 9,
    AQ_FAM_DOWNBLOCK = 10,
    AQ_FAM_STEM = 11,
    AQ_FAM_HEAD_DECODE = 12,
    AQ_FAM_ASM1X1 = 13             /* wide 1x1 in generated assembly (conv1x1_asm_nb13) */
};
int aq_engine_last_launch(aq_engine* e, int op, int* family, int* cfg);
/* Install a table that aq_engine_autotune produced earlier (or on another rank) for the SAME engine and (B,H,W): cfgs[n_ops], one id per
 * op as aq_engine_get_conv_config returns them (-1 for ops that are not tuned).  Used for every batch of that tile geometry (H, W),
 * whatever its size -- a ragged last batch runs the same kernels as the full ones --; other geometries keep the built-in heuristic.
 * Ids are validated per op. */
int aq_engine_set_tuned_table(aq_engine* e, int B, int H, int W, const int* cfgs, int n_ops);
int aq_conv_num_configs(void);
/* Diagnostics: arm (buf != NULL) or disarm a device buffer that the STAMPED builds of a few conv tile shapes fill with
 * per-wave phase cycle sums (8 x uint64 per wave); used by tools/stamp_conv.py only. */
int aq_debug_conv_stamp(void* buf_dev, size_t bytes);
/* Diagnostics: register-only bf16 MFMA loop (blocks x 4 waves x iters x 8 MFMAs) to read the sustained matrix rate. */
int aq_debug_mfma_peak(int blocks, int iters, void* out_dev, void* stream);
int aq_conv_config_tiles(int cfg, int* bm, int* bn);

/* ---- individual kernels --------------------------------------------------------------- */
/* Packs fp32 KRSC host weights into the device layout conv kernels read:
 * [cout_pad][k_pad] elements of `precision`, k = (ky, kx, cin) flattened, zero padded.  Returns the
 * byte count via *bytes when packed_dev == NULL. */
int aq_pack_conv_weights(const float* w_krsc_host, int cout, int k, int cin, int precision,
                         void* packed_dev, size_t* bytes, void* stream);
/* Implicit-GEMM convolution, NHWC, fused bias + SiLU + residual.  in/out/res element type = precision
 * (out is fp32 when out_f32 != 0).  Replaces Conv.forward_fuse / Bottleneck.forward of the reference's
 * yolov5 dependency [UPSTREAM models/common.py]. */
/* AQ_F16X3 packing: weights as fp16 hi / lo halves of w * 2^s[cout] (s: the power of two that puts the row's largest weight in
 * [2^14, 2^15)), per 8 channels 16 bytes of hi then 16 bytes of lo; bias_scale_dev receives float[2 * rows]: bias * 2^s, then 2^-s
 * (rows = *bias_floats / 2, the padded row count).  aq_conv2d with precision AQ_F16X3 takes THAT buffer as bias_dev.  cin % 8 == 0. */
int aq_pack_conv_weights_x3(const float* w_krsc_host, const float* bias_host, int cout, int k, int cin, void* packed_dev, size_t* bytes,
                            float* bias_scale_dev, size_t* bias_floats, void* stream);
int aq_conv2d(const void* in_dev, int in_ld, int in_choff, int cin,
              void* out_dev, int out_ld, int out_choff, int cout,
              const void* res_dev, int res_ld, int res_choff,
              const void* packed_w_dev, const float* bias_dev,
              int B, int H, int W, int k, int stride, int pad, int act,
              int precision, int out_f32, const void* zero_page_dev, void* stream);
/* Fused stem: uint8 RGB tiles [B][H][W][3] -> x/255 -> Conv(3, cout, k=6, s=2, p=2) + bias + SiLU -> NHWC [B][H/2][W/2][cout slice].
 * Replaces `im.float() / 255` + model.0 of the reference's yolov5 dependency [UPSTREAM detect.py, models/common.py Conv].
 * packed_w_dev comes from aq_pack_stem_weights (fp32 KRSC (cout, 6, 6, 3), BN folded); cout <= 64, H % 4 == 0, W even. */
int aq_pack_stem_weights(const float* w_krsc_host, int cout, int precision, void* packed_dev, size_t* bytes, void* stream);
int aq_stem_conv(const uint8_t* tiles_dev, void* out_dev, int out_ld, int out_choff, int cout, const void* packed_w_dev,
                 const float* bias_dev, int B, int H, int W, int act, int precision, void* stream);

/* Fused Bottleneck (bf16 only, C = 16, 32, 48, 64 or 96): y = (x +) SiLU(cv2_3x3(SiLU(cv1_1x1(x)))) in ONE launch, t never leaves the
 * chip.  Replaces [UPSTREAM models/common.py Bottleneck.forward] for the C3 stages whose hidden width fits
 * (yolov5m: model.2, model.4, model.17).
 * In the plan (op kind AQ_OP_BOTTLENECK): weight = cv1 KRSC [C][1][1][C] followed by cv2 KRSC [C][3][3][C]; bias = b1 | b2.
 * in/out: NHWC [B][H][W][ld] bf16 with the C channels at ch_off; out must not overlap in (neighbour tiles read halos). */
int aq_pack_bottleneck_weights(const float* w1_host, const float* w2_host, int C, void* packed_dev, size_t* bytes, void* stream);
int aq_bottleneck(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int C,
                  const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int shortcut, void* stream);

/* Fused down-sampling block (bf16 only, 48 -> 96 -> 96 channels): y = SiLU(Wb_1x1 . SiLU(Wa_3x3/s2 (*) x)) in ONE launch.
 * Replaces yolov5m's model.1 = Conv(48, 96, 3, 2) followed by the stacked model.2.cv1|cv2 1x1 convs
 * [UPSTREAM models/common.py Conv.forward_fuse, C3.forward].  In the plan (op kind AQ_OP_DOWNBLOCK): weight = wa KRSC
 * [96][3][3][48] followed by wb KRSC [96][1][1][96]; bias = ba | bb.  in: NHWC [B][H][W][ld]; out: [B][H/2][W/2][ld]. */
int aq_pack_downblock_weights(const float* wa_host, const float* wb_host, void* packed_dev, size_t* bytes, void* stream);
int aq_downblock(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff,
                 const void* packed_w_dev, const float* bias_dev, int B, int H, int W, void* stream);
/* The fused stem in front of it as well (yolov5m model.0 + model.1 + model.2.cv1|cv2 in ONE launch, bf16): the 17 x 33-pixel patch of
 * stem outputs a down-block tile needs is computed from the uint8 tile inside the kernel, so the stem's output tensor (629 MB per
 * 64-tile batch) is never written or read.  Takes exactly aq_stem_conv's and aq_downblock's weight images; bit-identical to the two
 * launches.  Hi, Wi: tile size (multiples of 4). */
int aq_stemdown_supported(int Hi, int Wi);
int aq_stemdown(const uint8_t* tiles_dev, void* out_dev, int out_ld, int out_choff, const void* stem_w_dev, const float* stem_bias_dev,
                const void* packed_w_dev, const float* bias_dev, int B, int Hi, int Wi, void* stream);

/* Direct 1x1 convolution (bf16; Cin -> Cout in {96->96, 192->192, 384->192, 384->384}): no K pipeline, whole-K pixel tiles by LDS-DMA, weights
 * in registers.  Same operation as aq_conv2d with k = 1; the engine's autotuner times it per layer against the implicit-GEMM tile
 * shapes under the config id AQ_CONV_CFG_DIRECT1X1. */
#define AQ_CONV_CFG_DIRECT1X1 1000
/* OR-ed into a tile configuration id (aq_conv2d, aq_engine_set_conv_config): launch one workgroup per output tile instead of the
 * persistent grid (one or two resident workgroups per CU striding over the tiles).  Tiles then reach CUs in the order CUs become
 * free; the autotuner times both forms per layer. */
#define AQ_CONV_CFG_ONE_TILE_PER_WG 4096
int aq_conv1x1_direct_supported(int cin, int cout);
int aq_pack_conv1x1_direct(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream);
int aq_conv1x1_direct(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int cin, int cout,
                      const void* packed_w_dev, const float* bias_dev, long long npix, int act, void* stream);

/* Wide 1x1 convolution in generated gfx950 assembly (bf16, SiLU; Cin a multiple of 96, Cout in {384, 768, 1536}: yolov5m's K >= 768 1x1 layers --
 * C3 cv1|cv2 / cv3 at 20x20, SPPF cv1 / cv2, model.10, model.13 cv1|cv2).  208-pixel x 384-channel tiles, eight waves, weights streamed from L2
 * to registers, pixels by buffer-descriptor LDS-DMA in 96-channel chunks (csrc/gen_conv1x1_asm.py).  Same operation as aq_conv2d with k = 1,
 * act = 1; autotuner candidate AQ_CONV_CFG_ASM1X1.  npix: pixels (B x H x W); in / out may be channel slices of wider tensors. */
#define AQ_CONV_CFG_ASM1X1 1004
int aq_conv1x1_asm_supported(int cin, int cout);
int aq_pack_conv1x1_asm(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream);
int aq_conv1x1_asm(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int cin, int cout,
                   const void* packed_w_dev, const float* bias_dev, long long npix, int act, void* stream);

/* Direct 3x3 / stride 2 / pad 1 convolution (bf16; 96 -> 192 channels: yolov5m's model.3), the plain form of the down-block kernel:
 * input patches by LDS-DMA, weights in registers (last k-steps in LDS), each input-row fragment loaded once for the output rows it
 * serves.  Autotuner candidate AQ_CONV_CFG_DIRECT3X3S2. */
#define AQ_CONV_CFG_DIRECT3X3S2 1001
int aq_conv3x3s2_direct_supported(int cin, int cout);
int aq_pack_conv3x3s2_direct(const float* w_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream);
int aq_conv3x3s2_direct(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff, int cin, int cout,
                        const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int act, void* stream);
/* 3x3 / stride 1 / pad 1 convolution of the wide Bottleneck layers (bf16; Cin a multiple of 64 and >= 128, Cout a multiple of 192:
 * yolov5m's 192 -> 192 and 384 -> 384 [UPSTREAM models/common.py Bottleneck.cv2]): four waves, one per SIMD, weights streamed from L2
 * straight into registers in MFMA-fragment order (aq_pack_conv3x3_pl), the input region staged in LDS as padded, slot-major planes
 * (conflict-free fragment reads for every tap with no masks), a three-buffer chunk ring with ONE workgroup barrier per 64-channel chunk.
 * (Round 3: on 20- and 40-pixel-wide images -- yolov5m's two widths -- the region is staged as 64-byte pixel-major rows through a buffer
 * descriptor instead, a quarter of the cache-line look-ups per LDS-DMA instruction; same tiles, packed weights and results.)
 * Same operation as aq_conv2d with k = 3, stride = 1, pad = 1; the engine's autotuner times it per layer under AQ_CONV_CFG_PL3X3.
 * in_dev: first channel of the input; element (pixel P, 16-byte channel group g) lives at in_dev + P * in_pixel_stride_b +
 * g * in_group_stride_b (NHWC: row bytes and 16).  out / res: NHWC bf16 slices as in aq_conv2d (res may alias out: in-place shortcut). */
#define AQ_CONV_CFG_PL3X3 1002
int aq_conv3x3_pl_supported(int cin, int cout);
/* 1 when the assembly family with `nb` 16-pixel blocks per tile is part of this build (13: always; 7, 8: two workgroups per CU, an
 * experiment that wins on no BASELINE geometry, built only with AQ_GEN_EXPERIMENTAL=1), else 0; negative aq_status on a HIP error. */
int aq_conv3x3_pl_asm_family(int nb);
int aq_pack_conv3x3_pl(const float* w_krsc_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream);
int aq_conv3x3_pl(const void* in_dev, long long in_pixel_stride_b, long long in_group_stride_b, int cin,
                  void* out_dev, int out_ld, int out_choff, int cout, const void* res_dev, int res_ld, int res_choff,
                  const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int act, void* stream);
/* 3x3 / stride 2 / pad 1 on the same planar scheme (round 3; bf16, Cin a multiple of 32 and >= 64, Cout a multiple of 192 with Cout / 192 a
 * power of two, even H and W, and an output row short enough for the tile's region: yolov5m's model.5 / 7 / 18 / 21, i.e.
 * [UPSTREAM models/common.py Conv.forward_fuse] with k = 3, s = 2): the input region is staged in LDS as the four PARITY planes of the
 * input (y, x odd / even), each in padded OUTPUT coordinates, so a tap is a plane plus an offset of -1 or 0 and every fragment read
 * stays conflict-free; 32-channel chunks, two ring buffers, one barrier per chunk, weights L2 -> registers.  Generated gfx950 assembly
 * (csrc/gen_conv3x3_pl_asm.py, family s2nb13).  Same operation as aq_conv2d with k = 3, stride = 2, pad = 1; autotuner candidate
 * AQ_CONV_CFG_PL3X3S2.  H, W: the INPUT's size. */
#define AQ_CONV_CFG_PL3X3S2 1003
int aq_conv3x3_pl_s2_supported(int cin, int cout, int B, int H, int W);
int aq_pack_conv3x3_pl_s2(const float* w_krsc_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream);
int aq_conv3x3_pl_s2(const void* in_dev, int in_ld, int in_choff, int cin, void* out_dev, int out_ld, int out_choff, int cout,
                     const void* packed_w_dev, const float* bias_dev, int B, int H, int W, int act, void* stream);
/* fp8 on BOTH MFMA operands (round 3; BASELINE.json configs[3]): the stride-1 planar kernel on v_mfma_f32_16x16x128_f8f6f4 -- input tensor =
 * OCP e4m3fn codes (NHWC, one byte per channel, value = code x act_scale, written by the producing kernel), weights quantised by the
 * packer with one scale per output channel (max |w| / 448), fp32 accumulate, epilogue x act_scale x w_scale[co] + bias, SiLU, shortcut,
 * bf16 out.  cin a multiple of 64, cout of 192 (cout / 192 a power of two).  in_ld / in_choff in BYTES (= channels). */
int aq_conv1x1_direct_f8out(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_pitch_bytes, int out_byte_off, int cin, int cout,
                            const void* packed_w_dev, const float* bias_dev, long long npix, int act, float out_scale, void* stream);
int aq_absmax_bf16(const void* t_dev, int ld, int choff, int c, long long npix, float* out_dev, void* stream);
unsigned char aq_f32_to_e4m3(float v);      /* round to nearest even, saturating at +-448 */
int aq_conv3x3_pl_f8_supported(int cin, int cout, int B, int H, int W);
int aq_pack_conv3x3_pl_f8(const float* w_krsc_host, const float* bias_host, int cin, int cout, float act_scale, void* packed_dev,
                          size_t* bytes, float* scale_bias_dev, void* stream);
int aq_conv3x3_pl_f8(const void* in_dev, int in_ld, int in_choff, int cin, void* out_dev, int out_ld, int out_choff, int cout,
                     const void* res_dev, int res_ld, int res_choff, const void* packed_w_dev, const float* scale_bias_dev,
                     int B, int H, int W, int act, void* stream);
/* The same kernel streaming e4m3fn weight codes (AQ_BF16_W8): bit-identical outputs to aq_conv3x3_pl on the dequantised weights.
 * `w` must lie on a per-output-channel grid code x 2^e (AQ_ERR_INVALID otherwise); scale_bias_dev: float[2048] written by the packer
 * (bias x 2^-e, then 2^e).  _supported: NB = 13 tiles fit the image and cout / 192 is a power of two. */
int aq_conv3x3_pl_w8_supported(int cin, int cout, int B, int H, int W);
int aq_pack_conv3x3_pl_w8(const float* w_krsc_host, const float* bias_host, int cin, int cout, void* packed_dev, size_t* bytes,
                          float* scale_bias_dev, void* stream);
int aq_conv3x3_pl_w8(const void* in_dev, long long in_sp, long long in_ss, int cin, void* out_dev, int out_ld, int out_choff,
                     int cout, const void* res_dev, int res_ld, int res_choff, const void* packed_w8_dev,
                     const float* scale_bias_dev, int B, int H, int W, int act, void* stream);
/* uint8 RGB NHWC -> 2x2 space-to-depth, 16 channels, value/255 ([UPSTREAM detect.py: im.float()/255]). */
int aq_preprocess_s2d(const uint8_t* tiles_dev, void* out_dev, int B, int H, int W, int precision, void* stream);
/* Letterbox on device (the real 1024x1024 tiles of reference src/load_data/tile_tifs.py:13 -> 640x640):
 * [UPSTREAM utils/augmentations.py letterbox] = cv2.resize(INTER_LINEAR) to (new_w, new_h) + constant border 114, uint8 RGB NHWC.
 * xtab_dev / ytab_dev: int32 [new_w][4] / [new_h][4] rows (i0, i1, w0, w1): OpenCV's 11-bit fixed-point coefficient tables, built
 * on the host (aquaculture_amd/dataloader.py resize tables).  UNPINNED: no OpenCV is available to check against. */
int aq_letterbox_u8(const uint8_t* src_dev, int B, int H0, int W0, uint8_t* dst_dev, int H, int W, int new_w, int new_h,
                    int top, int left, const int32_t* xtab_dev, const int32_t* ytab_dev, void* stream);
/* The same letterbox reading its B tiles (H0 x W0) out of ONE uint8 RGB raster in device memory: tile b starts tile_off[b] bytes into
 * the raster, rows are row_bytes apart.  Replaces the crop + re-encode of reference src/load_data/tile_tifs.py:13-47 (gdal.Translate
 * srcWin per tile) + :50-74 (JPEG) for the opt-in scene mode: the scene is uploaded once and never written back as tiles.  The offsets
 * are passed twice: tile_off_dev for the kernel, tile_off_host so that the call can refuse tiles that leave the raster. */
int aq_letterbox_tiles_u8(const uint8_t* scene_dev, long long scene_bytes, long long row_bytes, const long long* tile_off_dev,
                          const long long* tile_off_host, int B, int H0, int W0, uint8_t* dst_dev, int H, int W, int new_w, int new_h,
                          int top, int left, const int32_t* xtab_dev, const int32_t* ytab_dev, void* stream);
/* Split JPEG decode (round 3): the pixel half of what [UPSTREAM detect.py LoadImages -> cv2.imread] gets from
 * libjpeg(-turbo) -- dequantisation, the islow IDCT, h2v2 fancy chroma upsampling, YCbCr -> RGB -- bit for bit, on the device.  Input: the
 * quantised coefficient blocks of B baseline 4:2:0 JPEGs of one size as written by aq_jpeg_decode_coeffs (libaqjpeg.so, include/aq_jpeg.h:
 * the entropy decoder, which stays on the host), image b from coef_off_dev[b] (int16 units, a multiple of 64), qt_dev uint16 [B][3][64].
 * Output: uint8 RGB [B][H][W][3].  scratch_dev: aq_jpeg_scratch_bytes(B, H, W) bytes. */
size_t aq_jpeg_scratch_bytes(int B, int H, int W);
int aq_jpeg_idct_rgb(const int16_t* coef_dev, const long long* coef_off_dev, const uint16_t* qt_dev, int B, int H, int W,
                     void* scratch_dev, uint8_t* out_dev, void* stream);
/* GPU entropy decode (round 4): the Huffman stage of the same decode on the device, one lane per restart segment (= per image for files
 * without restart markers).  Replaces the host half's aq_jpeg_decode_coeffs ([UPSTREAM LoadImages -> cv2.imread]'s entropy stage;
 * producer of the files: reference src/load_data/tile_tifs.py:66-74).  streams_dev: the bytes aq_jpeg_prepare (include/aq_jpeg.h) wrote,
 * uploaded as they are; segs_dev: nseg descriptors of 32 bytes {u32 stream_off, u32 stream_len, u64 coef_off (int16 index of the image's
 * first coefficient in coef_dev), u32 mcu0, u32 n_mcu, u16 mcu_cols, u16 mcu_rows, u32 tabset}; tabsets_dev: table sets of six
 * aq_jpeg_gpu_tab each; coef_dev: ZEROED coefficient buffers in aq_jpeg_decode_coeffs's layout; status_dev: int32 per segment (0 ok, 2 corrupt).
 * The kernel reads the stream in 64-byte chunks from each segment's start and runs up to two chunks ahead: streams_dev needs 256 readable
 * bytes behind the last segment.  The caller validates every offset; the kernel trusts them. */
int aq_jpeg_huffman_decode(const void* streams_dev, const void* segs_dev, int nseg, const void* tabsets_dev, void* coef_dev,
                           void* status_dev, void* stream);
/* SPPF pools: y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2) written to channel slices c, 2c, 3c of the same buffer. */
int aq_sppf_pool(void* buf_dev, int ld, int ch_off, int c, int B, int H, int W, int precision, void* stream);
/* nearest 2x upsample of a channel slice into a channel slice. */
int aq_upsample2x(const void* in_dev, int in_ld, int in_choff, void* out_dev, int out_ld, int out_choff,
                  int c, int B, int H, int W, int precision, void* stream);
/* One Detect level fused with its decode (bf16, small heads: na * (nc + 5) <= 32, cin a multiple of 32 up to 1024): the 1x1 head conv,
 * sigmoid(objectness) > conf_thres, box decode and the append to the image's compact candidate list, with the arithmetic of
 * aq_detect_decode; the fp32 head maps are never written.  The caller zeroes the counters before the first level.
 * [UPSTREAM models/yolo.py Detect.forward + utils/general.py non_max_suppression's candidate filter] */
int aq_head_decode_supported(int cin, int na, int nc);
int aq_pack_head_weights(const float* w_host, const float* bias_host, int cin, int cout, void* packed_dev, size_t* bytes, void* stream);
int aq_head_decode(const void* in_dev, int in_ld, int in_choff, int cin, const void* packed_dev, int B, int ny, int nx,
                   int cand_off, float stride, const float* anchors_px, int nc, int na, float conf_thres,
                   int32_t* cand_dev, float* cand_rows_dev, int32_t* cand_count_dev, int count_stride, int cand_cap, void* stream);
/* image b's candidate counter lives at cand_count_dev[b * count_stride] (the engine spaces them 4 KB apart: adjacent counters share an L2
 * channel and its atomic unit); this copies them into the compact [B] array aq_nms reads. */
int aq_head_counts_gather(const int32_t* wide_dev, int count_stride, int32_t* compact_dev, int B, void* stream);

/* Detect.forward inference branch on raw fp32 head maps [B][ny][nx][head_ld] (channel = a*no + o):
 * writes pred [B][N][no] when pred_dev != NULL; when cand_dev != NULL also compacts the candidates with
 * obj > conf_thres: their candidate indices into cand_dev[B][cand_cap], their decoded rows into
 * cand_rows_dev[B][cand_cap][no] (optional) and their number into cand_count_dev[B]. */
int aq_detect_decode(const float* const head_dev[3], int head_ld, int B, int H, int W, int nc, int na,
                     const float* anchors_px /* [3][na][2] host */, const float* stride /* [3] host */,
                     float* pred_dev, float conf_thres, int32_t* cand_dev, float* cand_rows_dev,
                     int32_t* cand_count_dev, int cand_cap, void* stream);
/* S2: non_max_suppression(pred, conf, iou, classes=None, agnostic=False, multi_label=False, max_det).
 * rows_dev is either the full pred [B][N][no] (rows_per_tile = N, cand_dev = NULL) or the compact rows
 * aq_detect_decode wrote (rows_per_tile = cand_cap, with cand_dev / cand_count_dev).
 * scratch_dev: aq_nms_scratch_bytes(B, N).  Deterministic: ties in confidence by ascending candidate index;
 * no wall-clock time limit.  N < 131072. */
size_t aq_nms_scratch_bytes(int B, int N);
int aq_nms(const float* rows_dev, int rows_per_tile, int B, int N, int nc, float conf_thres, float iou_thres,
           int max_det, const int32_t* cand_dev, const int32_t* cand_count_dev, int cand_cap,
           void* scratch_dev, aq_det* dets_dev, int32_t* counts_dev, void* stream);
/* aq_nms with upstream's two other options: agnostic != 0 = class-agnostic suppression (no per-class box offset); classes_lo / classes_hi = bit
 * mask of the classes kept (bit c of lo, bit c - 64 of hi; all ones = no filter), applied after the confidence threshold as upstream's
 * `x[(x[:, 5:6] == classes).any(1)]` [UPSTREAM utils/general.py non_max_suppression(classes, agnostic)]. */
int aq_nms_opts(const float* rows_dev, int rows_per_tile, int B, int N, int nc, float conf_thres, float iou_thres,
                int max_det, const int32_t* cand_dev, const int32_t* cand_count_dev, int cand_cap,
                void* scratch_dev, aq_det* dets_dev, int32_t* counts_dev, int agnostic, unsigned long long classes_lo,
                unsigned long long classes_hi, void* stream);
/* The engine's NMS step (aq_engine_infer) with those options; defaults: agnostic = 0, every class. */
int aq_engine_set_nms_options(aq_engine* e, int agnostic, unsigned long long classes_lo, unsigned long long classes_hi);

/* Host helper: n label rows (cls xc yc w h conf, fp32, stride 6) -> the text detect.py --save-txt [--save-conf] writes
 * ("%g" per value, one line per row).  Returns bytes written or -(bytes needed). */
long aq_format_label_rows(const float* rows, int n, int save_conf, char* buf, size_t buflen);
/* A batch of label files per call (no interpreter lock held): tile t's rows are rows[offsets[t] .. offsets[t + 1]); tiles with rows get
 * <dir>/<stems[t]>.txt with aq_format_label_rows's bytes (truncating), tiles without get no file.  Returns files written, or -1 - t. */
long aq_write_label_files(const char* dir, const char* const* stems, const float* rows, const long long* offsets, int n_tiles,
                          int save_conf, int do_fsync);

#ifdef __cplusplus
}
#endif
#endif /* AQ_ENGINE_H */
