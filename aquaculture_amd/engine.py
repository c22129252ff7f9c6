"""ctypes binding of libaqengine.so (include/aq_engine.h) + the Python-side engine object.

Mirrors the two operator seams of the reference's ``yolov5/detect.py`` (reference README.md:77)
[UPSTREAM detect.py run()]:

    pred = model(im)                                   -> Engine.forward_raw(tiles_u8)
    pred = non_max_suppression(pred, conf, iou, ...)   -> Engine.nms(pred, ...)
    both, fused, from uint8 tiles                      -> Engine.infer(tiles_u8, ...)

PyTorch is used for device memory and streams only.  There is NO fallback: if the HIP library
cannot be loaded, or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import spec as _spec
from .checkpoint import Checkpoint, pack_plan_weights

AQ_BF16, AQ_FP32, AQ_BF16_W8, AQ_F16X3 = 0, 1, 2, 3
PRECISIONS = {"bf16": AQ_BF16, "fp32": AQ_FP32, "fp8w": AQ_BF16, "f16x3": AQ_F16X3}
"""Compute precision of the single-op helpers.  fp8w = fp8 (OCP e4m3fn) weights with per-output-channel power-of-two scales, bf16
activations (quant.py): values bf16 holds exactly, so every bf16 kernel runs them as they are."""
ENGINE_PRECISIONS = {"bf16": AQ_BF16, "fp32": AQ_FP32, "fp8w": AQ_BF16_W8, "f16x3": AQ_F16X3, "fp8": AQ_BF16}
"""aq_model_desc.precision.  AQ_BF16_W8 computes as AQ_BF16; kernels with an fp8-weight stream (the planar 3x3) load the e4m3 codes."""
_DTYPE_CODE = {"act": 0, "f32": 1, "u8": 2}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libaqengine.so")


class aq_tensor_desc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("down", C.c_int32), ("dtype", C.c_int32)]


class aq_slice(C.Structure):
    _fields_ = [("tensor", C.c_int32), ("ch_off", C.c_int32), ("channels", C.c_int32)]


class aq_op_desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("src", aq_slice), ("dst", aq_slice), ("res", aq_slice),
                ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("act", C.c_int32),
                ("level", C.c_int32), ("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
                ("flops_per_tile", C.c_double)]


class aq_model_desc(C.Structure):
    _fields_ = [("precision", C.c_int32), ("nc", C.c_int32), ("na", C.c_int32), ("nl", C.c_int32),
                ("anchors_px", C.c_float * 2 * 8 * 3), ("stride", C.c_float * 3),
                ("head_tensor", C.c_int32 * 3), ("input_tensor", C.c_int32),
                ("n_tensors", C.c_int32), ("n_ops", C.c_int32),
                ("tensors", C.POINTER(aq_tensor_desc)), ("ops", C.POINTER(aq_op_desc))]


class aq_det(C.Structure):
    _fields_ = [("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float),
                ("conf", C.c_float), ("cls", C.c_float)]


# every symbol include/aq_engine.h declares (tests check the library exports all of them)
EXPORTS = (
    "aq_last_error", "aq_version", "aq_engine_create", "aq_engine_destroy", "aq_engine_workspace_bytes",
    "aq_engine_infer", "aq_engine_forward_raw", "aq_engine_tensor_ptr", "aq_engine_profile",
    "aq_engine_op_times", "aq_engine_num_ops", "aq_engine_set_conv_config", "aq_engine_autotune", "aq_engine_set_tuned_table",
    "aq_engine_get_conv_config", "aq_conv_num_configs", "aq_debug_conv_stamp", "aq_debug_mfma_peak",
    "aq_conv_config_tiles", "aq_pack_conv_weights", "aq_pack_conv_weights_x3", "aq_conv2d", "aq_pack_stem_weights", "aq_stem_conv", "aq_pack_bottleneck_weights", "aq_bottleneck", "aq_pack_downblock_weights", "aq_downblock", "aq_stemdown_supported", "aq_stemdown", "aq_conv1x1_direct_supported", "aq_pack_conv1x1_direct", "aq_conv1x1_direct", "aq_conv1x1_asm_supported", "aq_pack_conv1x1_asm", "aq_conv1x1_asm", "aq_nms_opts", "aq_engine_set_nms_options",
    "aq_conv3x3s2_direct_supported", "aq_pack_conv3x3s2_direct", "aq_conv3x3s2_direct",
    "aq_conv3x3_pl_supported", "aq_conv3x3_pl_asm_family", "aq_pack_conv3x3_pl", "aq_conv3x3_pl", "aq_conv3x3_pl_s2_supported", "aq_pack_conv3x3_pl_s2", "aq_conv3x3_pl_s2", "aq_jpeg_scratch_bytes", "aq_jpeg_idct_rgb", "aq_f32_to_e4m3", "aq_conv1x1_direct_f8out", "aq_absmax_bf16", "aq_engine_calibrate_amax", "aq_engine_set_fp8_scales", "aq_engine_last_launch", "aq_conv3x3_pl_f8_supported", "aq_pack_conv3x3_pl_f8", "aq_conv3x3_pl_f8", "aq_conv3x3_pl_w8_supported", "aq_pack_conv3x3_pl_w8", "aq_conv3x3_pl_w8", "aq_head_decode_supported", "aq_pack_head_weights", "aq_head_decode", "aq_head_counts_gather", "aq_preprocess_s2d", "aq_sppf_pool",
    "aq_upsample2x", "aq_letterbox_u8", "aq_letterbox_tiles_u8", "aq_format_label_rows", "aq_detect_decode", "aq_nms_scratch_bytes", "aq_nms", "aq_jpeg_huffman_decode", "aq_write_label_files",
)

_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen libaqengine.so and set argument types.  Raises if it is missing (no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if path == LIB_PATH and os.environ.get("AQ_ENGINE_LIB"):
        path = os.environ["AQ_ENGINE_LIB"]             # A/B runs of two builds of the library in one session (tools/build_variant.py)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python -m aquaculture_amd.build` "
                           "(there is no CPU/PyTorch fallback for the detect path)")
    lib = C.CDLL(path)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.aq_last_error.restype = C.c_char_p
    lib.aq_engine_create.argtypes = [C.POINTER(aq_model_desc), i32, C.POINTER(vp)]
    lib.aq_engine_destroy.argtypes = [vp]
    lib.aq_engine_destroy.restype = None
    lib.aq_engine_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(sz)]
    lib.aq_engine_infer.argtypes = [vp, vp, i32, i32, i32, vp, sz, vp, vp, f32, f32, i32, vp]
    lib.aq_engine_forward_raw.argtypes = [vp, vp, i32, i32, i32, vp, sz, vp, vp]
    lib.aq_engine_tensor_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.aq_engine_profile.argtypes = [vp, i32, i32]
    lib.aq_engine_op_times.argtypes = [vp, C.POINTER(f32), i32, C.POINTER(i32)]
    lib.aq_engine_num_ops.argtypes = [vp]
    lib.aq_engine_set_conv_config.argtypes = [vp, i32, i32]
    lib.aq_engine_autotune.argtypes = [vp, vp, i32, i32, i32, vp, sz, i32, vp]
    lib.aq_engine_get_conv_config.argtypes = [vp, i32]
    lib.aq_engine_last_launch.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.aq_jpeg_huffman_decode.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    lib.aq_write_label_files.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(f32), C.POINTER(C.c_longlong), i32, i32, i32]
    lib.aq_write_label_files.restype = C.c_long
    lib.aq_engine_set_tuned_table.argtypes = [vp, i32, i32, i32, C.POINTER(i32), i32]
    lib.aq_engine_calibrate_amax.argtypes = [vp, vp, i32, i32, i32, vp, sz, C.POINTER(f32), i32, vp]
    lib.aq_engine_set_fp8_scales.argtypes = [vp, C.POINTER(f32), i32]
    lib.aq_conv1x1_direct_f8out.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, vp, vp, C.c_longlong, i32, f32, vp]
    lib.aq_absmax_bf16.argtypes = [vp, i32, i32, i32, C.c_longlong, vp, vp]
    lib.aq_debug_conv_stamp.argtypes = [vp, sz]
    lib.aq_debug_mfma_peak.argtypes = [i32, i32, vp, vp]
    lib.aq_conv_config_tiles.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
    lib.aq_pack_conv_weights.argtypes = [C.POINTER(f32), i32, i32, i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_pack_conv_weights_x3.argtypes = [C.POINTER(f32), C.POINTER(f32), i32, i32, i32, vp, C.POINTER(sz), vp, C.POINTER(sz), vp]
    lib.aq_conv2d.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, vp, vp,
                              i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.aq_pack_stem_weights.argtypes = [C.POINTER(f32), i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_stem_conv.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.aq_pack_bottleneck_weights.argtypes = [C.POINTER(f32), C.POINTER(f32), i32, vp, C.POINTER(sz), vp]
    lib.aq_bottleneck.argtypes = [vp, i32, i32, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.aq_pack_downblock_weights.argtypes = [C.POINTER(f32), C.POINTER(f32), vp, C.POINTER(sz), vp]
    lib.aq_downblock.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, i32, vp]
    lib.aq_conv1x1_direct_supported.argtypes = [i32, i32]
    lib.aq_pack_conv1x1_direct.argtypes = [C.POINTER(f32), i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_conv1x1_direct.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, vp, vp, C.c_longlong, i32, vp]
    lib.aq_nms_opts.argtypes = [vp, i32, i32, i32, i32, f32, f32, i32, vp, vp, i32, vp, vp, vp, i32, C.c_ulonglong, C.c_ulonglong, vp]
    lib.aq_engine_set_nms_options.argtypes = [vp, i32, C.c_ulonglong, C.c_ulonglong]
    lib.aq_conv1x1_asm_supported.argtypes = [i32, i32]
    lib.aq_pack_conv1x1_asm.argtypes = [C.POINTER(f32), i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_conv1x1_asm.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, vp, vp, C.c_longlong, i32, vp]
    lib.aq_conv3x3s2_direct_supported.argtypes = [i32, i32]
    lib.aq_pack_conv3x3s2_direct.argtypes = [C.POINTER(f32), i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_conv3x3s2_direct.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.aq_conv3x3_pl_supported.argtypes = [i32, i32]
    lib.aq_conv3x3_pl_asm_family.argtypes = [i32]
    lib.aq_pack_conv3x3_pl.argtypes = [C.POINTER(f32), i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_conv3x3_pl.argtypes = [vp, C.c_longlong, C.c_longlong, i32, vp, i32, i32, i32, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.aq_conv3x3_pl_w8.argtypes = lib.aq_conv3x3_pl.argtypes
    lib.aq_pack_conv3x3_pl_w8.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), i32, i32, vp, C.POINTER(C.c_size_t), vp, vp]
    lib.aq_conv3x3_pl_w8_supported.argtypes = [i32] * 5
    lib.aq_jpeg_scratch_bytes.argtypes = [i32, i32, i32]
    lib.aq_jpeg_scratch_bytes.restype = sz
    lib.aq_jpeg_idct_rgb.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.aq_f32_to_e4m3.argtypes = [f32]
    lib.aq_f32_to_e4m3.restype = C.c_ubyte
    lib.aq_conv3x3_pl_f8_supported.argtypes = [i32] * 5
    lib.aq_pack_conv3x3_pl_f8.argtypes = [C.POINTER(f32), C.POINTER(f32), i32, i32, f32, vp, C.POINTER(sz), vp, vp]
    lib.aq_conv3x3_pl_f8.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.aq_conv3x3_pl_s2_supported.argtypes = [i32] * 5
    lib.aq_pack_conv3x3_pl_s2.argtypes = [C.POINTER(f32), i32, i32, vp, C.POINTER(sz), vp]
    lib.aq_conv3x3_pl_s2.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.aq_preprocess_s2d.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.aq_sppf_pool.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.aq_upsample2x.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.aq_letterbox_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.aq_letterbox_tiles_u8.argtypes = [vp, C.c_longlong, C.c_longlong, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.aq_format_label_rows.argtypes = [C.POINTER(f32), i32, i32, C.c_char_p, sz]
    lib.aq_format_label_rows.restype = C.c_long
    lib.aq_detect_decode.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32),
                                     vp, f32, vp, vp, vp, i32, vp]
    lib.aq_nms_scratch_bytes.argtypes = [i32, i32]
    lib.aq_nms_scratch_bytes.restype = sz
    lib.aq_nms.argtypes = [vp, i32, i32, i32, i32, f32, f32, i32, vp, vp, i32, vp, vp, vp, vp]
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libaqengine error {rc}: {load_library().aq_last_error().decode()}")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("aquaculture_amd needs a ROCm GPU (MI355X/gfx950); there is no CPU fallback")


def _act_dtype(precision: int) -> torch.dtype:
    return torch.float32 if precision in (AQ_FP32, AQ_F16X3) else torch.bfloat16


class Engine:
    """YOLOv5 tile engine on one GPU.  Owns the C engine (packed weights) and a workspace tensor."""

    def __init__(self, ck: Checkpoint, precision: str = "bf16", device: int = 0, fused_stem: bool = True,
                 fused_bottleneck: Optional[bool] = None, fp8_calibration=None):
        """``fused_bottleneck``: None = on for bf16 engines (the fused kernel is bf16 only), off for fp32 parity engines.
        ``precision="fp8"`` (BASELINE.json configs[3]): a bf16 engine whose wide Bottleneck 3x3 layers run on the fp8 MFMA with e4m3 on
        both operands; the per-tensor activation scales are calibrated on ``fp8_calibration`` (uint8 tiles [B,H,W,3], numpy or CUDA tensor;
        default: synthetic tiles 0-7 at 640 px) -- call ``calibrate_fp8`` again to re-calibrate on other tiles."""
        _require_gpu()
        self.lib = load_library()
        self.ck = ck
        self.precision = ENGINE_PRECISIONS[precision]
        self.precision_name = precision
        self.device = torch.device("cuda", device)
        if fused_bottleneck is None:
            fused_bottleneck = precision in ("bf16", "fp8w", "fp8")
        if fused_bottleneck and precision in ("fp32", "f16x3"):
            raise ValueError("the fused Bottleneck kernel is bf16 only")
        self.plan = _spec.build_plan(ck.variant, ck.nc, ck.na, fused_stem=fused_stem, fused_bottleneck=fused_bottleneck)
        self.no = ck.nc + 5
        packed = pack_plan_weights(ck, self.plan, "fp8" if precision == "fp8w" else "native")
        self._keep = packed   # host arrays must outlive aq_engine_create only, kept for debugging
        tens = (aq_tensor_desc * len(self.plan.tensors))()
        for i, t in enumerate(self.plan.tensors):
            tens[i] = aq_tensor_desc(t.channels, t.down, _DTYPE_CODE[t.dtype])
        ops = (aq_op_desc * len(self.plan.ops))()
        self.plan.flops(640, 640)
        ci = 0
        for i, o in enumerate(self.plan.ops):
            d = aq_op_desc()
            d.kind = o.kind
            for name in ("src", "dst", "res"):
                s = getattr(o, name)
                setattr(d, name, aq_slice(s.tensor, s.ch_off, s.channels) if s is not None else aq_slice(-1, 0, 0))
            d.k, d.stride, d.pad, d.act, d.level = o.k, o.stride, o.pad, o.act, o.level
            d.flops_per_tile = o.flops_per_tile
            if o.kind in (_spec.OP_CONV, _spec.OP_STEM, _spec.OP_BOTTLENECK, _spec.OP_DOWNBLOCK):
                pw = packed[ci]
                ci += 1
                d.weight = pw.weight.ctypes.data_as(C.POINTER(C.c_float))
                d.bias = pw.bias.ctypes.data_as(C.POINTER(C.c_float))
            ops[i] = d
        desc = aq_model_desc()
        desc.precision, desc.nc, desc.na, desc.nl = self.precision, ck.nc, ck.na, 3
        ag = ck.anchor_grid_px().numpy()
        for l in range(3):
            desc.stride[l] = ck.stride[l]
            desc.head_tensor[l] = self.plan.head_tensors[l]
            for a in range(ck.na):
                desc.anchors_px[l][a][0] = float(ag[l, a, 0])
                desc.anchors_px[l][a][1] = float(ag[l, a, 1])
        desc.input_tensor = self.plan.input_tensor
        desc.n_tensors, desc.n_ops = len(self.plan.tensors), len(self.plan.ops)
        desc.tensors, desc.ops = tens, ops
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.aq_engine_create(C.byref(desc), device, C.byref(h)))
        self.handle = h
        self._ws: Optional[torch.Tensor] = None
        self._slots: Dict[int, torch.Tensor] = {}     # one workspace per in-flight batch (stream slot)
        self.fp8_scales: Dict[str, float] = {}        # fp8 engines: consumer op name -> e4m3 scale of its input tensor
        if precision == "fp8" and not (isinstance(fp8_calibration, str) and fp8_calibration == "defer"):
            # "defer": the caller calibrates on its own imagery (calibrate_fp8) or installs recorded scales (set_fp8_scales) before the first
            # batch -- yolov5/detect.py does, on tiles sampled from the sweep; until then every pair runs in bf16 (scale 0)
            if fp8_calibration is None:
                from . import tiles as _tiles
                fp8_calibration = _tiles.synthetic_batch(list(range(8)), 640)
            self.calibrate_fp8(fp8_calibration)

    def fp8_pairs(self) -> List[Tuple[int, int]]:
        """(producer op, consumer op) pairs the fp8 path covers: a Bottleneck's cv1 (1x1, direct kernel) feeding its cv2 (3x3 / stride 1,
        planar kernel) with 192 or 384 channels -- yolov5m: the 14 wide Bottleneck layers."""
        pairs = []
        ops = self.plan.ops
        for i, o in enumerate(ops):
            if (i and o.kind == _spec.OP_CONV and o.k == 3 and o.stride == 1 and o.name.endswith(".cv2") and ops[i - 1].name == o.name[:-1] + "1"
                    and ops[i - 1].kind == _spec.OP_CONV and ops[i - 1].k == 1 and o.src.channels == o.dst.channels and o.src.channels in (192, 384)
                    and (ops[i - 1].dst.tensor, ops[i - 1].dst.ch_off) == (o.src.tensor, o.src.ch_off)
                    and self.lib.aq_conv3x3_pl_f8_supported(o.src.channels, o.dst.channels, 1, 8, 8)):
                pairs.append((i - 1, i))
        return pairs

    def calibrate_fp8(self, tiles, margin: float = 1.0) -> Dict[str, float]:
        """One bf16 pass over ``tiles`` (uint8 [B,H,W,3]) records max |t| of every Bottleneck cv1 output; scale = margin x max / 448 (e4m3's
        largest value; the producer saturates beyond it).  Installs the scales (aq_engine_set_fp8_scales) and returns them by consumer name."""
        if not isinstance(tiles, torch.Tensor):
            tiles = torch.from_numpy(np.ascontiguousarray(tiles))
        tiles = tiles.to(self.device).contiguous()
        B, H, W = self._check_tiles(tiles)
        n = len(self.plan.ops)
        zero = (C.c_float * n)()
        _check(self.lib.aq_engine_set_fp8_scales(self.handle, zero, n))         # calibrate on the bf16 path
        ws = self.workspace(B, H, W)
        amax = (C.c_float * n)()
        _check(self.lib.aq_engine_calibrate_amax(self.handle, tiles.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), amax, n, _stream_ptr()))
        scales = (C.c_float * n)()
        self.fp8_scales = {}
        for prod, cons in self.fp8_pairs():
            a = float(amax[prod])
            if a > 0.0 and np.isfinite(a):
                scales[cons] = margin * a / 448.0
                self.fp8_scales[self.plan.ops[cons].name] = float(scales[cons])
        _check(self.lib.aq_engine_set_fp8_scales(self.handle, scales, n))
        return dict(self.fp8_scales)

    def set_fp8_scales(self, by_name: Dict[str, float]) -> None:
        """Installs e4m3 activation scales recorded elsewhere (consumer op name -> scale, as calibrate_fp8 returns them): the other ranks of a
        sweep take rank 0's, a resumed sweep takes the interrupted run's, so that every tile is quantised with the same scales."""
        n = len(self.plan.ops)
        scales = (C.c_float * n)()
        names = {self.plan.ops[cons].name: cons for _, cons in self.fp8_pairs()}
        unknown = sorted(set(by_name) - set(names))
        if unknown:
            raise ValueError(f"fp8 scales for ops this engine does not run in fp8: {unknown}")
        for name, v in by_name.items():
            if not (np.isfinite(v) and v > 0.0):
                raise ValueError(f"fp8 scale of {name} must be positive and finite, got {v}")
            scales[names[name]] = float(v)
        _check(self.lib.aq_engine_set_fp8_scales(self.handle, scales, n))
        self.fp8_scales = {k: float(v) for k, v in by_name.items()}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.aq_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- workspace ----
    def workspace(self, B: int, H: int, W: int, slot: int = 0) -> torch.Tensor:
        """Workspace of in-flight batch ``slot``: batches issued on different streams must not share one."""
        n = C.c_size_t()
        _check(self.lib.aq_engine_workspace_bytes(self.handle, B, H, W, C.byref(n)))
        ws = self._slots.get(slot)
        if ws is None or ws.numel() < n.value:
            self._slots.pop(slot, None)
            ws = self._slots[slot] = torch.empty(n.value, dtype=torch.uint8, device=self.device)
        self._ws = ws
        return ws

    def num_candidates(self, H: int, W: int) -> int:
        return _spec.num_candidates(H, W, self.ck.na)

    @staticmethod
    def _check_tiles(tiles: torch.Tensor) -> Tuple[int, int, int]:
        if tiles.dtype != torch.uint8 or tiles.dim() != 4 or tiles.shape[3] != 3 or not tiles.is_cuda or not tiles.is_contiguous():
            raise ValueError("tiles must be a contiguous CUDA uint8 tensor [B, H, W, 3] (RGB)")
        return int(tiles.shape[0]), int(tiles.shape[1]), int(tiles.shape[2])

    # ---- S1 + S2 ----
    def infer(self, tiles: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45, max_det: int = 1000,
              out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, slot: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """uint8 [B,H,W,3] -> (dets float32 [B,max_det,6] = x1,y1,x2,y2,conf,cls ; counts int32 [B]).
        Work is enqueued on torch's current stream; ``slot`` selects the workspace (use one slot per stream when
        several batches are in flight)."""
        B, H, W = self._check_tiles(tiles)
        ws = self.workspace(B, H, W, slot)
        if out is None:
            dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=self.device)
            counts = torch.empty((B,), dtype=torch.int32, device=self.device)
        else:
            dets, counts = out
        _check(self.lib.aq_engine_infer(self.handle, tiles.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(),
                                        dets.data_ptr(), counts.data_ptr(), conf_thres, iou_thres, max_det, _stream_ptr()))
        return dets, counts

    # ---- S1 ----
    def forward_raw(self, tiles: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] -> pred float32 [B, N, 5+nc] (what Detect.forward returns at inference)."""
        B, H, W = self._check_tiles(tiles)
        ws = self.workspace(B, H, W)
        pred = torch.empty((B, self.num_candidates(H, W), self.no), dtype=torch.float32, device=self.device)
        _check(self.lib.aq_engine_forward_raw(self.handle, tiles.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(),
                                              pred.data_ptr(), _stream_ptr()))
        return pred

    # ---- S2 ----
    def nms(self, pred: torch.Tensor, conf_thres=0.25, iou_thres=0.45, max_det=1000) -> Tuple[torch.Tensor, torch.Tensor]:
        return nms(pred, self.ck.nc, conf_thres, iou_thres, max_det)

    def set_nms_options(self, agnostic: bool = False, classes: Optional[Sequence[int]] = None) -> None:
        """detect.py's --agnostic-nms / --classes for the NMS step of infer() [UPSTREAM non_max_suppression(classes, agnostic)]."""
        lo, hi = class_mask(classes, self.ck.nc)
        _check(self.lib.aq_engine_set_nms_options(self.handle, int(bool(agnostic)), lo, hi))

    # ---- test / tuning hooks ----
    def tensor(self, tensor_id: int, B: int) -> torch.Tensor:
        """View of plan tensor ``tensor_id`` ([B,h,w,C]) inside the workspace after a call (tests only)."""
        p, c, h, w, eb = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.aq_engine_tensor_ptr(self.handle, tensor_id, C.byref(p), C.byref(c), C.byref(h), C.byref(w), C.byref(eb)))
        off = p.value - self._ws.data_ptr()
        nbytes = B * h.value * w.value * c.value * eb.value
        dt = {1: torch.uint8, 2: torch.bfloat16, 4: torch.float32}[eb.value]
        return self._ws[off:off + nbytes].view(dt).view(B, h.value, w.value, c.value)

    def tensor_by_name(self, name: str, B: int) -> torch.Tensor:
        ids = [i for i, t in enumerate(self.plan.tensors) if t.name == name]
        return self.tensor(ids[0], B)

    def set_conv_config(self, op: int, cfg: int) -> None:
        _check(self.lib.aq_engine_set_conv_config(self.handle, op, cfg))

    def set_tuned_table(self, B: int, H: int, W: int, cfgs) -> None:
        """Install an autotune result (from a cache file or another rank) for batches of exactly this geometry."""
        arr = (C.c_int * len(cfgs))(*[int(c) for c in cfgs])
        _check(self.lib.aq_engine_set_tuned_table(self.handle, B, H, W, arr, len(cfgs)))

    def tune_key(self, B: int, H: int, W: int) -> str:
        """Key of a tuned table: model, precision, geometry, library version + candidate count + plan shape (new kernels invalidate old
        tables), and the environment switches that remove kernels from the candidate set (such a run has a table of its own)."""
        key = (f"{self.ck.variant}:nc{self.ck.nc}:p{self.precision_name}:{B}x{H}x{W}:n{self.lib.aq_conv_num_configs()}"
               f":v{self.lib.aq_version()}:ops{len(self.plan.ops)}")
        off = ",".join(sorted(f"{k}={v}" for k, v in os.environ.items() if k.startswith("AQ_DISABLE_") or k in ("AQ_PL_W8", "AQ_PL_ASM", "AQ_PL_NB", "AQ_C1_ASM", "AQ_C1_ASM_NB")))
        return key + (":" + off if off else "")

    def autotune(self, tiles: torch.Tensor, reps: int = 3, cache: Optional[str] = None, shipped: bool = True) -> List[int]:
        """Pick the fastest tile configuration per conv op for this batch geometry (synchronises).
        ``cache``: optional JSON file; a stored table for the same model/precision/geometry is applied
        instead of re-timing (used to keep tuning launches out of rocprof traces).
        ``shipped``: on a cache miss look in the table that ships in-tree (aquaculture_amd/data/tuned_tables.json: the geometries of
        BASELINE.json's configs, timed on MI355X with this library version) before timing anything -- two runs of the same build then
        launch the same kernels and write the same bf16 label bytes (the timing sweep's picks between near-equal shapes flip from run to
        run); ``shipped=False`` (bench.py --retune, AQ_RETUNE=1) times regardless."""
        import json
        B, H, W = self._check_tiles(tiles)
        key = self.tune_key(B, H, W)
        table = {}
        if cache and os.path.exists(cache):
            try:
                with open(cache) as f:
                    table = json.load(f)
            except (OSError, ValueError):      # unreadable or half-written by another rank: tune again
                table = {}
            if key in table and len(table[key]) == len(self.plan.ops):
                try:
                    self.set_tuned_table(B, H, W, table[key])
                    return list(table[key])
                except RuntimeError:           # an entry this build / environment has no kernel for: tune again, replace the entry
                    pass
        if shipped and os.environ.get("AQ_RETUNE") != "1":
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tuned_tables.json")) as f:
                    ship = json.load(f)
                if key in ship and len(ship[key]) == len(self.plan.ops):
                    self.set_tuned_table(B, H, W, ship[key])
                    self.tuned_from = "shipped table"
                    return list(ship[key])
            except (OSError, ValueError, RuntimeError):
                pass
        self.tuned_from = "timed in this run"
        ws = self.workspace(B, H, W)
        _check(self.lib.aq_engine_autotune(self.handle, tiles.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), reps, _stream_ptr()))
        cfgs = [self.lib.aq_engine_get_conv_config(self.handle, i) for i in range(len(self.plan.ops))]
        if cache:
            table[key] = cfgs
            os.makedirs(os.path.dirname(os.path.abspath(cache)), exist_ok=True)
            tmp = f"{cache}.{os.getpid()}.tmp"     # ranks of one job may share the file: replace it atomically
            with open(tmp, "w") as f:
                json.dump(table, f)
            os.replace(tmp, cache)
        return cfgs

    FAMILIES = {0: "none", 1: "igemm_or_halo", 2: "pl3x3", 3: "pl3x3_w8", 4: "pl3x3s2", 5: "pl3x3_f8", 6: "direct1x1", 7: "direct1x1_f8out",
                8: "direct3x3s2", 9: "bottleneck", 10: "downblock", 11: "stem", 12: "head_decode", 13: "asm1x1"}

    def last_launches(self) -> List[Tuple[str, int]]:
        """(kernel family, tile-configuration id) of every op's most recent launch (aq_engine_last_launch) -- what actually ran, as opposed
        to what the tuned table asked for."""
        out = []
        fam, cfg = C.c_int(), C.c_int()
        for i in range(len(self.plan.ops)):
            _check(self.lib.aq_engine_last_launch(self.handle, i, C.byref(fam), C.byref(cfg)))
            out.append((self.FAMILIES.get(fam.value, str(fam.value)), cfg.value))
        return out

    def profile(self, enable: bool, ring: int = 32) -> None:
        _check(self.lib.aq_engine_profile(self.handle, int(enable), ring))

    def op_times_ms(self) -> Tuple[np.ndarray, int]:
        n = len(self.plan.ops)
        buf = (C.c_float * n)()
        calls = C.c_int()
        _check(self.lib.aq_engine_op_times(self.handle, buf, n, C.byref(calls)))
        return np.array(buf[:], dtype=np.float64), calls.value


# --------------------------------------------------------------------------------------
# individual kernels (used by the parity tests; same C entry points the engine uses)
# --------------------------------------------------------------------------------------
def class_mask(classes: Optional[Sequence[int]], nc: int) -> Tuple[int, int]:
    """--classes as the two 64-bit words aq_nms_opts takes (None = every class)."""
    if classes is None:
        return (1 << 64) - 1, (1 << 64) - 1
    m = 0
    for c in classes:
        if not 0 <= int(c) < min(nc, 128):
            raise ValueError(f"--classes {c}: the model has classes 0 .. {nc - 1}")
        m |= 1 << int(c)
    return m & ((1 << 64) - 1), m >> 64


def nms(pred: torch.Tensor, nc: int, conf_thres=0.25, iou_thres=0.45, max_det=1000, agnostic: bool = False,
        classes: Optional[Sequence[int]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """non_max_suppression(pred, conf, iou, classes, agnostic, multi_label=False, max_det) on device."""
    _require_gpu()
    lib = load_library()
    if pred.dtype != torch.float32 or pred.dim() != 3 or not pred.is_cuda or not pred.is_contiguous():
        raise ValueError("pred must be a contiguous CUDA float32 tensor [B, N, 5+nc]")
    B, N, no = pred.shape
    if no != nc + 5:
        raise ValueError(f"pred last dim {no} != nc + 5 = {nc + 5}")
    scratch = torch.empty(lib.aq_nms_scratch_bytes(B, N), dtype=torch.uint8, device=pred.device)
    dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=pred.device)
    counts = torch.empty((B,), dtype=torch.int32, device=pred.device)
    lo, hi = class_mask(classes, nc)
    _check(lib.aq_nms_opts(pred.data_ptr(), N, B, N, nc, conf_thres, iou_thres, max_det, None, None, 0,
                           scratch.data_ptr(), dets.data_ptr(), counts.data_ptr(), int(bool(agnostic)), lo, hi, _stream_ptr()))
    return dets, counts


_zero_pages: Dict[int, torch.Tensor] = {}


def _zero_page(device) -> torch.Tensor:
    key = device.index or 0
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(4096, dtype=torch.uint8, device=device)
    return _zero_pages[key]


def pack_conv_weights(w_oihw: torch.Tensor, precision: str, device) -> torch.Tensor:
    """fp32 (Cout,Cin,k,k) -> packed device buffer the conv kernels read."""
    lib = load_library()
    prec = PRECISIONS[precision]
    w = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    cout, k, _, cin = w.shape
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_conv_weights(wp, cout, k, cin, prec, None, C.byref(n), None))
    buf = torch.empty(n.value, dtype=torch.uint8, device=device)
    _check(lib.aq_pack_conv_weights(wp, cout, k, cin, prec, buf.data_ptr(), C.byref(n), _stream_ptr()))
    return buf


def conv2d_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, stride=1, pad=None, act=True,
                residual: Optional[torch.Tensor] = None, precision="bf16", out_f32=False,
                cfg: Optional[int] = None) -> torch.Tensor:
    """out = (residual +) SiLU(conv(x, w) + b) on NHWC tensors through aq_conv2d (tests)."""
    _require_gpu()
    lib = load_library()
    prec = PRECISIONS[precision]
    assert x.is_cuda and x.is_contiguous() and x.dtype == _act_dtype(prec)
    B, H, W, cin = x.shape
    cout, _, k, _ = w_oihw.shape
    pad = k // 2 if pad is None else pad
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if prec == AQ_F16X3:          # split mode: the packer writes the weights' fp16 halves and the bias / scale buffer the kernel reads
        wk = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
        bh = np.ascontiguousarray(bias.float().cpu().numpy())
        n, nb = C.c_size_t(), C.c_size_t()
        wp_, bp_ = wk.ctypes.data_as(C.POINTER(C.c_float)), bh.ctypes.data_as(C.POINTER(C.c_float))
        _check(lib.aq_pack_conv_weights_x3(wp_, bp_, cout, k, cin, None, C.byref(n), None, C.byref(nb), None))
        wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
        bbuf = torch.empty(nb.value, dtype=torch.float32, device=x.device)
        _check(lib.aq_pack_conv_weights_x3(wp_, bp_, cout, k, cin, wbuf.data_ptr(), C.byref(n), bbuf.data_ptr(), C.byref(nb), _stream_ptr()))
    else:
        wbuf = pack_conv_weights(w_oihw, precision, x.device)
        bbuf = torch.zeros(cout + 512, dtype=torch.float32, device=x.device)
        bbuf[:cout] = bias.float().to(x.device)
    odt = torch.float32 if (out_f32 or prec in (AQ_FP32, AQ_F16X3)) else torch.bfloat16
    out = torch.empty((B, Ho, Wo, cout), dtype=odt, device=x.device)
    res_ptr = residual.data_ptr() if residual is not None else None
    old = os.environ.get("AQ_CONV_CFG")
    if cfg is not None:
        os.environ["AQ_CONV_CFG"] = str(cfg)
    try:
        _check(lib.aq_conv2d(x.data_ptr(), cin, 0, cin, out.data_ptr(), cout, 0, cout, res_ptr, cout, 0,
                             wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, k, stride, pad, int(act), prec, int(out_f32),
                             _zero_page(x.device).data_ptr(), _stream_ptr()))
    finally:
        if cfg is not None:
            if old is None:
                os.environ.pop("AQ_CONV_CFG", None)
            else:
                os.environ["AQ_CONV_CFG"] = old
    torch.cuda.current_stream().synchronize()   # wbuf/bbuf are freed on return
    return out


_lb_tables: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor, Tuple[int, int, int, int, int, int]]] = {}


def letterbox_device(tiles0: torch.Tensor, new_shape=(640, 640), stride: int = 32, auto: bool = True) -> torch.Tensor:
    """uint8 CUDA [B,H0,W0,3] original tiles -> letterboxed uint8 [B,H,W,3] on the device (aq_letterbox_u8).
    Geometry and coefficient tables come from dataloader.letterbox_geometry / resize tables (host, cached per shape)."""
    from . import dataloader
    _require_gpu()
    lib = load_library()
    assert tiles0.is_cuda and tiles0.dtype == torch.uint8 and tiles0.dim() == 4 and tiles0.shape[3] == 3 and tiles0.is_contiguous()
    B, H0, W0, _ = tiles0.shape
    key = (H0, W0, tuple(new_shape), stride, auto, tiles0.device.index)
    if key not in _lb_tables:
        (nw, nh), (top, bottom, left, right) = dataloader.letterbox_geometry((H0, W0), new_shape, auto, True, stride)
        xt = np.stack(dataloader._axis_coeffs(W0, nw), 1).astype(np.int32)
        yt = np.stack(dataloader._axis_coeffs(H0, nh), 1).astype(np.int32)
        _lb_tables[key] = (torch.from_numpy(xt).to(tiles0.device), torch.from_numpy(yt).to(tiles0.device),
                           (nw, nh, top, left, nh + top + bottom, nw + left + right))
    xt, yt, (nw, nh, top, left, H, W) = _lb_tables[key]
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=tiles0.device)
    _check(lib.aq_letterbox_u8(tiles0.data_ptr(), B, H0, W0, out.data_ptr(), H, W, nw, nh, top, left,
                               xt.data_ptr(), yt.data_ptr(), _stream_ptr()))
    return out


def _letterbox_tables(H0: int, W0: int, new_shape, stride: int, auto: bool, device):
    from . import dataloader
    key = (H0, W0, tuple(new_shape), stride, auto, device.index)
    if key not in _lb_tables:
        (nw, nh), (top, bottom, left, right) = dataloader.letterbox_geometry((H0, W0), new_shape, auto, True, stride)
        xt = np.stack(dataloader._axis_coeffs(W0, nw), 1).astype(np.int32)
        yt = np.stack(dataloader._axis_coeffs(H0, nh), 1).astype(np.int32)
        _lb_tables[key] = (torch.from_numpy(xt).to(device), torch.from_numpy(yt).to(device),
                           (nw, nh, top, left, nh + top + bottom, nw + left + right))
    return _lb_tables[key]


def letterbox_scene_tiles(scene: torch.Tensor, origins, tile_hw, new_shape=(640, 640), stride: int = 32, auto: bool = True) -> torch.Tensor:
    """uint8 CUDA scene raster [Hs,Ws,3] + tile origins [(x0, y0), ...] of equal size tile_hw=(H0,W0) -> letterboxed uint8 [B,H,W,3]
    (aq_letterbox_tiles_u8): crop (reference src/load_data/tile_tifs.py:33-47) and letterbox in one pass, no tile copies."""
    _require_gpu()
    lib = load_library()
    assert scene.is_cuda and scene.dtype == torch.uint8 and scene.dim() == 3 and scene.shape[2] == 3 and scene.is_contiguous()
    Hs, Ws, _ = scene.shape
    H0, W0 = tile_hw
    row_b = Ws * 3
    offs = np.asarray([y0 * row_b + x0 * 3 for x0, y0 in origins], dtype=np.int64)
    B = int(offs.shape[0])
    xt, yt, (nw, nh, top, left, H, W) = _letterbox_tables(H0, W0, new_shape, stride, auto, scene.device)
    offs_dev = torch.from_numpy(offs).to(scene.device)
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=scene.device)
    _check(lib.aq_letterbox_tiles_u8(scene.data_ptr(), Hs * row_b, row_b, offs_dev.data_ptr(), offs.ctypes.data, B, H0, W0, out.data_ptr(),
                                     H, W, nw, nh, top, left, xt.data_ptr(), yt.data_ptr(), _stream_ptr()))
    return out


def format_label_rows(rows: np.ndarray, save_conf: bool = True) -> bytes:
    """rows float32 [n,6] (cls xc yc w h conf) -> label-file bytes via the C formatter (releases the GIL)."""
    lib = load_library()
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n = rows.shape[0]
    if n == 0:
        return b""
    buf = C.create_string_buffer(n * 6 * 16)
    k = lib.aq_format_label_rows(rows.ctypes.data_as(C.POINTER(C.c_float)), n, int(save_conf), buf, len(buf))
    if k < 0:
        buf = C.create_string_buffer(-k + 1)
        k = lib.aq_format_label_rows(rows.ctypes.data_as(C.POINTER(C.c_float)), n, int(save_conf), buf, len(buf))
    return buf.raw[:k]


def write_label_files(labels_dir: str, stems, rows: np.ndarray, offsets: np.ndarray, save_conf: bool = True, fsync: bool = False) -> int:
    """One C call per batch (aq_write_label_files; releases the GIL for formatting and file system calls): tile t's rows are
    rows[offsets[t]:offsets[t + 1]] (float32 [N, 6], file order); returns the number of label files written (tiles without rows get none)."""
    lib = load_library()
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(stems)
    assert offsets.shape[0] == n + 1 and (rows.shape[0] == 0 or rows.shape[1] == 6)
    arr = (C.c_char_p * n)(*[os.fsencode(s_) for s_ in stems])
    k = lib.aq_write_label_files(os.fsencode(labels_dir), arr, rows.ctypes.data_as(C.POINTER(C.c_float)), offsets.ctypes.data_as(C.POINTER(C.c_longlong)), n,
                                 int(save_conf), int(fsync))
    if k < 0:
        raise OSError(f"could not write the label file of {stems[-1 - k]} in {labels_dir}")
    return int(k)


def stem_conv_nhwc(tiles_u8: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, act: bool = True, precision: str = "bf16") -> torch.Tensor:
    """uint8 [B,H,W,3] -> SiLU(conv6x6/s2/p2(x / 255, w) + b) as NHWC [B,H/2,W/2,cout] through aq_stem_conv (tests)."""
    _require_gpu()
    lib = load_library()
    prec = PRECISIONS[precision]
    B, H, W, _ = tiles_u8.shape
    cout = w_oihw.shape[0]
    w = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_stem_weights(wp, cout, prec, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=tiles_u8.device)
    _check(lib.aq_pack_stem_weights(wp, cout, prec, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = torch.zeros(64, dtype=torch.float32, device=tiles_u8.device)
    bbuf[:cout] = bias.float().to(tiles_u8.device)
    out = torch.empty((B, H // 2, W // 2, cout), dtype=_act_dtype(prec), device=tiles_u8.device)
    _check(lib.aq_stem_conv(tiles_u8.data_ptr(), out.data_ptr(), cout, 0, cout, wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, int(act), prec,
                            _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def bottleneck_nhwc(x: torch.Tensor, w1_oihw: torch.Tensor, b1: torch.Tensor, w2_oihw: torch.Tensor, b2: torch.Tensor,
                    shortcut: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 NHWC [B,H,W,C] (may be a channel slice of a wider tensor) -> x + SiLU(conv3x3(SiLU(conv1x1(x)))) through
    aq_bottleneck (tests).  ``out`` may be a channel slice too; it must not overlap ``x``."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(3) == 1
    B, H, W, c = x.shape
    ld = x.stride(2)
    assert x.stride(1) == W * ld and x.stride(0) == H * W * ld, "x must be a channel slice of a dense NHWC tensor"
    w1 = np.ascontiguousarray(w1_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    w2 = np.ascontiguousarray(w2_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    p1, p2 = w1.ctypes.data_as(C.POINTER(C.c_float)), w2.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_bottleneck_weights(p1, p2, c, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_bottleneck_weights(p1, p2, c, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = torch.cat([b1.float(), b2.float()]).to(x.device).contiguous()
    if out is None:
        out = torch.empty((B, H, W, c), dtype=torch.bfloat16, device=x.device)
    old = out.stride(2)
    # data_ptr() of a channel slice already points at its first channel: pass ch_off = 0 with the parent's row length
    _check(lib.aq_bottleneck(x.data_ptr(), ld, 0, out.data_ptr(), old, 0, c, wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, int(shortcut),
                             _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def downblock_nhwc(x: torch.Tensor, wa_oihw: torch.Tensor, ba: torch.Tensor, wb_oihw: torch.Tensor, bb: torch.Tensor,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 NHWC [B,H,W,48] (may be a channel slice) -> SiLU(conv1x1(SiLU(conv3x3/s2(x)))) [B,H/2,W/2,96] through aq_downblock (tests)."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(3) == 1 and x.shape[3] == 48
    B, H, W, _ = x.shape
    ld = x.stride(2)
    assert x.stride(1) == W * ld and x.stride(0) == H * W * ld, "x must be a channel slice of a dense NHWC tensor"
    wa = np.ascontiguousarray(wa_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    wb = np.ascontiguousarray(wb_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    pa, pb = wa.ctypes.data_as(C.POINTER(C.c_float)), wb.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_downblock_weights(pa, pb, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_downblock_weights(pa, pb, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = torch.cat([ba.float(), bb.float()]).to(x.device).contiguous()
    if out is None:
        out = torch.empty((B, H // 2, W // 2, 96), dtype=torch.bfloat16, device=x.device)
    _check(lib.aq_downblock(x.data_ptr(), ld, 0, out.data_ptr(), out.stride(2), 0, wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


CONV_CFG_DIRECT1X1 = 1000   # AQ_CONV_CFG_DIRECT1X1
CONV_CFG_ASM1X1 = 1004      # AQ_CONV_CFG_ASM1X1
CONV_CFG_DIRECT3X3S2 = 1001  # AQ_CONV_CFG_DIRECT3X3S2
CONV_CFG_PL3X3 = 1002       # AQ_CONV_CFG_PL3X3
CONV_CFG_PL3X3S2 = 1003     # AQ_CONV_CFG_PL3X3S2
CONV_CFG_ONE_TILE_PER_WG = 4096  # AQ_CONV_CFG_ONE_TILE_PER_WG (OR-ed into a tile configuration id)


def conv3x3_pl_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, act: bool = True,
                    residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, w8: bool = False) -> torch.Tensor:
    """bf16 NHWC [B,H,W,cin] (may be a channel slice) -> (residual +) SiLU(conv3x3/s1/p1(x) + b) through aq_conv3x3_pl (tests).
    ``out`` and ``residual`` may be channel slices of wider tensors; ``residual`` may be ``out`` itself (in-place shortcut).
    ``w8``: through aq_conv3x3_pl_w8 (e4m3 weight stream; ``w_oihw`` must lie on an fp8 grid, quant.quantize_rows)."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(3) == 1
    B, H, W, cin = x.shape
    cout = w_oihw.shape[0]
    ld = x.stride(2)
    assert x.stride(1) == W * ld and x.stride(0) == H * W * ld, "x must be a channel slice of a dense NHWC tensor"
    w = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    if w8:
        if not lib.aq_conv3x3_pl_w8_supported(cin, cout, B, H, W):
            raise RuntimeError("aq_conv3x3_pl_w8 does not support this shape")
        bh = np.ascontiguousarray(bias.float().cpu().numpy())
        bp = bh.ctypes.data_as(C.POINTER(C.c_float))
        _check(lib.aq_pack_conv3x3_pl_w8(wp, bp, cin, cout, None, C.byref(n), None, None))
        wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
        bbuf = torch.empty(2048, dtype=torch.float32, device=x.device)
        _check(lib.aq_pack_conv3x3_pl_w8(wp, bp, cin, cout, wbuf.data_ptr(), C.byref(n), bbuf.data_ptr(), _stream_ptr()))
    else:
        _check(lib.aq_pack_conv3x3_pl(wp, cin, cout, None, C.byref(n), None))
        wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
        _check(lib.aq_pack_conv3x3_pl(wp, cin, cout, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
        bbuf = bias.float().to(x.device).contiguous()
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=x.device)
    assert out.stride(3) == 1 and (residual is None or residual.stride(3) == 1)
    _check((lib.aq_conv3x3_pl_w8 if w8 else lib.aq_conv3x3_pl)(x.data_ptr(), ld * 2, 16, cin, out.data_ptr(), out.stride(2), 0, cout,
                             residual.data_ptr() if residual is not None else None, residual.stride(2) if residual is not None else 0, 0,
                             wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, int(act), _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def conv1x1_direct_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, act: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [..., cin] pixels (may be a channel slice of a wider dense tensor) -> SiLU(W x + b) through aq_conv1x1_direct (tests)."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(-1) == 1
    cin, cout = x.shape[-1], w_oihw.shape[0]
    ld = x.stride(-2)
    npix = x.numel() // cin
    w = np.ascontiguousarray(w_oihw.reshape(cout, cin).float().cpu().numpy())
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_conv1x1_direct(wp, cin, cout, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_conv1x1_direct(wp, cin, cout, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = bias.float().to(x.device).contiguous()
    if out is None:
        out = torch.empty(x.shape[:-1] + (cout,), dtype=torch.bfloat16, device=x.device)
    _check(lib.aq_conv1x1_direct(x.data_ptr(), ld, 0, out.data_ptr(), out.stride(-2), 0, cin, cout, wbuf.data_ptr(), bbuf.data_ptr(), npix, int(act),
                                 _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def conv1x1_asm_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [..., cin] pixels (may be a channel slice of a wider dense tensor) -> SiLU(W x + b) through aq_conv1x1_asm, the generated-assembly
    wide 1x1 (tests, tools/time_conv1x1_asm.py)."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(-1) == 1
    cin, cout = x.shape[-1], w_oihw.shape[0]
    ld = x.stride(-2)
    npix = x.numel() // cin
    w = np.ascontiguousarray(w_oihw.reshape(cout, cin).float().cpu().numpy())
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_conv1x1_asm(wp, cin, cout, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_conv1x1_asm(wp, cin, cout, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = bias.float().to(x.device).contiguous()
    if out is None:
        out = torch.empty(x.shape[:-1] + (cout,), dtype=torch.bfloat16, device=x.device)
    _check(lib.aq_conv1x1_asm(x.data_ptr(), ld, 0, out.data_ptr(), out.stride(-2), 0, cin, cout, wbuf.data_ptr(), bbuf.data_ptr(), npix, 1, _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def conv3x3s2_direct_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, act: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 NHWC [B,H,W,cin] (may be a channel slice) -> SiLU(conv3x3/s2/p1(x) + b) [B,H/2,W/2,cout] through aq_conv3x3s2_direct (tests)."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(3) == 1
    B, H, W, cin = x.shape
    cout = w_oihw.shape[0]
    ld = x.stride(2)
    assert x.stride(1) == W * ld and x.stride(0) == H * W * ld, "x must be a channel slice of a dense NHWC tensor"
    w = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_conv3x3s2_direct(wp, cin, cout, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_conv3x3s2_direct(wp, cin, cout, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = bias.float().to(x.device).contiguous()
    if out is None:
        out = torch.empty((B, H // 2, W // 2, cout), dtype=torch.bfloat16, device=x.device)
    _check(lib.aq_conv3x3s2_direct(x.data_ptr(), ld, 0, out.data_ptr(), out.stride(2), 0, cin, cout, wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, int(act),
                                   _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def conv3x3_pl_s2_nhwc(x: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, act: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 NHWC [B,H,W,cin] (may be a channel slice) -> SiLU(conv3x3/s2/p1(x) + b) [B,H/2,W/2,cout] (may be a channel slice) through
    aq_conv3x3_pl_s2, the planar stride-2 kernel (tests, tools)."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(3) == 1
    B, H, W, cin = x.shape
    cout = w_oihw.shape[0]
    ld = x.stride(2)
    assert x.stride(1) == W * ld and x.stride(0) == H * W * ld, "x must be a channel slice of a dense NHWC tensor"
    w = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_conv3x3_pl_s2(wp, cin, cout, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_conv3x3_pl_s2(wp, cin, cout, wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = torch.zeros((cout + 255) // 256 * 256 + 1024, dtype=torch.float32, device=x.device)    # the kernel's bias tile over-reads up to 1024 floats
    bbuf[:cout] = bias.float().to(x.device)
    if out is None:
        out = torch.empty((B, H // 2, W // 2, cout), dtype=torch.bfloat16, device=x.device)
    assert out.stride(3) == 1 and out.stride(1) == (W // 2) * out.stride(2)
    # channel offsets are expressed through the base pointers (slices): in_choff = out_choff = 0
    _check(lib.aq_conv3x3_pl_s2(x.data_ptr(), ld, 0, cin, out.data_ptr(), out.stride(2), 0, cout, wbuf.data_ptr(), bbuf.data_ptr(), B, H, W, int(act),
                                _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def conv3x3_pl_f8_nhwc(xq: torch.Tensor, act_scale: float, w_oihw: torch.Tensor, bias: torch.Tensor, act: bool = True,
                       residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """e4m3 codes NHWC [B,H,W,cin] (uint8 or float8_e4m3fn view; may be a channel slice; value = code x act_scale) ->
    (residual +) SiLU(conv3x3/s1/p1 + b) as bf16 [B,H,W,cout] through aq_conv3x3_pl_f8 (fp8 MFMA on both operands; tests, tools)."""
    _require_gpu()
    lib = load_library()
    xq = xq.view(torch.uint8)
    assert xq.stride(3) == 1
    B, H, W, cin = xq.shape
    cout = w_oihw.shape[0]
    ld = xq.stride(2)
    assert xq.stride(1) == W * ld and xq.stride(0) == H * W * ld, "x must be a channel slice of a dense NHWC tensor"
    w = np.ascontiguousarray(w_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    bh = np.ascontiguousarray(bias.float().cpu().numpy())
    n = C.c_size_t()
    wp, bp = w.ctypes.data_as(C.POINTER(C.c_float)), bh.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib.aq_pack_conv3x3_pl_f8(wp, bp, cin, cout, float(act_scale), None, C.byref(n), None, None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=xq.device)
    sb = torch.empty(2048, dtype=torch.float32, device=xq.device)
    _check(lib.aq_pack_conv3x3_pl_f8(wp, bp, cin, cout, float(act_scale), wbuf.data_ptr(), C.byref(n), sb.data_ptr(), _stream_ptr()))
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=xq.device)
    assert out.dtype == torch.bfloat16 and out.stride(3) == 1 and (residual is None or (residual.dtype == torch.bfloat16 and residual.stride(3) == 1))
    _check(lib.aq_conv3x3_pl_f8(xq.data_ptr(), ld, 0, cin, out.data_ptr(), out.stride(2), 0, cout,
                                residual.data_ptr() if residual is not None else None, residual.stride(2) if residual is not None else 0, 0,
                                wbuf.data_ptr(), sb.data_ptr(), B, H, W, int(act), _stream_ptr()))
    torch.cuda.current_stream().synchronize()
    return out


def jpeg_idct_rgb(coef: torch.Tensor, coef_off: torch.Tensor, qt: torch.Tensor, H: int, W: int, out: Optional[torch.Tensor] = None,
                  scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The device half of the split JPEG decode (aq_jpeg_idct_rgb): coef int16 CUDA (the images' coefficient blocks), coef_off int64 [B]
    (first value of each image, multiples of 64), qt uint16 [B,3,64]  ->  uint8 RGB [B,H,W,3], the pixels libjpeg(-turbo) produces."""
    _require_gpu()
    lib = load_library()
    B = int(coef_off.shape[0])
    assert coef.is_cuda and coef.dtype == torch.int16 and coef_off.dtype == torch.int64 and qt.dtype in (torch.uint16, torch.int16) and qt.numel() == B * 192
    if out is None:
        out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=coef.device)
    n = lib.aq_jpeg_scratch_bytes(B, H, W)
    if scratch is None or scratch.numel() < n:
        scratch = torch.empty(n, dtype=torch.uint8, device=coef.device)
    _check(lib.aq_jpeg_idct_rgb(coef.data_ptr(), coef_off.data_ptr(), qt.data_ptr(), B, H, W, scratch.data_ptr(), out.data_ptr(), _stream_ptr()))
    return out


def jpeg_huffman_decode(streams: torch.Tensor, segs: torch.Tensor, tabsets: torch.Tensor, coef: torch.Tensor, status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GPU entropy decode (aq_jpeg_huffman_decode): ``streams`` uint8 CUDA = the upload buffer jpeg.GpuDecodeBatch filled, ``segs`` uint8
    CUDA [nseg, 32] = its segment descriptors, ``tabsets`` uint8 CUDA [n_sets, jpeg.TABSET_BYTES]; ``coef`` int16 CUDA, ZEROED by the caller,
    receives the quantised coefficient blocks in the host decoder's layout.  Returns the per-segment status tensor (int32; 0 = ok)."""
    _require_gpu()
    lib = load_library()
    assert streams.is_cuda and streams.dtype == torch.uint8 and segs.is_cuda and segs.dtype == torch.uint8 and segs.dim() == 2 and segs.shape[1] == 32
    assert tabsets.is_cuda and tabsets.dtype == torch.uint8 and coef.is_cuda and coef.dtype == torch.int16
    nseg = int(segs.shape[0])
    if status is None:
        status = torch.empty(nseg, dtype=torch.int32, device=streams.device)
    _check(lib.aq_jpeg_huffman_decode(streams.data_ptr(), segs.data_ptr(), nseg, tabsets.data_ptr(), coef.data_ptr(), status.data_ptr(), _stream_ptr()))
    return status


def jpeg_slots_to_rgb(slots: torch.Tensor, H: int, W: int, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A batch as the decode workers' coefficient mode delivers it -- uint8 CUDA [b, jpeg.slot_bytes(H, W)]: per image the int16 coefficient
    blocks, then three uint16 quantisation tables -- to RGB tiles uint8 [b, H, W, 3] on the device."""
    from . import jpeg
    assert slots.is_cuda and slots.dtype == torch.uint8 and slots.dim() == 2 and slots.is_contiguous() and slots.shape[1] == jpeg.slot_bytes(H, W)
    b = slots.shape[0]
    nco = jpeg.coef_count(H, W)
    qt = slots[:, 2 * nco:2 * nco + 384].contiguous().view(torch.int16)
    off = torch.arange(b, dtype=torch.int64, device=slots.device) * (slots.shape[1] // 2)
    return jpeg_idct_rgb(slots.view(torch.int16).reshape(-1), off, qt, H, W, scratch=scratch)


def head_decode_level(x: torch.Tensor, w_oi: torch.Tensor, bias: torch.Tensor, cand_off: int, stride: float, anchors_px, nc: int,
                      conf_thres: float, cap: int):
    """One Detect level through aq_head_decode (tests): x bf16 NHWC [B, ny, nx, cin] (may be a channel slice), w [na * (nc + 5), cin].
    Returns (counts [B] int32, cand [B, cap] int32, rows [B, cap, nc + 5] float32); the order within an image is unspecified."""
    _require_gpu()
    lib = load_library()
    assert x.dtype == torch.bfloat16 and x.stride(3) == 1
    B, ny, nx, cin = x.shape
    ld = x.stride(2)
    assert x.stride(1) == nx * ld and x.stride(0) == ny * nx * ld
    na = len(anchors_px)
    cout = na * (nc + 5)
    w = np.ascontiguousarray(w_oi.float().cpu().numpy().reshape(cout, cin))
    bh = np.ascontiguousarray(bias.float().cpu().numpy())
    n = C.c_size_t()
    fp = C.POINTER(C.c_float)
    _check(lib.aq_pack_head_weights(w.ctypes.data_as(fp), bh.ctypes.data_as(fp), cin, cout, None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=x.device)
    _check(lib.aq_pack_head_weights(w.ctypes.data_as(fp), bh.ctypes.data_as(fp), cin, cout, C.c_void_p(wbuf.data_ptr()), C.byref(n), C.c_void_p(_stream_ptr())))
    counts = torch.zeros(B, dtype=torch.int32, device=x.device)
    cand = torch.full((B, cap), -1, dtype=torch.int32, device=x.device)
    rows = torch.zeros((B, cap, nc + 5), dtype=torch.float32, device=x.device)
    anch = np.ascontiguousarray(np.asarray(anchors_px, np.float32).reshape(-1))
    _check(lib.aq_head_decode(C.c_void_p(x.data_ptr()), ld, 0, cin, C.c_void_p(wbuf.data_ptr()), B, ny, nx, cand_off, C.c_float(stride),
                              anch.ctypes.data_as(fp), nc, na, C.c_float(conf_thres), C.c_void_p(cand.data_ptr()), C.c_void_p(rows.data_ptr()),
                              C.c_void_p(counts.data_ptr()), 1, cap, C.c_void_p(_stream_ptr())))
    torch.cuda.current_stream().synchronize()
    return counts, cand, rows


def stemdown_nhwc(tiles_u8: torch.Tensor, ws_oihw: torch.Tensor, bs: torch.Tensor, wa_oihw: torch.Tensor, ba: torch.Tensor,
                  wb_oihw: torch.Tensor, bb: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [B,Hi,Wi,3] -> stem (6x6/s2, 3 -> 48) -> 3x3/s2 (48 -> 96) -> 1x1 (96 -> 96), SiLU after each, as bf16 NHWC
    [B,Hi/4,Wi/4,96] through aq_stemdown (tests): one launch, the stem's output never leaves the CU."""
    _require_gpu()
    lib = load_library()
    B, Hi, Wi, _ = tiles_u8.shape
    assert tiles_u8.dtype == torch.uint8 and tiles_u8.is_contiguous() and ws_oihw.shape[0] == 48
    dev = tiles_u8.device
    fp = C.POINTER(C.c_float)
    ws = np.ascontiguousarray(ws_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    n = C.c_size_t()
    _check(lib.aq_pack_stem_weights(ws.ctypes.data_as(fp), 48, AQ_BF16, None, C.byref(n), None))
    wsbuf = torch.empty(n.value, dtype=torch.uint8, device=dev)
    _check(lib.aq_pack_stem_weights(ws.ctypes.data_as(fp), 48, AQ_BF16, wsbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bsbuf = torch.zeros(64, dtype=torch.float32, device=dev)
    bsbuf[:48] = bs.float().to(dev)
    wa = np.ascontiguousarray(wa_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    wb = np.ascontiguousarray(wb_oihw.permute(0, 2, 3, 1).float().cpu().numpy())
    _check(lib.aq_pack_downblock_weights(wa.ctypes.data_as(fp), wb.ctypes.data_as(fp), None, C.byref(n), None))
    wbuf = torch.empty(n.value, dtype=torch.uint8, device=dev)
    _check(lib.aq_pack_downblock_weights(wa.ctypes.data_as(fp), wb.ctypes.data_as(fp), wbuf.data_ptr(), C.byref(n), _stream_ptr()))
    bbuf = torch.cat([ba.float(), bb.float()]).to(dev).contiguous()
    if out is None:
        out = torch.empty((B, Hi // 4, Wi // 4, 96), dtype=torch.bfloat16, device=dev)
    _check(lib.aq_stemdown(C.c_void_p(tiles_u8.data_ptr()), C.c_void_p(out.data_ptr()), out.stride(2), 0, C.c_void_p(wsbuf.data_ptr()),
                           C.c_void_p(bsbuf.data_ptr()), C.c_void_p(wbuf.data_ptr()), C.c_void_p(bbuf.data_ptr()), B, Hi, Wi, C.c_void_p(_stream_ptr())))
    torch.cuda.current_stream().synchronize()
    return out
