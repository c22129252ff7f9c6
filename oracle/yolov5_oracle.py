"""CPU oracle for the YOLOv5 detect.py hot path.  TEST INFRASTRUCTURE ONLY.

*** PARITY UNPINNED ***  The reference's implementation of this path is not in
the reference tree: /root/reference/yolov5/yolov5/ is an empty, un-vendored
ultralytics/yolov5 submodule (pin unknown; bracketed to ~v7.0 by
reference requirements.txt:238,241,151), the trained weights are a missing blob
(reference .MISSING_LARGE_BLOBS:3) and the reference holds no tests or golden
vectors for the path.  This file is therefore a
restatement of the published upstream algorithm, anchored on the reference's
call site ``python3 yolov5/detect.py --weights ... --source ... --nosave
--save-txt --save-conf`` (reference README.md:77) and on the output grammar the
reference's consumer parses (reference src/process_yolo/geocode_results.py:140-172).
Its structure is pinned only by architecture invariants (param/FLOP counts,
candidate count) and by cross-checks between independent restatements
(torch.nn.functional vs the plain-C loops in oracle/ref_kernels.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (aquaculture_amd) never does.

Every function names the upstream function it restates ([UPSTREAM] = ultralytics/yolov5
v6.x/v7.0, not present under /root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MAX_WH = 7680.0     # [UPSTREAM utils/general.py non_max_suppression: max_wh]
MAX_NMS = 30000     # [UPSTREAM non_max_suppression: max_nms]


# --------------------------------------------------------------------------------------
# precision models
# --------------------------------------------------------------------------------------
def q_fp32(t: torch.Tensor) -> torch.Tensor:
    return t


def q_bf16(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16 and back: the storage rounding of the engine's bf16 mode."""
    return t.to(torch.bfloat16).to(torch.float32)


def wq_fp8_e4m3(w: torch.Tensor) -> torch.Tensor:
    """The fp8w mode's weight quantiser, restated with torch's own float8 type: per output channel (dim 0) the smallest power of two s
    with max|w| / s <= 448, then w -> float8_e4m3fn(w / s) * s (round to nearest even; OCP e4m3fn, the gfx950 fp8).  Independent of
    aquaculture_amd/quant.py (numpy arithmetic), which the tests hold to it."""
    flat = w.reshape(w.shape[0], -1).double()
    amax = flat.abs().amax(1)
    e = torch.where(amax > 0, torch.ceil(torch.log2(amax.clamp(min=1e-300) / 448.0)), torch.zeros_like(amax))
    s = torch.pow(2.0, e).view(-1, *([1] * (w.dim() - 1)))
    return ((w.double() / s).float().to(torch.float8_e4m3fn).float().double() * s).float()


def q_fp8_act(t: torch.Tensor, scale: float) -> torch.Tensor:
    """The fp8 mode's activation quantiser (engine: the Bottleneck cv1 epilogue, aquaculture_amd/csrc/conv1x1_direct.hip F8OUT):
    e4m3fn CODES (as floats) of t / scale, saturating at +-448, round to nearest even.  The tensor's value is code x scale."""
    return (t.float() / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def wq_fp8_real(w: torch.Tensor):
    """The fp8 mode's weight quantiser (engine: aq_pack_conv3x3_pl_f8): per output channel s = max|w| / 448 in fp32, codes =
    float8_e4m3fn(w / s).  Returns (codes as floats, s [cout])."""
    amax = w.float().abs().amax(dim=tuple(range(1, w.dim())))
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    codes = (w.float() / s.view(-1, *([1] * (w.dim() - 1)))).to(torch.float8_e4m3fn).float()
    return codes, s


def q_bf16_f64(t: torch.Tensor) -> torch.Tensor:
    """The same storage rounding with the arithmetic between two roundings carried in fp64: against ``q_bf16`` (fp32 accumulation in
    PyTorch's summation order) it measures how far results move when ONLY the accumulation changes -- the floor any other
    implementation of the same bf16 pipeline (the HIP engine: MFMA order, fp32) is entitled to (tests/test_gpu_bf16_deviation.py)."""
    return t.to(torch.bfloat16).to(torch.float64)


# --------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------
def fuse_conv_and_bn(state: Dict[str, torch.Tensor], key: str, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """[UPSTREAM utils/torch_utils.py fuse_conv_and_bn] in fp32."""
    w = state[f"{key}.conv.weight"].float()
    gamma, beta = state[f"{key}.bn.weight"].float(), state[f"{key}.bn.bias"].float()
    mean, var = state[f"{key}.bn.running_mean"].float(), state[f"{key}.bn.running_var"].float()
    w_bn = torch.diag(gamma.div(torch.sqrt(eps + var)))
    w_f = torch.mm(w_bn, w.view(w.shape[0], -1)).view(w.shape)
    b_bn = beta - gamma.mul(mean).div(torch.sqrt(var + eps))
    b_f = torch.mm(w_bn, torch.zeros(w.shape[0], 1)).reshape(-1) + b_bn
    return w_f, b_f


class OracleModel:
    """Fused yolov5 (v6 architecture) forward built from torch.nn.functional CPU ops.

    ``quant`` models activation/weight storage rounding (identity = the fp32 reference semantics
    of detect.py without --half; ``q_bf16`` = the engine's bf16 mode: bf16 weights and activations,
    fp32 accumulation, fp32 bias/SiLU/residual epilogue, fp32 detect head)."""

    def __init__(self, state: Dict[str, torch.Tensor], nc: int, anchors_grid: torch.Tensor,
                 stride: Sequence[float] = (8.0, 16.0, 32.0), bn_eps: float = 1e-3,
                 quant: Callable[[torch.Tensor], torch.Tensor] = q_fp32, wquant: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 f8_scales: Optional[Dict[str, float]] = None):
        self.state = state
        self.nc, self.no = nc, nc + 5
        self.anchors = anchors_grid.float()          # (nl, na, 2) grid units
        self.stride = [float(s) for s in stride]
        self.na = int(self.anchors.shape[1])
        self.eps = bn_eps
        self.q = quant
        self.f8 = dict(f8_scales or {})   # fp8 mode (configs[3]): "<bottleneck key>.cv2" -> e4m3 scale of that layer's INPUT (the cv1 output);
        #                                  those 3x3 layers multiply e4m3 codes on both sides (q_fp8_act x wq_fp8_real), everything else as ``quant``
        self.wq = wquant          # weight-only quantiser of the conv layers (fp8w mode), applied to the fused weights before ``quant``
        self._fused: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.taps: Optional[Dict[str, torch.Tensor]] = None   # per-module outputs when tracing

    # -- blocks [UPSTREAM models/common.py] --
    def _wb(self, key):
        if key not in self._fused:
            w, b = fuse_conv_and_bn(self.state, key, self.eps)
            if self.wq is not None:
                w = self.wq(w)
            w = self.q(w)
            self._fused[key] = (w, b.to(w.dtype))
        return self._fused[key]

    def conv(self, x, key, k, s, p=None):
        """Conv.forward_fuse: act(conv(x)), act = SiLU, autopad p = k // 2."""
        w, b = self._wb(key)
        y = F.conv2d(x, w, b, stride=s, padding=k // 2 if p is None else p)
        return F.silu(y)

    def bottleneck(self, x, key, shortcut):
        """Bottleneck: x + cv2(cv1(x)) if shortcut else cv2(cv1(x)); cv1 1x1, cv2 3x3."""
        sa = self.f8.get(f"{key}.cv2")
        if sa:
            # fp8 pair: cv1's fp32 epilogue output goes straight to e4m3 codes (no bf16 rounding in between); cv2 multiplies codes by
            # codes (exact products, fp32 accumulate) and rescales by act_scale x w_scale[co] before bias and SiLU
            tq = q_fp8_act(self.conv(x, f"{key}.cv1", 1, 1), sa)
            w, b = fuse_conv_and_bn(self.state, f"{key}.cv2", self.eps)
            wc, ws = wq_fp8_real(w)
            y = F.silu(F.conv2d(tq, wc, None, padding=1) * (sa * ws).view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
            return self.q(x + y if shortcut else y)
        t = self.q(self.conv(x, f"{key}.cv1", 1, 1))
        y = self.conv(t, f"{key}.cv2", 3, 1)
        return self.q(x + y if shortcut else y)

    def c3(self, x, key, n, shortcut):
        """C3: cv3(cat(m(cv1(x)), cv2(x)), 1)."""
        a = self.q(self.conv(x, f"{key}.cv1", 1, 1))
        for j in range(n):
            a = self.bottleneck(a, f"{key}.m.{j}", shortcut)
        b = self.q(self.conv(x, f"{key}.cv2", 1, 1))
        return self.q(self.conv(torch.cat((a, b), 1), f"{key}.cv3", 1, 1))

    def sppf(self, x, key):
        """SPPF: x = cv1(x); y1 = m(x); y2 = m(y1); cv2(cat(x, y1, y2, m(y2))), m = MaxPool2d(5, 1, 2)."""
        x = self.q(self.conv(x, f"{key}.cv1", 1, 1))
        y1 = F.max_pool2d(x, 5, 1, 2)
        y2 = F.max_pool2d(y1, 5, 1, 2)
        y3 = F.max_pool2d(y2, 5, 1, 2)
        return self.q(self.conv(torch.cat((x, y1, y2, y3), 1), f"{key}.cv2", 1, 1))

    def _depth(self, idx):
        return len({k.split(".")[3] for k in self.state if k.startswith(f"model.{idx}.m.")})

    def features(self, x: torch.Tensor) -> List[torch.Tensor]:
        """[UPSTREAM models/yolo.py BaseModel._forward_once] over models/yolov5*.yaml (v6.0 graph)."""
        tap = (lambda name, t: self.taps.__setitem__(name, t)) if self.taps is not None else (lambda n, t: None)
        x = self.q(x)
        x0 = self.q(self.conv(x, "model.0", 6, 2, 2)); tap("model.0", x0)
        x1 = self.q(self.conv(x0, "model.1", 3, 2)); tap("model.1", x1)
        x2 = self.c3(x1, "model.2", self._depth(2), True); tap("model.2", x2)
        x3 = self.q(self.conv(x2, "model.3", 3, 2)); tap("model.3", x3)
        x4 = self.c3(x3, "model.4", self._depth(4), True); tap("model.4", x4)
        x5 = self.q(self.conv(x4, "model.5", 3, 2)); tap("model.5", x5)
        x6 = self.c3(x5, "model.6", self._depth(6), True); tap("model.6", x6)
        x7 = self.q(self.conv(x6, "model.7", 3, 2)); tap("model.7", x7)
        x8 = self.c3(x7, "model.8", self._depth(8), True); tap("model.8", x8)
        x9 = self.sppf(x8, "model.9"); tap("model.9", x9)
        x10 = self.q(self.conv(x9, "model.10", 1, 1)); tap("model.10", x10)
        x12 = torch.cat((F.interpolate(x10, scale_factor=2.0, mode="nearest"), x6), 1)
        x13 = self.c3(x12, "model.13", self._depth(13), False); tap("model.13", x13)
        x14 = self.q(self.conv(x13, "model.14", 1, 1)); tap("model.14", x14)
        x16 = torch.cat((F.interpolate(x14, scale_factor=2.0, mode="nearest"), x4), 1)
        x17 = self.c3(x16, "model.17", self._depth(17), False); tap("model.17", x17)
        x18 = self.q(self.conv(x17, "model.18", 3, 2)); tap("model.18", x18)
        x20 = self.c3(torch.cat((x18, x14), 1), "model.20", self._depth(20), False); tap("model.20", x20)
        x21 = self.q(self.conv(x20, "model.21", 3, 2)); tap("model.21", x21)
        x23 = self.c3(torch.cat((x21, x10), 1), "model.23", self._depth(23), False); tap("model.23", x23)
        return [x17, x20, x23]

    def detect(self, feats: List[torch.Tensor]) -> torch.Tensor:
        """[UPSTREAM models/yolo.py Detect.forward (inference branch), Detect._make_grid]."""
        z = []
        for i, x in enumerate(feats):
            w = self.q(self.state[f"model.24.m.{i}.weight"].float())
            b = self.state[f"model.24.m.{i}.bias"].float().to(w.dtype)
            x = F.conv2d(x, w, b)
            if self.taps is not None:
                self.taps[f"model.24.m.{i}"] = x
            bs, _, ny, nx = x.shape
            x = x.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            yv, xv = torch.meshgrid(torch.arange(ny, dtype=x.dtype), torch.arange(nx, dtype=x.dtype), indexing="ij")
            grid = torch.stack((xv, yv), 2).expand(1, self.na, ny, nx, 2) - 0.5
            anchor_grid = (self.anchors[i] * self.stride[i]).to(x.dtype).view(1, self.na, 1, 1, 2).expand(1, self.na, ny, nx, 2)
            xy, wh, conf = x.sigmoid().split((2, 2, self.nc + 1), 4)
            xy = (xy * 2 + grid) * self.stride[i]
            wh = (wh * 2) ** 2 * anchor_grid
            y = torch.cat((xy, wh, conf), 4)
            z.append(y.view(bs, self.na * nx * ny, self.no))
        return torch.cat(z, 1)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: float32 (B,3,H,W) in [0,1], RGB  ->  (B, sum na*ny*nx, 5+nc): xywh px, obj, cls."""
        return self.detect(self.features(x))


def preprocess(tiles_u8_nhwc: np.ndarray) -> torch.Tensor:
    """[UPSTREAM detect.py run]: im = torch.from_numpy(im).float(); im /= 255 (CHW, RGB)."""
    im = torch.from_numpy(np.ascontiguousarray(tiles_u8_nhwc)).permute(0, 3, 1, 2).float()
    im /= 255
    return im


# --------------------------------------------------------------------------------------
# NMS  [UPSTREAM utils/general.py non_max_suppression + torchvision/csrc/ops/cpu/nms_kernel.cpp]
# --------------------------------------------------------------------------------------
def xywh2xyxy(x: np.ndarray) -> np.ndarray:
    """[UPSTREAM utils/general.py xywh2xyxy] fp32: x1 = x - w/2 ..."""
    y = np.empty_like(x)
    y[..., 0] = x[..., 0] - x[..., 2] / np.float32(2)
    y[..., 1] = x[..., 1] - x[..., 3] / np.float32(2)
    y[..., 2] = x[..., 0] + x[..., 2] / np.float32(2)
    y[..., 3] = x[..., 1] + x[..., 3] / np.float32(2)
    return y


def greedy_nms(boxes: np.ndarray, iou_thres: float) -> List[int]:
    """torchvision nms_kernel_impl restated: boxes already in descending score order (fp32 xyxy).
    Suppress j when inter / (area_i + area_j - inter) > thr (strict), all fp32."""
    n = boxes.shape[0]
    x1, y1, x2, y2 = (boxes[:, i].astype(np.float32) for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    thr = np.float32(iou_thres)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = np.float32(0)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):   # 0/0 for zero-area boxes -> NaN -> "not > thr", as in C
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return keep


def non_max_suppression(pred: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        max_det: int = 1000, agnostic: bool = False, classes=None) -> List[np.ndarray]:
    """[UPSTREAM non_max_suppression(pred, conf, iou, classes=None, agnostic, multi_label=False, max_det)].

    Returns per image an (n,6) fp32 array [x1,y1,x2,y2,conf,cls] in descending confidence.
    Deliberate deviations: no wall-clock ``time_limit`` break; confidence ties are
    ordered by ascending candidate index (upstream's argsort is unstable, i.e. unspecified)."""
    pred = np.asarray(pred, dtype=np.float32)
    out = []
    ct = np.float32(conf_thres)
    for x in pred:
        x = x[x[:, 4] > ct]
        if not x.shape[0]:
            out.append(np.zeros((0, 6), np.float32))
            continue
        x = x.copy()
        x[:, 5:] *= x[:, 4:5]                      # conf = obj_conf * cls_conf
        box = xywh2xyxy(x[:, :4])
        j = x[:, 5:].argmax(1)                     # first index on ties, like torch.max
        conf = x[np.arange(x.shape[0]), 5 + j]
        m = conf > ct
        box, conf, j = box[m], conf[m], j[m].astype(np.float32)
        if classes is not None:                    # [UPSTREAM: x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]]
            m = np.isin(j, np.asarray(classes, dtype=np.float32))
            box, conf, j = box[m], conf[m], j[m]
        if not box.shape[0]:
            out.append(np.zeros((0, 6), np.float32))
            continue
        order = np.argsort(-conf, kind="stable")[:MAX_NMS]
        box, conf, j = box[order], conf[order], j[order]
        c = j * np.float32(0.0 if agnostic else MAX_WH)
        keep = greedy_nms(box + c[:, None], iou_thres)[:max_det]
        out.append(np.concatenate((box[keep], conf[keep, None], j[keep, None]), 1).astype(np.float32))
    return out


# --------------------------------------------------------------------------------------
# box rescale + label text   [UPSTREAM utils/general.py scale_boxes/clip_boxes/xyxy2xywh, detect.py writer]
# --------------------------------------------------------------------------------------
def scale_boxes(img1_shape, boxes: np.ndarray, img0_shape) -> np.ndarray:
    """Letterbox inverse in fp32, then clip to the original image."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    b = torch.from_numpy(np.array(boxes, dtype=np.float32, copy=True))
    b[..., [0, 2]] -= pad[0]
    b[..., [1, 3]] -= pad[1]
    b[..., :4] /= gain
    b[..., 0].clamp_(0, img0_shape[1])
    b[..., 1].clamp_(0, img0_shape[0])
    b[..., 2].clamp_(0, img0_shape[1])
    b[..., 3].clamp_(0, img0_shape[0])
    return b.numpy()


def label_lines(det: np.ndarray, img1_shape, img0_shape) -> List[str]:
    """detect.py --save-txt --save-conf: per detection ``cls xc yc w h conf`` each ``%g``; rows written
    for ``reversed(det)`` i.e. ascending confidence; boxes rounded (half-to-even) in original pixels
    and normalised by gn = (w0, h0, w0, h0)."""
    if det.shape[0] == 0:
        return []
    d = torch.from_numpy(det.astype(np.float32).copy())
    d[:, :4] = torch.from_numpy(scale_boxes(img1_shape, d[:, :4].numpy(), img0_shape)).round()
    gn = torch.tensor(img0_shape)[[1, 0, 1, 0]]
    lines = []
    for row in reversed(d):
        xyxy, conf, cls = row[:4], row[4], row[5]
        x = xyxy.view(1, 4)
        y = x.clone()
        y[..., 0] = (x[..., 0] + x[..., 2]) / 2
        y[..., 1] = (x[..., 1] + x[..., 3]) / 2
        y[..., 2] = x[..., 2] - x[..., 0]
        y[..., 3] = x[..., 3] - x[..., 1]
        xywh = (y / gn).view(-1).tolist()
        line = (cls, *xywh, conf)
        lines.append(("%g " * len(line)).rstrip() % line)
    return lines


def detect_tiles(model: OracleModel, tiles_u8_nhwc: np.ndarray, conf_thres=0.25, iou_thres=0.45,
                 max_det=1000, batch: int = 1) -> List[np.ndarray]:
    """End to end for 'no letterbox needed' tiles (H, W multiples of 32): (n,6) detections per tile."""
    dets: List[np.ndarray] = []
    for s in range(0, tiles_u8_nhwc.shape[0], batch):
        pred = model.forward(preprocess(tiles_u8_nhwc[s:s + batch])).numpy()
        dets.extend(non_max_suppression(pred, conf_thres, iou_thres, max_det))
    return dets


def model_from_checkpoint(ck, quant=q_fp32, wquant=None, f8_scales=None) -> OracleModel:
    """``ck``: any object with .state, .nc, .anchors (grid units), .stride, .bn_eps.  ``wquant=wq_fp8_e4m3`` with ``quant=q_bf16`` is
    the engine's fp8w mode; ``f8_scales`` (with ``quant=q_bf16``) its fp8 mode: {"model.6.m.0.cv2": act_scale, ...}."""
    return OracleModel(ck.state, ck.nc, ck.anchors, ck.stride, ck.bn_eps, quant, wquant, f8_scales)
