"""YOLOv5 (v6.x/v7.0 architecture) -> flat execution plan for the HIP engine.

The reference runs ``yolov5/detect.py`` (reference README.md:77) on a YOLOv5m
checkpoint trained from ``yolov5m.pt`` (reference README.md:52).  The module
graph itself lives in the un-vendored ultralytics/yolov5 submodule
(/root/reference/yolov5/yolov5 is empty); its topology is restated here from
SURVEY.md section 3.3 (models/yolov5m.yaml + models/yolo.py parse_model
[UPSTREAM]) and must be re-checked against a real checkpoint's pickled ``yaml``
the first time one is available (``checkpoint.load_checkpoint`` does that check).

This module is *planning only*: it decides which HBM buffer every layer output
lives in so that every ``Concat`` is free (producers write channel slices of the
concatenated buffer), every C3's ``cv1`` and ``cv2`` become one stacked 1x1
convolution, the Bottleneck chain runs in place, and the 6x6/s2 stem becomes a
3x3/s1 convolution over a 2x2 space-to-depth image.  No arithmetic happens here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

# op kinds (mirrors enum aq_op_kind in include/aq_engine.h)
OP_PREPROCESS = 0   # u8 NHWC(3) -> space-to-depth 16ch, /255
OP_CONV = 1         # implicit-GEMM conv (+bias, +SiLU, +residual)
OP_SPPF_POOL = 2    # three chained 5x5/s1/p2 max pools -> 3 channel slices
OP_UPSAMPLE2X = 3   # nearest 2x into a channel slice
OP_DECODE = 4       # sigmoid + grid/anchor decode + obj threshold + compaction
OP_NMS = 5          # per-tile class-offset greedy NMS
OP_STEM = 6         # fused u8 -> /255 -> Conv(3, C, 6, 2, 2) + SiLU (stem_conv.hip)
OP_BOTTLENECK = 7   # fused Bottleneck (1x1 -> 3x3 -> + shortcut) for small hidden widths, bf16 engines only
OP_DOWNBLOCK = 8    # fused 3x3/s2 (48 -> 96) + stacked 1x1 (96 -> 96): yolov5m's model.1 + model.2.cv1|cv2, bf16 engines only

VARIANTS = {
    # name: (depth_multiple, width_multiple)   [UPSTREAM models/yolov5{n,s,m,l,x}.yaml]
    "yolov5n": (0.33, 0.25),
    "yolov5s": (0.33, 0.50),
    "yolov5m": (0.67, 0.75),
    "yolov5l": (1.00, 1.00),
    "yolov5x": (1.33, 1.25),
}

DEFAULT_ANCHORS = (  # pixels, [UPSTREAM models/yolov5m.yaml]; real ones come from the checkpoint
    (10, 13, 16, 30, 33, 23),
    (30, 61, 62, 45, 59, 119),
    (116, 90, 156, 198, 373, 326),
)
STRIDES = (8, 16, 32)
STEM_S2D_CH = 16  # 2x2x3 = 12 real channels, zero-padded to one 16-channel K block


def make_divisible(x: float, divisor: int = 8) -> int:
    """[UPSTREAM utils/general.py make_divisible]"""
    return int(math.ceil(x / divisor) * divisor)


def scaled_depth(n: int, gd: float) -> int:
    """[UPSTREAM models/yolo.py parse_model]: n = max(round(n * gd), 1) if n > 1 else n"""
    return max(round(n * gd), 1) if n > 1 else n


@dataclass
class TensorSpec:
    """One HBM buffer: NHWC, ``channels`` contiguous, spatial = input / ``down``."""
    name: str
    channels: int
    down: int            # spatial downsample factor relative to the network input
    dtype: str = "act"   # "act" (engine precision), "f32", "u8"


@dataclass
class Slice:
    tensor: int
    ch_off: int
    channels: int


@dataclass
class Op:
    kind: int
    name: str
    src: Optional[Slice] = None
    dst: Optional[Slice] = None
    res: Optional[Slice] = None       # residual input (Bottleneck shortcut)
    k: int = 1
    stride: int = 1
    pad: int = 0
    act: int = 1                       # 1 = SiLU, 0 = identity
    weight_keys: Tuple[str, ...] = ()  # upstream module paths whose fused (W, b) are stacked along Cout
    level: int = -1                    # detect level for DECODE / head conv
    flops_per_tile: float = 0.0        # algorithmic 2*MAC at the plan's imgsz (filled by Plan.finalize)
    meta: Dict[str, object] = field(default_factory=dict)


@dataclass
class Plan:
    variant: str
    nc: int
    no: int
    na: int
    channels: Tuple[int, ...]
    tensors: List[TensorSpec]
    ops: List[Op]
    head_tensors: Tuple[int, int, int]
    input_tensor: int

    def conv_ops(self) -> List[Op]:
        """Ops that carry weights (implicit-GEMM convs and the fused stem), in plan order."""
        return [o for o in self.ops if o.kind in (OP_CONV, OP_STEM, OP_BOTTLENECK, OP_DOWNBLOCK)]

    def flops(self, h: int, w: int) -> Dict[str, float]:
        """Algorithmic FLOPs per tile (2 x MAC, true Cin of the upstream layer, no padding)."""
        out = {"total": 0.0, "conv3x3": 0.0, "conv1x1": 0.0, "stem": 0.0, "detect": 0.0}
        for o in self.conv_ops():
            down = self.tensors[o.dst.tensor].down
            ho, wo = h // down, w // down
            if o.kind == OP_DOWNBLOCK:        # 3x3/s2 (src -> dst channels) + 1x1 (dst -> dst channels)
                f3 = 2.0 * ho * wo * o.dst.channels * o.src.channels * 9
                f1 = 2.0 * ho * wo * o.dst.channels * o.dst.channels
                o.flops_per_tile = f3 + f1
                out["total"] += f3 + f1
                out["conv3x3"] += f3
                out["conv1x1"] += f1
                continue
            if o.kind == OP_BOTTLENECK:       # cv1 (1x1) + cv2 (3x3), both C -> C
                f1 = 2.0 * ho * wo * o.dst.channels * o.src.channels
                o.flops_per_tile = 10.0 * f1
                out["total"] += 10.0 * f1
                out["conv1x1"] += f1
                out["conv3x3"] += 9.0 * f1
                continue
            cin = o.meta.get("true_cin", o.src.channels)
            kk = o.meta.get("true_k", o.k)
            f = 2.0 * ho * wo * o.dst.channels * cin * kk * kk
            if "true_cout" in o.meta:
                f = 2.0 * ho * wo * o.meta["true_cout"] * cin * kk * kk
            o.flops_per_tile = f
            out["total"] += f
            cls = o.meta.get("class", "conv1x1")
            out[cls] += f
        return out


FUSED_BOTTLENECK_WIDTHS = (16, 32, 48, 64, 96)   # hidden widths csrc/bottleneck.hip keeps in registers


class _Builder:
    def __init__(self, fused_bottleneck: bool = False):
        self.tensors: List[TensorSpec] = []
        self.ops: List[Op] = []
        self.fused_bottleneck = fused_bottleneck

    def tensor(self, name, channels, down, dtype="act") -> int:
        self.tensors.append(TensorSpec(name, channels, down, dtype))
        return len(self.tensors) - 1

    def conv(self, name, src: Slice, dst: Slice, k, stride, keys, act=1, res=None, **meta) -> Slice:
        self.ops.append(Op(OP_CONV, name, src=src, dst=dst, res=res, k=k, stride=stride,
                           pad=k // 2, act=act, weight_keys=tuple(keys), meta=dict(meta)))
        return dst

    def c3(self, idx: int, src: Slice, dst: Slice, c2: int, n: int, shortcut: bool, down: int, down_from: Optional[Slice] = None) -> Slice:
        """C3: cv3(cat(m(cv1 x), cv2 x)) [UPSTREAM models/common.py C3, Bottleneck].

        cv1 and cv2 read the same x, so they run as ONE 1x1 conv whose Cout is
        [cv1 | cv2] writing the whole concat buffer; the n Bottlenecks then run in
        place on channels [0, c_): t = cv1_1x1(x1) -> tmp, x1 = (x1 +) cv2_3x3(t).
        """
        c_ = c2 // 2
        cat = self.tensor(f"m{idx}.cat", 2 * c_, down)
        tmp = self.tensor(f"m{idx}.tmp", c_, down)
        p = f"model.{idx}"
        if down_from is not None:
            # the 3x3/s2 conv in front of this C3 and the stacked cv1|cv2 run as ONE op (csrc/downblock.hip): `src` (the
            # down-sampling conv's own output buffer) is never written
            self.ops.append(Op(OP_DOWNBLOCK, f"model.{idx - 1}+{p}.cv1|cv2", src=down_from, dst=Slice(cat, 0, 2 * c_), k=3, stride=2, pad=1,
                               act=1, weight_keys=(f"model.{idx - 1}", f"{p}.cv1", f"{p}.cv2"), meta={"class": "downblock"}))
        else:
            self.conv(f"{p}.cv1|cv2", src, Slice(cat, 0, 2 * c_), 1, 1, (f"{p}.cv1", f"{p}.cv2"))
        x1 = Slice(cat, 0, c_)
        # Fused form (bf16 engines, small c_): Bottlenecks run in PAIRS, ping-ponging cat[:c_] -> tmp -> cat[:c_]
        # (the fused kernel reads halos, so it cannot run in place); an odd one out runs in the two-kernel form.
        n_fused = 2 * (n // 2) if (self.fused_bottleneck and c_ in FUSED_BOTTLENECK_WIDTHS) else 0
        for j in range(n_fused):
            a, bdst = (x1, Slice(tmp, 0, c_)) if j % 2 == 0 else (Slice(tmp, 0, c_), x1)
            self.ops.append(Op(OP_BOTTLENECK, f"{p}.m.{j}", src=a, dst=bdst, res=a if shortcut else None, k=3, stride=1, pad=1, act=1,
                               weight_keys=(f"{p}.m.{j}.cv1", f"{p}.m.{j}.cv2"), meta={"class": "bottleneck"}))
        for j in range(n_fused, n):
            self.conv(f"{p}.m.{j}.cv1", x1, Slice(tmp, 0, c_), 1, 1, (f"{p}.m.{j}.cv1",))
            self.conv(f"{p}.m.{j}.cv2", Slice(tmp, 0, c_), x1, 3, 1, (f"{p}.m.{j}.cv2",),
                      res=x1 if shortcut else None, **{"class": "conv3x3"})
        self.conv(f"{p}.cv3", Slice(cat, 0, 2 * c_), dst, 1, 1, (f"{p}.cv3",))
        return dst


def build_plan(variant: str = "yolov5m", nc: int = 5, na: int = 3, fused_stem: bool = True, fused_bottleneck: bool = False) -> Plan:
    """Flatten the v6 architecture into ops over sliced NHWC buffers."""
    gd, gw = VARIANTS[variant]
    ch = tuple(make_divisible(c * gw) for c in (64, 128, 256, 512, 1024))
    c1, c2, c3, c4, c5 = ch
    n3, n6, n9 = scaled_depth(3, gd), scaled_depth(6, gd), scaled_depth(9, gd)
    no = nc + 5
    b = _Builder(fused_bottleneck)

    # concat buffers that upstream builds with torch.cat (layers 12, 16, 19, 22): producers write slices.
    cat12 = b.tensor("cat12", c4 + c4, 16)   # [up(10) | out6]
    cat16 = b.tensor("cat16", c3 + c3, 8)    # [up(14) | out4]
    cat19 = b.tensor("cat19", c3 + c3, 16)   # [out18 | out14]
    cat22 = b.tensor("cat22", c4 + c4, 32)   # [out21 | out10]

    # 0: Conv(3, c1, 6, 2, 2).  Default: ONE fused kernel from the uint8 tiles (stem_conv.hip).  Alternative (kept for
    # A/B and for channel counts above 64): preprocess to a 2x2 space-to-depth image (12 -> 16 ch) + generic 3x3/s1 conv.
    t_in = b.tensor("tiles_u8", 3, 1, "u8")
    t0 = b.tensor("out0", c1, 2)
    if fused_stem and c1 <= 64:
        b.ops.append(Op(OP_STEM, "model.0", src=Slice(t_in, 0, 3), dst=Slice(t0, 0, c1), k=6, stride=2, pad=2, act=1,
                        weight_keys=("model.0",), meta={"class": "stem"}))
        x = Slice(t0, 0, c1)
    else:
        t_s2d = b.tensor("s2d", STEM_S2D_CH, 2)
        b.ops.append(Op(OP_PREPROCESS, "preprocess", src=Slice(t_in, 0, 3), dst=Slice(t_s2d, 0, STEM_S2D_CH)))
        x = b.conv("model.0", Slice(t_s2d, 0, STEM_S2D_CH), Slice(t0, 0, c1), 3, 1, ("model.0",),
                   **{"class": "stem", "true_cin": 3, "true_k": 6, "stem_s2d": True})
    # 1: Conv(c1, c2, 3, 2);  2: C3(c2, c2, n3).  bf16 engines with yolov5m's widths fuse model.1 with the C3's stacked cv1|cv2.
    t1 = b.tensor("out1", c2, 4)
    t2 = b.tensor("out2", c2, 4)
    if fused_bottleneck and (c1, c2) == (48, 96):
        x = b.c3(2, Slice(t1, 0, c2), Slice(t2, 0, c2), c2, n3, True, 4, down_from=x)
    else:
        x = b.conv("model.1", x, Slice(t1, 0, c2), 3, 2, ("model.1",), **{"class": "conv3x3"})
        x = b.c3(2, x, Slice(t2, 0, c2), c2, n3, True, 4)
    # 3: Conv(c2, c3, 3, 2)
    t3 = b.tensor("out3", c3, 8)
    x = b.conv("model.3", x, Slice(t3, 0, c3), 3, 2, ("model.3",), **{"class": "conv3x3"})
    # 4: C3(c3, c3, n6) -> P3 skip, lives in cat16[c3:]
    x = out4 = b.c3(4, x, Slice(cat16, c3, c3), c3, n6, True, 8)
    # 5: Conv(c3, c4, 3, 2)
    t5 = b.tensor("out5", c4, 16)
    x = b.conv("model.5", x, Slice(t5, 0, c4), 3, 2, ("model.5",), **{"class": "conv3x3"})
    # 6: C3(c4, c4, n9) -> P4 skip, lives in cat12[c4:]
    x = out6 = b.c3(6, x, Slice(cat12, c4, c4), c4, n9, True, 16)
    # 7: Conv(c4, c5, 3, 2)
    t7 = b.tensor("out7", c5, 32)
    x = b.conv("model.7", x, Slice(t7, 0, c5), 3, 2, ("model.7",), **{"class": "conv3x3"})
    # 8: C3(c5, c5, n3)
    t8 = b.tensor("out8", c5, 32)
    x = b.c3(8, x, Slice(t8, 0, c5), c5, n3, True, 32)
    # 9: SPPF(c5, c5, 5): cv1 -> slice 0 of [x|y1|y2|y3]; pools fill slices 1..3; cv2 reads all
    ch_ = c5 // 2
    sppf = b.tensor("m9.cat", 4 * ch_, 32)
    b.conv("model.9.cv1", x, Slice(sppf, 0, ch_), 1, 1, ("model.9.cv1",))
    b.ops.append(Op(OP_SPPF_POOL, "model.9.m", src=Slice(sppf, 0, ch_), dst=Slice(sppf, ch_, 3 * ch_), k=5))
    t9 = b.tensor("out9", c5, 32)
    x = b.conv("model.9.cv2", Slice(sppf, 0, 4 * ch_), Slice(t9, 0, c5), 1, 1, ("model.9.cv2",))
    # 10: Conv(c5, c4, 1, 1) -> saved for cat22[c4:]
    x = out10 = b.conv("model.10", x, Slice(cat22, c4, c4), 1, 1, ("model.10",))
    # 11/12: Upsample + Concat(-1, 6)
    b.ops.append(Op(OP_UPSAMPLE2X, "model.11", src=out10, dst=Slice(cat12, 0, c4)))
    # 13: C3(2*c4, c4, n3, False)
    t13 = b.tensor("out13", c4, 16)
    x = b.c3(13, Slice(cat12, 0, 2 * c4), Slice(t13, 0, c4), c4, n3, False, 16)
    # 14: Conv(c4, c3, 1, 1) -> saved for cat19[c3:]
    x = out14 = b.conv("model.14", x, Slice(cat19, c3, c3), 1, 1, ("model.14",))
    # 15/16: Upsample + Concat(-1, 4)
    b.ops.append(Op(OP_UPSAMPLE2X, "model.15", src=out14, dst=Slice(cat16, 0, c3)))
    # 17: C3(2*c3, c3, n3, False) -> P3 out
    t17 = b.tensor("out17", c3, 8)
    p3 = b.c3(17, Slice(cat16, 0, 2 * c3), Slice(t17, 0, c3), c3, n3, False, 8)
    # 18: Conv(c3, c3, 3, 2) -> cat19[:c3]
    b.conv("model.18", p3, Slice(cat19, 0, c3), 3, 2, ("model.18",), **{"class": "conv3x3"})
    # 20: C3(2*c3, c4, n3, False) -> P4 out
    t20 = b.tensor("out20", c4, 16)
    p4 = b.c3(20, Slice(cat19, 0, 2 * c3), Slice(t20, 0, c4), c4, n3, False, 16)
    # 21: Conv(c4, c4, 3, 2) -> cat22[:c4]
    b.conv("model.21", p4, Slice(cat22, 0, c4), 3, 2, ("model.21",), **{"class": "conv3x3"})
    # 23: C3(2*c4, c5, n3, False) -> P5 out
    t23 = b.tensor("out23", c5, 32)
    p5 = b.c3(23, Slice(cat22, 0, 2 * c4), Slice(t23, 0, c5), c5, n3, False, 32)

    # 24: Detect: per level a 1x1 conv (bias, no activation) to na*no channels, kept in fp32,
    # channel-padded to a multiple of 32 so it is one MFMA M block; then decode + threshold.
    head_c = make_divisible(na * no, 32)
    heads = []
    for lvl, (feat, down) in enumerate(((p3, 8), (p4, 16), (p5, 32))):
        th = b.tensor(f"head{lvl}", head_c, down, "f32")
        b.conv(f"model.24.m.{lvl}", feat, Slice(th, 0, head_c), 1, 1, (f"model.24.m.{lvl}",), act=0,
               **{"class": "detect", "true_cout": na * no, "level": lvl})
        b.ops[-1].level = lvl
        heads.append(th)
    b.ops.append(Op(OP_DECODE, "model.24.decode"))
    b.ops.append(Op(OP_NMS, "nms"))

    return Plan(variant=variant, nc=nc, no=no, na=na, channels=ch, tensors=b.tensors, ops=b.ops,
                head_tensors=tuple(heads), input_tensor=t_in)


def num_candidates(h: int, w: int, na: int = 3) -> int:
    return na * sum((h // s) * (w // s) for s in STRIDES)
