"""Generate the committed golden fixtures under tests/golden/ (SURVEY.md 8c G1-G6).

The reference holds no tests, golden vectors or runnable implementation for the detect path
(/root/reference/yolov5/yolov5 is empty; weights are a missing blob), so these vectors are produced by the
CPU oracle (oracle/yolov5_oracle.py) in this container and pin it against regressions; the oracle itself is
"parity unpinned" (see its header).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from aquaculture_amd import checkpoint, spec, tiles  # noqa: E402
from oracle import yolov5_oracle as O  # noqa: E402

torch.set_num_threads(8)
ck = checkpoint.synthetic_checkpoint("yolov5m", 5)
model = O.model_from_checkpoint(ck)

# G1: identity of the synthetic checkpoint (seed + digest of the fused, packed weights)
plan = spec.build_plan("yolov5m", 5)
h = hashlib.sha256()
for pw in checkpoint.pack_plan_weights(ck, plan):
    h.update(pw.weight.tobytes())
    h.update(pw.bias.tobytes())
g1 = {"variant": "yolov5m", "nc": 5, "seed": checkpoint.SYNTH_SEED, "packed_sha256": h.hexdigest(),
      "n_conv_ops": len(plan.conv_ops()), "flops_640": plan.flops(640, 640)}

# G2: per-module outputs for tile 0 at 64x64 (fp32)
x64 = tiles.synthetic_batch([0], 64)
model.taps = {}
pred64 = model.forward(O.preprocess(x64))
g2 = {k.replace(".", "_"): v.numpy() for k, v in model.taps.items()}
g2["pred"] = pred64.numpy()
model.taps = None
np.savez_compressed(os.path.join(HERE, "g2_modules_64.npz"), **g2)

# G3: final detections + label text for the 16 config-1 tiles at 640x640
x640 = tiles.synthetic_batch(range(16), 640)
dets = O.detect_tiles(model, x640, batch=4)
g3 = {f"det_{i}": d for i, d in enumerate(dets)}
labels = {tiles.tile_name(i): "\n".join(O.label_lines(d, (640, 640), (640, 640))) for i, d in enumerate(dets)}
np.savez_compressed(os.path.join(HERE, "g3_detections_640.npz"), **g3)
with open(os.path.join(HERE, "g3_labels_640.json"), "w") as f:
    json.dump(labels, f)

# G4: NMS unit cases (input pred rows + expected output)
rng = np.random.default_rng(42)
NC = 5


def rand_pred(n, spread=600.0):
    p = np.zeros((1, n, 5 + NC), np.float32)
    p[..., 0:2] = rng.uniform(20, spread, (1, n, 2))
    p[..., 2:4] = rng.uniform(4, 80, (1, n, 2))
    p[..., 4] = rng.uniform(0.3, 1, (1, n))
    p[..., 5:] = rng.uniform(0.3, 1, (1, n, NC))
    return p.astype(np.float32)


cases = {}
p = rand_pred(400); cases["dense400"] = (p, 0.25, 0.45, 1000)
p = rand_pred(50); p[..., 4] = 0.1; cases["none_pass"] = (p, 0.25, 0.45, 1000)
p = rand_pred(8); p[0, :, :4] = [[100 + 50 * i, 100, 20, 20] for i in range(8)]; p[..., 4] = 0.9
p[0, :, 5:] = [0.1, 0.8, 0.8, 0.2, 0.1]; cases["ties"] = (p, 0.25, 0.45, 1000)
p = rand_pred(2); p[0, :, :4] = [[100, 100, 40, 40], [100, 110, 40, 20]]; p[0, :, 4] = [0.9, 0.8]
p[0, :, 5:] = [0.9, 0, 0, 0, 0]; cases["iou_exactly_half_thr050"] = (p, 0.25, 0.5, 1000); cases["iou_exactly_half_thr045"] = (p.copy(), 0.25, 0.45, 1000)
p = rand_pred(3); p[0, :, :4] = [[200, 200, 50, 50], [200, 200, 50, 50], [201, 200, 50, 50]]; p[..., 4] = 0.9
p[0, :, 5:] = [[0.9, 0, 0, 0, 0], [0, 0.9, 0, 0, 0], [0.8, 0, 0, 0, 0]]; cases["cross_class"] = (p, 0.25, 0.45, 1000)
p = rand_pred(600, 3000.0); p[..., 2:4] = 3.0; p[..., 5] = 0.99; cases["max_det_100"] = (p, 0.25, 0.45, 100)
g4 = {}
for name, (p, ct, it, md) in cases.items():
    out = O.non_max_suppression(p, ct, it, md)[0]
    g4[name + "__pred"] = p
    g4[name + "__args"] = np.array([ct, it, md], np.float64)
    g4[name + "__out"] = out
np.savez_compressed(os.path.join(HERE, "g4_nms_cases.npz"), **g4)

# G5: %g formatting of fp32 values (exponent forms, integers, 6 significant digits)
vals = [0.0, 1.0, 0.5, 0.25165, 1e-5, 9.99999e-5, 0.000123456, 123456.0, 1234567.0, 0.1, 1 / 3, 2 / 3, 0.0257812, 639 / 640,
        1 / 1280, 1023.5 / 1024, 4.0, 0.999999, 0.9999999]
g5 = [{"value_f32_bits": int(np.float32(v).view(np.uint32)), "text": "%g" % float(np.float32(v))} for v in vals]

# G6: scale_boxes / clip / round-half-even / xyxy2xywh rows for 640 and 1024 originals
g6 = []
for (img1, img0) in (((640, 640), (640, 640)), ((640, 640), (1024, 1024)), ((480, 640), (500, 700))):
    d = np.array([[10.5, 20.5, 30.5, 41.5, 0.9, 1], [-3.2, 5.0, 700.0, 650.0, 0.8, 0], [100.49, 100.5, 101.5, 102.5, 0.7, 4],
                  [0.3125, 0.9375, 639.6875, 639.0625, 0.6, 2]], np.float32)
    g6.append({"img1": img1, "img0": img0, "det": d.tolist(), "lines": O.label_lines(d, img1, img0)})

with open(os.path.join(HERE, "g1_g5_g6.json"), "w") as f:
    json.dump({"g1": g1, "g5": g5, "g6": g6}, f, indent=1)
for fn in sorted(os.listdir(HERE)):
    print(f"{fn:32s} {os.path.getsize(os.path.join(HERE, fn)):9d} B")
