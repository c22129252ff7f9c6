#!/usr/bin/env python3
"""CPU experiment: does the fp8w mode's power-of-two per-channel weight scale cost accuracy against a real-valued
scale (max|w| / 448)?  The oracle (bf16 activations, fp32 accumulate) with each weight quantiser, against the fp32 oracle, on synthetic
golden tiles: |d conf|, |d box|, post-NMS count differences.  Usage: python tools/fp8_scale_experiment.py [--tiles 4] [--size 640]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaculture_amd import checkpoint, tiles
from oracle import yolov5_oracle as O


def wq_real(w):
    codes, s = O.wq_fp8_real(w)
    return codes * s.view(-1, *([1] * (w.dim() - 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=4)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ck = checkpoint.synthetic_checkpoint("yolov5m", 5)
    x = tiles.synthetic_batch(range(a.tiles), a.size)
    models = {"fp32": O.model_from_checkpoint(ck), "bf16": O.model_from_checkpoint(ck, O.q_bf16),
              "fp8w pow2 scale": O.model_from_checkpoint(ck, O.q_bf16, O.wq_fp8_e4m3), "fp8w real scale": O.model_from_checkpoint(ck, O.q_bf16, wq_real)}
    preds = {k: torch.cat([m.forward(O.preprocess(x[i:i + 1])) for i in range(a.tiles)], 0) for k, m in models.items()}
    counts = {k: [r.shape[0] for r in O.non_max_suppression(p.numpy())] for k, p in preds.items()}
    ref = preds["fp32"]
    res = {"tiles": a.tiles, "size": a.size, "counts": counts}
    for k in list(models)[1:]:
        dc = (preds[k][..., 4:] - ref[..., 4:]).abs().flatten()
        db = (preds[k][..., :4] - ref[..., :4]).abs().flatten()
        res[k + " vs fp32"] = {"dconf_mean": float(dc.mean()), "dconf_max": float(dc.max()), "dbox_mean_px": float(db.mean()),
                               "count_diff_sum": sum(abs(p - q) for p, q in zip(counts[k], counts["fp32"])), "boxes_total": sum(counts["fp32"])}
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
