#!/usr/bin/env python3
"""Throughput of the geocoding batch op (aquaculture_amd/geocode.py).  CPU only; prints one JSON line.  The comparison with the
per-detection loop of reference src/process_yolo/geocode_results.py:158-192 (the oracle's form) lives in
tests/test_geocode.py::test_batch_throughput_vs_scalar_loop -- nothing outside tests/ may touch oracle/.
Sample: N synthetic label rows spread over the tile names of tests/golden/g7_geocode.json (the reference's own scene table)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aquaculture_amd import geocode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    a = ap.parse_args()
    with open(os.path.join(ROOT, "tests", "golden", "g7_geocode.json")) as f:
        g = json.load(f)
    bboxes = {int(k): tuple(v) for k, v in g["wanted_bboxes"].items()}
    names = sorted({ft["image"][:-5] for ft in g["features"]})
    rng = np.random.default_rng(0)
    per = 340                                                # detections per tile of the synthetic checkpoint
    n_img = a.rows // per
    stems = [names[i % len(names)] for i in range(n_img)]
    counts = np.full(n_img, per)
    rows = np.zeros((n_img * per, 6))
    rows[:, 0] = rng.integers(0, 5, rows.shape[0]); rows[:, 1:3] = rng.uniform(0, 1, (rows.shape[0], 2))
    rows[:, 3:5] = rng.uniform(0.001, 0.3, (rows.shape[0], 2)); rows[:, 5] = rng.uniform(0.25, 1, rows.shape[0])
    t0 = time.perf_counter()
    t = geocode.geocode_detections(stems, counts, rows, bboxes)
    dt = time.perf_counter() - t0
    print(json.dumps({"op": "geocode_detections", "rows": int(rows.shape[0]), "seconds": round(dt, 4), "rows_per_s": round(rows.shape[0] / dt),
                      "checksum_xmin": int(t["xmin"].sum())}))


if __name__ == "__main__":
    main()
