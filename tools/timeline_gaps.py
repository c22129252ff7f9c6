#!/usr/bin/env python3
"""GPU timeline of a rocprofv3 --kernel-trace run: how much of the wall time between the first and last kernel of the busiest window has
no kernel running, one, or two and more (two batches in flight); per-stream gap between consecutive kernels.
usage: python tools/timeline_gaps.py <..._kernel_trace.csv> [--last-ms 80]"""
import argparse
import csv
import sys

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--last-ms", type=float, default=80.0, help="analyse this much time before the last conv kernel ends (the timed region of bench.py)")
ap.add_argument("--skip-tail-ms", type=float, default=0.0)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id") or "0"))
rows.sort()
t_end = max(e for _, e, _, _ in rows) - int(a.skip_tail_ms * 1e6)
t0 = t_end - int(a.last_ms * 1e6)
win = [(max(s, t0), min(e, t_end), n, q) for s, e, n, q in rows if e > t0 and s < t_end]
ev = []
for s, e, _, _ in win:
    ev += [(s, 1), (e, -1)]
ev.sort()
lvl, last, acc = 0, t0, {}
for t, d in ev:
    acc[min(lvl, 2)] = acc.get(min(lvl, 2), 0) + (t - last)
    lvl += d
    last = t
acc[min(lvl, 2)] = acc.get(min(lvl, 2), 0) + (t_end - last)
tot = sum(acc.values())
print(f"window {tot / 1e6:.2f} ms, {len(win)} kernels: idle {acc.get(0, 0) / tot:.1%}, one kernel {acc.get(1, 0) / tot:.1%}, two or more {acc.get(2, 0) / tot:.1%}; "
      f"sum of kernel durations {sum(e - s for s, e, _, _ in win) / 1e6:.2f} ms")
byq = {}
for s, e, n, q in win:
    byq.setdefault(q, []).append((s, e, n))
for q, ks in sorted(byq.items()):
    ks.sort()
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    if gaps:
        gaps_s = sorted(gaps)
        print(f"queue {q}: {len(ks)} kernels, busy {sum(e - s for s, e, _ in ks) / 1e6:.2f} ms, gap between consecutive kernels median {gaps_s[len(gaps_s) // 2] / 1e3:.2f} us, "
              f"mean {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 {gaps_s[int(0.9 * len(gaps_s))] / 1e3:.2f} us, total {sum(max(g, 0) for g in gaps) / 1e6:.2f} ms")
