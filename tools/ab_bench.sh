#!/usr/bin/env bash
# Same-box A/B of library builds / environment switches through bench.py (interleaved repetitions):
#   tools/ab_bench.sh OUTDIR REPS "NAME1:ENV1=V ENV2=V" "NAME2:..." ...
# prints value, ms_per_step, roofline frac, step_ms_by_kind and the fused-Bottleneck time of every run; per-op lines go to OUTDIR/ops_NAME_REP.txt
out=$1; reps=$2; shift 2
mkdir -p "$out"
for rep in $(seq 1 "$reps"); do
    for spec in "$@"; do
        name=${spec%%:*}; envs=${spec#*:}
        env $envs AQ_BENCH_OPS=1 timeout -k 10 300 python bench.py --steps 60 --no-cpu-baseline --parity-steps 0 --e2e-images 0 ${AB_ARGS:-} 2> "$out/ops_${name}_${rep}.txt" > "$out/line_${name}_${rep}.json"
        python - "$name" "$rep" "$out/line_${name}_${rep}.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3])); r = d["roofline"]
    print(f"{sys.argv[1]:14s} rep {sys.argv[2]}: {d['value']:9.1f} tiles/s {d['ms_per_step']:.3f} ms  3x3 frac {r['frac']:.4f} all_3x3 {r['all_3x3']['frac']:.4f} {r['step_ms_by_kind']} btl {r['fused_bottleneck']['ms_per_step']}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "failed:", e)
PY
    done
done
