#!/usr/bin/env bash
# Re-collects every digest-tagged file under profiles/ for the current build (run ON the GPU box, from the repository root, after
# `python __graft_entry__.py`): the rocprofv3 trace / PMC passes, the end-to-end runs the bench line quotes as `e2e`, and the bench lines of
# every BASELINE configuration.  Writes under gpurun_out/collect/; copy what should be judged into profiles/ (names in profiles/README.md).
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
set -u
OUT=gpurun_out/collect
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 1000 python tools/profile_pmc.py --out "$OUT/pmc" > "$OUT/pmc.log" 2>&1 || echo "profile_pmc failed (see $OUT/pmc.log)"
cp "$OUT/pmc/hbm_traffic.json" profiles/hbm_traffic_latest.json 2>/dev/null      # so that the bench lines below carry `traffic`
if [ -z "${SKIP_E2E:-}" ]; then
rm -f "$OUT/e2e_latest.json"
for sz in 640 1024; do
    for jd in split gpu host; do
        n=$([ "$sz" = 640 ] && echo 24576 || echo 16384)
        timeout -k 10 300 python tools/bench_e2e.py --size "$sz" --n "$n" --workers 14 --precision bf16 --jpeg-decode "$jd" --json "$OUT/e2e_latest.json" 2>&1 | grep -E "steady"
    done
done
for jd in split gpu host; do
    timeout -k 10 300 python tools/bench_e2e.py --size 1024 --n 16384 --workers 14 --precision bf16 --jpeg-decode "$jd" --noise 10 --json "$OUT/e2e_latest.json" 2>&1 | grep -E "steady"
done
timeout -k 10 300 python tools/bench_e2e.py --scenes 200 --workers 14 --precision bf16 --json "$OUT/e2e_latest.json" 2>&1 | grep -E "steady"
fi      # (SKIP_E2E=1: keep the e2e file of an earlier call of the same build)
cp "$OUT/e2e_latest.json" profiles/e2e_latest.json 2>/dev/null                   # ... and `e2e`
timeout -k 10 300 python bench.py > "$OUT/bench_default.json" 2>/dev/null
AQ_PL_PM=0 timeout -k 10 300 python bench.py --no-cpu-baseline --parity-steps 0 > "$OUT/bench_default_slot_major.json" 2>/dev/null
timeout -k 10 300 python bench.py --precision fp8 --batch 128 --no-cpu-baseline --parity-steps 0 > "$OUT/bench_yolov5m_fp8_b128.json" 2>/dev/null
timeout -k 10 300 python bench.py --batch 128 --no-cpu-baseline --parity-steps 0 > "$OUT/bench_yolov5m_bf16_b128.json" 2>/dev/null
timeout -k 10 300 python bench.py --precision fp8 --no-cpu-baseline --parity-steps 0 > "$OUT/bench_yolov5m_fp8_b64.json" 2>/dev/null
timeout -k 10 300 python bench.py --precision f16x3 --steps 30 --no-cpu-baseline --parity-steps 0 > "$OUT/bench_yolov5m_f16x3.json" 2>/dev/null
timeout -k 10 300 python bench.py --variant yolov5x --size 1280 --batch 16 --steps 30 --no-cpu-baseline --parity-steps 0 > "$OUT/bench_yolov5x_1280_b16.json" 2>/dev/null
for f in "$OUT"/bench_*.json; do
    python -c "
import json, sys
d = json.load(open('$f')); r = d['roofline']
print('$f'.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], 'traffic' if r['traffic'] else 'no traffic', 'e2e' if 'e2e' in d else 'no e2e')"
done
