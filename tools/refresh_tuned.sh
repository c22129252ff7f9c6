#!/usr/bin/env bash
# Re-times the tile configurations of every geometry the shipped table covers (BASELINE.json's configurations and the batch sizes a sweep's
# ragged tail produces) and writes a fresh table (run ON the GPU box after `python __graft_entry__.py`; needed whenever a kernel family
# joins the tuner's candidates -- the table key carries aq_version()):
#   gpurun --timeout 1200 -- 'bash tools/refresh_tuned.sh gpurun_out/tuned_tables.json'   then copy it to aquaculture_amd/data/tuned_tables.json
set -u
out=${1:-gpurun_out/tuned_tables.json}
mkdir -p "$(dirname "$out")"
echo '{}' > "$out"
common="--retune --write-tuned $out --no-cpu-baseline --parity-steps 0 --e2e-images 0 --steps 20 --warmup 3"
for args in "" "--batch 128" "--batch 32" "--batch 16" "--precision fp8" "--precision fp8 --batch 128" "--precision f16x3 --steps 6" \
            "--variant yolov5x --size 1280 --batch 16 --steps 6"; do
    timeout -k 10 600 python bench.py $common $args > /dev/null 2> "$out.log" || { echo "bench.py $args failed"; tail -5 "$out.log"; exit 1; }
    echo "tuned: ${args:-default}"
done
python - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, "asm1x1 layers:", sum(1 for c in v if c == 1004), "planar:", sum(1 for c in v if c in (1002, 1003)))
PY
