cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3 &&
AQ_BENCH_OPS=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 2>gpurun_out/bench_ops.txt | cut -c1-160
grep "^# op" gpurun_out/bench_ops.txt | awk '{printf "%s %s %s %s %s | ", $3,$5,$6,$7,$9} NR%3==0 {print ""}' | grep -E "cv2 |model.(1|3|5|7|18|21) "
python -c "
import json,sys
" ; python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['step_ms_by_kind'])"
