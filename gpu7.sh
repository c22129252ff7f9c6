cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 &&
AQ_BENCH_OPS=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/bench_ops.txt | cut -c1-200
grep "^# op" gpurun_out/bench_ops.txt | awk '{printf "%s %s %s %s %s | ", $3,$5,$6,$7,$9} NR%3==0 {print ""}'
