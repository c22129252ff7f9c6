cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "direct" 2>&1 | tail -4 &&
AQ_BENCH_OPS=1 python bench.py --no-cpu-baseline > gpurun_out/b_fused.log 2>&1 ; grep -E "\"metric\"" gpurun_out/b_fused.log | cut -c1-200; grep "^# op" gpurun_out/b_fused.log | grep -E "cfg 1000" | head -12
