cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3 &&
export AQ_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tc3.json
for S in 1 2 1 2; do python bench.py --no-cpu-baseline --streams $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=$S', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['step_ms_by_kind'])"; done
