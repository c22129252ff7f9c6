cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tee gpurun_out/bench_default.json | cut -c1-1500
