cd $GRAFT_REPO_ROOT
for B in 8 16 32 64; do
AQ_BENCH_OPS=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 --batch $B 2>gpurun_out/ops_b$B.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', d['value'], 'tiles/s', d['ms_per_step'], 'ms/step')"
done
