cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['step_ms_by_kind'], d['cpu_baseline']['value'])"
