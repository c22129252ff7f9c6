cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "ragged or rejects" 2>&1 | tail -12
