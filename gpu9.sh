set -e
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -k "test_conv_matches_reference" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1
timeout -k 10 300 python tools/sweep_conv_cfg.py --top 5 --ops model.6.m.0.cv2 model.8.m.0.cv2 model.7 model.9.cv2 model.24.m.0
