set -e
D=/tmp/aq_e2e
python tools/bench_e2e.py --size 1024 --n 4096 --workers 14 --precision bf16 --jpeg-decode split --noise 10 > /dev/null 2>&1 || true
python -m cProfile -o gpurun_out/e2e.prof yolov5/detect.py --weights $D/synth.pt --source $D/jpegs_1024_n10 --nosave --save-txt --save-conf --project $D/runs --name prof --batch-size 64 --workers 14 --precision bf16 --quiet --jpeg-decode split > gpurun_out/e2e_prof.log 2>&1
python - <<'PY'
import pstats
p = pstats.Stats("gpurun_out/e2e.prof")
p.sort_stats("tottime").print_stats(22)
PY
