#!/usr/bin/env python3
"""CPU time of an end-to-end sweep, split into the main process and the decode workers (getrusage), beside the images/s the CLI reports.
Run tools/bench_e2e.py --size 1024 first (it writes the jpegs and the checkpoint under /tmp/aq_e2e).  KEY=VALUE arguments are added to the
environment (OMP_NUM_THREADS=1, AQ_TRACE_LOADER=1 ...); AQ_E2E_EXTRA="--conf-thres 0.6" adds CLI flags."""
import os, resource, subprocess, sys, time
D = "/tmp/aq_e2e"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ)
for k, v in [a.split("=", 1) for a in sys.argv[1:] if "=" in a]:
    env[k] = v
cmd = [sys.executable, "-c", f"""
import resource, runpy, sys, os, time
sys.argv = ['detect.py', '--weights', '{D}/synth.pt', '--source', '{D}/jpegs_1024', '--nosave', '--save-txt', '--save-conf', '--project', '{D}/runs', '--name', 'cpu',
            '--batch-size', '64', '--workers', '14', '--precision', 'bf16', '--quiet', '--jpeg-decode', 'split'] + os.environ.get('AQ_E2E_EXTRA', '').split()
t0 = time.perf_counter()
try:
    runpy.run_path('{ROOT}/yolov5/detect.py', run_name='__main__')
except SystemExit:
    pass
dt = time.perf_counter() - t0
s, c = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
print(f'CPU main process: user {{s.ru_utime:.1f}} s sys {{s.ru_stime:.1f}} s; decode workers (children): user {{c.ru_utime:.1f}} sys {{c.ru_stime:.1f}}; wall {{dt:.1f}} s')
"""]
r = subprocess.run(cmd, env=env, capture_output=True, text=True)
ls = r.stdout.splitlines()
print("\n".join(l for l in ls if "images/s" in l or "CPU main" in l or "steady" in l))
tr = [l for l in ls if l.startswith("loader batch")]
print("\n".join(tr[20:36]))
if r.returncode:
    print(r.stderr[-1500:])
