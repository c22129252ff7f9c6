#!/usr/bin/env python3
"""Drop-in for the reference's ``python3 yolov5/detect.py ...`` entry point (reference README.md:77).

Same path, same flags, same ``runs/detect/exp*/labels/*.txt`` output; the work happens in the MI355X HIP
engine (aquaculture_amd).  Multi-GPU: ``python -m torch.distributed.run --nproc-per-node 8 yolov5/detect.py ...``.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# The CPU side of the sweep is decode worker processes, label formatting in C and a little numpy: nothing wants an OpenMP / BLAS team.
# Left at their defaults (one thread per logical CPU: 256 on an MI355X host) the pools spin after every small operation -- the main
# process burnt 20 CPU-seconds in a 5-second sweep, more than all fourteen decode workers together (tools/e2e_cpu_time.py).  The
# libraries read these variables when they are loaded, so they are set here, before numpy and torch come in.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, os.environ.get("AQ_CPU_THREADS", "1"))

from aquaculture_amd.detect import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
