#!/usr/bin/env python3
"""Drop-in for the reference's ``python3 yolov5/detect.py ...`` entry point (reference README.md:77).

Same path, same flags, same ``runs/detect/exp*/labels/*.txt`` output; the work happens in the MI355X HIP
engine (aquaculture_amd).  Multi-GPU: ``python -m torch.distributed.run --nproc-per-node 8 yolov5/detect.py ...``.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from aquaculture_amd.detect import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
