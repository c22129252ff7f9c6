"""aquaculture_amd: MI355X-native YOLOv5 tile-sweep inference path.

Drop-in for the one compute-heavy step of reglab/aquaculture: the
``yolov5/detect.py --weights W --source DIR --nosave --save-txt --save-conf``
sweep (reference README.md:77).  Host code is Python; every arithmetic step of
the path (preprocess, convolutions, SPPF pools, upsample, detect decode, NMS)
runs in hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/aq_engine.h``.  There is no CPU or PyTorch-eager fallback: if the HIP
library is missing the engine raises.
"""

__version__ = "0.1.0"
