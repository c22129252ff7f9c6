import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from aquaculture_amd import checkpoint, tiles, engine
ck = checkpoint.synthetic_checkpoint("yolov5x", 5)
eng = engine.Engine(ck, "bf16")
x = torch.from_numpy(tiles.synthetic_batch(range(16), 1280)).cuda()
pred = eng.forward_raw(x).float()
obj = pred[..., 4]; best = (pred[..., 5:] * obj[..., None]).amax(-1)
n = ((obj > 0.25) & (best > 0.25)).sum(1)
print("candidates per tile:", n.tolist())
import time
for _ in range(3):
    d, c = engine.nms(pred.contiguous(), 5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    d, c = engine.nms(pred.contiguous(), 5)
torch.cuda.synchronize(); print("nms ms per call (16 tiles, full pred rows):", (time.perf_counter() - t0) / 10 * 1e3, "kept:", c.tolist())
