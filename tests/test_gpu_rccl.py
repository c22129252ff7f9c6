"""RCCL on the hardware a one-GPU box has: the path's collectives (detection gather to rank 0, counters, barriers) run
through the "nccl" backend in a world of ONE rank (AQ_DIST_FORCE=1) -- communicator set-up, device tensors through all_gather / gather /
all_reduce / barrier, the CLI and bench.py end to end.  What it cannot show is a second rank; the gloo tests (tests/test_dist_gloo.py,
world 2-8) cover the protocol, tests/test_bench_launcher.py::test_bench_gpus_2_as_typed_nccl the real thing where two GPUs exist."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

GATHER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from aquaculture_amd import dist as aqdist
rank, world, local = aqdist.init("nccl")
assert (rank, world) == (0, 1) and aqdist.active() and torch.distributed.get_backend() == "nccl"
torch.cuda.set_device(0)
g = aqdist.DetectionGather("cuda:0", keep=True)
assert g.on and g.device.type == "cuda"
gen = torch.Generator().manual_seed(3)
sent_i, sent_r = [], []
for k in range(5):
    n = [7, 0, 300, 1, 64][k]
    idx = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, generator=gen)
    rows = torch.rand((n, 6), generator=gen)
    g.add(idx.cuda(), rows.cuda())
    sent_i.append(idx); sent_r.append(rows)
    if k %% 2 == 1:
        assert g.flush(more=True) is True
g.finish()
ti, tr = g.table()
assert torch.equal(ti, torch.cat(sent_i)) and torch.equal(tr, torch.cat(sent_r)), "rows changed on their way through RCCL"
assert g.total == 372 and g.flushes == 3
assert aqdist.reduce_counters(5, 4, 372, 1.5, torch.device("cuda", 0)) == (5, 4, 372, 1.5)
aqdist.barrier()
torch.distributed.destroy_process_group()
print("RCCL_OK")
"""


def _env(port):
    return dict(os.environ, AQ_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_detection_gather_over_rccl_in_a_world_of_one():
    r = subprocess.run([sys.executable, "-c", GATHER % ROOT], env=_env(29631), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_line_with_the_collectives_on_rccl(lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "8", "--size", "128",
                        "--no-cpu-baseline", "--parity-steps", "0", "--no-autotune"], env=_env(29632), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collectives"] == "nccl" and line["value"] > 0
