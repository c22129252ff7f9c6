"""The N>1 path on CPU: world_size 2 over gloo.  Shard -> per-rank detections -> one gather -> same table as 1 rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aquaculture_amd import dist as aqdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_dets(tile):
    rng = np.random.default_rng(tile)
    n = int(rng.integers(0, 6)) if tile % 3 else 0          # some tiles have no detections at all
    return torch.from_numpy(rng.uniform(0, 640, (n, 6)).astype(np.float32))


def _worker(rank, world, port, n_tiles, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = aqdist.init("gloo")
    assert (r, w) == (rank, world)
    rows, seen = [], 0
    for t in aqdist.shard_indices(n_tiles, rank, world):
        d = _fake_dets(t)
        seen += 1
        rows.append(aqdist.pack_rows(torch.full((d.shape[0],), t), d))
    local = torch.cat(rows, 0) if rows else torch.zeros((0, aqdist.ROW))
    allrows = aqdist.gather_rows(local)
    tiles, labels, dets, elapsed = aqdist.reduce_counters(seen, seen // 2, local.shape[0], 1.0 + rank, torch.device("cpu"))
    aqdist.barrier()
    if rank == 0:
        torch.save({"rows": allrows, "tiles": tiles, "dets": dets, "elapsed": elapsed}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,n_tiles", [(2, 23), (4, 23), (4, 3), (3, 1)])     # incl. ranks without tiles / without detections
def test_gather_equals_single_process(tmp_path, world, n_tiles):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), n_tiles, out), nprocs=world, join=True)
    got = torch.load(out)
    want = torch.cat([aqdist.pack_rows(torch.full((_fake_dets(t).shape[0],), t), _fake_dets(t)) for t in range(n_tiles)], 0)
    assert got["tiles"] == n_tiles and got["dets"] == want.shape[0] and got["elapsed"] == float(world)
    key = lambda r: r[torch.argsort(r[:, 0] * 1e6 + r[:, 6] * 1e3 + r[:, 2], stable=True)]
    assert torch.equal(key(got["rows"]), key(want))
    assert sorted(set(got["rows"][:, 0].int().tolist())) == sorted(t for t in range(n_tiles) if _fake_dets(t).shape[0])


def test_shard_is_a_partition():
    for n, w in ((10, 1), (10, 3), (7, 8), (1000003, 8)):
        parts = [aqdist.shard_indices(n, r, w) for r in range(w)]
        assert sum(len(p) for p in parts) == n
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert aqdist.env_rank_world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0)))
    assert aqdist.gather_rows(torch.ones(3, aqdist.ROW)).shape == (3, aqdist.ROW)     # no process group: identity
