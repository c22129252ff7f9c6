"""The N>1 path on CPU over gloo (world sizes 2 to 8).  Shard -> per-rank detections -> bounded gather to rank 0 -> same table as 1 rank."""
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


def _fake_dets(tile, skew=False):
    rng = np.random.default_rng(tile)
    n = int(rng.integers(0, 6)) if tile % 3 else 0          # some tiles have no detections at all
    if skew and tile % 37 == 5:
        n = 3000 + tile                                     # a cage-dense coastal tile: thousands of rows on one rank
    return torch.from_numpy(rng.uniform(0, 640, (n, 6)).astype(np.float32))


def _worker(rank, world, port, n_tiles, out, flush_every, skew):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = aqdist.init("gloo")
    assert (r, w) == (rank, world)
    g = aqdist.DetectionGather(torch.device("cpu"))
    seen = ndet = 0
    for k, t in enumerate(aqdist.shard_indices(n_tiles, rank, world)):
        d = _fake_dets(t, skew)
        seen += 1
        ndet += d.shape[0]
        g.add(torch.full((d.shape[0],), t, dtype=torch.int32), aqdist.pack_rows(d))
        if (k + 1) % flush_every == 0:
            g.flush(more=True)
    g.finish()
    tiles, labels, dets, elapsed = aqdist.reduce_counters(seen, seen // 2, ndet, 1.0 + rank, torch.device("cpu"))
    aqdist.barrier()
    if rank == 0:
        idx, rows = g.table()
        torch.save({"idx": idx, "rows": rows, "tiles": tiles, "dets": dets, "elapsed": elapsed, "flushes": g.flushes, "peak": g.max_rows_per_flush}, out)
    else:
        assert g.total == 0                 # nobody but rank 0 receives rows
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,n_tiles,flush_every,skew", [
    (2, 23, 4, False), (4, 23, 2, False), (4, 3, 16, False), (3, 1, 1, False),   # incl. ranks without tiles / without detections / never a periodic flush
    (8, 203, 3, True),                                                          # 8 ranks, skewed: a few tiles carry thousands of detections
])
def test_gather_equals_single_process(tmp_path, world, n_tiles, flush_every, skew):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), n_tiles, out, flush_every, skew), nprocs=world, join=True)
    got = torch.load(out)
    want_i = torch.cat([torch.full((_fake_dets(t, skew).shape[0],), t, dtype=torch.int32) for t in range(n_tiles)])
    want_r = torch.cat([aqdist.pack_rows(_fake_dets(t, skew)) for t in range(n_tiles)], 0)
    assert got["tiles"] == n_tiles and got["dets"] == want_r.shape[0] and got["elapsed"] == float(world)
    assert got["idx"].dtype == torch.int32 and got["rows"].shape == want_r.shape
    key = lambda i, r: r[torch.argsort(i.double() * 1e6 + r[:, 5].double() * 1e3 + r[:, 1].double(), stable=True)]
    assert torch.equal(key(got["idx"], got["rows"]), key(want_i, want_r))
    assert sorted(set(got["idx"].tolist())) == sorted(t for t in range(n_tiles) if _fake_dets(t, skew).shape[0])
    # bounded: no flush carried more than the rows one rank produced between two of its flushes
    per_flush = max(sum(_fake_dets(t, skew).shape[0] for t in list(aqdist.shard_indices(n_tiles, r, world))[k:k + flush_every])
                    for r in range(world) for k in range(0, max(1, len(aqdist.shard_indices(n_tiles, r, world))), flush_every))
    assert got["peak"] <= per_flush


def _failing_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    aqdist.init("gloo")
    g = aqdist.DetectionGather(torch.device("cpu"))
    try:
        if rank == 1:
            g.finish(failed=True)           # what detect.py does when its writer thread raised
        else:
            for k in range(40):
                g.add(torch.zeros(2, dtype=torch.int32), torch.ones(2, aqdist.ROW))
                g.flush(more=True)
            g.finish()
        res = "finished"
    except aqdist.RankFailed as e:
        res = f"aborted after {g.flushes} flush(es): {e}"
    with open(f"{out}.{rank}", "w") as f:
        f.write(res)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_failed_rank_aborts_the_others(tmp_path):
    """A rank that failed reports it through the gather; the others stop at their next flush instead of hanging in a collective."""
    out = str(tmp_path / "res")
    mp.spawn(_failing_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    res = [open(f"{out}.{r}").read() for r in range(3)]
    assert all(r.startswith("aborted after 1 flush") for r in res), res


def _main_loop_failure_worker(rank, world, port, out):
    """The tail of detect.run: the failing rank's error comes from its MAIN loop, is recorded, reported through the gather
    inside try/except RankFailed, and then re-raised as itself; the others see RankFailed at their next flush and their own tail's
    finish() starts no further collective (the failing rank has left by then)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    aqdist.init("gloo")
    g = aqdist.DetectionGather(torch.device("cpu"))
    err = []
    try:
        for k in range(40):
            if rank == 1 and k == 5:
                raise ValueError("engine error in the main loop")
            g.add(torch.zeros(2, dtype=torch.int32), torch.ones(2, aqdist.ROW))
            if (k + 1) % 4 == 0:
                g.flush(more=True)
    except BaseException as e:
        err.append(e)
    res = "finished"
    try:
        try:
            g.finish(failed=bool(err))
        except aqdist.RankFailed:
            if not err:
                raise
        if err:
            raise err[0]
    except BaseException as e:
        res = f"{type(e).__name__}: {e} [flushes {g.flushes}, aborted {g.aborted}]"
    with open(f"{out}.{rank}", "w") as f:
        f.write(res)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_main_loop_failure_keeps_its_own_exception(tmp_path):
    out = str(tmp_path / "res")
    mp.spawn(_main_loop_failure_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    res = [open(f"{out}.{r}").read() for r in range(3)]
    assert res[1].startswith("ValueError: engine error in the main loop") and "aborted True" in res[1], res
    for r in (0, 2):                                  # stopped at the flush the failing rank answered (its 2nd: one periodic flush, then finish)
        assert res[r].startswith("RankFailed") and "flushes 2" in res[r], res


def test_shard_is_a_partition():
    for n, w in ((10, 1), (10, 3), (7, 8), (1000003, 8)):
        parts = [aqdist.shard_indices(n, r, w) for r in range(w)]
        assert sum(len(p) for p in parts) == n
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert aqdist.env_rank_world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0)))
    g = aqdist.DetectionGather()                                   # no process group: everything stays local
    g.add(torch.arange(3, dtype=torch.int32), torch.ones(3, aqdist.ROW))
    g.finish()
    assert g.total == 3 and g.table()[1].shape == (3, aqdist.ROW)


def _rank0_worker(rank, world, port, fail, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    aqdist.init("gloo")
    calls = []

    def fn():
        calls.append(rank)
        if fail:
            raise ValueError("directory was written with other settings")
        return {"model.6.m.0.cv2": 0.125, "n": 3}
    try:
        got = aqdist.on_rank0(fn)
        outcome = ("ok", got)
    except Exception as e:
        outcome = (type(e).__name__, str(e))
    assert calls == ([0] if rank == 0 else [])          # fn runs on rank 0 only
    aqdist.barrier()                                     # every rank is still in step afterwards (nobody is stuck in a collective)
    torch.save(outcome, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("fail", [False, True])
def test_on_rank0_hands_the_result_or_the_failure_to_every_rank(tmp_path, fail):
    """rank 0's run-directory check could raise before a barrier the other ranks were already waiting in.  on_rank0
    broadcasts the outcome: a result reaches every rank, a failure raises on every rank (the original exception on rank 0)."""
    world = 3
    mp.spawn(_rank0_worker, args=(world, _free_port(), fail, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(str(tmp_path / f"r{r}.pt")) for r in range(world)]
    if not fail:
        assert all(o == ("ok", {"model.6.m.0.cv2": 0.125, "n": 3}) for o in outs)
    else:
        assert outs[0] == ("ValueError", "directory was written with other settings")
        assert all(o[0] == "RuntimeError" and "rank 0 refused the run: ValueError: directory was written" in o[1] for o in outs[1:])


def test_on_rank0_without_a_process_group_is_a_plain_call():
    assert aqdist.on_rank0(lambda: 7) == 7
    with pytest.raises(KeyError):
        aqdist.on_rank0(lambda: {}["x"])
