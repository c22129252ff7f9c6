"""Tile-level data parallelism: one process per GPU, strided shard, one gather at the end.

The reference runs detect.py as one process on one device (reference README.md:77); tiles are independent
(no halo: reference src/load_data/tile_tifs.py:33-47; one label file per tile), so the path shards with NO
data-path collective.  The only exchange is the final detection gather: counts, then
padded rows to rank 0, then counters -- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

ROW = 6   # float32 columns of a gathered detection: cls, x1, y1, x2, y2, conf (the tile index travels beside them as int32)


class RankFailed(RuntimeError):
    """Another rank of the run reported an error through the gather: abort instead of waiting for a collective that never comes."""


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun-style environment; (0, 1, 0) when not launched by it."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def forced() -> bool:
    """AQ_DIST_FORCE=1: run the collectives even in a world of one rank -- how a one-GPU box exercises the RCCL ("nccl") code path
    (communicator set-up, device tensors through all_gather / gather / all_reduce / barrier) that otherwise needs a second GPU."""
    return os.environ.get("AQ_DIST_FORCE") == "1"


def active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, world, local = env_rank_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("AQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n: int, rank: int, world: int) -> range:
    """Tiles i = rank (mod world) of the sorted list: balances cage-dense scenes that cluster in name order."""
    return range(rank, n, world)


def pack_rows(dets: torch.Tensor) -> torch.Tensor:
    """dets float32 [n,6] as NMS returns them (x1,y1,x2,y2,conf,cls) -> float32 [n,6] gather rows (cls,x1,y1,x2,y2,conf)."""
    if dets.shape[0] == 0:
        return dets.new_zeros((0, ROW))
    return torch.cat((dets[:, 5:6], dets[:, :4], dets[:, 4:5]), 1)


class DetectionGather:
    """The path's one collective, bounded: every rank adds (tile index int32, rows float32 [n,6]) as batches finish and
    calls ``flush`` every few batches; a flush is one small ``all_gather`` of (count, still-working, failed) per rank followed by one
    padded ``gather`` of the rows and one of the tile indices TO RANK 0 ONLY (RCCL over xGMI when the backend is nccl).  So neither the
    host lists nor the device padding grow with the sweep, nobody but rank 0 receives rows, and a tile index is an exact int32.

    Ranks need not have the same number of batches: ``finish`` keeps a finished rank answering flushes (with nothing) until every
    rank has reported that it is done.  A rank that hit an error calls ``finish(failed=True)``; the next flush raises RankFailed on
    all the others instead of leaving them in a collective until the backend's timeout.

    On rank 0 ``tile_index`` / ``rows`` hold what was gathered (kept only when ``keep`` is true) and ``total`` the row count."""

    def __init__(self, device=None, keep: bool = True):
        self.on = active()
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        gloo = self.on and dist.get_backend() == "gloo"
        self.device = torch.device("cpu") if (gloo or device is None) else torch.device(device)
        self.keep = keep
        self._idx: List[torch.Tensor] = []
        self._rows: List[torch.Tensor] = []
        self.tile_index: List[torch.Tensor] = []
        self.rows: List[torch.Tensor] = []
        self.total = 0
        self.flushes = 0
        self.max_rows_per_flush = 0
        self.aborted = False          # a flush has raised RankFailed: every rank saw the same flag in the same all_gather, none flushes again

    def add(self, tile_index: torch.Tensor, rows: torch.Tensor) -> None:
        if rows.shape[0]:
            self._idx.append(tile_index.to(torch.int32).reshape(-1))
            self._rows.append(rows.to(torch.float32).reshape(-1, ROW))

    def pending(self) -> int:
        return sum(int(r.shape[0]) for r in self._rows)

    def flush(self, more: bool = True, failed: bool = False) -> bool:
        """Collective.  Returns True while any rank still has batches to process."""
        idx = torch.cat(self._idx) if self._idx else torch.zeros((0,), dtype=torch.int32)
        rows = torch.cat(self._rows) if self._rows else torch.zeros((0, ROW), dtype=torch.float32)
        self._idx, self._rows = [], []
        if self.aborted:              # the run is over on every rank; a later flush / finish (an error path's tail) must not start a collective
            return False              # that the ranks which have already left would never join
        self.flushes += 1
        if not self.on:
            self._keep(idx, rows)
            return False
        idx, rows = idx.to(self.device), rows.to(self.device)
        me = torch.tensor([rows.shape[0], int(more), int(failed)], dtype=torch.int64, device=self.device)
        every = [torch.zeros_like(me) for _ in range(self.world)]
        dist.all_gather(every, me)
        counts = [int(e[0]) for e in every]
        if any(int(e[2]) for e in every):
            self.aborted = True
            raise RankFailed(f"rank(s) {[r for r, e in enumerate(every) if int(e[2])]} failed; rank {self.rank} stops at flush {self.flushes}")
        m = max(counts)
        self.max_rows_per_flush = max(self.max_rows_per_flush, m)
        if m:
            prow = rows.new_zeros((m, ROW))
            prow[: rows.shape[0]] = rows
            pidx = idx.new_zeros((m,))
            pidx[: idx.shape[0]] = idx
            if self.rank == 0:
                out_r = [torch.empty_like(prow) for _ in range(self.world)]
                out_i = [torch.empty_like(pidx) for _ in range(self.world)]
                dist.gather(prow, out_r, dst=0)
                dist.gather(pidx, out_i, dst=0)
                for r in range(self.world):
                    self._keep(out_i[r][: counts[r]].cpu(), out_r[r][: counts[r]].cpu())
            else:
                dist.gather(prow, None, dst=0)
                dist.gather(pidx, None, dst=0)
        return any(int(e[1]) for e in every)

    def finish(self, failed: bool = False) -> None:
        """Collective tail: this rank has no more batches; answer flushes until no rank has."""
        while self.flush(more=False, failed=failed):
            pass

    def _keep(self, idx: torch.Tensor, rows: torch.Tensor) -> None:
        self.total += int(rows.shape[0])
        if self.keep and rows.shape[0]:
            self.tile_index.append(idx)
            self.rows.append(rows)

    def table(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(tile_index int32 [n], rows float32 [n,6]) gathered so far on this rank (rank 0 holds everything)."""
        if not self.rows:
            return torch.zeros((0,), dtype=torch.int32), torch.zeros((0, ROW), dtype=torch.float32)
        return torch.cat(self.tile_index), torch.cat(self.rows)


def reduce_counters(tiles: int, labels: int, dets: int, elapsed_s: float, device) -> Tuple[int, int, int, float]:
    """C3: sums of (tiles, label files, detections) and the max of elapsed seconds over ranks."""
    if not active():
        return tiles, labels, dets, elapsed_s
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    s = torch.tensor([tiles, labels, dets], dtype=torch.int64, device=device)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s[0]), int(s[1]), int(s[2]), float(t[0])


def barrier() -> None:
    if active():
        dist.barrier()


def on_rank0(fn):
    """Runs ``fn`` on rank 0 and hands its (picklable) result to every rank.  If it raises there, EVERY rank raises -- rank 0 the original
    exception, the others a RuntimeError quoting it -- instead of the others waiting in their next collective until the launcher's timeout
    kills them (the run-directory check of a --resume, the fp8 calibration record).  One broadcast; a plain call in a world of one."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    result, error = None, None
    if rank == 0:
        try:
            result = fn()
        except Exception as e:
            error = e
    if active():
        wire = [result, None if error is None else f"{type(error).__name__}: {error}"]
        dist.broadcast_object_list(wire, src=0)
        if rank != 0:
            result, error = wire[0], None if wire[1] is None else RuntimeError(f"rank 0 refused the run: {wire[1]}")
    if error is not None:
        raise error
    return result
