"""Tile-level data parallelism: one process per GPU, strided shard, one gather at the end.

The reference runs detect.py as one process on one device (reference README.md:77); tiles are independent
(no halo: reference src/load_data/tile_tifs.py:33-47; one label file per tile), so the path shards with NO
data-path collective.  The only exchange is the final detection gather: counts, then
padded rows, then counters -- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

ROW = 7   # (tile_index, cls, x1, y1, x2, y2, conf) as float32; tile_index < 2^24 is exact in fp32


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun-style environment; (0, 1, 0) when not launched by it."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("AQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n: int, rank: int, world: int) -> range:
    """Tiles i = rank (mod world) of the sorted list: balances cage-dense scenes that cluster in name order."""
    return range(rank, n, world)


def pack_rows(tile_index: torch.Tensor, dets: torch.Tensor) -> torch.Tensor:
    """tile_index int [n], dets float32 [n,6] (x1,y1,x2,y2,conf,cls) -> float32 [n,7] gather rows."""
    if dets.shape[0] == 0:
        return dets.new_zeros((0, ROW))
    return torch.cat((tile_index.to(dets.dtype).view(-1, 1), dets[:, 5:6], dets[:, :4], dets[:, 4:5]), 1)


def gather_rows(rows: torch.Tensor) -> torch.Tensor:
    """All ranks get every rank's rows (rank order).  C1: all_gather(count); C2: padded all_gather(rows)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        rows = rows.cpu()          # rehearsal backend: host tensors
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    if m == 0:
        return rows.new_zeros((0, ROW))
    padded = rows.new_zeros((m, ROW))
    padded[: rows.shape[0]] = rows
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def reduce_counters(tiles: int, labels: int, dets: int, elapsed_s: float, device) -> Tuple[int, int, int, float]:
    """C3: sums of (tiles, label files, detections) and the max of elapsed seconds over ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tiles, labels, dets, elapsed_s
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    s = torch.tensor([tiles, labels, dets], dtype=torch.int64, device=device)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s[0]), int(s[1]), int(s[2]), float(t[0])


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
