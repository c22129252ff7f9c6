"""Done-manifest of a tile sweep: which tiles have been fully processed, so that a restarted sweep skips them.

The reference's only recovery mechanism is skip-if-exists on its outputs (reference src/load_data/tile_tifs.py:40-41, :73).  That
idiom does not carry over to the label files of detect.py: a tile WITHOUT detections gets no file (the consumer relies on it,
reference src/process_yolo/geocode_results.py:123), so the label directory cannot say whether such a tile was processed.  Each rank
therefore appends the names of the tiles it has finished (label file written and closed, or known to have none) to its own text file
in the run directory, fsync'd once per batch; `--resume` reads every rank's file (the world size may differ between the two runs) and
drops those tiles from the listing before sharding.  A record is one line = one tile stem; a line cut short by a crash (no trailing
newline) is ignored, so a tile is either recorded after its label file is complete or processed again (label files are written with
"wb": processing a tile twice leaves the same bytes).

Durability: the manifest line vouches for the label file, so the file must reach the disk first -- detect.py makes every label file of
a batch and the labels directory's new entries durable (one syncfs(2) of that filesystem per batch; per-file fsync + directory fsync where
syncfs is unavailable) before `add`, which fsyncs the manifest; a node crash or power loss
can therefore lose unrecorded work but never leave a recorded tile without its label bytes.  `run_params.json` beside the manifests
holds what the labels depend on (weights digest, thresholds, image size, precision); `--resume` refuses a directory written with
other settings instead of mixing two runs' labels.
"""
from __future__ import annotations

import glob
import hashlib
import json
import os
from typing import Dict, Iterable, Set


def fsync_dir(path: str) -> None:
    """New directory entries (files created in `path`) reach the disk."""
    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
    finally:
        os.close(fd)


_libc = None


def sync_filesystem_of(path: str) -> bool:
    """syncfs(2) on the filesystem holding `path`: every dirty page and directory entry of that filesystem reaches the disk -- ONE call
    makes a whole batch of freshly written label files durable (64 fsyncs, one journal commit each, cost a multiple of the sweep's own
    time on a real disk).  False when the call is unavailable: the caller falls back to per-file fsync."""
    global _libc
    try:
        if _libc is None:
            import ctypes
            _libc = ctypes.CDLL("libc.so.6", use_errno=True)
        fd = os.open(path, os.O_RDONLY)
        try:
            return _libc.syncfs(fd) == 0
        finally:
            os.close(fd)
    except (OSError, AttributeError):
        return False


def file_digest(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for piece in iter(lambda: f.read(1 << 20), b""):
            h.update(piece)
    return h.hexdigest()


class RunParamsMismatch(RuntimeError):
    """--resume into a directory whose labels were written with other settings."""


def check_run_params(directory: str, params: Dict, resume: bool) -> None:
    """Rank 0, before any tile is processed: on --resume compare `params` with the interrupted run's record and refuse a mismatch
    (a directory without a record -- written by an older build -- is accepted and gets one); otherwise (re)write the record."""
    path = os.path.join(directory, "run_params.json")
    if resume and os.path.exists(path):
        with open(path) as f:
            old = json.load(f)
        diff = {k: (old.get(k), params.get(k)) for k in sorted(set(old) | set(params)) if old.get(k) != params.get(k)}
        if diff:
            raise RunParamsMismatch(f"--resume: {directory} was written with other settings (recorded, now): {diff}")
        return
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(params, f, indent=1, sort_keys=True)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)
    fsync_dir(directory)


class DoneManifest:
    def __init__(self, directory: str, rank: int = 0):
        self.directory = directory
        self.path = os.path.join(directory, f"done.rank{rank}.txt")
        self._f = None

    def open(self) -> None:
        os.makedirs(self.directory, exist_ok=True)
        if os.path.exists(self.path):          # a crash may have left a record cut short: drop it (a truncated name could be another tile's)
            with open(self.path, "r+b") as r:
                data = r.read()
                r.truncate(data.rfind(b"\n") + 1)
        self._f = open(self.path, "ab")

    def add(self, stems: Iterable[str]) -> None:
        """One batch: durable when this returns (the label files of these tiles must be on disk -- fsync'd, and their directory too --
        before the call)."""
        data = "".join(s + "\n" for s in stems).encode()
        if not data:
            return
        self._f.write(data)
        self._f.flush()
        os.fsync(self._f.fileno())

    def close(self) -> None:
        if self._f is not None:
            self._f.close()
            self._f = None

    @staticmethod
    def load(directory: str) -> Set[str]:
        """Union over every rank's manifest in ``directory``; ignores a trailing record without its newline."""
        done: Set[str] = set()
        for path in sorted(glob.glob(os.path.join(directory, "done.rank*.txt"))):
            with open(path, "rb") as f:
                data = f.read()
            end = data.rfind(b"\n")
            if end < 0:
                continue
            done.update(line for line in data[:end].decode(errors="replace").split("\n") if line)
        return done
