"""Scene mode: whole aerial scenes in, detections per 1024-px tile out, without the jpeg intermediate.

The reference prepares its detector input in two passes over disk (src/load_data/tile_tifs.py): ``split_all_tiffs`` cuts every scene
``*.tif`` into ``tilesize`` x ``tilesize`` GeoTIFF tiles with ``gdal.Translate(srcWin=[i, j, w, h])`` (:13-47, i = x offset outer
loop, j = y offset inner loop, edge tiles smaller), ``create_jpegs`` re-encodes each tile as JPEG (:50-74); detect.py then decodes
the jpegs again (README.md:77).  Here the scene is read once, uploaded once, and the tiles are cut on the device by the letterbox
kernel itself (``aq_letterbox_tiles_u8``): tile pixels never exist as separate images.  Opt-in (``detect.py --tile-scenes``):
pixel values differ from the reference's pipeline by exactly the JPEG round trip that is skipped.

Tile names follow the shipped data (output/cf_images.csv, four ``_`` fields: ``<scene stem>_<x offset>_<y offset>``), which is what
the consumer parses (src/utils.py:385-389); the ``.tif_`` infix that tile_tifs.py:37,69 would produce as written is not reproduced.
"""
from __future__ import annotations

import glob
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Iterator, List, Sequence, Tuple

import numpy as np

SCENE_FORMATS = ("tif", "tiff")


def list_scenes(source: str) -> List[str]:
    """Sorted scene rasters of a directory (or one file / a glob), like tile_tifs.py:23 glob('*.tif') but in a defined order."""
    p = str(Path(source).resolve())
    if "*" in p:
        files = sorted(glob.glob(p))
    elif os.path.isdir(p):
        files = sorted(glob.glob(os.path.join(p, "*.*")))
    elif os.path.isfile(p):
        files = [p]
    else:
        raise FileNotFoundError(f"{p} does not exist")
    scenes = [f for f in files if f.rsplit(".", 1)[-1].lower() in SCENE_FORMATS]
    if not scenes:
        raise FileNotFoundError(f"No scene rasters (*.tif) found in {p}")
    return scenes


def tile_grid(width: int, height: int, tilesize: int = 1024) -> List[Tuple[int, int, int, int]]:
    """(x offset, y offset, w, h) of every tile in the reference's order (tile_tifs.py:33-36): x outer, y inner, edge tiles cut short."""
    if width <= 0 or height <= 0 or tilesize <= 0:
        raise ValueError("tile_grid: sizes must be positive")
    return [(i, j, min(i + tilesize, width) - i, min(j + tilesize, height) - j)
            for i in range(0, width, tilesize) for j in range(0, height, tilesize)]


def tile_stem(scene_path: str, x_off: int, y_off: int) -> str:
    """``<scene stem>_<x>_<y>``: the four-field form of the shipped tile names once the scene stem is ``ORTHO...<year>_<bbox>``."""
    return f"{Path(scene_path).stem}_{x_off}_{y_off}"


def _mapped_strips(path: str):
    """Uncompressed, chunky 8-bit RGB TIFF whose strips lie back to back (what GDAL's GTiff driver writes by default): the pixel
    block is mapped, not decoded.  None if the file is anything else."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            t = im.tag_v2
            w, h = im.size
            if (im.mode != "RGB" or t.get(259, 1) != 1 or t.get(284, 1) != 1 or tuple(t.get(258, ())) != (8, 8, 8) or 322 in t or 273 not in t
                    or 279 not in t):
                return None
            offs, cnts = list(t[273]), list(t[279])
    except Exception:
        return None
    if sum(cnts) != w * h * 3 or any(offs[k] + cnts[k] != offs[k + 1] for k in range(len(offs) - 1)):
        return None
    if offs[0] + w * h * 3 > os.path.getsize(path):
        return None
    return np.memmap(path, dtype=np.uint8, mode="r", offset=offs[0], shape=(h, w, 3))


def read_scene(path: str, mapped: bool = False) -> np.ndarray:
    """Scene raster -> uint8 [H, W, 3] RGB (what ``gdal.Translate -ot Byte -of JPEG`` encodes, tile_tifs.py:74).  8-bit 3-band rasters
    only: the JPEG driver the reference uses would refuse anything else as well.  ``mapped``: return a read-only memory map of the
    pixel block when the file layout allows it (no decode, no copy) -- same bytes either way."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None            # scenes are 6144 x 6144 and larger: not a decompression bomb
    if mapped:
        m = _mapped_strips(path)
        if m is not None:
            return m
    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA", "P", "L"):
            raise ValueError(f"{path}: mode {im.mode} is not an 8-bit raster")
        if im.mode == "RGBA":
            raise ValueError(f"{path}: four bands; the tile_tifs.py JPEG step handles 3-band rasters only")
        arr = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(arr)


def group_tiles(grid: Sequence[Tuple[int, int, int, int]]):
    """Tiles of equal size batch together: {(h, w): [grid index, ...]} in grid order (full tiles first for any real scene)."""
    groups = {}
    for k, (_, _, w, h) in enumerate(grid):
        groups.setdefault((h, w), []).append(k)
    return groups


class SceneTiles:
    """Iterates scenes (sharded ``i % world == rank`` over the sorted scene list) and yields, per scene and tile size,
    ``(stems, scene uint8 [H,W,3] numpy, [(x0, y0), ...], (h, w))``; the caller uploads the scene and cuts the tiles on the device."""

    def __init__(self, source: str, tilesize: int = 1024, shard: Tuple[int, int] = (0, 1), workers: int = 4, batch_size: int = 64,
                 pinned: bool = False):
        scenes = list_scenes(source)
        self.pinned = pinned
        self.slot_of = {}
        self._inflight = []
        self.tilesize = int(tilesize)
        rank, world = shard
        self.scene_indices = list(range(rank, len(scenes), world))
        self.scenes = [scenes[i] for i in self.scene_indices]
        self.n_scenes_total = len(scenes)
        self.workers = max(1, workers)
        self.batch_size = int(batch_size)
        # global tile numbering needs every scene's size (header only), on every rank
        from PIL import Image
        Image.MAX_IMAGE_PIXELS = None
        self.tile_offset, n = [], 0
        for f in scenes:
            with Image.open(f) as im:
                w, h = im.size
            self.tile_offset.append(n)
            n += len(tile_grid(w, h, self.tilesize))
        self.total = n

    def __len__(self):
        return len(self.scenes)

    def __iter__(self) -> Iterator[Tuple[str, np.ndarray]]:
        """(path, raster) in shard order; with ``pinned=True`` the raster is a numpy view of a page-locked buffer (slot in
        ``self.slot_of[path]``) that the caller hands back with ``uploaded(slot, event)`` once its H2D copy is enqueued."""
        if not self.pinned:
            if self.workers == 1 or len(self.scenes) <= 1:
                for p in self.scenes:
                    yield p, read_scene(p)
                return
            depth = min(self.workers, 3)        # a few scenes decode ahead of the GPU (PIL releases the GIL); a scene is > 100 MB
            with ThreadPoolExecutor(depth) as ex:
                pending = []
                for p in self.scenes:
                    pending.append((p, ex.submit(read_scene, p)))
                    if len(pending) >= depth:
                        q, f = pending.pop(0)
                        yield q, f.result()
                for q, f in pending:
                    yield q, f.result()
            return
        import torch
        nslots = 4
        free = list(range(nslots))
        bufs = [None] * nslots
        self._inflight = []

        def get_slot():
            if free:
                return free.pop()
            if not self._inflight:
                raise RuntimeError("scene ring: every page-locked buffer is out and none was handed back with uploaded()")
            ev, slot = self._inflight.pop(0)
            ev.synchronize()
            return slot

        def job(path, slot):
            arr = read_scene(path, mapped=True)
            if bufs[slot] is None or bufs[slot].numel() < arr.size:
                bufs[slot] = torch.empty(arr.size, dtype=torch.uint8).pin_memory()
            view = bufs[slot][: arr.size].view(arr.shape).numpy()
            np.copyto(view, arr)                 # page cache / decoded array -> page-locked memory (memcpy, GIL released)
            return view

        depth = min(self.workers, nslots - 1)
        with ThreadPoolExecutor(depth) as ex:
            pending = []
            for p in self.scenes:
                slot = get_slot()
                pending.append((p, slot, ex.submit(job, p, slot)))
                if len(pending) >= depth:
                    q, sl, f = pending.pop(0)
                    self.slot_of = {q: sl}
                    yield q, f.result()
            for q, sl, f in pending:
                self.slot_of = {q: sl}
                yield q, f.result()

    def uploaded(self, slot: int, event) -> None:
        """The H2D copy out of page-locked buffer ``slot`` has been enqueued; ``event`` (recorded after it) guards its reuse."""
        self._inflight.append((event, slot))

    def batches(self):
        """-> (scene path, scene array, stems, origins, (h, w), global tile indices) per (scene, tile size, chunk of batch_size tiles)."""
        for si, (path, arr) in zip(self.scene_indices, self):
            H, W = arr.shape[:2]
            base = self.tile_offset[si]
            grid = tile_grid(W, H, self.tilesize)
            for (h, w), idx in group_tiles(grid).items():
                for c in range(0, len(idx), self.batch_size):
                    part = idx[c:c + self.batch_size]
                    yield (path, arr, [tile_stem(path, grid[k][0], grid[k][1]) for k in part], [(grid[k][0], grid[k][1]) for k in part], (h, w),
                           [base + k for k in part])
