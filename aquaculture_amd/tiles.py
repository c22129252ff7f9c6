"""Synthetic ocean tiles and the reference's tile file-name codec.

The producer of the hot path's input is reference src/load_data/tile_tifs.py:13-74
(6144^2 GeoTIFF -> 1024^2 tiles -> jpeg).  Names follow the shipped data
(reference output/cf_images.csv; reference src/utils.py:372-389):
``ORTHOIMAGERY.ORTHOPHOTOS{year}_{bbox}_{xoff}_{yoff}.jpeg`` with exactly four
``_``-separated fields, which the consumer relies on
(reference src/process_yolo/geocode_results.py:89,143).

Synthetic tiles follow SURVEY.md 8d: seeded per tile, ocean-coloured, low-frequency
swell, pixel noise, and on 5% of tiles a grid of ring-shaped "cages".
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List

import numpy as np

TILE_SEED_BASE = 0xA9CA_0000


def tile_name(i: int, year: int = 2015, extension: str = "jpeg") -> str:
    """File name of synthetic tile i (reference src/utils.py:372-383 generate_image_file_name_str)."""
    name = f"{year}_{i // 36}_{1024 * (i % 6)}_{1024 * ((i // 6) % 6)}"
    prefix = "ORTHOIMAGERY.ORTHOPHOTOS.ORTHO-EXPRESS." if int(year) == 2021 else "ORTHOIMAGERY.ORTHOPHOTOS"
    return f"{prefix}{name}.{extension}"


def parse_tile_name(file: str) -> Dict[str, str]:
    """Inverse (reference src/utils.py:386-389 generate_image_specs_from_file_name): 4 fields or ValueError."""
    stem = os.path.basename(file)
    for ext in (".jpeg", ".jpg", ".txt"):
        if stem.endswith(ext):
            stem = stem[: -len(ext)]
    name, bbox_ind, x_offset, y_offset = stem.split("_")
    return {"name": name, "bbox_ind": bbox_ind, "x_offset": x_offset, "y_offset": y_offset, "year": name[-4:]}


def _bilinear_up(field: np.ndarray, size: int) -> np.ndarray:
    n = field.shape[0]
    pos = (np.arange(size, dtype=np.float64) + 0.5) * n / size - 0.5
    i0 = np.clip(np.floor(pos).astype(np.int64), 0, n - 1)
    i1 = np.clip(i0 + 1, 0, n - 1)
    f = np.clip(pos - i0, 0.0, 1.0)
    rows = field[i0, :] * (1 - f)[:, None] + field[i1, :] * f[:, None]
    return rows[:, i0] * (1 - f)[None, :] + rows[:, i1] * f[None, :]


def synthetic_tile(i: int, size: int = 640) -> np.ndarray:
    """uint8 (size, size, 3) RGB tile, deterministic in ``i``."""
    rng = np.random.Generator(np.random.PCG64(TILE_SEED_BASE + i))
    base = rng.normal((28.0, 72.0, 108.0), (6.0, 8.0, 10.0))
    swell = _bilinear_up(rng.normal(0.0, 6.0, (20, 20)), size)
    img = base[None, None, :] + swell[:, :, None] + rng.normal(0.0, 3.0, (size, size, 3))
    if rng.random() < 0.05:
        n = int(rng.integers(1, 13))
        circ = rng.random() < 0.5
        pitch = int(rng.integers(44, 72))
        lo = min(40, size // 4)   # small test tiles: keep the draw valid (640-px tiles are unchanged)
        x0, y0 = int(rng.integers(lo, size // 2)), int(rng.integers(lo, size // 2))
        yy, xx = np.mgrid[0:size, 0:size]
        for k in range(n):
            cx, cy = x0 + (k % 4) * pitch, y0 + (k // 4) * pitch
            if circ:
                r = float(rng.integers(6, 21))
                d = np.hypot(xx - cx, yy - cy)
                ring, inside = (np.abs(d - r) <= 1.0), d < r - 1
            else:
                s = float(rng.integers(10, 37)) / 2
                d = np.maximum(np.abs(xx - cx), np.abs(yy - cy))
                ring, inside = (np.abs(d - s) <= 1.0), d < s - 1
            img[inside] = img[inside] * 0.8 - 6.0
            img[ring] = (200.0, 205.0, 210.0)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def synthetic_batch(indices: Iterable[int], size: int = 640) -> np.ndarray:
    return np.stack([synthetic_tile(int(i), size) for i in indices], 0)


def write_synthetic_jpegs(directory: str, indices: Iterable[int], size: int = 640, quality: int = 75) -> List[str]:
    """JPEG q=75 is GDAL's JPEG-driver default (reference src/load_data/tile_tifs.py:74)."""
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i in indices:
        p = os.path.join(directory, tile_name(int(i)))
        Image.fromarray(synthetic_tile(int(i), size)).save(p, format="JPEG", quality=quality)
        paths.append(p)
    return paths
