"""Label rows -> pixel boxes -> geocoded boxes, as ONE vectorised pass over all detections of a sweep.

The consumer of the hot path's label files, reference src/process_yolo/geocode_results.py:106-197 (`geocode_all_detections`) with
its helper :71-101 (`convert_pix_to_m_bboxes`), walks the label files one detection at a time in Python.  This module does the same
arithmetic -- IEEE double throughout, in the reference's operation order, including its quirks -- on whole arrays:

  * pixel box  = int(1024 * (xc -/+ w / 2)), int(1024 * (yc -/+ h / 2)): truncation toward zero, and a FIXED 1024-px tile size
    (reference src/utils.py:18-19) whatever the image's real size;
  * tile name  = `<prefix><year>_<bbox_ind>_<x_offset>_<y_offset>`: exactly four `_`-separated fields (:90), year = last four
    characters of the first (:143);
  * metres     = affine map through the EPSG:3857 bounds of the 6144-px parent scene `bbox_ind` (:92-99);
  * geometry   = box(xmin_m, ymin_m, xmax_m, ymax_m) in EPSG:3857, delivered in EPSG:4326 (`df.to_crs(4326)`, :195);
  * area CRS   = EPSG:3035 corner coordinates (:176-177).

Pinned against the reference's own output for the pixel -> EPSG:3857 -> EPSG:4326 chain (tests/golden/g7_geocode.json).  The
EPSG:3035 columns follow the published IOGP formulas and are checked against the published worked examples only (pyproj is not
available here): they are returned as explicit easting / northing, because which of the reference's `xmin_m`/`ymin_m` columns
receives which depends on pyproj's axis-order handling for EPSG:3035 (northing-first by authority), which cannot be checked
without pyproj.  Out of scope, as in the reference's later steps: the land filter, de-duplication, facility clustering.
"""
from __future__ import annotations

import csv
import glob
import json
import os
import re
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

LARGE_TIF_SIZE = 1024 * 6          # reference src/utils.py:17
IM_WIDTH = 1024                    # reference src/utils.py:18
IM_HEIGHT = 1024                   # reference src/utils.py:19
REVERSE_CLASS_MAPPING = {0: "circle_farm", 1: "square_farm", 2: "triangle_farm", 3: "other_farm", 4: "rectangle_farm"}
"""reference src/process_yolo/geocode_results.py:24-30"""

WGS84_A = 6378137.0
GRS80_A = 6378137.0
GRS80_F = 1.0 / 298.257222101
LAEA_LAT0 = np.radians(52.0)
LAEA_LON0 = np.radians(10.0)
LAEA_FE = 4321000.0
LAEA_FN = 3210000.0

_NUM = re.compile(r"-?\d+\.?\d*(?:[eE][-+]?\d+)?")


def load_wanted_bboxes(path: str) -> Dict[int, Tuple[float, float, float, float]]:
    """`data/wanted_bboxes.csv` (reference src/utils.py:25-43: index column + WKT POLYGON in EPSG:3857) -> {bbox_ind: bounds}.
    Only the bounds are used downstream (`large_tif_bbox.bounds`, reference geocode_results.py:96)."""
    out = {}
    with open(path, newline="") as f:
        r = csv.reader(f)
        header = next(r)
        gcol = header.index("geometry")
        for row in r:
            nums = [float(v) for v in _NUM.findall(row[gcol])]
            if len(nums) < 6:
                raise ValueError(f"{path}: row {row[0]!r} holds no polygon")
            xs, ys = nums[0::2], nums[1::2]
            out[int(row[0])] = (min(xs), min(ys), max(xs), max(ys))
    return out


def parse_stems(stems: Sequence[str]) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Tile stems -> (bbox_ind, x_offset, y_offset, year) int64 arrays.  A stem without exactly four `_` fields is an error, as the
    reference's tuple unpacking makes it."""
    ind, xo, yo, yr = [], [], [], []
    for s in stems:
        parts = s.split("_")
        if len(parts) != 4:
            raise ValueError(f"tile name {s!r}: expected <prefix><year>_<bbox_ind>_<x_offset>_<y_offset>")
        ind.append(int(parts[1])); xo.append(int(parts[2])); yo.append(int(parts[3])); yr.append(int(parts[0][-4:]))
    return (np.asarray(ind, np.int64), np.asarray(xo, np.int64), np.asarray(yo, np.int64), np.asarray(yr, np.int64))


def rows_from_label_text(text: str) -> np.ndarray:
    """What `np.loadtxt(label)` gives the reference: float64 [n, 6] parsed from the `%g` text."""
    vals = np.array(text.split(), dtype=np.float64)
    if vals.size % 6:
        raise ValueError("label text is not rows of 6 numbers (detect.py --save-txt --save-conf)")
    return vals.reshape(-1, 6)


def mercator_to_lonlat(x: np.ndarray, y: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """EPSG:3857 -> EPSG:4326 (IOGP GN7-2 method 1024, reverse)."""
    lon = np.degrees(x / WGS84_A)
    lat = np.degrees(np.pi / 2.0 - 2.0 * np.arctan(np.exp(-y / WGS84_A)))
    return lon, lat


def _laea_q(sin_phi, e):
    return (1.0 - e * e) * (sin_phi / (1.0 - e * e * sin_phi * sin_phi) - (1.0 / (2.0 * e)) * np.log((1.0 - e * sin_phi) / (1.0 + e * sin_phi)))


def lonlat_to_laea_europe(lon_deg: np.ndarray, lat_deg: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """EPSG:4326 -> EPSG:3035 (easting, northing): IOGP GN7-2 method 9820, oblique aspect, GRS80, origin 52 N 10 E."""
    e2 = GRS80_F * (2.0 - GRS80_F)
    e = np.sqrt(e2)
    phi, lam = np.radians(lat_deg), np.radians(lon_deg)
    qp = _laea_q(1.0, e)
    q0 = _laea_q(np.sin(LAEA_LAT0), e)
    q = _laea_q(np.sin(phi), e)
    beta0 = np.arcsin(q0 / qp)
    beta = np.arcsin(q / qp)
    rq = GRS80_A * np.sqrt(qp / 2.0)
    d = GRS80_A * (np.cos(LAEA_LAT0) / np.sqrt(1.0 - e2 * np.sin(LAEA_LAT0) ** 2)) / (rq * np.cos(beta0))
    b = rq * np.sqrt(2.0 / (1.0 + np.sin(beta0) * np.sin(beta) + np.cos(beta0) * np.cos(beta) * np.cos(lam - LAEA_LON0)))
    east = LAEA_FE + b * d * (np.cos(beta) * np.sin(lam - LAEA_LON0))
    north = LAEA_FN + (b / d) * (np.cos(beta0) * np.sin(beta) - np.sin(beta0) * np.cos(beta) * np.cos(lam - LAEA_LON0))
    return east, north


def geocode_detections(stems: Sequence[str], counts: Sequence[int], rows: np.ndarray,
                       wanted_bboxes: Dict[int, Tuple[float, float, float, float]]) -> Dict[str, np.ndarray]:
    """All detections of a sweep at once.

    stems[i] = tile name of image i (no extension), counts[i] = its number of label rows, rows = the concatenated float64
    [sum(counts), 6] rows `cls xc yc w h conf` AS PARSED FROM THE LABEL TEXT (use rows_from_label_text / label_dir_rows: the
    reference reads the `%g` text, so the 6-significant-digit rounding is part of the arithmetic).
    Returns columns (one entry per detection, label-file order): image (index into stems), xmin, xmax, ymin, ymax (int64 pixels),
    xmin_3857 .. ymax_3857, lon_min/lon_max/lat_min/lat_max (the EPSG:4326 box corners), e_min_3035 .. n_max_3035, cls, year, det_conf.
    """
    rows = np.asarray(rows, np.float64).reshape(-1, 6)
    counts = np.asarray(counts, np.int64)
    if counts.sum() != rows.shape[0] or len(stems) != counts.shape[0]:
        raise ValueError("counts do not add up to the number of rows")
    ind, xo, yo, yr = parse_stems(stems)
    missing = sorted(set(int(i) for i in ind) - set(wanted_bboxes))
    if missing:
        raise KeyError(f"bbox_ind {missing[:5]} not in the wanted_bboxes table")
    b = np.array([wanted_bboxes[int(i)] for i in ind], np.float64).reshape(-1, 4)
    img = np.repeat(np.arange(len(stems), dtype=np.int64), counts)
    xc, yc, w, h = rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4]
    # reference geocode_results.py:160-163 -- int() truncates toward zero
    xmin = np.trunc(IM_WIDTH * (xc - w / 2)).astype(np.int64)
    ymin = np.trunc(IM_HEIGHT * (yc - h / 2)).astype(np.int64)
    xmax = np.trunc(IM_WIDTH * (xc + w / 2)).astype(np.int64)
    ymax = np.trunc(IM_HEIGHT * (yc + h / 2)).astype(np.int64)
    bx0, by0, bx1, by1 = b[img, 0], b[img, 1], b[img, 2], b[img, 3]
    sx = (bx1 - bx0) / LARGE_TIF_SIZE                        # m per pixel (reference geocode_results.py:98-99)
    sy = (by1 - by0) / LARGE_TIF_SIZE
    xoff, yoff = xo[img], yo[img]
    xmin_m = (xmin + xoff) * sx + bx0
    ymax_m = by1 - (ymin + yoff) * sy                        # image y grows downward: the pixel ymin is the northern edge
    xmax_m = (xmax + xoff) * sx + bx0
    ymin_m = by1 - (ymax + yoff) * sy
    lon_min, lat_max = mercator_to_lonlat(xmin_m, ymax_m)
    lon_max, lat_min = mercator_to_lonlat(xmax_m, ymin_m)
    e_min, n_max = lonlat_to_laea_europe(lon_min, lat_max)
    e_max, n_min = lonlat_to_laea_europe(lon_max, lat_min)
    return {"image": img, "xmin": xmin, "xmax": xmax, "ymin": ymin, "ymax": ymax,
            "xmin_3857": xmin_m, "ymin_3857": ymin_m, "xmax_3857": xmax_m, "ymax_3857": ymax_m,
            "lon_min": lon_min, "lon_max": lon_max, "lat_min": lat_min, "lat_max": lat_max,
            "e_min_3035": e_min, "e_max_3035": e_max, "n_min_3035": n_min, "n_max_3035": n_max,
            "cls": rows[:, 0].astype(np.int64), "year": yr[img], "det_conf": rows[:, 5].copy()}


def label_dir_rows(labels_dir: str) -> Tuple[List[str], List[int], np.ndarray]:
    """Reads `labels_dir/*.txt` (what detect.py --save-txt --save-conf wrote) -> (stems, counts, rows) for geocode_detections.
    Files are visited in sorted order (the reference's glob order is unspecified; its output is order-independent downstream)."""
    stems, counts, chunks = [], [], []
    for path in sorted(glob.glob(os.path.join(labels_dir, "*.txt"))):
        with open(path) as f:
            r = rows_from_label_text(f.read())
        if r.shape[0] == 0:
            continue
        stems.append(os.path.basename(path)[:-4]); counts.append(r.shape[0]); chunks.append(r)
    rows = np.concatenate(chunks, 0) if chunks else np.zeros((0, 6), np.float64)
    return stems, counts, rows


def write_geojson(path: str, stems: Sequence[str], table: Dict[str, np.ndarray]) -> int:
    """FeatureCollection in EPSG:4326 with the reference's property names where they are unambiguous (image, xmin, xmax, ymin,
    ymax, type, year, det_conf) and explicit e/n names for the EPSG:3035 corners.  Returns the number of features."""
    feats = []
    n = table["image"].shape[0]
    for k in range(n):
        x0, x1, y0, y1 = (float(table[c][k]) for c in ("lon_min", "lon_max", "lat_min", "lat_max"))
        feats.append({"type": "Feature",
                      "properties": {"image": stems[int(table["image"][k])] + ".jpeg",
                                     "xmin": int(table["xmin"][k]), "xmax": int(table["xmax"][k]),
                                     "ymin": int(table["ymin"][k]), "ymax": int(table["ymax"][k]),
                                     "e_min_3035": float(table["e_min_3035"][k]), "e_max_3035": float(table["e_max_3035"][k]),
                                     "n_min_3035": float(table["n_min_3035"][k]), "n_max_3035": float(table["n_max_3035"][k]),
                                     "type": REVERSE_CLASS_MAPPING[int(table["cls"][k])], "year": int(table["year"][k]),
                                     "det_conf": float(table["det_conf"][k])},
                      # shapely.geometry.box(minx, miny, maxx, maxy) ring order: (maxx, miny), (maxx, maxy), (minx, maxy), (minx, miny)
                      "geometry": {"type": "Polygon", "coordinates": [[[x1, y0], [x1, y1], [x0, y1], [x0, y0], [x1, y0]]]}})
    with open(path, "w") as f:
        json.dump({"type": "FeatureCollection", "crs": {"type": "name", "properties": {"name": "urn:ogc:def:crs:OGC:1.3:CRS84"}},
                   "features": feats}, f)
    return n


def geocode_label_dir(labels_dir: str, wanted_bboxes_csv: str, out_geojson: str | None = None) -> Dict[str, np.ndarray]:
    """labels directory + wanted_bboxes.csv -> table (and optionally the GeoJSON file): the whole of the reference's
    geocode_all_detections for a sweep, minus the image existence check (it only needs the images to skip unreadable ones)."""
    stems, counts, rows = label_dir_rows(labels_dir)
    table = geocode_detections(stems, counts, rows, load_wanted_bboxes(wanted_bboxes_csv))
    table["stems"] = np.asarray(stems, dtype=object)
    if out_geojson:
        write_geojson(out_geojson, stems, table)
    return table
