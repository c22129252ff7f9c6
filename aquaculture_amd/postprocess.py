"""Box rescale + YOLO label text: the output contract of the hot path.

Restates, in numpy fp32 with the same operation order, what ``yolov5/detect.py`` does after NMS
[UPSTREAM detect.py run(): ``det[:, :4] = scale_boxes(im.shape[2:], det[:, :4], im0.shape).round()``,
then per detection ``xywh = (xyxy2xywh(xyxy.view(1,4)) / gn).view(-1).tolist()`` and
``('%g ' * len(line)).rstrip() % line`` with ``line = (cls, *xywh, conf)`` under ``--save-conf``;
utils/general.py scale_boxes / clip_boxes / xyxy2xywh].

The consumer is reference src/process_yolo/geocode_results.py:140-172 (``np.loadtxt`` rows
``cls xc yc w h conf``; one file per image stem, *no file when there are no detections*,
reference src/process_yolo/geocode_results.py:46-55).  Because the consumer truncates
``int(IM_WIDTH * (xc - w / 2))`` on the printed 6-significant-digit values
(reference src/process_yolo/geocode_results.py:160-163), label lines must be byte-identical to
upstream's whenever the rounded pixel boxes agree; this module therefore works on the exact fp32
values and prints with ``%g``.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, Tuple

import numpy as np

F32 = np.float32


def scale_boxes(img1_shape: Sequence[int], boxes: np.ndarray, img0_shape: Sequence[int]) -> np.ndarray:
    """Letterbox inverse (fp32) + clip.  img1 = network input (h, w), img0 = original image (h, w)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = ((img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2)
    b = np.array(boxes, dtype=F32, copy=True)
    b[..., [0, 2]] -= F32(pad[0])
    b[..., [1, 3]] -= F32(pad[1])
    b[..., :4] /= F32(gain)
    np.clip(b[..., 0], 0, img0_shape[1], out=b[..., 0])
    np.clip(b[..., 1], 0, img0_shape[0], out=b[..., 1])
    np.clip(b[..., 2], 0, img0_shape[1], out=b[..., 2])
    np.clip(b[..., 3], 0, img0_shape[0], out=b[..., 3])
    return b


def detections_to_rows(det: np.ndarray, img1_shape, img0_shape) -> np.ndarray:
    """(n,6) [x1,y1,x2,y2,conf,cls] in network pixels (descending conf) -> (n,6) float32 rows
    [cls, xc, yc, w, h, conf] normalised by the ORIGINAL size, in file order (ascending conf)."""
    det = np.asarray(det, dtype=F32)
    if det.shape[0] == 0:
        return np.zeros((0, 6), F32)
    xyxy = np.rint(scale_boxes(img1_shape, det[:, :4], img0_shape)).astype(F32)   # torch.round = half-to-even
    h0, w0 = F32(img0_shape[0]), F32(img0_shape[1])
    rows = np.empty((det.shape[0], 6), F32)
    rows[:, 0] = det[:, 5]
    rows[:, 1] = ((xyxy[:, 0] + xyxy[:, 2]) / F32(2)) / w0
    rows[:, 2] = ((xyxy[:, 1] + xyxy[:, 3]) / F32(2)) / h0
    rows[:, 3] = (xyxy[:, 2] - xyxy[:, 0]) / w0
    rows[:, 4] = (xyxy[:, 3] - xyxy[:, 1]) / h0
    rows[:, 5] = det[:, 4]
    return rows[::-1].copy()   # `for *xyxy, conf, cls in reversed(det)`


def batch_rows(det_all: np.ndarray, counts: np.ndarray, img1_shape, img0_shape):
    """detections_to_rows for a whole batch of tiles of ONE original size in a single pass: det_all [B, max_det, 6], counts [B] ->
    (rows float32 [N, 6] with every tile's rows in file order, tile after tile; offsets int64 [B + 1]).  The arithmetic is elementwise,
    so the values equal the per-tile function's bit for bit; what it saves is ~20 numpy calls per tile on threads that share one GIL."""
    counts = np.asarray(counts, dtype=np.int64)
    B = counts.shape[0]
    offsets = np.zeros(B + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    n = int(offsets[-1])
    if n == 0:
        return np.zeros((0, 6), F32), offsets
    valid = np.arange(det_all.shape[1])[None, :] < counts[:, None]
    det = np.asarray(det_all[:B], dtype=F32)[valid]                     # [N, 6], tile-major, descending confidence inside a tile
    xyxy = np.rint(scale_boxes(img1_shape, det[:, :4], img0_shape)).astype(F32)
    h0, w0 = F32(img0_shape[0]), F32(img0_shape[1])
    rows = np.empty((n, 6), F32)
    rows[:, 0] = det[:, 5]
    rows[:, 1] = ((xyxy[:, 0] + xyxy[:, 2]) / F32(2)) / w0
    rows[:, 2] = ((xyxy[:, 1] + xyxy[:, 3]) / F32(2)) / h0
    rows[:, 3] = (xyxy[:, 2] - xyxy[:, 0]) / w0
    rows[:, 4] = (xyxy[:, 3] - xyxy[:, 1]) / h0
    rows[:, 5] = det[:, 4]
    tile = np.repeat(np.arange(B), counts)
    pos = np.arange(n)
    rev = offsets[tile] + (counts[tile] - 1) - (pos - offsets[tile])     # `for *xyxy, conf, cls in reversed(det)`, per tile
    return rows[rev], offsets


def format_rows(rows: np.ndarray, save_conf: bool = True) -> str:
    """Text of one label file.  Each value through ``%g`` of the double that equals the fp32 value
    (one C-level format call for the whole file: the same conversions as upstream's per-line ``%``)."""
    n = 6 if save_conf else 5
    k = rows.shape[0]
    if k == 0:
        return ""
    line = ("%g " * n).rstrip() + "\n"
    return (line * k) % tuple(np.asarray(rows[:, :n], dtype=np.float64).ravel().tolist())


def write_label_file(labels_dir: str, stem: str, rows: np.ndarray, save_conf: bool = True) -> bool:
    """Appends like upstream (``open(f'{txt_path}.txt', 'a')``); writes nothing for zero detections."""
    if rows.shape[0] == 0:
        return False
    with open(os.path.join(labels_dir, stem + ".txt"), "a") as f:
        f.write(format_rows(rows, save_conf))
    return True


def class_summary(det_cls: np.ndarray, names) -> str:
    """The per-image log fragment upstream prints: "3 circle_farms, 1 square_farm, "."""
    s = ""
    for c in np.unique(det_cls):
        n = int((det_cls == c).sum())
        s += f"{n} {names[int(c)]}{'s' * (n > 1)}, "
    return s
