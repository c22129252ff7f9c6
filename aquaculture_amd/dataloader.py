"""Image directory -> letterboxed uint8 RGB tiles.

Restates [UPSTREAM utils/dataloaders.py LoadImages, utils/augmentations.py letterbox] as invoked by
``yolov5/detect.py --source DIR`` (reference README.md:77).  The directory is what the reference's tiler
writes: ``*.jpeg``, 8-bit, 3-band (reference src/load_data/tile_tifs.py:66-74).

Differences that are deliberate and documented in DESIGN.md:
  * decode is PIL/libjpeg-turbo (cv2 is not installed here); upstream reads BGR and flips to RGB, we read RGB;
  * the INTER_LINEAR resize (only taken when the tile is not already 640-fitting) is a numpy restatement of
    OpenCV's 8-bit fixed-point bilinear kernel -- UNPINNED until a cv2 is available to check it against.
"""
from __future__ import annotations

import glob
import math
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

IMG_FORMATS = ("bmp", "dng", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp", "pfm")  # [UPSTREAM dataloaders.IMG_FORMATS]
PAD_VALUE = 114
_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def list_images(source: str) -> List[str]:
    """[UPSTREAM LoadImages.__init__]: sorted(glob(dir/*.*)) filtered by extension (a file or glob also works)."""
    p = str(Path(source).resolve())
    if "*" in p:
        files = sorted(glob.glob(p, recursive=True))
    elif os.path.isdir(p):
        files = sorted(glob.glob(os.path.join(p, "*.*")))
    elif os.path.isfile(p):
        files = [p]
    else:
        raise FileNotFoundError(f"{p} does not exist")
    images = [x for x in files if x.split(".")[-1].lower() in IMG_FORMATS]
    if not images:
        raise FileNotFoundError(f"No images found in {p}. Supported formats are: {IMG_FORMATS}")
    return images


def _axis_coeffs(src: int, dst: int):
    """OpenCV resize INTER_LINEAR 8u coefficient table for one axis: index, (w0, w1) as int16 in 2^11 units."""
    scale = src / dst
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    w1 = np.rint(f * np.float32(_COEF_SCALE)).astype(np.int32)            # saturate_cast<short>(cvRound)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(_COEF_SCALE)).astype(np.int32)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, w0, w1


def resize_linear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC (restated, unpinned)."""
    h, w = img.shape[:2]
    x0, x1, a0, a1 = _axis_coeffs(w, new_w)
    y0, y1, b0, b1 = _axis_coeffs(h, new_h)
    src = img.astype(np.int32)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]        # HResize, scale 2^11
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(shape: Tuple[int, int], new_shape=(640, 640), auto=True, scaleup=True, stride=32):
    """Returns (new_unpad (w, h), (top, bottom, left, right)) exactly as upstream letterbox computes them."""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, (top, bottom, left, right)


def letterbox(im: np.ndarray, new_shape=(640, 640), auto=True, scaleup=True, stride=32) -> np.ndarray:
    """Resize + pad to a stride multiple with 114 [UPSTREAM letterbox].  640^2 and 1024^2 tiles need no padding."""
    shape = im.shape[:2]
    new_unpad, (top, bottom, left, right) = letterbox_geometry(shape, new_shape, auto, scaleup, stride)
    if shape[::-1] != new_unpad:
        im = resize_linear_u8(im, new_unpad[0], new_unpad[1])
    if top or bottom or left or right:
        im = np.pad(im, ((top, bottom), (left, right), (0, 0)), mode="constant", constant_values=PAD_VALUE)
    return np.ascontiguousarray(im)


def check_img_size(imgsz, s=32):
    """[UPSTREAM utils/general.py check_img_size]: round each side up to a multiple of the max stride."""
    if isinstance(imgsz, int):
        return max(math.ceil(imgsz / s) * s, 0)
    return [max(math.ceil(x / s) * s, 0) for x in imgsz]


def read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class LoadImages:
    """Iterates (path, letterboxed uint8 HWC RGB, original (h, w)) in sorted order; ``shard`` = (rank, world)
    keeps images ``i % world == rank`` (strided tile shard, one process per GPU)."""

    def __init__(self, source: str, img_size=640, stride=32, auto=True, shard: Tuple[int, int] = (0, 1), workers: int = 8,
                 raw: bool = False, skip_stems=None):
        """``raw``: yield the decoded image as is (the caller letterboxes on the device).  ``skip_stems``: tiles a previous run of
        the sweep has finished (manifest.DoneManifest); they keep their place in the global numbering and are simply not loaded."""
        self.raw = raw
        files = list_images(source)
        self.total = len(files)
        rank, world = shard
        self.indices = list(range(rank, len(files), world))
        self.skipped = 0
        if skip_stems:
            keep = [i for i in self.indices if Path(files[i]).stem not in skip_stems]
            self.skipped = len(self.indices) - len(keep)
            self.indices = keep
        self.files = [files[i] for i in self.indices]
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.stride, self.auto = stride, auto
        self.workers = max(1, workers)

    def __len__(self):
        return len(self.files)

    def scan_sizes(self) -> List[Tuple[int, int]]:
        """(width, height) of every image of this shard from its header (no decode), in file order."""
        from PIL import Image

        def size(path):
            with Image.open(path) as im:
                return im.size
        with ThreadPoolExecutor(self.workers) as ex:
            return list(ex.map(size, self.files))

    def scan_split_decodable(self) -> List[Optional[Tuple[int, int]]]:
        """Per file: (width, height) when the split JPEG decoder covers it (aquaculture_amd/jpeg.py: baseline 4:2:0), else None."""
        from . import jpeg
        if len(self.files) >= 16384 and self.workers > 1 and os.environ.get("AQ_SCAN_PROCS", "1") != "0":
            try:                                   # a long sweep: scanner processes (threads do not scale: the per-file Python holds the GIL)
                return jpeg.scan_files_in_processes(self.files, min(self.workers, 16))
            except Exception:                      # the threaded scan below still works
                pass
        with ThreadPoolExecutor(self.workers) as ex:
            return list(ex.map(jpeg.scan_file, self.files))

    def subset(self, positions: Sequence[int]) -> "LoadImages":
        """The same dataset restricted to some of its files (positions into ``files``); global indices travel with them."""
        import copy
        sub = copy.copy(self)
        sub.files = [self.files[i] for i in positions]
        sub.indices = [self.indices[i] for i in positions]
        return sub

    def load(self, path: str):
        im0 = read_rgb(path)
        if self.raw:
            return path, np.ascontiguousarray(im0), im0.shape[:2]
        return path, letterbox(im0, self.img_size, self.auto, True, self.stride), im0.shape[:2]

    def __iter__(self) -> Iterator[Tuple[str, np.ndarray, Tuple[int, int]]]:
        if self.workers == 1:
            for p in self.files:
                yield self.load(p)
            return
        with ThreadPoolExecutor(self.workers) as ex:   # PIL releases the GIL while decoding
            yield from ex.map(self.load, self.files)

    def batches(self, batch_size: int):
        """Groups consecutive images of equal letterboxed shape: yields (paths, uint8 [b,H,W,3], [orig shapes])."""
        paths, ims, shapes = [], [], []
        for p, im, s0 in self:
            if ims and (im.shape != ims[0].shape or len(ims) == batch_size):
                yield paths, np.stack(ims, 0), shapes
                paths, ims, shapes = [], [], []
            paths.append(p); ims.append(im); shapes.append(s0)
        if ims:
            yield paths, np.stack(ims, 0), shapes


    def pinned_batches(self, batch_size: int, n_buffers: int = 3, processes: Optional[int] = None, coef: bool = False):
        """Raw-mode fast path for tile sweeps (every image the same size, as the reference's tiler produces them): decoders write
        straight into pre-pinned uint8 batch buffers [batch, H0, W0, 3] (no stack / pin copies).
        Yields (paths, pinned torch uint8 tensor [b,H0,W0,3], [orig shapes], buffer index); the caller must be done with
        buffer ``i`` (its H2D copy enqueued AND completed) before the (i + n_buffers)-th batch is produced -- it signals
        that by calling ``release(i)`` on the returned generator's ``release`` attribute.  Images of a different size end
        the fast path with a ValueError (use ``batches`` for mixed directories).

        ``processes`` (default: ``self.workers``; 0 = decode threads in this process): decode worker PROCESSES
        (aquaculture_amd/_decode_worker.py) that write into a shared-memory ring registered as pinned host memory.  The threaded
        path stops scaling after ~2 threads -- the PIL -> numpy conversion and the copies run under the GIL (8 threads: 231
        images/s for 1024-px jpegs on 8 cores, 8 processes: 620) -- SURVEY.md 8f rank 2.

        ``coef`` (worker processes only; every file must be split-decodable, ``scan_split_decodable``): the workers undo only the
        Huffman coding and the batch is yielded as a uint8 tensor [b, jpeg.slot_bytes(H0, W0)] of coefficient blocks + quantisation
        tables; the caller turns it into pixels on the device (engine.jpeg_slots_to_rgb)."""
        import threading

        import torch
        assert self.raw, "pinned_batches is the raw (device letterbox) path"
        if not self.files:
            return
        if coef:
            from . import jpeg
            wh = jpeg.scan_file(self.files[0])
            assert wh is not None, "coef mode needs split-decodable files"
            W0, H0 = wh
        else:
            first = read_rgb(self.files[0])
            H0, W0 = first.shape[:2]
        free = [threading.Semaphore(1) for _ in range(n_buffers)]
        self.release = lambda i: free[i].release()
        nproc = self.workers if processes is None else processes
        if (nproc and nproc > 1) or coef:
            yield from self._pinned_batches_procs(batch_size, n_buffers, max(nproc or 1, 1), H0, W0, free, coef)
            return
        bufs = [torch.empty((batch_size, H0, W0, 3), dtype=torch.uint8).pin_memory() for _ in range(n_buffers)]
        views = [b.numpy() for b in bufs]

        def job(args):
            path, dst = args
            im = read_rgb(path)
            if im.shape != dst.shape:
                raise ValueError(f"{path}: {im.shape[:2]} differs from the first image {dst.shape[:2]}; mixed sizes need batches()")
            dst[...] = im

        with ThreadPoolExecutor(self.workers) as ex:
            k = 0
            for s in range(0, len(self.files), batch_size):
                paths = self.files[s:s + batch_size]
                i = k % n_buffers
                free[i].acquire()
                list(ex.map(job, [(p, views[i][j]) for j, p in enumerate(paths)]))
                yield paths, bufs[i][:len(paths)], [(H0, W0)] * len(paths), i
                k += 1

    def _pinned_batches_procs(self, batch_size, n_buffers, nproc, H0, W0, free, coef=False):
        """Decode worker processes + shared-memory ring (see pinned_batches)."""
        import subprocess
        import sys
        from multiprocessing import shared_memory

        import torch
        n_slots = n_buffers * batch_size
        if coef:
            from . import jpeg
            slot = jpeg.slot_bytes(H0, W0)
            nbytes = n_slots * slot
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
            ring = np.ndarray((n_buffers, batch_size, slot), dtype=np.uint8, buffer=shm.buf)
        else:
            nbytes = n_slots * H0 * W0 * 3
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
            ring = np.ndarray((n_buffers, batch_size, H0, W0, 3), dtype=np.uint8, buffer=shm.buf)
        ring_t = torch.from_numpy(ring)
        # Page-lock the ring in place (cudaHostRegister = hipHostRegister on ROCm) so that H2D copies from it are asynchronous DMA.
        # Measured on the MI355X box: DMA reads from a registered shm mapping run at only ~4 GB/s (hipHostMalloc memory: ~55 GB/s),
        # yet that still beats the alternative -- a memcpy per batch from the ring into a hipHostMalloc staging buffer (1,660 images/s for
        # 1024-px jpegs against 1,290 with the copy spread over 8 threads and 755 with a single copy).  The threaded staging copy
        # remains as the fallback if registration fails (AQ_DECODE_REGISTER=0 forces it).
        registered = False
        if os.environ.get("AQ_DECODE_REGISTER", "1") != "0":
            try:
                rc = torch.cuda.cudart().cudaHostRegister(ring_t.data_ptr(), nbytes, 0)
                registered = int(rc) == 0
            except Exception:
                registered = False
        stage = None if registered else [torch.empty(tuple(ring.shape[1:]), dtype=torch.uint8).pin_memory() for _ in range(n_buffers)]
        ncopy = max(1, min(8, nproc))
        copiers = None if registered else ThreadPoolExecutor(ncopy)
        nproc = max(1, min(nproc, batch_size))
        env = dict(os.environ)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):     # one decoder = one core: no BLAS / OpenMP teams in the workers
            env.setdefault(var, "1")
        procs = [subprocess.Popen([sys.executable, "-m", "aquaculture_amd._decode_worker", shm.name, str(n_slots), str(H0), str(W0)] + (["coef"] if coef else []),
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1, env=env) for _ in range(nproc)]
        try:
            # The workers run one batch AHEAD of the batch being collected: the requests of batch k + 1 go out (when its ring buffer is
            # free) before the answers of batch k are awaited, so the decoders never idle while the parent collects, uploads and launches
            # (round 3: with one batch in flight the decode stage sat idle for most of every batch interval).  A worker answers its
            # requests in order, so batch k's answers precede batch k + 1's on every pipe.
            starts = list(range(0, len(self.files), batch_size))

            def dispatch(k_, blocking):
                i_ = k_ % n_buffers
                if not free[i_].acquire(blocking=blocking):
                    return None
                paths_ = self.files[starts[k_]:starts[k_] + batch_size]
                sent_ = [0] * nproc
                for j, p in enumerate(paths_):           # round-robin over the workers; they decode concurrently
                    w = j % nproc
                    procs[w].stdin.write(f"{i_ * batch_size + j} {p}\n")
                    sent_[w] += 1
                for w in range(nproc):
                    if sent_[w]:
                        procs[w].stdin.write("flush\n")          # the worker answers once per batch: "ok <count>" or the first error
                        procs[w].stdin.flush()
                return paths_, i_, sent_

            ahead = None
            trace = os.environ.get("AQ_TRACE_LOADER") == "1"     # per batch: ms spent dispatching, waiting for the decoders' answers, and in the consumer
            import time as _time
            t_out = _time.perf_counter()
            for k in range(len(starts)):
                t_a = _time.perf_counter()
                paths, i, sent = ahead if ahead is not None else dispatch(k, True)
                ahead = dispatch(k + 1, False) if k + 1 < len(starts) else None
                t_b = _time.perf_counter()
                for w in range(nproc):
                    if not sent[w]:
                        continue
                    ans = procs[w].stdout.readline()
                    if not ans:
                        raise RuntimeError(f"decode worker {w} died (exit code {procs[w].poll()})")
                    if ans.startswith("err"):
                        raise ValueError(ans.rstrip().split(" ", 2)[2])
                    if ans.split() != ["ok", str(sent[w])]:
                        raise RuntimeError(f"decode worker {w}: unexpected answer {ans!r} to a batch of {sent[w]}")
                if registered:
                    batch = ring_t[i][:len(paths)]
                else:                                        # parallel memcpy (releases the GIL) into the hipHostMalloc buffer
                    n = len(paths)
                    step = max(1, (n + ncopy - 1) // ncopy)
                    list(copiers.map(lambda a: stage[i][a:min(a + step, n)].copy_(ring_t[i][a:min(a + step, n)]), range(0, n, step)))
                    batch = stage[i][:n]
                if trace:
                    t_c = _time.perf_counter()
                    print(f"loader batch {k}: consumer {1e3 * (t_a - t_out):.1f} ms, dispatch {1e3 * (t_b - t_a):.1f} ms (next batch {'sent' if ahead else 'NOT sent: buffer busy'}), "
                          f"answers {1e3 * (t_c - t_b):.1f} ms", flush=True)
                    t_out = _time.perf_counter()
                yield paths, batch, [(H0, W0)] * len(paths), i
                if ahead is None and k + 1 < len(starts):
                    ahead = dispatch(k + 1, True)            # its buffer was still in use a moment ago: wait for it now
        finally:
            for pr in procs:
                try:
                    pr.stdin.close()
                except Exception:
                    pass
            for pr in procs:
                try:
                    pr.wait(timeout=10)
                except Exception:
                    pr.kill()
            if copiers is not None:
                copiers.shutdown(wait=True)
            if registered:
                try:
                    torch.cuda.cudart().cudaHostUnregister(ring_t.data_ptr())
                except Exception:
                    pass
            del ring_t, ring
            shm.close()
            shm.unlink()
