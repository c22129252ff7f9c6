"""Split JPEG decode: entropy decoding on the host (libaqjpeg.so, csrc/jpeg_coef.c), everything after it on the GPU (aq_jpeg_idct_rgb).

[UPSTREAM utils/dataloaders.py LoadImages.__next__ -> cv2.imread] spends most of a host core's 2-3 ms per 640-px tile in the inverse DCT,
the chroma upsampling and the colour conversion; the tiles of reference src/load_data/tile_tifs.py:66-74 are baseline 4:2:0 JPEGs, for
which those stages are restated bit for bit on the device (csrc/jpeg_idct.hip).  This module is the host side: it imports neither torch
nor the HIP library, so the decode worker processes can use it (aquaculture_amd/_decode_worker.py, mode "coef").
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

AQJ_OK, AQJ_UNSUPPORTED, AQJ_CORRUPT, AQJ_SPACE = 0, -1, -2, -3
_LIB = None


class JpegInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32), ("mcu_cols", C.c_int32), ("mcu_rows", C.c_int32),
                ("y_blocks_w", C.c_int32), ("y_blocks_h", C.c_int32), ("total_blocks", C.c_int32), ("qt", (C.c_uint16 * 64) * 3)]


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libaqjpeg.so")


def load_lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(lib_path())
        lib.aq_jpeg_decode_coeffs.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(JpegInfo)]
        lib.aq_jpeg_scan.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(JpegInfo)]
        lib.aq_jpeg_prepare_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_uint64,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.aq_jpeg_prepare.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(StreamInfo)]
        _LIB = lib
    return _LIB


class GpuTab(C.Structure):
    _fields_ = [("look", C.c_uint16 * 512), ("maxcode", C.c_int32 * 18), ("valoff", C.c_int32 * 18), ("vals", C.c_uint8 * 256)]


class StreamInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("mcu_cols", C.c_int32), ("mcu_rows", C.c_int32), ("restart", C.c_int32),
                ("nseg", C.c_int32), ("stream_bytes", C.c_uint32), ("pad", C.c_uint32), ("tab_hash", C.c_uint64), ("qt", (C.c_uint16 * 64) * 3),
                ("tabs", GpuTab * 6)]


SEG_DTYPE = np.dtype([("stream_off", "<u4"), ("stream_len", "<u4"), ("coef_off", "<u8"), ("mcu0", "<u4"), ("n_mcu", "<u4"),
                      ("mcu_cols", "<u2"), ("mcu_rows", "<u2"), ("tabset", "<u4")])
assert SEG_DTYPE.itemsize == 32
TABSET_BYTES = 6 * C.sizeof(GpuTab)


def coef_count(H: int, W: int) -> int:
    """int16 values of one image's coefficient blocks (4:2:0): 1.5 x the padded pixel count."""
    return ((H + 15) // 16 * 16) * ((W + 15) // 16 * 16) * 3 // 2


def slot_bytes(H: int, W: int) -> int:
    """Bytes of one image's slot in the decode workers' ring: coefficient blocks (int16), then the three quantisation tables (384 B),
    padded so that slots stay 128-byte aligned."""
    return 2 * coef_count(H, W) + 512


def scan_file(path: str) -> Optional[Tuple[int, int]]:
    """(width, height) when the split decoder covers the file (baseline / extended sequential Huffman, 8-bit, YCbCr 4:2:0), else None."""
    try:
        with open(path, "rb", buffering=0) as f:
            head = f.read(4096)                 # the tables and frame header sit in front of the scan: ~600 bytes for a GDAL / PIL file
            info = scan(head)                   # (a sweep scans every file: 64 KB each was 8.6 GB of page-cache copies for 131 k tiles)
            if info is None and len(head) == 4096:
                head += f.read((1 << 16) - 4096)        # APPn / ICC segments in front: up to 64 KB, then the whole file
                info = scan(head)
                if info is None and len(head) == (1 << 16):
                    info = scan(head + f.read())
        return (info.width, info.height) if info is not None else None
    except OSError:
        return None


def scan_files_in_processes(paths, nproc: int) -> list:
    """scan_file over a long list in `nproc` child processes (`python -m aquaculture_amd.jpeg`: paths on stdin, "<w> <h>" or "-" per line
    back) -- plain children that import numpy and ctypes only, never a fork of a process that may have initialised the GPU.  The per-file
    Python (open, read, ctypes call) holds the GIL, so threads do not scale: 46 us per file, 6 s for a 131 k-tile sweep in one process."""
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor
    nproc = max(1, min(nproc, len(paths)))
    step = (len(paths) + nproc - 1) // nproc
    chunks = [paths[i:i + step] for i in range(0, len(paths), step)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")

    def run(chunk):
        r = subprocess.run([sys.executable, "-m", "aquaculture_amd.jpeg"], input="\n".join(chunk) + "\n", capture_output=True, text=True, env=env)
        lines = r.stdout.splitlines()
        if r.returncode != 0 or len(lines) != len(chunk):
            raise RuntimeError(f"jpeg header scanner failed (exit code {r.returncode}): {r.stderr[-500:]}")
        return [None if l == "-" else tuple(int(v) for v in l.split()) for l in lines]

    with ThreadPoolExecutor(len(chunks)) as ex:
        return [x for part in ex.map(run, chunks) for x in part]


def scan(data: bytes) -> Optional[JpegInfo]:
    """Headers of a baseline 4:2:0 JPEG, or None when the split decoder does not cover the file (the caller decodes it in software)."""
    info = JpegInfo()
    return info if load_lib().aq_jpeg_scan(data, len(data), C.byref(info)) == AQJ_OK and info.ncomp == 3 else None


def decode_coeffs(data: bytes, coef_out: np.ndarray, qt_out: np.ndarray) -> Tuple[int, JpegInfo]:
    """Entropy-decode `data` into coef_out (int16, C-contiguous, >= coef_count values) and its three quantisation tables into qt_out
    (uint16 [3][64]).  Returns (status, info)."""
    info = JpegInfo()
    rc = load_lib().aq_jpeg_decode_coeffs(data, len(data), coef_out.ctypes.data, coef_out.size, C.byref(info))
    if rc == AQJ_OK:
        if info.ncomp != 3:
            return AQJ_UNSUPPORTED, info
        qt_out[...] = np.ctypeslib.as_array(info.qt).reshape(3, 64)
    return rc, info


class GpuDecodeBatch:
    """Host side of the GPU entropy decode for a batch of same-size JPEG files: every file's scan -- byte stuffing removed, cut at restart
    markers -- back to back in one (page-locked, if the caller passes such a buffer) upload buffer, the 32-byte segment descriptors the
    device kernel reads (aq_jpeg_huffman_decode, include/aq_engine.h), the quantisation tables, and the distinct Huffman table sets.
    ``add`` is thread-safe per slot: callers prepare different slots from different threads (the C call releases the GIL)."""

    MAX_SEG = 4096                     # restart segments per image (a DRI of one MCU row on a 1024-px tile gives 64)

    def __init__(self, n_images: int, H: int, W: int, stream_buf: Optional[np.ndarray] = None, bytes_per_image: Optional[int] = None):
        self.n, self.H, self.W = n_images, H, W
        self.per = bytes_per_image or stream_capacity(H, W)
        self.streams = stream_buf if stream_buf is not None else np.zeros(n_images * self.per + 256, np.uint8)      # (+ 256: the device reads whole 64-byte chunks, two ahead)
        assert self.streams.dtype == np.uint8 and self.streams.size >= n_images * self.per + 256
        self.qt = np.zeros((n_images, 3, 64), np.uint16)
        self.nco = coef_count(H, W)
        self._segs = [None] * n_images
        self._hash = [0] * n_images
        self._tabs = [None] * n_images

    def add(self, slot: int, data: bytes) -> int:
        """Prepare image `slot` from the file's bytes.  Returns AQJ_OK or the status that makes the caller fall back / fail."""
        info = StreamInfo()
        off = np.zeros(self.MAX_SEG, np.uint32)
        ln = np.zeros(self.MAX_SEG, np.uint32)
        base = slot * self.per
        rc = load_lib().aq_jpeg_prepare(data, len(data), self.streams[base:].ctypes.data, self.per, off.ctypes.data, ln.ctypes.data, self.MAX_SEG, C.byref(info))
        if rc != AQJ_OK:
            return rc
        if (info.width, info.height) != (self.W, self.H):
            raise ValueError(f"{info.height}x{info.width} differs from the batch's {self.H}x{self.W}; mixed sizes need batches()")
        n = info.nseg
        segs = np.zeros(n, SEG_DTYPE)
        segs["stream_off"] = off[:n] + base
        segs["stream_len"] = ln[:n]
        segs["coef_off"] = slot * self.nco
        per_seg = info.restart if info.restart else info.mcu_cols * info.mcu_rows
        segs["mcu0"] = np.arange(n, dtype=np.uint32) * per_seg
        segs["n_mcu"] = np.minimum(per_seg, info.mcu_cols * info.mcu_rows - segs["mcu0"])
        segs["mcu_cols"], segs["mcu_rows"] = info.mcu_cols, info.mcu_rows
        self.qt[slot] = np.ctypeslib.as_array(info.qt).reshape(3, 64)
        self._segs[slot], self._hash[slot] = segs, int(info.tab_hash)
        self._tabs[slot] = bytes(info.tabs)
        return AQJ_OK

    SEG_CAP = 64                       # restart segments per image in the bulk path (prepare_files)

    def prepare_files(self, paths, nthreads: int = 8):
        """Slots 0 .. len(paths) - 1 from files, in C threads (aq_jpeg_prepare_files: file reads, header parsing, byte-stuffing removal -- no
        interpreter per image).  Returns (segs, table sets, first segment per image) like ``finish``; raises ValueError naming the first file
        the decoder does not take."""
        n = len(paths)
        assert 0 < n <= self.n
        if not hasattr(self, "_bulk"):
            self._bulk = (np.zeros((self.n, self.SEG_CAP), SEG_DTYPE), np.zeros(self.n, np.int32), np.zeros(self.n, np.int32),
                          np.zeros(self.n, np.uint64), np.zeros((self.n, TABSET_BYTES), np.uint8))
        segs, status, nseg, hashes, tabs = self._bulk
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        bad = load_lib().aq_jpeg_prepare_files(arr, n, self.H, self.W, self.streams.ctypes.data, self.per, segs.ctypes.data, self.SEG_CAP, self.nco,
                                               status.ctypes.data, nseg.ctypes.data, self.qt.ctypes.data, hashes.ctypes.data, tabs.ctypes.data, nthreads)
        if bad:
            i = int(np.nonzero(status[:n])[0][0]) if bad > 0 else 0
            why = {AQJ_UNSUPPORTED: "not a baseline 8-bit 4:2:0 JPEG of the batch's size", AQJ_CORRUPT: "corrupt or truncated (libjpeg / Pillow refuse the file too)",
                   AQJ_SPACE: f"more than {self.per * 8 / (self.H * self.W):.1f} bits per pixel or more than {self.SEG_CAP} restart segments: raise "
                              "AQ_JPEG_GPU_BPP or use --jpeg-decode split"}.get(int(status[i]), f"status {int(status[i])}")
            raise ValueError(f"{paths[i]}: GPU JPEG decode preparation failed: {why}")
        uniq, first_idx, inverse = np.unique(hashes[:n], return_index=True, return_inverse=True)
        counts = nseg[:n]
        if (counts == 1).all():
            out = segs[:n, 0].copy()
            out["tabset"] = inverse
        else:
            keep = np.arange(self.SEG_CAP)[None, :] < counts[:, None]
            out = segs[:n][keep]
            out["tabset"] = np.repeat(inverse, counts)
        first = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        return out, tabs[first_idx].copy(), first

    def finish(self, count: Optional[int] = None):
        """(segment descriptors [nseg] of SEG_DTYPE, table sets uint8 [n_sets][TABSET_BYTES], first segment of every image [count + 1]) for
        images 0 .. count - 1, table sets de-duplicated by hash."""
        count = self.n if count is None else count
        sets, index = [], {}
        parts, first = [], [0]
        for i in range(count):
            h = self._hash[i]
            if h not in index:
                index[h] = len(sets)
                sets.append(np.frombuffer(self._tabs[i], np.uint8))
            sg = self._segs[i]
            sg["tabset"] = index[h]
            parts.append(sg)
            first.append(first[-1] + sg.shape[0])
        return np.concatenate(parts), np.stack(sets), np.asarray(first, np.int64)


def stream_capacity(H: int, W: int, bits_per_pixel: Optional[float] = None) -> int:
    """Upload-buffer bytes reserved per image (a multiple of 128).  Default 2 bits per pixel + slack (AQ_JPEG_GPU_BPP overrides): GDAL's
    q = 75 tiles take 0.3 (smooth water) to 1.1 (detailed imagery) bits per pixel; a file that needs more makes GpuDecodeBatch.add return
    AQJ_SPACE and the caller decodes the sweep with the host decoder instead (tests pass bits_per_pixel = 12: q = 100 noise)."""
    bpp = bits_per_pixel if bits_per_pixel is not None else float(os.environ.get("AQ_JPEG_GPU_BPP", 2.0))
    px = (H + 15) // 16 * 16 * ((W + 15) // 16 * 16)
    return (int(px * bpp / 8) + 8192 + 127) // 128 * 128


if __name__ == "__main__":                      # header scanner process (scan_files_in_processes)
    import sys
    out = []
    for line in sys.stdin:
        r = scan_file(line.rstrip("\n"))
        out.append("-" if r is None else f"{r[0]} {r[1]}")
    sys.stdout.write("\n".join(out) + ("\n" if out else ""))
