#!/usr/bin/env python3
"""Where does the 4 GB/s ceiling of H2D copies from the decode workers' shared-memory ring come from?
H2D rate of 64-tile batches (640 px: 78.6 MB) from differently allocated host memory, same copy call (tensor.to(device, non_blocking=True)
/ hipMemcpyAsync), on the GPU box:
  pinned      torch pin_memory() = hipHostMalloc
  pageable    plain numpy memory (the driver stages it)
  shm+reg     multiprocessing.shared_memory (tmpfs pages) + hipHostRegister -- what dataloader.pinned_batches does today
  shm+reg+thp the same mapping after madvise(MADV_HUGEPAGE)
  anon+reg    an anonymous private mmap + hipHostRegister (not shareable with workers: tells tmpfs from registration)
  shm->pinned workers' shm copied into a hipHostMalloc buffer by N memcpy threads (numpy releases the GIL), then DMA: the staging design
Usage: python tools/h2d_bench.py"""
import ctypes
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from multiprocessing import shared_memory

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

NB = 64 * 640 * 640 * 3
REPS = 20


def rate(host_t, dev_t, label, pre=None):
    torch.cuda.synchronize()
    for _ in range(2):
        if pre:
            pre()
        dev_t.copy_(host_t, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        if pre:
            pre()
        dev_t.copy_(host_t, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS
    print(f"{label:34s} {NB / dt / 1e9:7.2f} GB/s  ({dt * 1e3:.2f} ms per 64-tile batch = {64 / dt:,.0f} tiles/s)", flush=True)


def main():
    dev = torch.device("cuda", 0)
    d = torch.empty(NB, dtype=torch.uint8, device=dev)
    rt = torch.cuda.cudart()
    src = np.random.default_rng(0).integers(0, 255, NB, dtype=np.uint8)
    pinned = torch.from_numpy(src.copy()).pin_memory()
    rate(pinned, d, "pinned (hipHostMalloc)")
    rate(torch.from_numpy(src.copy()), d, "pageable numpy")
    shm = shared_memory.SharedMemory(create=True, size=NB)
    a = np.ndarray((NB,), dtype=np.uint8, buffer=shm.buf)
    a[:] = src
    t = torch.from_numpy(a)
    rate(t, d, "shm, not registered")
    rc = rt.cudaHostRegister(t.data_ptr(), NB, 0)
    print("  hipHostRegister(shm) ->", rc)
    rate(t, d, "shm + hipHostRegister")
    rt.cudaHostUnregister(t.data_ptr())
    try:
        libc = ctypes.CDLL("libc.so.6", use_errno=True)
        r = libc.madvise(ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(NB), 14)     # MADV_HUGEPAGE
        print("  madvise(MADV_HUGEPAGE) ->", r, os.strerror(ctypes.get_errno()) if r else "")
        a[:] = src
        rt.cudaHostRegister(t.data_ptr(), NB, 0)
        rate(t, d, "shm + MADV_HUGEPAGE + register")
        rt.cudaHostUnregister(t.data_ptr())
    except Exception as e:
        print("  thp variant failed:", e)
    m = mmap.mmap(-1, NB + (2 << 20))
    b = np.frombuffer(m, dtype=np.uint8)
    off = (-b.ctypes.data) % (2 << 20)
    b = b[off:off + NB]
    b[:] = src
    tb = torch.from_numpy(b)
    rt.cudaHostRegister(tb.data_ptr(), NB, 0)
    rate(tb, d, "anonymous mmap + register")
    rt.cudaHostUnregister(tb.data_ptr())
    for nth in (1, 2, 4, 8):
        stage = torch.empty(NB, dtype=torch.uint8).pin_memory()
        sn = stage.numpy()
        cuts = [(i * NB // nth, (i + 1) * NB // nth) for i in range(nth)]
        ex = ThreadPoolExecutor(nth)

        def pre():
            list(ex.map(lambda c: np.copyto(sn[c[0]:c[1]], a[c[0]:c[1]]), cuts))
        rate(stage, d, f"shm -> pinned by {nth} memcpy thread(s)", pre)
        ex.shutdown()
    del t, a
    shm.close()
    shm.unlink()


if __name__ == "__main__":
    main()
