"""jpeg decode worker process (decode scaling).

Started by dataloader.LoadImages.pinned_batches as `python -m aquaculture_amd._decode_worker <shm name> <n_slots> <H> <W>`:
a plain child process (never a fork of the GPU process) that imports only the standard library, numpy and PIL -- no torch, no HIP.
Protocol, one line each way: parent -> `<slot> <path>`; worker decodes the image as RGB uint8 straight into slot `slot` of the
shared-memory ring [n_slots][H][W][3] and answers `ok <slot>` or `err <slot> <message>`.  EOF on stdin ends the worker.

[UPSTREAM utils/dataloaders.py LoadImages.__next__ -> cv2.imread]: same decoded pixels as the threaded path (PIL / libjpeg-turbo).
"""
import sys
from multiprocessing import shared_memory

import numpy as np
from PIL import Image


def main() -> int:
    name, n_slots, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    shm = shared_memory.SharedMemory(name=name)
    try:
        # the parent owns the segment: keep Python's resource tracker from unlinking it when this process exits
        from multiprocessing import resource_tracker
        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:
        pass
    ring = np.ndarray((n_slots, H, W, 3), dtype=np.uint8, buffer=shm.buf)
    out = sys.stdout
    for line in sys.stdin:
        line = line.rstrip("\n")
        if not line:
            continue
        slot_s, path = line.split(" ", 1)
        slot = int(slot_s)
        try:
            with Image.open(path) as im:
                if im.mode != "RGB":
                    im = im.convert("RGB")
                if im.size != (W, H):
                    raise ValueError(f"{im.size[1]}x{im.size[0]} differs from the first image {H}x{W}; mixed sizes need batches()")
                im.load()
                ring[slot] = np.asarray(im)
            out.write(f"ok {slot}\n")
        except Exception as e:  # reported to the parent, which raises
            out.write(f"err {slot} {type(e).__name__}: {e}\n".replace("\r", " "))
        out.flush()
    del ring
    shm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
