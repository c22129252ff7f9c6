"""jpeg decode worker process (decode scaling).

Started by dataloader.LoadImages.pinned_batches as `python -m aquaculture_amd._decode_worker <shm name> <n_slots> <H> <W> [coef]`:
a plain child process (never a fork of the GPU process) that imports only the standard library, numpy and PIL -- no torch, no HIP.
Protocol: parent -> `<slot> <path>` per image, then `flush` after a batch's last one; the worker decodes each image as RGB uint8 straight
into slot `slot` of the shared-memory ring [n_slots][H][W][3] and answers the `flush` with ONE line, `ok <images decoded>` or
`err <slot> <path>: <message>` for the first failure of the batch (one pipe read per worker and batch on the parent's single thread, not one
per image).  EOF on stdin ends the worker.

[UPSTREAM utils/dataloaders.py LoadImages.__next__ -> cv2.imread]: same decoded pixels as the threaded path (PIL / libjpeg-turbo).

Mode `coef` (round 3, the split decode): the worker undoes only the Huffman coding (libaqjpeg.so through aquaculture_amd/jpeg.py) and
writes the image's quantised DCT coefficient blocks and quantisation tables into its slot of a byte ring [n_slots][jpeg.slot_bytes(H, W)];
the GPU does the rest (aq_jpeg_idct_rgb), byte-identical to this process's PIL decode.  The parent only selects this mode for file sets
whose headers say baseline 4:2:0 throughout; a file that still fails here is an error, as a PIL failure is.
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
    coef_mode = len(sys.argv) > 5 and sys.argv[5] == "coef"
    if coef_mode:
        from aquaculture_amd import jpeg
        nco = jpeg.coef_count(H, W)
        ring = np.ndarray((n_slots, jpeg.slot_bytes(H, W)), dtype=np.uint8, buffer=shm.buf)
    else:
        ring = np.ndarray((n_slots, H, W, 3), dtype=np.uint8, buffer=shm.buf)
    out = sys.stdout
    done, failed = 0, None             # answers are held back until the parent's "flush" line: one write + one read per worker and batch
    for line in sys.stdin:
        line = line.rstrip("\n")
        if not line:
            continue
        if line == "flush":
            out.write(failed if failed is not None else f"ok {done}\n")
            out.flush()
            done, failed = 0, None
            continue
        if failed is not None:         # the batch is lost already: skip the rest of it
            continue
        slot_s, path = line.split(" ", 1)
        slot = int(slot_s)
        try:
            if coef_mode:
                with open(path, "rb") as f:
                    data = f.read()
                rc, info = jpeg.decode_coeffs(data, ring[slot, :2 * nco].view(np.int16), ring[slot, 2 * nco:2 * nco + 384].view(np.uint16).reshape(3, 64))
                if rc != 0:
                    raise ValueError(f"split JPEG decoder status {rc} (-1 unsupported coding, -2 corrupt data)")
                if (info.width, info.height) != (W, H):
                    raise ValueError(f"{info.height}x{info.width} differs from the first image {H}x{W}; mixed sizes need batches()")
                done += 1
                continue
            with Image.open(path) as im:
                if im.mode != "RGB":
                    im = im.convert("RGB")
                if im.size != (W, H):
                    raise ValueError(f"{im.size[1]}x{im.size[0]} differs from the first image {H}x{W}; mixed sizes need batches()")
                im.load()
                ring[slot] = np.asarray(im)
            done += 1
        except Exception as e:  # reported to the parent (at the batch's flush), which raises
            failed = f"err {slot} {path}: {type(e).__name__}: {e}".replace("\r", " ").replace("\n", " ") + "\n"
    del ring
    shm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
