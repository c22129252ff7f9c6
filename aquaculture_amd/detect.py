"""``yolov5/detect.py`` re-implemented on the HIP engine: same flags, same output files.

Reference call (reference README.md:77):

    python3 yolov5/detect.py --weights output/model_weights/multilabel_farms_exp2.pt \\
        --source data/jpegs --nosave --save-txt --save-conf

Output contract: ``<project>/<name>[N]/labels/<image stem>.txt``, one line per detection
``cls xc yc w h conf`` (``%g`` each, normalised by the original image size, ascending confidence), no file
for an image without detections -- consumed unchanged by reference src/process_yolo/geocode_results.py:123-172.

What differs from upstream's loop ([UPSTREAM detect.py run()]): images are decoded by a thread pool and run
through the engine in batches (``--batch-size``); with ``WORLD_SIZE > 1`` (torchrun) every rank takes the tiles
``i % world == rank`` and writes its own label files; detections are gathered over RCCL once at the end.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from . import dist as aqdist
from . import postprocess
from .checkpoint import load_checkpoint
from .dataloader import LoadImages, check_img_size

UNSUPPORTED = ("view_img", "save_crop", "augment", "visualize", "update", "dnn")


def increment_path(path, exist_ok=False, sep="", mkdir=False) -> Path:
    """[UPSTREAM utils/general.py increment_path]: runs/detect/exp -> exp2, exp3, ..."""
    path = Path(path)
    if path.exists() and not exist_ok:
        for n in range(2, 9999):
            p = f"{path}{sep}{n}"
            if not os.path.exists(p):
                break
        path = Path(p)
    if mkdir:
        path.mkdir(parents=True, exist_ok=True)
    return path


def parse_opt(argv: Optional[List[str]] = None) -> argparse.Namespace:
    """Upstream's flag set plus --batch-size / --precision / --workers."""
    root = Path.cwd()
    p = argparse.ArgumentParser()
    p.add_argument("--weights", nargs="+", type=str, default="yolov5s.pt", help="model path")
    p.add_argument("--source", type=str, default="data/images", help="file/dir/glob")
    p.add_argument("--data", type=str, default="data/coco128.yaml", help="(unused: class names come from the checkpoint)")
    p.add_argument("--imgsz", "--img", "--img-size", nargs="+", type=int, default=[640], help="inference size h,w")
    p.add_argument("--conf-thres", type=float, default=0.25, help="confidence threshold")
    p.add_argument("--iou-thres", type=float, default=0.45, help="NMS IoU threshold")
    p.add_argument("--max-det", type=int, default=1000, help="maximum detections per image")
    p.add_argument("--device", default="", help="GPU ordinal (default: LOCAL_RANK or 0)")
    p.add_argument("--view-img", action="store_true")
    p.add_argument("--save-txt", action="store_true", help="save results to *.txt")
    p.add_argument("--save-conf", action="store_true", help="save confidences in --save-txt labels")
    p.add_argument("--save-crop", action="store_true")
    p.add_argument("--nosave", action="store_true", help="do not save images/videos")
    p.add_argument("--classes", nargs="+", type=int, help="filter by class: --classes 0, or --classes 0 2 3")
    p.add_argument("--agnostic-nms", action="store_true")
    p.add_argument("--augment", action="store_true")
    p.add_argument("--visualize", action="store_true")
    p.add_argument("--update", action="store_true")
    p.add_argument("--project", default=str(root / "runs/detect"), help="save results to project/name")
    p.add_argument("--name", default="exp", help="save results to project/name")
    p.add_argument("--exist-ok", action="store_true", help="existing project/name ok, do not increment")
    p.add_argument("--line-thickness", default=3, type=int)
    p.add_argument("--hide-labels", default=False, action="store_true")
    p.add_argument("--hide-conf", default=False, action="store_true")
    p.add_argument("--half", action="store_true", help="reduced precision (bf16 on MI355X; upstream: fp16)")
    p.add_argument("--dnn", action="store_true")
    p.add_argument("--vid-stride", type=int, default=1)
    p.add_argument("--batch-size", type=int, default=64, help="tiles per engine call")
    p.add_argument("--precision", choices=("fp32", "bf16", "fp8w", "f16x3", "fp8"), default=None,
                   help="default fp32 (detect.py without --half); fp8w = OCP e4m3 weights with per-channel power-of-two scales, bf16 activations")
    p.add_argument("--workers", type=int, default=8, help="jpeg decoders (worker processes for uniform tile directories, threads otherwise)")
    p.add_argument("--decode-threads", action="store_true", help="decode in threads of this process instead of worker processes")
    p.add_argument("--quiet", action="store_true", help="no per-image log line (the summary lines are still printed)")
    p.add_argument("--geocode-bboxes", default=None, metavar="CSV",
                   help="reference data/wanted_bboxes.csv: after the sweep, geocode every detection (the arithmetic of the "
                        "reference's src/process_yolo/geocode_results.py geocode_all_detections, as one batch op) -> --geocode-out")
    p.add_argument("--geocode-out", default=None, metavar="GEOJSON", help="default <save_dir>/detections.geojson")
    p.add_argument("--autotune", choices=("auto", "on", "off"), default="auto",
                   help="time the conv kernels' tile configurations on the first full batch and keep the fastest per layer "
                        "(cached in $AQ_TUNE_CACHE or ~/.cache/aquaculture_amd/); auto = only for sweeps of >= 8 batches per GPU")
    p.add_argument("--jpeg-decode", choices=("auto", "host", "split", "gpu"), default="auto",
                   help="split: the decode workers undo only the Huffman coding, the GPU does the inverse DCT, chroma upsampling and colour "
                        "conversion (byte-identical to libjpeg-turbo; baseline 4:2:0 JPEGs, which is what the reference's tiler writes); "
                        "host: full software decode in the workers; "
                        "gpu: the Huffman stage on the GPU as well (one lane per image, super-batches of AQ_JPEG_GPU_SUPERBATCH = 2048 tiles in "
                        "flight): the host only reads the files and strips byte stuffing, H2D carries the files' entropy-coded bytes; "
                        "auto: split when every image of the sweep qualifies, gpu when this rank's share is also >= AQ_JPEG_GPU_AUTO_MIN = "
                        "32768 images (measured on 65,536 1024-px tiles: gpu 10.4 k images/s steady and 0.84 s of start-up against split's 9.4 k "
                        "and 0.48 s -- 9.19 k vs 8.81 k over the whole sweep, break-even near 35 k images; same label bytes)")
    p.add_argument("--resume", action="store_true",
                   help="continue an interrupted sweep in project/name (implies --exist-ok): tiles recorded in the run directory's "
                        "done.rank*.txt manifests are skipped, also those that produced no label file")
    p.add_argument("--tile-scenes", nargs="?", type=int, const=1024, default=0, metavar="TILESIZE",
                   help="opt-in scene mode: --source holds whole scene rasters (*.tif); they are cut into TILESIZE (default 1024) tiles "
                        "on the GPU, in the order and with the names of reference src/load_data/tile_tifs.py, skipping its jpeg step")
    opt = p.parse_args(argv)
    opt.imgsz *= 2 if len(opt.imgsz) == 1 else 1
    return opt


def run(weights, source, imgsz=(640, 640), conf_thres=0.25, iou_thres=0.45, max_det=1000, device="",
        save_txt=False, save_conf=False, nosave=False, classes=None, agnostic_nms=False,
        project="runs/detect", name="exp", exist_ok=False, half=False, batch_size=64, precision=None,
        workers=8, decode_threads=False, quiet=False, geocode_bboxes=None, geocode_out=None, tile_scenes=0, autotune="auto", resume=False,
        jpeg_decode="auto", log=print, **unsupported):
    from .engine import Engine, format_label_rows, write_label_files, jpeg_idct_rgb, jpeg_slots_to_rgb, letterbox_device, letterbox_scene_tiles   # raises if the HIP library or the GPU is missing: there is no fallback

    for k in UNSUPPORTED:
        if unsupported.get(k):
            raise NotImplementedError(f"--{k.replace('_', '-')} is not part of the tile-sweep path (reference README.md:77)")
    if not nosave:
        log("note: annotated images are never written (the reference runs with --nosave)")
    weights = weights[0] if isinstance(weights, (list, tuple)) else weights
    precision = precision or ("bf16" if half else "fp32")

    # (yolov5/detect.py has set OMP_NUM_THREADS / OPENBLAS_NUM_THREADS / MKL_NUM_THREADS to 1 before numpy and torch were loaded: the CPU
    # side of the sweep wants no thread teams; here the same for a caller that imported torch first)
    torch.set_num_threads(max(1, int(os.environ.get("AQ_CPU_THREADS", "1"))))
    rank, world, local = aqdist.init()
    multi = world > 1 or aqdist.forced()                  # the collectives run (AQ_DIST_FORCE=1: also in a world of one -- RCCL on a one-GPU box)
    dev = int(device) if str(device).strip().isdigit() else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)

    # directories: rank 0 picks the run directory, everyone uses it
    if resume and tile_scenes:
        raise NotImplementedError("--resume works on tile directories (the manifest records tile names), not with --tile-scenes")
    if rank == 0:
        save_dir = increment_path(Path(project) / name, exist_ok=exist_ok or resume)
        (save_dir / "labels" if save_txt else save_dir).mkdir(parents=True, exist_ok=True)
    if multi:
        box = [str(save_dir) if rank == 0 else None]
        torch.distributed.broadcast_object_list(box, src=0)
        save_dir = Path(box[0])
    labels_dir = str(save_dir / "labels")
    # done-manifest: every rank records the tiles it has finished (fsync'd per batch); --resume skips what ANY rank of the
    # interrupted run recorded (reference idiom: skip-if-exists, src/load_data/tile_tifs.py:40-41 -- which label files cannot express)
    from .manifest import DoneManifest, check_run_params, file_digest, fsync_dir, sync_filesystem_of
    durable = not tile_scenes and os.environ.get("AQ_NO_FSYNC") != "1"     # (AQ_NO_FSYNC=1: only process kills are covered, as before round 3)
    # what the label bytes depend on; --resume refuses to continue a directory written with anything else.  Collective: nobody processes
    # a tile before rank 0 has accepted the directory, and a refusal reaches every rank.
    aqdist.on_rank0(lambda: None if tile_scenes else check_run_params(
        str(save_dir), {"weights_sha256": file_digest(weights) if os.path.isfile(str(weights)) else str(weights),
                        "conf_thres": float(conf_thres), "iou_thres": float(iou_thres), "max_det": int(max_det),
                        "imgsz": [int(v) for v in imgsz], "precision": precision, "save_conf": bool(save_conf),
                        **({"classes": sorted(int(c) for c in classes)} if classes is not None else {}),
                        **({"agnostic_nms": True} if agnostic_nms else {})}, resume))
    done_before = DoneManifest.load(str(save_dir)) if resume else set()
    manifest = DoneManifest(str(save_dir), rank)
    if not tile_scenes:
        manifest.open()

    ck = load_checkpoint(weights)
    eng = Engine(ck, precision, dev, fp8_calibration="defer")
    if classes is not None or agnostic_nms:
        eng.set_nms_options(agnostic=agnostic_nms, classes=classes)            # [UPSTREAM non_max_suppression(pred, conf, iou, classes, agnostic_nms, max_det)]
    imgsz = check_img_size(list(imgsz), s=int(max(ck.stride)))
    if precision == "fp8":
        # e4m3 activation scales from THIS sweep's imagery and tile size (the engine's default calibrates on synthetic 640-px
        # tiles; real tiles beyond that range would saturate silently).  Rank 0 calibrates on tiles sampled across the whole file list, every
        # rank installs the same scales, and they are recorded next to run_params.json so that --resume quantises the rest of the sweep
        # exactly as the interrupted run did.
        import json
        rec = save_dir / "fp8_scales.json"

        def calibrate():
            if resume and rec.exists():
                return json.loads(rec.read_text())
            sc = eng.calibrate_fp8(_calibration_tiles(source, tile_scenes, imgsz, int(max(ck.stride)), dev))
            tmp = str(rec) + ".tmp"
            with open(tmp, "w") as f:
                json.dump(sc, f, indent=1, sort_keys=True)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, rec)
            return sc
        scales = aqdist.on_rank0(calibrate)
        eng.set_fp8_scales(scales)
        if rank == 0:
            log(f"fp8: {len(scales)} activation scales calibrated on this sweep's tiles (recorded in {rec}); "
                f"|t| range covered {min(scales.values()) * 448:.3g} .. {max(scales.values()) * 448:.3g}" if scales else
                "fp8: no layer of this model runs in fp8 at this geometry (bf16 throughout)")
    # decoded images go to the GPU as they are; the letterbox (resize INTER_LINEAR + pad 114) runs on the device
    if tile_scenes:
        from .scenes import SceneTiles
        dataset = SceneTiles(source, tilesize=int(tile_scenes), shard=(rank, world), workers=workers, batch_size=batch_size, pinned=True)
    else:
        dataset = LoadImages(source, img_size=imgsz, stride=int(max(ck.stride)), auto=True, shard=(rank, world), workers=workers, raw=True,
                             skip_stems=done_before)
        if resume:
            log(f"resume: {len(done_before)} tiles recorded as done in {save_dir}; rank {rank} skips {dataset.skipped} of its share")

    # Pipeline: decode threads -> [main thread: H2D, letterbox, engine, async D2H] -> [writer thread: rescale, format, files].
    # Up to `depth` batches are in flight, each with its own workspace slot, pinned result buffers and stream.
    import queue
    import threading
    depth = max(2, int(os.environ.get("AQ_PIPELINE_DEPTH", 3)))
    streams = [torch.cuda.Stream() for _ in range(depth)]
    slot_free = [threading.Semaphore(1) for _ in range(depth)]   # a slot's pinned result buffers are reused only after its writer is done
    stats = {"seen": 0, "labels": 0, "dets": 0, "t_post": 0.0}
    gather = aqdist.DetectionGather(dev, keep=False)      # rank 0 counts what arrives; the label files are the product
    gather_lock = threading.Lock()
    flush_every = 16                                       # batches between flushes of the detection gather
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    err: List[BaseException] = []

    def writer():
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                ev, counts_h, dets_h, paths, shapes0, hw, gidx, t_inf, slot_id = item
                ev.synchronize()
                t0 = time.perf_counter()
                nlab = ndet = 0
                released = False
                H, W = hw
                cnt = counts_h.numpy()
                det_all = dets_h.numpy()
                written = []
                same = all(sh == shapes0[0] for sh in shapes0)       # the pinned path: one original size per batch -> one numpy pass for all tiles
                if same:
                    rows_all, offs = postprocess.batch_rows(det_all, cnt[:len(paths)], (H, W), shapes0[0])
                bulk = same and quiet and save_txt                  # a whole batch of label files in ONE C call (no interpreter lock held; round 4)
                if bulk:
                    ndet = int(offs[-1])
                    if multi and ndet:                              # the batch's rows for the detection gather, in one piece
                        cnt_b = cnt[:len(paths)].astype(np.int64)
                        keep_ = np.arange(det_all.shape[1])[None, :] < cnt_b[:, None]
                        with gather_lock:
                            gather.add(torch.from_numpy(np.repeat(np.asarray(gidx, np.int32), cnt_b)), aqdist.pack_rows(torch.from_numpy(det_all[:len(paths)][keep_].copy())))
                    # the slot's pinned result buffers have been read (batch_rows copied what it needs): hand the slot back BEFORE formatting and
                    # writing -- ~20 ms of C per batch with the synthetic checkpoint's 340 detections per tile; held across it, three slots capped the
                    # sweep at ~5.5 k tiles/s whatever the decode path (round 4)
                    slot_free[slot_id].release()
                    released = True
                    fs_fallback = durable and not can_syncfs[0]
                    nlab = write_label_files(labels_dir, [os.path.splitext(os.path.basename(p))[0] for p in paths], rows_all, offs, save_conf, fsync=fs_fallback)
                    if durable and nlab and can_syncfs[0] and not sync_filesystem_of(labels_dir):
                        can_syncfs[0] = False               # (no syncfs here: this batch's files are synced one by one now, later batches inside the C call)
                        for p in paths:
                            fp_ = os.path.join(labels_dir, os.path.splitext(os.path.basename(p))[0] + ".txt")
                            if os.path.exists(fp_):
                                fd_ = os.open(fp_, os.O_RDONLY)
                                os.fsync(fd_)
                                os.close(fd_)
                    if durable and nlab and not can_syncfs[0]:
                        fsync_dir(labels_dir)
                for b, p in enumerate(paths if not bulk else ()):
                    det = det_all[b, : cnt[b]]
                    ndet += det.shape[0]
                    rows = rows_all[offs[b]:offs[b + 1]] if same else postprocess.detections_to_rows(det, (H, W), shapes0[b])
                    if save_txt and rows.shape[0]:       # no detections => no file (the consumer relies on it)
                        f = open(os.path.join(labels_dir, os.path.splitext(os.path.basename(p))[0] + ".txt"), "wb")     # "wb": a tile processed again after a crash leaves the same bytes
                        f.write(format_label_rows(rows, save_conf))
                        f.flush()
                        written.append(f)
                        nlab += 1
                    if not quiet:
                        s = f"image {gidx[b] + 1}/{dataset.total} {p}: {H}x{W} "
                        s += postprocess.class_summary(det[:, 5], ck.names) if det.shape[0] else "(no detections), "
                        log(f"{s}{t_inf * 1e3 / len(paths):.1f}ms")
                    if multi and det.shape[0]:
                        with gather_lock:
                            gather.add(torch.full((det.shape[0],), gidx[b], dtype=torch.int32), aqdist.pack_rows(torch.from_numpy(det.copy())))
                if not released:
                    slot_free[slot_id].release()
                # the manifest line below vouches for these bytes: on disk first (files and their directory entries), then the record -- a
                # node crash must not leave a recorded tile without its label file (it would look like "no detections" for good).
                # One syncfs per batch; per-file fsync + directory fsync where that is unavailable.
                fds = [f.fileno() for f in written]
                for f in written:
                    f.flush()
                if durable and written and not sync_filesystem_of(labels_dir):
                    for fd in fds:
                        os.fsync(fd)
                    fsync_dir(labels_dir)
                for f in written:
                    f.close()
                if not tile_scenes:
                    with manifest_lock:
                        manifest.add(Path(p).stem for p in paths)
                with lock:
                    stats["seen"] += len(paths)
                    stats["labels"] += nlab
                    stats["dets"] += ndet
                    stats["t_post"] += time.perf_counter() - t0
        except BaseException as e:   # surfaced by the main thread
            err.append(e)
            for sf in slot_free:     # never leave the producer blocked on a slot
                sf.release()
            while True:              # keep draining so q.put() cannot block either
                if q.get() is None:
                    return

    lock = threading.Lock()
    manifest_lock = threading.Lock()
    can_syncfs = [True]
    n_writers = max(4, min(8, workers // 2)) if quiet else 1          # (per-image log lines stay in order with a single writer; the quiet path's work is C without the interpreter lock)
    depth_q = depth
    wts = [threading.Thread(target=writer, daemon=True) for _ in range(n_writers)]
    for w_ in wts:
        w_.start()
    t_pre = t_inf = 0.0
    t_start = time.perf_counter()
    shape_str = ""
    pinned = [None] * depth
    k = 0
    def scene_source():
        """Scene mode: (tile names, ("scene", path, raster, origins, (h, w)), shapes, None, global tile indices) per batch."""
        for path, arr, stems, origins, hw, gids in dataset.batches():
            names = [os.path.join(os.path.dirname(path), st_ + ".tif") for st_ in stems]     # label file = <tile stem>.txt
            yield names, ("scene", path, arr, origins, hw), [hw] * len(stems), None, gids

    gpu_superbatch = max(batch_size, int(os.environ.get("AQ_JPEG_GPU_SUPERBATCH", 2048)) // batch_size * batch_size)

    def gpu_jpeg_source(sub):
        """--jpeg-decode gpu: (paths, ("gpu_coef", coef int16 CUDA [b * nco], qt int16 CUDA [b, 192], decode-done event, done-hook, ...), shapes,
        None) per batch.  D + 2 super-batch buffers rotate: while the engine consumes the batches of super-batch k, the Huffman kernels of
        k + 1 .. k + D run (one launch each: a launch takes as long as its slowest image's serial decode, 50-200 ms for 1024-px tiles,
        whatever the number of tiles in it) and C threads read and prepare the files of k + 3 (aq_jpeg_prepare_files)."""
        from concurrent.futures import ThreadPoolExecutor
        from . import jpeg as aqjpeg
        from .engine import jpeg_huffman_decode
        W0, H0 = aqjpeg.scan_file(sub.files[0])
        nco = aqjpeg.coef_count(H0, W0)
        D = max(1, int(os.environ.get("AQ_JPEG_GPU_INFLIGHT", 3)))     # decode launches in flight ahead of the one being consumed
        NB = D + 2
        SB = min(gpu_superbatch, (len(sub.files) + batch_size - 1) // batch_size * batch_size)
        per = aqjpeg.stream_capacity(H0, W0)
        if SB * per + 256 >= 1 << 32:
            raise ValueError("--jpeg-decode gpu: AQ_JPEG_GPU_SUPERBATCH x bytes per image exceeds 4 GiB; lower it")
        # super-batches ramp up (4 x batch, 8 x batch, ... SB): the first tiles reach the engine after one small decode launch instead of a
        # full one -- a launch takes as long as its slowest image whatever its size, so small launches only cost throughput at the start
        chunks, s_, size = [], 0, min(SB, 4 * batch_size)
        while s_ < len(sub.files):
            chunks.append(sub.files[s_:s_ + size])
            s_ += size
            size = min(SB, size * 2)
        NB = min(NB, len(chunks))
        # Buffer set i is created when super-batch i is first prepared -- the pool thread is already reading files into set 0 while the main
        # thread allocates set 1 -- and sized for the largest super-batch it will ever hold (i, i + NB, ...: the ramp's small ones stay small
        # in a short sweep).  Pinned memory is allocated as such (torch.empty(pin_memory=True)): `zeros().pin_memory()` touched and copied
        # 2.7 GB at start-up, 1.4 of the 2.0 s the mode lost to the split path before its first batch (65 k-image sweep, round 4).
        host, batches_h, dev_streams, dev_coef, status_h = [None] * NB, [None] * NB, [None] * NB, [None] * NB, [None] * NB

        def make_set(i):
            cap = max(len(chunks[k_]) for k_ in range(i, len(chunks), NB))
            cap = (cap + batch_size - 1) // batch_size * batch_size
            host[i] = torch.empty(cap * per + 256, dtype=torch.uint8, pin_memory=True)
            host[i][-256:].zero_()
            batches_h[i] = aqjpeg.GpuDecodeBatch(cap, H0, W0, stream_buf=host[i].numpy(), bytes_per_image=per)
            dev_streams[i] = torch.empty(cap * per + 256, dtype=torch.uint8, device=dev)
            dev_coef[i] = torch.empty(cap * nco, dtype=torch.int16, device=dev)
            status_h[i] = torch.zeros(cap * aqjpeg.GpuDecodeBatch.SEG_CAP, dtype=torch.int32, pin_memory=True)
        dstreams = [torch.cuda.Stream() for _ in range(NB)]
        consumed = [[] for _ in range(NB)]                 # events of the batches that read buffer i: its next decode waits for them
        nthreads = max(2, min(sub.workers, 16))
        pool = ThreadPoolExecutor(1)                       # one orchestrating thread; the parallelism is inside the C call

        def launch(k, prepared):
            i = k % NB
            segs, sets, first = prepared
            n = len(chunks[k])
            ds = dstreams[i]
            for ev_ in consumed[i]:
                ds.wait_event(ev_)
            consumed[i] = []
            with torch.cuda.stream(ds):
                used = (int(segs["stream_off"][-1]) + int(segs["stream_len"][-1]) + 256 + 127) // 128 * 128
                dev_streams[i][:used].copy_(host[i][:used], non_blocking=True)
                segs_d = torch.from_numpy(segs.view(np.uint8).reshape(-1, 32)).pin_memory().to(dev, non_blocking=True)
                sets_d = torch.from_numpy(np.ascontiguousarray(sets)).pin_memory().to(dev, non_blocking=True)
                qt_d = torch.from_numpy(batches_h[i].qt[:n].view(np.int16).reshape(n, 192).copy()).pin_memory().to(dev, non_blocking=True)
                dev_coef[i][:n * nco].zero_()
                st_d = torch.empty(segs.shape[0], dtype=torch.int32, device=dev)
                jpeg_huffman_decode(dev_streams[i], segs_d, sets_d, dev_coef[i], st_d)
                status_h[i][:segs.shape[0]].copy_(st_d, non_blocking=True)
                ev_ = torch.cuda.Event()
                ev_.record(ds)
            return ev_, qt_d, first, segs.shape[0], (segs_d, sets_d, st_d)

        try:
            futures, launched = {}, {}
            next_prep = next_launch = 0
            for k in range(len(chunks)):
                while next_prep < len(chunks) and next_prep < k + NB:          # (buffer next_prep % NB was last used by super-batch next_prep - NB < k)
                    if host[next_prep % NB] is None:
                        make_set(next_prep % NB)
                    futures[next_prep] = pool.submit(batches_h[next_prep % NB].prepare_files, chunks[next_prep], nthreads)
                    next_prep += 1
                while next_launch < len(chunks) and next_launch <= k + D:
                    launched[next_launch] = launch(next_launch, futures.pop(next_launch).result())
                    next_launch += 1
                ev_, qt_d, first, nseg, keep = launched.pop(k)
                ev_.synchronize()                                              # this super-batch's coefficients are in HBM; its statuses on the host
                st = status_h[k % NB][:nseg].numpy()
                if st.any():
                    bad_seg = int(np.nonzero(st)[0][0])
                    bad_img = int(np.searchsorted(first, bad_seg, side="right") - 1)
                    raise ValueError(f"{chunks[k][bad_img]}: corrupt JPEG scan (GPU entropy decode status {int(st[bad_seg])}; libjpeg / Pillow refuse "
                                     "the file as truncated)")
                i = k % NB
                paths_k = chunks[k]
                for s_ in range(0, len(paths_k), batch_size):
                    b_ = min(batch_size, len(paths_k) - s_)
                    hook = (lambda e, i=i: consumed[i].append(e))
                    yield (paths_k[s_:s_ + b_], ("gpu_coef", dev_coef[i][s_ * nco:(s_ + b_) * nco], qt_d[s_:s_ + b_], ev_, hook, (H0, W0), nco),
                           [(H0, W0)] * b_, None)
        finally:
            pool.shutdown(wait=False)

    def image_source():
        """Every header is read once (no decode): the images of the most common size take the zero-copy pinned path, any others
        (the reference's tiler cuts smaller edge tiles for scenes that are not a multiple of 1024, src/load_data/tile_tifs.py:35-36)
        follow through the generic path, grouped by size -- a stray size can no longer end the sweep half way."""
        parts = []
        if len(dataset):
            try:
                sizes = None
                if jpeg_decode != "host" and not decode_threads:       # one header pass serves both questions when every file qualifies
                    split_sizes = dataset.scan_split_decodable()
                    if all(r is not None for r in split_sizes):
                        sizes, all_split[0] = split_sizes, True
                if sizes is None:
                    sizes = dataset.scan_sizes()
                common = max(set(sizes), key=sizes.count)
                main = [i for i, sz in enumerate(sizes) if sz == common]
                odd = [i for i, sz in enumerate(sizes) if sz != common]
            except Exception as e:                       # unreadable header: let the generic path report the file
                log(f"note: header scan failed ({e}); using the generic loader")
                main, odd = [], list(range(len(dataset)))
            if main:
                parts.append((dataset.subset(main), True))
            if odd:
                log(f"note: {len(odd)} of {len(dataset)} images differ in size from the rest; they take the generic loader")
                parts.append((dataset.subset(sorted(odd, key=lambda i: (sizes[i] if main else (0, 0), i))), False))
        for sub, fast in parts:
            if fast:
                # split JPEG decode (Huffman on the host, the rest on the GPU) when every file of this part is a baseline 4:2:0 JPEG
                split = False
                if jpeg_decode != "host" and not decode_threads:
                    scan = None if all_split[0] else sub.scan_split_decodable()
                    split = all_split[0] or all(r is not None for r in scan)
                    if jpeg_decode == "split" and not split:
                        bad = [sub.files[i] for i, r in enumerate(scan) if r is None]
                        raise ValueError(f"--jpeg-decode split: {len(bad)} of {len(scan)} images are not baseline 4:2:0 JPEGs the split decoder reads "
                                         f"(use auto or host): {', '.join(bad[:3])}{' ...' if len(bad) > 3 else ''}")
                if jpeg_decode == "gpu" and not split:
                    raise ValueError("--jpeg-decode gpu needs baseline 4:2:0 JPEGs throughout (use auto or host)")
                use_gpu = jpeg_decode == "gpu" or (jpeg_decode == "auto" and split and len(sub) >= int(os.environ.get("AQ_JPEG_GPU_AUTO_MIN", 32768)))
                if split and not split_note[0]:
                    split_note[0] = True
                    log(f"jpeg decode: split (entropy decoding in {sub.workers} worker processes, IDCT / upsampling / colour conversion on the GPU)"
                        if not use_gpu else
                        f"jpeg decode: gpu (entropy decoding on the GPU, one lane per image, super-batches of {gpu_superbatch} tiles; "
                        f"{sub.workers} reader threads strip byte stuffing; IDCT / upsampling / colour conversion on the GPU)")
                if use_gpu:
                    it = gpu_jpeg_source(sub)
                else:
                    gen = sub.pinned_batches(batch_size, depth + 2, processes=0 if decode_threads else None, coef=split)     # one buffer more than the batches in flight: the workers decode one batch ahead
                    release_of[0] = lambda i, sub=sub: sub.release(i)
                    it = gen
            else:
                it = ((p_, torch.from_numpy(b_).pin_memory(), s_, None) for p_, b_, s_ in sub.batches(batch_size))
            n_ = 0
            for p_, h_, s_, bi_ in it:
                yield p_, h_, s_, bi_, sub.indices[n_: n_ + len(p_)]
                n_ += len(p_)

    release_of = [None]
    split_note = [False]
    t_steady, n_steady, n_fed = [None], [0], [0]
    all_split = [False]
    jpeg_scratch = [None] * depth
    source_iter = scene_source() if tile_scenes else image_source()
    scene_dev, scene_path, scene_ev = None, None, None
    # Autotune on the first full batch (rank 0 times, every rank installs the same table: identical kernels on all GPUs of a run).
    per_rank = dataset.total // max(world, 1)
    tune = autotune == "on" or (autotune == "auto" and per_rank >= 8 * batch_size)
    if tune and world > 1 and ((dataset.n_scenes_total if tile_scenes else dataset.total) < world or resume):
        tune = False                                   # some rank may have no batch to meet the broadcast with
    tune_cache = os.environ.get("AQ_TUNE_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "aquaculture_amd", "tune.json")
    copy_done = []
    diag = {"source": 0.0, "slot_wait": 0.0, "put": 0.0, "writer_busy": 0.0} if os.environ.get("AQ_E2E_DIAG") == "1" else None
    t_src = time.perf_counter()
    try:
        for paths, host, shapes0, buf_i, gidx in source_iter:
            if err:
                break
            t0 = time.perf_counter()
            if diag is not None:
                diag["source"] += t0 - t_src
            slot = k % depth
            slot_free[slot].acquire()
            if diag is not None:
                diag["slot_wait"] += time.perf_counter() - t0
            st = streams[slot]
            with torch.cuda.stream(st):
                if isinstance(host, tuple) and host[0] == "scene":      # scene mode: one upload per scene, tiles cut by the letterbox kernel
                    _, spath, sarr, origins, thw = host
                    if spath != scene_path:                # page-locked source: async copy, its buffer goes back once the copy is done
                        scene_dev, scene_path = torch.from_numpy(sarr).to(dev, non_blocking=True), spath
                        scene_ev = torch.cuda.Event()
                        scene_ev.record(st)
                        dataset.uploaded(dataset.slot_of[spath], scene_ev)
                    else:
                        st.wait_event(scene_ev)            # another stream uploaded this scene
                    scene_dev.record_stream(st)
                    tiles = letterbox_scene_tiles(scene_dev, origins, thw, tuple(imgsz), int(max(ck.stride)), True)
                elif isinstance(host, tuple) and host[0] == "gpu_coef":   # --jpeg-decode gpu: the coefficient blocks are in HBM already
                    _, coef_v, qt_v, dec_ev, hook, (h0_, w0_), nco_ = host
                    st.wait_event(dec_ev)
                    if jpeg_scratch[slot] is None:
                        jpeg_scratch[slot] = torch.empty(eng.lib.aq_jpeg_scratch_bytes(batch_size, h0_, w0_), dtype=torch.uint8, device=dev)
                    off_ = torch.arange(qt_v.shape[0], dtype=torch.int64, device=dev) * nco_
                    tiles = jpeg_idct_rgb(coef_v, off_, qt_v, h0_, w0_, scratch=jpeg_scratch[slot])
                    used_ev = torch.cuda.Event()
                    used_ev.record(st)
                    hook(used_ev)                          # the super-batch's buffer is free for its next decode once this batch has read it
                    tiles = letterbox_device(tiles, tuple(imgsz), int(max(ck.stride)), True)
                else:
                    tiles = host.to(dev, non_blocking=True)
                    if buf_i is not None:                  # hand the pinned buffer back once its H2D copy has completed
                        h2d = torch.cuda.Event()
                        h2d.record(st)
                        copy_done.append((h2d, buf_i))
                    if tiles.dim() == 2:                   # coefficient blocks from the split JPEG decode: the pixel half runs here
                        h0_, w0_ = shapes0[0]
                        if jpeg_scratch[slot] is None:
                            jpeg_scratch[slot] = torch.empty(eng.lib.aq_jpeg_scratch_bytes(batch_size, h0_, w0_), dtype=torch.uint8, device=tiles.device)
                        tiles = jpeg_slots_to_rgb(tiles, h0_, w0_, scratch=jpeg_scratch[slot])
                    tiles = letterbox_device(tiles, tuple(imgsz), int(max(ck.stride)), True)
                if tune:                                   # once, before the pipeline fills
                    tune = False
                    geom = [int(tiles.shape[0]), int(tiles.shape[1]), int(tiles.shape[2])]
                    t_tune = time.perf_counter()
                    box = [geom, eng.autotune(tiles, cache=tune_cache) if rank == 0 else None]
                    if multi:
                        torch.distributed.broadcast_object_list(box, src=0)
                        if rank != 0 and box[0] == geom:
                            eng.set_tuned_table(*geom, box[1])
                    if rank == 0:
                        log(f"autotuned {sum(1 for c in box[1] if c >= 0)} conv layers for batch {geom[0]} x {geom[1]}x{geom[2]} in "
                            f"{time.perf_counter() - t_tune:.1f}s (table: {tune_cache})")
                t1 = time.perf_counter()
                dets, counts = eng.infer(tiles, conf_thres, iou_thres, max_det, slot=slot)
                B = tiles.shape[0]
                if pinned[slot] is None or pinned[slot][0].shape[0] < B:
                    pinned[slot] = (torch.empty((max(B, batch_size),), dtype=torch.int32).pin_memory(),
                                    torch.empty((max(B, batch_size), max_det, 6), dtype=torch.float32).pin_memory())
                counts_h, dets_h = pinned[slot][0][:B], pinned[slot][1][:B]
                counts_h.copy_(counts, non_blocking=True)
                dets_h.copy_(dets, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st)
            H, W = int(tiles.shape[1]), int(tiles.shape[2])
            shape_str = f"(1, 3, {H}, {W})"
            t2 = time.perf_counter()
            q.put((ev, counts_h, dets_h, paths, shapes0, (H, W), list(gidx), t2 - t1, slot))
            if diag is not None:
                diag["put"] += time.perf_counter() - t2
            n_fed[0] += len(paths)
            while copy_done and (copy_done[0][0].query() or len(copy_done) > depth):
                ev_, bi_ = copy_done.pop(0)
                ev_.synchronize()
                (release_of[0] or dataset.release)(bi_)
            t_pre += t1 - t0
            t_inf += t2 - t1
            k += 1
            if k == 2:                                     # steady state: from the hand-over of the second batch (header scan, worker start-up,
                t_steady[0], n_steady[0] = time.perf_counter(), sum(1 for _ in gidx)        # autotune and the pipeline fill are behind us)
            if multi and k % flush_every == 0:         # bounded gather: rows leave the host lists every few batches (collective)
                with gather_lock:
                    gather.flush(more=True, failed=bool(err))
            t_src = time.perf_counter()
    except BaseException as e:   # anything the main loop raises (engine, autotune, a decode error, another rank's RankFailed) ends the sweep
        err.append(e)            # HERE, but only after the writers are joined and the other ranks have been told (the collective tail below)
    for _ in wts:
        q.put(None)
    for w_ in wts:
        w_.join()
    if multi:                                          # collective tail; a failed rank takes the others down with it at once
        try:
            with gather_lock:
                gather.finish(failed=bool(err))
        except aqdist.RankFailed:
            if not err:                                # another rank failed: that IS this rank's error
                raise
    if err:                                            # this rank's own error, with its traceback (not the RankFailed its flag just raised here too)
        raise err[0]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    seen, n_labels, n_dets, t_post = stats["seen"], stats["labels"], stats["dets"], stats["t_post"]

    if multi:       # the one collective of the path (detection gather over RCCL/xGMI) has run in bounded pieces; now the counters
        seen_all, labels_all, dets_all, elapsed = aqdist.reduce_counters(seen, n_labels, n_dets, elapsed, dev)
        assert rank != 0 or gather.total == dets_all, "gathered detection rows do not add up"
    else:
        seen_all, labels_all, dets_all = seen, n_labels, n_dets
    if rank == 0:
        per = lambda t: t * 1e3 / max(seen, 1)
        log(f"Speed: {per(t_pre):.1f}ms pre-process, {per(t_inf):.1f}ms inference, {per(t_post):.1f}ms NMS-out/post-process "
            f"per image at shape {shape_str} (host time per stage; stages overlap)")
        log(f"{seen_all} images, {dets_all} detections, {seen_all / max(elapsed, 1e-9):.1f} images/s on {world} GPU(s) [{precision}]")
        if diag is not None:
            log(f"diag (main thread, seconds of {elapsed:.2f}): waiting for the source {diag['source']:.2f}, for a free slot {diag['slot_wait']:.2f}, "
                f"in q.put {diag['put']:.2f}, pre {t_pre:.2f}, enqueue {t_inf:.2f}; writers' post time summed over threads {t_post:.2f}")
        if t_steady[0] is not None and seen > 4 * batch_size:
            # this rank's rate once the sweep is under way (everything up to the second batch -- header scan, decode worker start-up,
            # tile-configuration timing, pipeline fill -- excluded): what a long sweep converges to
            t_end = t_start + (time.perf_counter() - t_start if world > 1 else elapsed)
            log(f"steady state: {(seen - 2 * batch_size) / max(t_end - t_steady[0], 1e-9):.1f} images/s on this GPU after the first two batches")
        if save_txt:
            log(f"Results saved to {save_dir}\n{labels_all} labels saved to {save_dir / 'labels'}")
        if geocode_bboxes:
            # the consumer's next step (reference src/process_yolo/geocode_results.py:106-197) on the label files just written
            if not save_txt or not save_conf:
                raise ValueError("--geocode-bboxes reads the label files: it needs --save-txt --save-conf")
            from . import geocode
            t_g = time.perf_counter()
            out = geocode_out or str(save_dir / "detections.geojson")
            table = geocode.geocode_label_dir(labels_dir, geocode_bboxes, out)
            log(f"{table['image'].shape[0]} detections geocoded to {out} in {time.perf_counter() - t_g:.2f}s")
    manifest.close()
    eng.close()
    return save_dir


def _calibration_tiles(source, tile_scenes, imgsz, stride, dev, n=16):
    """Up to `n` letterboxed tiles sampled evenly over the sweep's sorted file list (not its first files, which all come from one scene):
    uint8 CUDA [b, H, W, 3], every tile of the first sample's original size.  Host decode -- it runs once per sweep."""
    from .engine import letterbox_device
    if tile_scenes:
        from .scenes import list_scenes, read_scene, tile_grid
        scenes = list_scenes(source)
        picks = []
        for sp in scenes[:: max(1, len(scenes) // 4)][:4]:
            arr = read_scene(sp)
            grid = tile_grid(arr.shape[1], arr.shape[0], int(tile_scenes))
            grid = [g for g in grid if (g[2], g[3]) == (int(tile_scenes), int(tile_scenes))] or grid     # (scenes smaller than one tile: edge tiles)
            for x, y, w, h in grid[:: max(1, len(grid) // 4)][:4]:
                picks.append(np.ascontiguousarray(arr[y:y + h, x:x + w, :3]))
    else:
        from .dataloader import list_images, read_rgb
        files = list_images(source)
        picks = [read_rgb(f) for f in files[:: max(1, len(files) // n)][:n]]
    picks = [p for p in picks if p.shape == picks[0].shape]
    if not picks:
        raise ValueError("fp8 calibration: no tiles to calibrate on")
    return letterbox_device(torch.from_numpy(np.stack(picks)).to(dev), tuple(imgsz), stride, True)


def main(argv: Optional[List[str]] = None) -> int:
    opt = parse_opt(argv)
    run(**vars(opt))
    return 0


if __name__ == "__main__":
    sys.exit(main())
