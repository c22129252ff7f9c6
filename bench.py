#!/usr/bin/env python3
"""Headline benchmark: 640-px tiles/s (whole node), YOLOv5m bf16, MI355X -- BASELINE.json `metric`.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One *step* = one pass of the whole hot path (preprocess -> backbone/neck/head convs -> decode -> NMS) over one
batch of 64 synthetic 640x640 ocean tiles that already reside in HBM (BASELINE.json configs[1]); each rank runs
its own batches (tile-sharded, weak scaling) and the only collective is the final detection gather.
Rank 0 prints ONE JSON line with the metric, a `roofline` object for the dominant kernel (the implicit-GEMM conv
on the 3x3 layers, MFMA-bound, timed live with HIP events on the launch stream) and a `cpu_baseline` object
(the CPU oracle -- a PyTorch-CPU restatement of detect.py, NOT the reference's own detect.py, which is absent from
the reference tree -- timed on this host's cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)     # ~0.5 s timed at batch 64: the first tens of steps still see the clocks settle
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--size", type=int, default=640)
    p.add_argument("--variant", default="yolov5m")
    p.add_argument("--precision", default="bf16", choices=("bf16", "fp32", "fp8w", "f16x3", "fp8"),
                   help="fp8 = BASELINE.json configs[3] as written: the wide Bottleneck 3x3 layers on the fp8 MFMA, e4m3 on both operands "
                        "(activation scales calibrated on the first batch), everything else bf16; fp8w = e4m3 weight VALUES on the bf16 MFMA "
                        "(round 2); f16x3 = the fast parity mode (fp32 activations, three fp16 MFMAs per product)")
    p.add_argument("--pool", type=int, default=8, help="distinct synthetic batches kept in HBM and cycled (8 x 64 tiles = 629 MB of input, "
                                                       "beyond the 256 MB Infinity Cache)")
    p.add_argument("--roof-steps", type=int, default=10, help="steps of the single-stream HIP-event pass that feeds `roofline`")
    p.add_argument("--streams", type=int, default=int(os.environ.get("AQ_BENCH_STREAMS", 2)),
                   help="independent batches in flight (one HIP stream + workspace each)")
    p.add_argument("--two-kernel-bottleneck", action="store_true", help="A/B: 1x1 + 3x3 launches instead of the fused Bottleneck kernel")
    p.add_argument("--two-kernel-stem", action="store_true", help="A/B: preprocess + space-to-depth conv instead of the fused stem kernel")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cu-split", default="", help="experiment (measured: no gain -- 14.2 k tiles/s with `low`, 13.2-13.3 k with the others, against 14.5 k without): run the "
                   "batches in flight on disjoint halves of the CUs (hipExtStreamCreateWithCUMask); value = layout of the even streams' "
                   "mask: low | even | pairs | xcd (odd streams get the complement)")
    p.add_argument("--parity-steps", type=int, default=3, help="steps of the fp32 parity-mode engine timed beside the bf16 metric (0 = skip)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-oracle sample budget")
    p.add_argument("--e2e-images", type=int, default=8192,
                   help="N = 1 only: after the timed region, write this many synthetic 1024-px jpeg tiles (the reference's tile size, src/utils.py:17-19) and "
                        "sweep them with yolov5/detect.py --half in a child process; its images/s go into the line as `e2e` (0 = skip)")
    p.add_argument("--no-autotune", action="store_true", help="use the built-in tile heuristic instead of timing configs")
    p.add_argument("--retune", action="store_true", help="time the tile configurations in this run even when the in-tree table (aquaculture_amd/data/"
                                                         "tuned_tables.json) has this geometry")
    p.add_argument("--write-tuned", default="", help="merge this run's tuned table into the given JSON file (maintainers: refresh the shipped table)")
    p.add_argument("--no-profile", action="store_true", help="skip per-op HIP events (roofline becomes null)")
    p.add_argument("--launcher-selftest", action="store_true", help="run only the rank launch / rendezvous / reduction plumbing (no GPU work)")
    p.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic_latest.json"),
                   help="optional PMC-derived HBM bytes per launch for the dominant kernel")
    return p.parse_args()


def host_threads() -> int:
    """CPU threads we may use: the affinity mask, capped at the GPU box's per-GPU share of 16."""
    if os.environ.get("AQ_CPU_THREADS"):
        return int(os.environ["AQ_CPU_THREADS"])
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def make_tiles(rank: int, batch: int, pool: int, size: int) -> np.ndarray:
    from concurrent.futures import ThreadPoolExecutor
    from aquaculture_amd import tiles
    idx = [rank * 100003 + i for i in range(batch * pool)]   # disjoint synthetic tile ids per rank
    with ThreadPoolExecutor(host_threads()) as ex:
        arr = list(ex.map(lambda i: tiles.synthetic_tile(i, size), idx))
    return np.stack(arr, 0).reshape(pool, batch, size, size, 3)


def cpu_baseline(ck, size: int, budget_s: float) -> dict:
    """The oracle (kind 'port') on this host's cores: B=1, fp32, all threads, 2 warm-up tiles, then tiles until
    the time budget is used."""
    from aquaculture_amd import tiles
    from oracle import yolov5_oracle as O
    torch.set_num_threads(host_threads())
    m = O.model_from_checkpoint(ck)
    for i in range(2):
        O.detect_tiles(m, tiles.synthetic_batch([i], size))
    n, t0, per = 0, time.perf_counter(), []
    while n < 64 and (time.perf_counter() - t0) < budget_s:
        x = tiles.synthetic_batch([n], size)
        t1 = time.perf_counter()
        O.detect_tiles(m, x)
        per.append(time.perf_counter() - t1)
        n += 1
    return {"value": round(n / sum(per), 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} synthetic {size}x{size} tiles, batch 1, fp32 PyTorch-CPU restatement of detect.py (oracle/), "
                      f"median {1e3 * float(np.median(per)):.0f} ms/tile, {torch.get_num_threads()} threads of os.cpu_count()={os.cpu_count()}"}


def e2e_live(a) -> dict:
    """The second leg of SURVEY 8d, measured by THIS command: jpeg directory -> label files through the preserved
    entry point, in a child process (this process has released its engine; the decode workers are the child's).  Tiles are 1024-px
    baseline jpegs at GDAL's default quality, as reference src/load_data/tile_tifs.py:66-74 writes them; the checkpoint is the seeded
    synthetic one in upstream's pickle format.  Never part of `value`."""
    import re
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from aquaculture_amd import checkpoint, tiles
    d = tempfile.mkdtemp(prefix="aq_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t0 = time.perf_counter()
        src = os.path.join(d, "jpegs")
        os.makedirs(src)
        base = 64                                          # distinct tiles rendered; the rest of the directory repeats their bytes under other names
        n = a.e2e_images
        with ThreadPoolExecutor(host_threads()) as ex:
            list(ex.map(lambda i: tiles.write_synthetic_jpegs(src, [i], size=1024), range(min(base, n))))
        names = sorted(os.listdir(src))
        for i in range(base, n):
            shutil.copyfile(os.path.join(src, names[i % base]), os.path.join(src, tiles.tile_name(i)))
        wts = os.path.join(d, "synthetic_yolov5m.pt")
        checkpoint.write_synthetic_checkpoint(wts, "yolov5m", 5)
        t_make = time.perf_counter() - t0
        workers = max(1, host_threads() - 2)
        cmd = [sys.executable, os.path.join(ROOT, "yolov5", "detect.py"), "--weights", wts, "--source", src, "--nosave", "--save-txt", "--save-conf",
               "--half", "--quiet", "--batch-size", "64", "--workers", str(workers), "--project", os.path.join(d, "runs"), "--name", "e2e"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, AQ_NO_FSYNC="1"))
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        whole = re.search(r"(\d+) images, (\d+) detections, ([0-9.]+) images/s", r.stdout)
        steady = re.search(r"steady state: ([0-9.]+) images/s", r.stdout)
        labels = len(os.listdir(os.path.join(d, "runs", "e2e", "labels")))
        return {"images": n, "tile_px": 1024, "images_per_s_steady": float(steady.group(1)) if steady else None,
                "images_per_s_whole_sweep": float(whole.group(3)) if whole else None, "detections": int(whole.group(2)) if whole else None,
                "label_files": labels, "child_wall_s": round(wall, 2), "make_inputs_s": round(t_make, 2), "decode_workers": workers,
                "jpeg_decode": "split" if "jpeg decode: split" in r.stdout else "host",
                "note": f"measured live by this command in a child process: yolov5/detect.py --half over {n} 1024-px q75 jpegs ({min(base, n)} distinct tiles) in "
                        "shared memory, label files written without fsync; whole-sweep rate includes checkpoint load, engine creation, worker start-up and "
                        "pipeline fill; steady = after the first two batches; bound by the host's CPU quota, not by the GPU"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def baseline_config(a) -> str:
    """Which entry of BASELINE.json `configs` this command line measures (by model, tile size, precision and batch), or that it is none."""
    if a.variant == "yolov5m" and a.size == 640 and a.precision == "bf16" and a.batch == 64:
        return "BASELINE.json configs[1]" if a.gpus == 1 else f"BASELINE.json configs[2] geometry on {a.gpus} GPUs" if a.gpus != 8 else "BASELINE.json configs[2]"
    if a.variant == "yolov5m" and a.size == 640 and a.precision in ("fp8", "fp8w"):
        return "BASELINE.json configs[3]" + ("" if a.batch == 128 else f" at batch {a.batch} instead of 128") + (" with fp8 weight VALUES on the bf16 MFMA" if a.precision == "fp8w" else "")
    if a.variant == "yolov5x" and a.size == 1280 and a.precision == "bf16":
        return "BASELINE.json configs[4]" + ("" if a.batch == 16 else f" at batch {a.batch} instead of 16") + (f", {a.gpus} of its 8 GPUs" if a.gpus != 8 else "")
    return "not a BASELINE.json configuration"


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` typed by hand (no torchrun around it): the parent starts N children of this same command line, one
    per GPU, with the torchrun-style environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), forwards rank 0's JSON
    line and exits with the worst child's code.  The parent itself never touches a GPU (no HIP call, no torch.cuda query) and never
    exec()s; a failed child is reported, the others are stopped by PID, nothing is retried."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port (closed again before the children bind it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this image
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)    # rank 0 prints the one line
    reader.start()
    rcs = [None] * n
    deadline = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad and deadline is None:
            deadline = time.monotonic() + 30.0        # the survivors get half a minute to notice (RankFailed / backend error) ...
        if deadline is not None and time.monotonic() > deadline:
            for r, p in enumerate(procs):             # ... then are stopped, each by its own PID
                if rcs[r] is None:
                    p.kill()
                    rcs[r] = p.wait()
        time.sleep(0.05)
    worst = max(rcs, key=lambda rc: (rc != 0, abs(rc)))
    if worst != 0:
        print(f"bench: rank exit codes {rcs}", file=sys.stderr)
        return worst if worst > 0 else 1
    reader.join(10.0)
    for line in (got[0].decode() if got else "").splitlines():
        # the JSON line to stdout; whatever a backend chose to print on rank 0's stdout (gloo's connection notes) to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return 0


def launcher_selftest() -> int:
    """--launcher-selftest: what the N > 1 launch path does around the GPU work -- rendezvous, barrier, max-over-ranks reduction, one
    line from rank 0 -- with nothing that needs a GPU in between (tests/test_bench_launcher.py runs it on the CPU with gloo)."""
    from aquaculture_amd import dist as aqdist
    a = parse()
    rank, world, _ = aqdist.init(os.environ.get("AQ_DIST_BACKEND") or "gloo")
    if world != a.gpus:
        print(f"bench: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if os.environ.get("AQ_SELFTEST_FAIL_RANK") == str(rank):
        return 7
    aqdist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    aqdist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest", "n_gpus": world, "max_over_ranks": float(t[0])}), flush=True)
    return 0


def main() -> int:
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a.gpus)
    if a.launcher_selftest:
        return launcher_selftest()
    from aquaculture_amd import checkpoint, dist as aqdist, spec
    from aquaculture_amd import engine as aqengine
    from aquaculture_amd.engine import Engine

    # one rank per GPU over RCCL ("nccl"); AQ_DIST_BACKEND=gloo is the single-GPU rehearsal of the N > 1 code path
    rank, world, local = aqdist.init((os.environ.get("AQ_DIST_BACKEND") or "nccl") if int(os.environ.get("WORLD_SIZE", 1)) > 1 or aqdist.forced() else None)
    local = local % max(torch.cuda.device_count(), 1)
    if world != a.gpus:
        print(f"bench: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if a.cu_split:
        ncu = torch.cuda.get_device_properties(local).multi_processor_count
        os.environ["AQ_NUM_CUS"] = str(ncu // 2)          # persistent grids sized for half the chip (read once, before the first launch)
    ck = checkpoint.synthetic_checkpoint(a.variant, 5)
    B, K, W = a.batch, a.steps, a.warmup
    tiles_dev = torch.from_numpy(make_tiles(rank, B, a.pool, a.size)).to(dev)
    eng = Engine(ck, a.precision, local, fused_stem=not a.two_kernel_stem,
                 fused_bottleneck=(a.precision in ("bf16", "fp8w", "fp8") and not a.two_kernel_bottleneck),
                 fp8_calibration=tiles_dev[0][:min(B, 16)] if a.precision == "fp8" else None)
    max_det = 1000
    dets = torch.empty((K, B, max_det, 6), dtype=torch.float32, device=dev)
    counts = torch.zeros((K, B), dtype=torch.int32, device=dev)

    streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)] if a.streams > 1 else [torch.cuda.current_stream()]
    if a.cu_split:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        words = (ncu + 31) // 32
        bit = {"low": lambda i: i < ncu // 2, "even": lambda i: i % 2 == 0, "pairs": lambda i: (i // 2) % 2 == 0,
               "xcd": lambda i: (i % 8) < 4}[a.cu_split]
        streams = []
        for si in range(max(2, a.streams)):
            side = si % 2
            m = (C.c_uint32 * words)()
            for i in range(ncu):
                if bit(i) == (side == 0):
                    m[i // 32] |= 1 << (i % 32)
            h = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(words), m)
            if rc != 0:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
            streams.append(torch.cuda.ExternalStream(h.value, device=dev))

    def step(k: int, slot: int):
        # independent batches alternate over the streams: batch k+1's high-resolution layers fill the CUs that
        # batch k's low-resolution layers (200-400 tiles on 256 CUs) leave idle
        with torch.cuda.stream(streams[k % len(streams)]):
            eng.infer(tiles_dev[k % a.pool], float(os.environ.get("AQ_BENCH_CONF", 0.25)), 0.45, max_det, out=(dets[slot], counts[slot]), slot=k % len(streams))

    def join():
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)

    def gather_all():
        # final detection gather: the path's one collective (to rank 0; RCCL over xGMI when world > 1), in pieces of 8 steps so
        # that neither the padding nor rank 0's receive buffers grow with the run
        g = aqdist.DetectionGather(dev, keep=False)
        for k0 in range(0, K, 8):
            k1 = min(K, k0 + 8)
            keep = torch.arange(max_det, device=dev).view(1, 1, -1) < counts[k0:k1].view(k1 - k0, B, 1)
            tile_id = ((torch.arange(k0 * B, k1 * B, device=dev, dtype=torch.int32).view(k1 - k0, B, 1)) * world + rank).expand(k1 - k0, B, max_det)[keep]
            g.add(tile_id, aqdist.pack_rows(dets[k0:k1][keep]))
            g.flush(more=k1 < K)
        g.finish()
        t = torch.tensor([g.total if rank == 0 else int(counts.sum())], dtype=torch.int64)
        return t

    with torch.cuda.stream(streams[0]):                   # (a CU-masked stream times the candidates on the CUs they will run on)
        cfgs = eng.autotune(tiles_dev[0], cache=os.environ.get("AQ_TUNE_CACHE"), shipped=not a.retune) if not a.no_autotune else None
    torch.cuda.synchronize()
    if a.write_tuned and cfgs is not None and rank == 0:
        key = eng.tune_key(B, a.size, a.size)
        try:
            with open(a.write_tuned) as f:
                tab = json.load(f)
        except (OSError, ValueError):
            tab = {}
        tab[key] = [int(c) for c in cfgs]
        with open(a.write_tuned, "w") as f:
            json.dump(tab, f, indent=0)
    for k in range(max(W, a.streams)):
        step(k, 0)
    join()
    gather_all()          # untimed: loads torch's indexing kernels / opens the RCCL channels once
    counts.zero_()
    torch.cuda.synchronize()
    aqdist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(K):
        step(k, k)
    join()
    ev1.record()                                          # this rank's K steps are done here (no host synchronisation: the gather follows on the stream)
    gathered = gather_all()
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0                     # this rank alone: its steps + its part of the gather
    aqdist.barrier()
    elapsed = time.perf_counter() - t0
    n_dets_total = int(gathered[0])
    ranks_info = None
    if aqdist.active():
        cdev = "cpu" if torch.distributed.get_backend() == "gloo" else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])
        # straggler / collective visibility for the first real multi-GPU run: every rank's own step time (HIP events
        # around its K steps) and the time it then spent in the detection gather, collected outside the timed region
        mine = torch.tensor([ev0.elapsed_time(ev1) * 1e-3, t_rank], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        comp = [float(e[0]) for e in every]
        tot = [float(e[1]) for e in every]
        ranks_info = {"dist_world_size": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                      "per_rank_tiles_per_s": {"min": round(K * B / max(comp), 1), "max": round(K * B / min(comp), 1),
                                               "slowest_rank": int(np.argmax(comp))},
                      "gather_share_of_timed_region": round(max(0.0, max(tt - c for tt, c in zip(tot, comp))) / elapsed, 4),
                      "note": "per-rank rates from HIP events around each rank's own K steps; the gather share is the largest (rank time - step time) "
                              "over the max-over-ranks region that `value` is computed from"}

    # roofline pass: the same steps on ONE stream with a HIP event recorded on the launch stream before every op
    # (with several batches in flight the per-kernel event intervals of one stream would include the other's kernels)
    R = 0
    if not a.no_profile:
        R = max(1, min(K, a.roof_steps))
        eng.profile(True, ring=R)
        for k in range(R):
            eng.infer(tiles_dev[k % a.pool], 0.25, 0.45, max_det, out=(dets[k], counts[k]), slot=0)
        torch.cuda.synchronize()
    if rank != 0:
        return 0

    tiles_total = world * K * B
    value = tiles_total / elapsed
    plan = eng.plan
    fl = plan.flops(a.size, a.size)
    roof = None
    ops_ms = None
    if not a.no_profile:
        ms, calls = eng.op_times_ms()
        ops_ms = ms
        idx3 = [i for i, o in enumerate(plan.ops) if o.kind == spec.OP_CONV and o.meta.get("class") == "conv3x3"]
        idxc = [i for i, o in enumerate(plan.ops) if o.kind in (spec.OP_CONV, spec.OP_STEM, spec.OP_BOTTLENECK, spec.OP_DOWNBLOCK)]
        t3 = float(ms[idx3].sum()) * 1e-3       # seconds per step in the 3x3 conv launches
        tc = float(ms[idxc].sum()) * 1e-3
        # fp8w: fp8 VALUES on the bf16 MFMA (quant.py), so the bf16 peak; fp8: the family mixes bf16 and fp8 launches -- priced against the bf16
        # peak here, the fp8 launches alone against the fp8 peak in `fp8_layers`; f16x3: three fp16 MFMAs per product on the bf16-rate pipe
        peak = PEAK_F32_TFLOPS if a.precision == "fp32" else PEAK_BF16_TFLOPS
        f3 = float(sum(plan.ops[i].flops_per_tile for i in idx3)) * B     # the 3x3 layers launched as implicit-GEMM convs
        idxb = [i for i, o in enumerate(plan.ops) if o.kind == spec.OP_BOTTLENECK]
        tb = float(ms[idxb].sum()) * 1e-3
        fb = float(sum(plan.ops[i].flops_per_tile for i in idxb)) * B
        ach = f3 / t3 / 1e12
        # within that family: the one kernel instantiation (autotuned config id) that takes the most time per step
        single = None
        if cfgs:
            groups = {}
            for i in idx3:
                groups.setdefault(int(cfgs[i]), []).append(i)
            cbest = max(groups, key=lambda c: float(ms[groups[c]].sum()))
            ii = groups[cbest]
            tt = float(ms[ii].sum()) * 1e-3
            ff = float(sum(plan.ops[i].flops_per_tile for i in ii)) * B
            n_ig = eng.lib.aq_conv_num_configs()
            one_per_wg = cbest >= aqengine.CONV_CFG_ONE_TILE_PER_WG          # flag OR-ed into a tile-shape id
            shape = cbest & (aqengine.CONV_CFG_ONE_TILE_PER_WG - 1)
            if not one_per_wg and cbest == aqengine.CONV_CFG_PL3X3:
                kname = "conv3x3_pl_asm_nb13 / conv3x3_pl_kernel (planar 3x3/s1: weights streamed to registers, slot-major region in LDS; generated gfx950 assembly build, HIP-source fallback)"
            elif not one_per_wg and cbest == aqengine.CONV_CFG_PL3X3S2:
                kname = "conv3x3_pl_asm_s2nb13 (planar 3x3/s2: parity planes of the input as pixel-major swizzled region rows, LDS-DMA through a buffer descriptor; generated gfx950 assembly)"
            elif not one_per_wg and cbest >= 1000:
                kname = "downblock_kernel<96, 192> (direct 3x3/s2)"
            else:
                import ctypes as _C
                bm, bn = _C.c_int(), _C.c_int()
                eng.lib.aq_conv_config_tiles(shape, _C.byref(bm), _C.byref(bn))
                n_igemm = 21                                  # conv_igemm.hip's shape table; halo shapes follow (tools/stamp_conv.py)
                kname = (f"{'conv3x3_halo_kernel' if shape >= n_igemm else 'conv_igemm_kernel'} tile {bm.value}x{bn.value} "
                         f"(shape {shape} of {n_ig}, {'one workgroup per tile' if one_per_wg else 'persistent grid'})")
            single = {"config": cbest, "kernel": kname, "launches_per_step": len(ii), "avg_launch_ms": round(1e3 * tt / len(ii), 4),
                      "tflops": round(ff / tt / 1e12, 1), "frac": round(ff / tt / 1e12 / peak, 4)}
        # PMC figure of a separate `rocprofv3 --pmc` pass (tools/profile_pmc.py): reported only for the workload AND the library build
        # it was collected on, and labelled as what it is (this run measures no counters)
        traffic, traffic_src = None, None
        try:
            with open(a.traffic_json) as f:
                tj = json.load(f)
            with open(os.path.join(ROOT, "aquaculture_amd", "csrc", "libaqengine.so.sha256")) as f:
                digest = f.read().strip()
            mine = [int(cfgs[i]) for i in idx3] if cfgs is not None else None
            # the tuner's pick between two implicit-GEMM shapes of a stride-2 layer flips from run to run (a few us apart); the PMC figure
            # is still this build's and this workload's when at most two of the 19 layers differ -- and the label says how many do
            compared = mine is not None and bool(tj.get("conv3x3_configs"))
            differ = 0 if not compared else min(
                (sum(x != y for x, y in zip(mine, v)) if len(v) == len(mine) else 99) for v in tj["conv3x3_configs"].values())
            if (a.variant == "yolov5m" and a.size == 640 and a.precision in ("bf16", "fp8w") and int(tj.get("batch", -1)) == B
                    and tj.get("library_source_digest") == digest and differ <= 2):
                traffic = tj.get("bytes_per_launch")
                traffic_src = (f"{os.path.relpath(a.traffic_json, ROOT)} (separate rocprofv3 --pmc pass of this library build {digest[:12]}; "
                               + ("tuned 3x3 kernels NOT compared: the record holds no configuration list)" if not compared else
                                  "same tuned 3x3 kernels, compared layer by layer)" if differ == 0 else
                                  f"the tuner picked another tile shape on {differ} of the {len(mine)} layers in this run)"))
            else:
                traffic_src = f"null: {os.path.relpath(a.traffic_json, ROOT)} was collected on another build / workload / tuned table"
        except (OSError, ValueError):
            pass
        roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": f"conv3x3_pl_asm_pm13w40 / pm13w20 (planar 3x3/s1, pixel-major assembly builds; nb13 = slot-major for other widths) / conv3x3_pl_asm_s2nb13 (planar 3x3/s2, assembly) / downblock_kernel / conv_igemm_kernel (3x3 convs, kernel and tile shape autotuned per layer) on the {len(idx3)} 3x3 "
                          f"layers launched as plain convs ({len(idxb)} Bottlenecks run in bottleneck_kernel, reported separately)",
                "launches_per_step": len(idx3), "avg_launch_ms": round(1e3 * t3 / len(idx3), 4),
                "flops_per_step": f3, "steps_timed": calls,
                "largest_single_kernel": single,
                "fused_bottleneck": {"launches_per_step": len(idxb), "ms_per_step": round(1e3 * tb, 3),
                                     "tflops": round(fb / tb / 1e12, 1) if tb > 0 else None},
                "pass": f"{calls} single-stream steps with HIP events right after the timed region (same process, same buffers; "
                        f"the timed region keeps {a.streams} batches in flight)",
                # every kernel that CONTAINS a 3x3 layer (the 19 plain launches, the 8 fused Bottlenecks, the down-block with model.1): all 28
                # 3x3 layers of the network, priced with the whole FLOPs of those launches (their fused 1x1 stages included) over their time
                "all_3x3": (lambda ii: {"launches_per_step": len(ii), "layers_3x3": len(idx3) + len(idxb) + sum(1 for i in ii if plan.ops[i].kind == spec.OP_DOWNBLOCK),
                                         "flops_per_step": float(sum(plan.ops[i].flops_per_tile for i in ii)) * B,
                                         "ms_per_step": round(float(ms[ii].sum()), 3),
                                         "tflops": round(float(sum(plan.ops[i].flops_per_tile for i in ii)) * B / (float(ms[ii].sum()) * 1e-3) / 1e12, 1),
                                         "frac": round(float(sum(plan.ops[i].flops_per_tile for i in ii)) * B / (float(ms[ii].sum()) * 1e-3) / 1e12 / peak, 4)})(
                    idx3 + idxb + [i for i, o in enumerate(plan.ops) if o.kind == spec.OP_DOWNBLOCK]),
                **({"fp8_layers": (lambda ii: {"launches_per_step": len(ii), "ms_per_step": round(float(ms[ii].sum()), 3),
                                                "tflops": round(float(sum(plan.ops[i].flops_per_tile for i in ii)) * B / (float(ms[ii].sum()) * 1e-3) / 1e12, 1),
                                                "peak": 5000.0, "frac": round(float(sum(plan.ops[i].flops_per_tile for i in ii)) * B / (float(ms[ii].sum()) * 1e-3) / 1e12 / 5000.0, 4),
                                                "kernel": "conv3x3_pl_asm_f8nb13 (v_mfma_f32_16x16x128_f8f6f4, e4m3 x e4m3), fed by conv1x1_direct_kernel<F8OUT>",
                                                "act_scales": {k_: round(v_, 6) for k_, v_ in eng.fp8_scales.items()}})(
                    [c_ for _, c_ in eng.fp8_pairs()])} if a.precision == "fp8" else {}),
                "all_conv_tflops": round((fl["total"]) * B / tc / 1e12, 1),
                "step_ms_by_kind": {"conv3x3": round(1e3 * t3, 3), "conv_other": round(1e3 * (tc - t3), 3),
                                    "rest": round(float(ms.sum()) - 1e3 * tc, 3)}}
    out = {
        "metric": "640px tiles/sec (whole node) YOLOv5m bf16" if (a.size == 640 and a.variant == "yolov5m" and a.precision == "bf16")
                  else f"{a.size}px tiles/sec (whole node) {a.variant} {a.precision}",
        "value": round(value, 1), "unit": "tiles/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16 x fp8-e4m3 weights (bf16 MFMA)" if a.precision == "fp8w" else
                  "bf16 + fp8-e4m3 x fp8-e4m3 on the fp8 MFMA (14 wide Bottleneck 3x3 layers)" if a.precision == "fp8" else a.precision, "data": "synthetic",
        "config": {"workload": f"{a.variant} {a.precision}, 1xMI355X per rank, batch={B}, synthetic {a.size}x{a.size} ocean tiles "
                               f"resident in HBM ({a.pool} distinct batches cycled), seeded random-init weights nc=5 "
                               f"({baseline_config(a)})",
                   "tile_configs": getattr(eng, "tuned_from", "heuristic") if not a.no_autotune else "heuristic",
                   "batch_per_gpu": B, "tile_px": a.size, "parallelism": f"tile-sharded dp{world}", **({"collectives": torch.distributed.get_backend(), "ranks": ranks_info} if aqdist.active() else {}), "batches_in_flight": a.streams, **({"cu_split": a.cu_split} if a.cu_split else {}),
                   "detections_gathered": n_dets_total},
        "roofline": roof,
    }
    # SURVEY 8d's second leg: the same build end to end from an image directory (tools/bench_e2e.py, a separate run on the same pool of
    # boxes): quoted only when it was measured on THIS library build; never part of `value`
    try:
        with open(os.path.join(ROOT, "profiles", "e2e_latest.json")) as f:
            e2e = json.load(f)
        with open(os.path.join(ROOT, "aquaculture_amd", "csrc", "libaqengine.so.sha256")) as f:
            out["e2e_separate_run"] = dict(e2e, note="NOT measured by this command: tools/bench_e2e.py sweeps of 8-25 k images on this library "
                                                      "build, another run on the same pool of boxes") if e2e.get("library_source_digest") == f.read().strip() else None
    except (OSError, ValueError):
        out["e2e_separate_run"] = None
    if world == 1 and a.e2e_images > 0 and a.variant == "yolov5m":
        eng.close()
        out["e2e"] = e2e_live(a)
    if world == 1 and a.precision == "bf16" and a.parity_steps > 0:
        # what the 1e-4 parity gate costs: the same workload through the fp32 engine (exact-fp32 MFMA, 157 TFLOP/s peak), heuristic tile shapes
        eng.close()
        e32 = Engine(ck, "fp32", local)
        d32 = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
        c32 = torch.zeros((B,), dtype=torch.int32, device=dev)
        e32.infer(tiles_dev[0], 0.25, 0.45, max_det, out=(d32, c32))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.parity_steps):
            e32.infer(tiles_dev[k % a.pool], 0.25, 0.45, max_det, out=(d32, c32))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.parity_steps
        p32 = {"dtype": "fp32", "tiles_per_s": round(B / dt, 1), "ms_per_step": round(1e3 * dt, 2),
               "frac_of_157TF": round(fl["total"] * B / dt / 1e12 / PEAK_F32_TFLOPS, 4), "steps": a.parity_steps,
               "note": "fp32 engine (the --precision fp32 parity gate: boxes/conf within 1e-4 of the oracle), one batch in flight"}
        e32.close()
        del e32
        # the FAST parity mode (fp32 activations, three fp16 MFMAs per product; tests/test_gpu_engine.py::test_infer_f16x3_matches_oracle_detections
        # holds it to the same 1e-4 / identical-count gate): the number to set beside `value` when results must match the reference
        e16 = Engine(ck, "f16x3", local)
        e16.autotune(tiles_dev[0], cache=os.environ.get("AQ_TUNE_CACHE"), shipped=not a.retune)
        e16.infer(tiles_dev[0], 0.25, 0.45, max_det, out=(d32, c32))
        torch.cuda.synchronize()
        n16 = max(a.parity_steps, 6)
        t0 = time.perf_counter()
        for k in range(n16):
            e16.infer(tiles_dev[k % a.pool], 0.25, 0.45, max_det, out=(d32, c32))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n16
        out["parity_mode"] = {"dtype": "f16x3", "tiles_per_s": round(B / dt, 1), "ms_per_step": round(1e3 * dt, 2), "steps": n16,
                              "tile_configs": getattr(e16, "tuned_from", "heuristic"),
                              "note": "fastest mode that meets north_star's gate (boxes / conf within 1e-4 of the fp32 oracle, identical post-NMS counts): fp32 "
                                      "activations, fp16 hi/lo split of both MFMA operands, one batch in flight; `fp32` = the exact-fp32 MFMA engine",
                              "fp32": p32}
        e16.close()
    if world == 1 and not a.no_cpu_baseline:
        eng.close()
        out["cpu_baseline"] = cpu_baseline(ck, a.size, a.cpu_seconds)
    print(json.dumps(out), flush=True)
    if ops_ms is not None and os.environ.get("AQ_BENCH_OPS"):
        for i, o in enumerate(plan.ops):
            print(f"# op {i:3d} cfg {cfgs[i] if cfgs else -1:2d} {o.name:24s} {ops_ms[i]:8.4f} ms  {o.flops_per_tile * B / max(ops_ms[i], 1e-9) / 1e9:9.1f} TFLOP/s",
                  file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
