"""Builds libaqengine.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
``aquaculture_amd/csrc/libaqengine.so`` travels with the tree to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libaqengine.so")
ARCH = "gfx950"

SOURCES = [
    # (file, extra flags)
    ("conv_igemm.hip", []),
    ("conv_halo.hip", []),
    ("stem_conv.hip", []),
    ("bottleneck.hip", []),
    ("downblock.hip", []),
    ("conv1x1_direct.hip", []),
    ("conv3x3_pl.hip", []),
    ("pointwise.hip", []),
    ("detect_nms.hip", ["-ffp-contract=off"]),   # bit-level parity with the oracle's fp32 op order
    ("head_decode.hip", ["-ffp-contract=off"]),  # the same decode arithmetic, fused behind the Detect head convs
    ("engine.cpp", ["-x", "hip"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libaqengine.so")
    return exe


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".cpp", ".h", "_asm.py")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "aq_engine.h"), "rb").read())
    h.update(repr((SOURCES, COMMON)).encode())
    return h.hexdigest()


def _assemble_planar(verbose: bool) -> None:
    """The hand-scheduled assembly build of the planar 3x3 kernel: generate the .s (gen_conv3x3_pl_asm.py), assemble and link it into a
    gfx950 code object with the ROCm clang / lld, and write it as a byte list that conv3x3_pl.hip embeds (hipModuleLoadData)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    src, obj, co = (os.path.join(CSRC, n) for n in ("conv3x3_pl_asm.s", "conv3x3_pl_asm.o", "conv3x3_pl_asm.hsaco"))
    cmds = [[sys.executable, os.path.join(CSRC, "gen_conv3x3_pl_asm.py"), src],
            [os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", src, "-o", obj],
            [os.path.join(llvm, "ld.lld"), "-shared", obj, "-o", co]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{cmd[0]} failed:\n{r.stdout}\n{r.stderr}")
    data = open(co, "rb").read()
    with open(os.path.join(CSRC, "conv3x3_pl_asm_hsaco.inc"), "w") as f:
        for i in range(0, len(data), 32):
            f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")


def build(force: bool = False, verbose: bool = False) -> str:
    stamp = LIB + ".sha256"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    cc = hipcc()
    objs = []
    _assemble_planar(verbose)

    def compile_one(item):
        src, extra = item
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [cc, *COMMON, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
