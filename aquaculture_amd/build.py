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
JPEG_LIB = os.path.join(CSRC, "libaqjpeg.so")    # host half of the split JPEG decode: plain C (gcc), no HIP -- the decode worker processes load it
ARCH = "gfx950"

SOURCES = [
    # (file, extra flags)
    ("conv_igemm.hip", []),
    ("conv_halo.hip", []),
    ("stem_conv.hip", []),
    ("bottleneck.hip", []),
    ("downblock.hip", []),
    ("conv1x1_direct.hip", []),
    ("conv1x1_asm.hip", []),                     # host side of the generated-assembly wide 1x1 (gen_conv1x1_asm.py; round 4)
    ("conv3x3_pl.hip", []),
    ("pointwise.hip", []),
    ("detect_nms.hip", ["-ffp-contract=off"]),   # bit-level parity with the oracle's fp32 op order
    ("head_decode.hip", ["-ffp-contract=off"]),  # the same decode arithmetic, fused behind the Detect head convs
    ("jpeg_idct.hip", []),                       # device half of the split JPEG decode (IDCT, chroma upsampling, colour conversion)
    ("jpeg_huff.hip", []),                       # GPU entropy decode (round 4): the Huffman stage, one lane per restart segment
    ("engine.cpp", ["-x", "hip"]),
]
# -packed-fp32-ops: no v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 in compiled kernels.  The SiLU epilogues run beside other waves' MFMAs (two
# or three waves per SIMD in the fused kernels and the direct 1x1 kernel); there a packed fp32 instruction takes 20.7 cycles against 8.3 for a
# plain one (tools/ubench/valu_issue.hip, profiles/NOTES_r04.md): measured in the engine, same box, interleaved: C = 96 Bottlenecks - 2.5 %, direct
# 1x1 layers - 3 %, whole step + 0.5-1.5 % tiles/s.  (The host pass of each compile warns that it does not know the feature; it is a device feature.)
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libaqengine.so")
    return exe


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".cpp", ".c", ".h", "_asm.py")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "aq_engine.h"), "rb").read())
    h.update(repr((SOURCES, COMMON)).encode())
    # the assembly generator reads experiment switches from the environment at build time: a different kernel is a different digest
    h.update(repr(sorted((k, v) for k, v in os.environ.items() if k.startswith("AQ_GEN_"))).encode())
    return h.hexdigest()


def llvm_bin(cc: str) -> str:
    """Directory of the ROCm clang / ld.lld that belong to the hipcc in use (ROCM_PATH, or beside hipcc, or /opt/rocm)."""
    cands = []
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(cc))), "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for d in cands:
        if os.path.exists(os.path.join(d, "clang")) and os.path.exists(os.path.join(d, "ld.lld")):
            return d
    raise RuntimeError(f"ROCm clang / ld.lld not found in any of {cands}")


def _assemble_planar(verbose: bool, cc: str) -> None:
    """The hand-scheduled assembly kernels: generate each .s (gen_conv3x3_pl_asm.py: the planar 3x3 families; gen_bottleneck_asm.py: the fused
    C = 48 Bottleneck; gen_conv1x1_asm.py: the wide 1x1), assemble and link it into a gfx950 code object with the ROCm clang / lld, and write it as a byte list that the
    .hip file of the same family embeds (hipModuleLoadData)."""
    llvm = llvm_bin(cc)
    for gen, stem in (("gen_conv3x3_pl_asm.py", "conv3x3_pl_asm"), ("gen_bottleneck_asm.py", "bottleneck_asm"), ("gen_bottleneck96_asm.py", "bottleneck96_asm"),
                      ("gen_conv1x1_asm.py", "conv1x1_asm")):
        src, obj, co = (os.path.join(CSRC, stem + ext) for ext in (".s", ".o", ".hsaco"))
        cmds = [[sys.executable, os.path.join(CSRC, gen), src],
                [os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", src, "-o", obj],
                [os.path.join(llvm, "ld.lld"), "-shared", obj, "-o", co]]
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"{cmd[0]} failed:\n{r.stdout}\n{r.stderr}")
        data = open(co, "rb").read()
        inc = os.path.join(CSRC, stem + "_hsaco.inc")
        with open(inc + ".tmp", "w") as f:
            for i in range(0, len(data), 32):
                f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
        os.replace(inc + ".tmp", inc)


def build_jpeg_lib() -> str:
    """libaqjpeg.so: csrc/jpeg_coef.c with the host C compiler (no GPU toolchain involved)."""
    cc = shutil.which("gcc") or shutil.which("cc") or shutil.which("clang")
    if not cc:
        raise RuntimeError("no C compiler for libaqjpeg.so")
    r = subprocess.run([cc, "-O3", "-shared", "-fPIC", "-Wall", "-pthread", "-o", JPEG_LIB + ".tmp", os.path.join(CSRC, "jpeg_coef.c")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{cc} failed on jpeg_coef.c:\n{r.stdout}\n{r.stderr}")
    os.replace(JPEG_LIB + ".tmp", JPEG_LIB)
    return JPEG_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    stamp = LIB + ".sha256"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(JPEG_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    # ranks (or test workers) that find a stale library at the same time: one builds, the others wait for the lock and find it fresh
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB) and os.path.exists(JPEG_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
            return LIB
        return _build_locked(dig, stamp, verbose)


def _build_locked(dig: str, stamp: str, verbose: bool) -> str:
    cc = hipcc()
    objs = []
    _assemble_planar(verbose, cc)

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
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)              # a process that has the old library mapped keeps its inode
    build_jpeg_lib()
    with open(stamp + ".tmp", "w") as f:
        f.write(dig)
    os.replace(stamp + ".tmp", stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
