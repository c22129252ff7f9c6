#!/usr/bin/env python3
"""Build a VARIANT of libaqengine.so whose generated planar 3x3 assembly was produced with other AQ_GEN_* switches, for same-box A/B
runs (AQ_ENGINE_LIB=<path> selects it at load time; tools/time_conv3x3.py --lib <path>):

    python tools/build_gen_variant.py aacc AQ_GEN_A_ACC=1            # -> build/variant_aacc/libaqengine.so

The in-tree library must be built first: every object but conv3x3_pl.o is reused from aquaculture_amd/csrc/."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aquaculture_amd import build as B  # noqa: E402


def run(cmd, env=None):
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    return r.stdout


def main():
    name, switches = sys.argv[1], sys.argv[2:]
    B.build()
    out = os.path.join(ROOT, "build", f"variant_{name}")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ)
    for kv in switches:
        k, v = kv.split("=", 1)
        assert k.startswith("AQ_GEN_"), k
        env[k] = v
    cc = B.hipcc()
    llvm = B.llvm_bin(cc)
    stem = "conv3x3_pl_asm"
    src, obj, co = (os.path.join(out, stem + ext) for ext in (".s", ".o", ".hsaco"))
    print(run([sys.executable, os.path.join(B.CSRC, "gen_conv3x3_pl_asm.py"), src], env).strip().splitlines()[-1])
    run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={B.ARCH}", "-c", src, "-o", obj])
    run([os.path.join(llvm, "ld.lld"), "-shared", obj, "-o", co])
    data = open(co, "rb").read()
    with open(os.path.join(out, stem + "_hsaco.inc"), "w") as f:
        for i in range(0, len(data), 32):
            f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
    # the .hip file includes "conv3x3_pl_asm_hsaco.inc" relative to itself: compile a copy that sits beside the variant's code object
    for h in os.listdir(B.CSRC):
        if h.endswith(".h") or h == "conv3x3_pl.hip":
            shutil.copy(os.path.join(B.CSRC, h), os.path.join(out, h))
    pl_obj = os.path.join(out, "conv3x3_pl.o")
    run([cc, *B.COMMON, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(out, "conv3x3_pl.hip"), "-o", pl_obj])
    objs = [pl_obj if s == "conv3x3_pl.hip" else os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s, _ in B.SOURCES]
    lib = os.path.join(out, "libaqengine.so")
    run([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, *objs])
    print(lib)


if __name__ == "__main__":
    main()
