#!/usr/bin/env python3
"""Build a VARIANT of libaqengine.so beside the in-tree one, for same-box A/B runs (AQ_ENGINE_LIB=<path> selects it at load time):

    python tools/build_variant.py nopk -Xclang -target-feature -Xclang -packed-fp32-ops        # -> build/variant_nopk/libaqengine.so
    python tools/build_variant.py nopk --only bottleneck.hip downblock.hip -- -Xclang ...       # extra flags for some sources only

The in-tree library must be built first (this reuses its generated assembly includes)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aquaculture_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    rest = sys.argv[2:]
    only = None
    if rest and rest[0] == "--only":
        i = rest.index("--")
        only, rest = set(rest[1:i]), rest[i + 1:]
    B.build()
    out = os.path.join(ROOT, "build", f"variant_{name}")
    os.makedirs(out, exist_ok=True)
    cc = B.hipcc()

    def one(item):
        src, extra = item
        obj = os.path.join(out, os.path.splitext(src)[0] + ".o")
        flags = rest if (only is None or src in only) else []
        r = subprocess.run([cc, *B.COMMON, *extra, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{src}: {r.stderr[-2000:]}")
        return obj
    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(one, B.SOURCES))
    lib = os.path.join(out, "libaqengine.so")
    r = subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    print(lib)


if __name__ == "__main__":
    main()
