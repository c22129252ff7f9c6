"""Build-time check of the one place where registers are loaded by inline asm and waited for by hand.

`bottleneck_kernel` (C = 96 shape) re-reads the shortcut from global memory with `asm volatile("global_load_dwordx2 ...")` so that the
compiler does not put `s_waitcnt vmcnt(0)` in front of the MFMA loop (that wait also drained the next patch's LDS-DMA).  The price: the
compiler does not know those registers are in flight until the hand-written `s_waitcnt vmcnt(0)` after the loop.  This test compiles the
file to gfx950 assembly (no GPU needed) and checks that NO instruction between the loads and that wait reads or writes them -- a
register-allocator copy or spill there would read stale data.  It also checks that the compiler inserted no vmcnt(0) of its own in the
steady-state loop of that kernel."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(text):
    r = set()
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", text):
        r |= {int(m.group(1))} if m.group(1) else set(range(int(m.group(2)), int(m.group(3)) + 1))
    return r


@pytest.fixture(scope="module")
def bottleneck_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "bottleneck.s"
    src = os.path.join(ROOT, "aquaculture_amd", "csrc", "bottleneck.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{os.path.dirname(src)}", "--cuda-device-only",
                    "-S", src, "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


def test_asm_loaded_shortcut_registers_are_untouched_until_the_wait(bottleneck_asm):
    kernels = re.split(r"\n(?=_ZN\S*bottleneck_kernel\S*:\s*; @)", bottleneck_asm)[1:]
    checked = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if name.endswith("ELb1EEEvNS_9BtlParamsE"):
            continue                                        # stamped diagnostic builds (tools/stamp_conv.py): timing probes, not the product
        body = k.split("s_endpgm")[0].split("\n")
        loads = [i for i, l in enumerate(body) if "global_load_dwordx2" in l and "ASMSTART" in body[i - 1]]
        if not loads:
            continue                                        # shapes that read the shortcut from LDS
        loaded = set()
        for i in loads:
            loaded |= _regs(body[i].split(",")[0])
        wait = next(i for i, l in enumerate(body) if i > loads[-1] and "s_waitcnt vmcnt(0)" in l and "ASMSTART" in body[i - 1])
        for i in range(loads[0] + 1, wait):
            line = body[i].split(";")[0]
            if i in loads:
                assert not (_regs(line.split(",", 1)[1]) & loaded - _regs(line.split(",")[0])), body[i]   # address regs are not in-flight ones
                continue
            assert not (_regs(line) & loaded), f"{body[i].strip()} touches an in-flight register"
        # no compiler-inserted vmcnt(0) inside the loops of this kernel (the hand-written ones sit between ASMSTART / ASMEND)
        in_loop = False
        for i, l in enumerate(body):
            if l.startswith(".LBB"):
                in_loop = "Loop" in l
            if in_loop and "s_waitcnt" in l and "vmcnt(0)" in l:
                assert "ASMSTART" in body[i - 1], f"compiler-inserted {l.strip()} in a loop"
        checked += 1
    assert checked >= 1


# ----------------------------------------------------------------------------------------------------------------------------------
# conv3x3_pl.hip: EVERY vector-memory operation of the kernel is inline asm counted by hand, so the compiler must never (a) add a
# vector-memory operation of its own (a scratch spill counts on vmcnt), (b) touch an asm-loaded register before the wait that covers it
# (a copy or an AGPR spill there reads the old contents), (c) move accumulators around inside the MFMA stream.
@pytest.fixture(scope="module")
def planar_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "conv3x3_pl.s"
    src = os.path.join(ROOT, "aquaculture_amd", "csrc", "conv3x3_pl.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{os.path.dirname(src)}", "--cuda-device-only",
                    "-S", src, "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


def test_planar_kernel_hand_counted_memory_operations(planar_asm):
    import bisect
    kernels = re.split(r"\n(?=_ZN\S*conv3x3_pl_kernel\S*:\s*; @)", planar_asm)[1:]
    assert len(kernels) >= 6
    for k in kernels:
        name = k.split(":", 1)[0]
        body = k.split(".Lfunc_end")[0].split("\n")
        assert not [l for l in body if "scratch_" in l], f"{name}: scratch spill (a vector-memory operation the hand counts do not know)"
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        lo, hi = mf[0], mf[-1]
        assert not [l for l in body[lo:hi] if "v_accvgpr" in l], f"{name}: accumulator traffic inside the MFMA stream"
        for l in body[lo:hi + 1]:
            m = re.search(r"v_mfma\S+ (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], (a\[\d+:\d+\])", l)
            assert m is None or m.group(1) == m.group(2), f"{name}: {l.strip()} does not accumulate in place"
        waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\(\d+\)\s*$", l) and "ASMSTART" in body[i - 1]]
        # weight fragments (dwordx4, issued in tap T for tap T + 2): untouched until the second hand-written wait after them;
        # residual loads (dwordx2): batch 0 at the chunk top until the third wait after it, the epilogue batches until the next wait
        for i, l in enumerate(body):
            m = re.search(r"global_load_dwordx([24]) v\[(\d+):(\d+)\]", l)
            if not m or "ASMSTART" not in body[i - 1]:
                continue
            dest = set(range(int(m.group(2)), int(m.group(3)) + 1))
            kth = bisect.bisect_right(waits, i)
            if m.group(1) == "4":
                if not lo <= i <= hi:
                    continue                                 # prologue loads: waited for by the vmcnt(12) right behind them
                end = min(waits[kth + 1], hi) if kth + 1 < len(waits) else hi
            else:
                end = waits[kth + 2] if i < lo else (waits[kth] if kth < len(waits) else len(body))
            for j in range(i + 1, end):
                line = body[j].split(";")[0]
                if not line.strip() or "global_load_dwordx" in line:
                    continue
                assert not (_regs(line) & dest), f"{name}: '{line.strip()}' touches a register that '{l.strip()}' is still loading"


# ----------------------------------------------------------------------------------------------------------------------------------
# The generated assembly build of the planar kernel (csrc/gen_conv3x3_pl_asm.py): it must assemble for gfx950 with the ROCm clang, and
# every family must fit the occupancy it is built for (registers per wave, LDS per workgroup) -- the generator asserts the same, this
# checks what the assembler was actually told.
def test_generated_planar_assembly_fits_its_occupancy(tmp_path):
    import sys
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("ROCm clang not available")
    src = tmp_path / "pl.s"
    gen = os.path.join(ROOT, "aquaculture_amd", "csrc", "gen_conv3x3_pl_asm.py")
    subprocess.run([sys.executable, gen, str(src)], check=True, capture_output=True, env=dict(os.environ, AQ_GEN_EXPERIMENTAL="1"))   # every family, shipped or not
    subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "pl.o")],
                   check=True, capture_output=True)
    text = src.read_text()
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    names = [k for k, _ in kernels]
    for want in ("conv3x3_pl_asm_nb13_res0", "conv3x3_pl_asm_nb13_res1", "conv3x3_pl_asm_nb7_res1", "conv3x3_pl_asm_nb8_res0", "conv3x3_pl_asm_nb13_res1_w8",
                 "conv3x3_pl_asm_s2nb13_res0", "conv3x3_pl_asm_f8nb13_res0", "conv3x3_pl_asm_f8nb13_res1", "conv3x3_pl_asm_pm13_res0", "conv3x3_pl_asm_pm13_res1"):     # round 3: stride-2 and fp8 families
        assert want in names
    for name, body in kernels:
        regs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        sgpr = int(re.search(r"\.amdhsa_next_free_sgpr (\d+)", body).group(1))
        occ = 1 if ("nb13_" in name or "_pm13" in name) else 2       # (nb13, pm13*, s2nb13, f8nb13: one workgroup per CU)
        assert regs <= 512 // occ and lds * occ <= 160 * 1024 and sgpr <= 102, (name, regs, lds, sgpr)
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) == 0      # no scratch: every memory operation is counted by hand
    # every hand-counted wait fits the 6-bit vmcnt field, and no kernel relies on a compiler: there is none
    assert all(int(n) <= 63 for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)", text))


# The 16-byte output stores of the pixel-major / stride-2 / fp8 families (gen_conv3x3_pl_asm.py W16): 13 + 6 stores of
# 16 bytes and one of 8 per tile and activation branch instead of 39 of 8, and the hazards the assembler does not pad -- a VALU write
# of either operand is at least two instructions ahead of the v_permlane16_swap that reads it, and the data registers of a 16-byte
# store are not written by the instruction right behind it.
def test_generated_planar_assembly_wide_stores(tmp_path):
    import sys
    src = tmp_path / "pl.s"
    gen = os.path.join(ROOT, "aquaculture_amd", "csrc", "gen_conv3x3_pl_asm.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("AQ_GEN_")}
    subprocess.run([sys.executable, gen, str(src)], check=True, capture_output=True, env=env)
    text = src.read_text()
    for fam, copies in (("pm13w40_res1", 2), ("pm13w20_res0", 2), ("s2nb13_res0", 2), ("f8nb13_res1", 2)):
        body = text.split(f"\nconv3x3_pl_asm_{fam}:\n", 1)[1].split("s_endpgm", 1)[0].split("\n")
        ins = [l.split(";")[0].strip() for l in body if l.startswith("\t")]
        out_stores = [l for l in ins if l.startswith("global_store_dwordx") and "offset:" in l or l.startswith("global_store_dwordx4")]
        assert sum(l.startswith("global_store_dwordx4") for l in out_stores) == 19 * copies, fam          # (act / no-act copies of the epilogue)
        assert sum(l.startswith("global_store_dwordx2") for l in ins) == 1 * copies, fam
        for i, l in enumerate(ins):
            if l.startswith("v_permlane16_swap_b32"):
                ops = _regs(l)
                for back in (1, 2):
                    prev = ins[i - back]
                    if prev.startswith(("v_", "ds_read", "global_load")) and not prev.startswith("v_permlane16_swap"):
                        dst = re.match(r"\S+ (v\d+|v\[\d+:\d+\])", prev)
                        assert not dst or not (_regs(dst.group(1)) & ops), f"{fam}: '{prev}' writes an operand of '{l}' {back} instruction(s) ahead"
            if l.startswith("global_store_dwordx4"):
                data = _regs(l.split(",")[1])
                for ahead in (1, 2):                     # gfx940+: two wait states between a store of more than 8 bytes and a VALU write of its data
                    nxt = re.match(r"v_\S+ (v\d+|v\[\d+:\d+\])", ins[i + ahead])
                    assert not nxt or not (_regs(nxt.group(1)) & data), f"{fam}: '{ins[i + ahead]}' rewrites the data of '{l}'"
