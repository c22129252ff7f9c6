#!/usr/bin/env python3
"""Hardware counters per kernel NAME for an arbitrary command (runs ON the GPU box): one `rocprofv3 --pmc` pass per counter group
(kernel-trace only), averaged over the dispatches of each kernel whose name contains --match.

    python tools/pmc_kernel.py --out gpurun_out/pmc_pl --match conv3x3_pl -- python3 tools/time_conv3x3.py --reps 3 --old '' --nbs ''
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

GROUPS = {
    "sq1": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE",
    "sq2": "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM",
    "sq3": "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS_LOAD_DMA",
    "ic": "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES",
    "ic2": "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES",
    "tcp": "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum",
    "ta": "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "l2": "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--match", default="")
    ap.add_argument("--groups", nargs="+", default=["sq1", "sq2", "fetch", "write"])
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    cwd = os.getcwd()
    table = defaultdict(lambda: defaultdict(list))
    for g in a.groups:
        d = os.path.join(out, g)
        os.makedirs(d, exist_ok=True)
        full = ["rocprofv3", "--kernel-trace", "--pmc", *GROUPS[g].split(), "--output-format", "csv", "-d", d, "-o", "pmc", "--", *cmd]
        r = subprocess.run(full, cwd=cwd, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        with open(os.path.join(d, "log.txt"), "w") as f:
            f.write(r.stdout[-6000:] + "\n---\n" + r.stderr[-6000:])
        if r.returncode != 0:
            print(f"group {g}: rocprofv3 rc={r.returncode}: {r.stderr[-400:]}", file=sys.stderr)
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print(f"group {g}: no counter file", file=sys.stderr)
            continue
        per = defaultdict(lambda: defaultdict(float))
        meta = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if a.match and a.match not in row["Kernel_Name"]:
                    continue
                did = int(row["Dispatch_Id"])
                per[did][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[did] = (row["Kernel_Name"], row.get("Grid_Size"), row.get("LDS_Block_Size"), row.get("VGPR_Count"), row.get("Accum_VGPR_Count"))
        for did, cs in per.items():
            key = meta[did][0][:90] + f" grid={meta[did][1]}"
            for k, v in cs.items():
                table[key][k].append(v)
    res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_dispatches": max(len(v) for v in cs.values())} for k, cs in table.items()}
    with open(os.path.join(out, "pmc_summary.json"), "w") as f:
        json.dump(res, f, indent=1)
    for k, cs in res.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"    {c:36s} {v:16.1f}")


if __name__ == "__main__":
    main()
