#!/usr/bin/env python
"""Per-kernel averages of the derived rocprofv3 counters MfmaUtil / LDSBankConflict / LdsUtil (one --pmc pass each,
tools/profile_round.sh) -> profiles/<round>_pmc_mfma_lds.md.
Usage: python tools/summarize_pmc.py gpurun_out/<tag>_pmc_ <round>"""
import collections
import csv
import glob
import os
import sys

prefix, rnd = sys.argv[1], sys.argv[2]
names = ("MfmaUtil", "LDSBankConflict", "LdsUtil")
vals = {n: collections.defaultdict(list) for n in names}
for n in names:
    for f in glob.glob(f"{prefix}{n}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == n:
                vals[n][r["Kernel_Name"]].append(float(r["Counter_Value"]))
kernels = sorted(vals["MfmaUtil"], key=lambda k: -sum(vals["MfmaUtil"][k]))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "profiles", f"{rnd}_pmc_mfma_lds.md"), "w") as out:
    out.write(f"# rocprofv3 --pmc MfmaUtil / LDSBankConflict / LdsUtil ({rnd}), per-dispatch averages (%)\n")
    out.write("command (one pass per counter): `rocprofv3 --pmc <counter> --output-format csv -- python bench.py --steps 2 --warmup 1 "
              "--no-graph --no-cpu-baseline` (tools/profile_round.sh)\n")
    out.write("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs); LDSBankConflict = share of GPU time the LDS is stalled by "
              "bank conflicts; LdsUtil = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE * CUs).  Rows ordered by total MFMA-busy time.\n\n")
    out.write("| kernel | dispatches | MfmaUtil | LDSBankConflict | LdsUtil |\n|---|---:|---:|---:|---:|\n")
    avg = lambda n, k: (sum(vals[n][k]) / len(vals[n][k])) if vals[n].get(k) else float("nan")
    for k in kernels:
        out.write(f"| `{k[:80]}` | {len(vals['MfmaUtil'][k])} | {avg('MfmaUtil', k):.1f} | {avg('LDSBankConflict', k):.2f} | {avg('LdsUtil', k):.1f} |\n")
print("ok")
