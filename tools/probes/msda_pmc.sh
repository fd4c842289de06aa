#!/bin/bash
# SQ counters of the MSDeformAttn gather variants (tuning aid): bash tools/probes/msda_pmc.sh
repo=$(cd "$(dirname "$0")/../.." && pwd)
out=$repo/gpurun_out/msda_pmc
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out/a" -o a -- python "$repo/tools/probes/msda_variants.py" > "$out/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d "$out/b" -o b -- python "$repo/tools/probes/msda_variants.py" > "$out/b.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t" -o t -- python "$repo/tools/probes/msda_variants.py" > "$out/t.log" 2>&1
python3 - "$out" <<'PY'
import csv, collections, sys, glob
out = sys.argv[1]
for tag in ("a", "b"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        if "msda" in k or "enc_block" in k:
            print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}, len(next(iter(v.values()))))
for f in glob.glob(f"{out}/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda" in r["Name"] or "enc_block" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
