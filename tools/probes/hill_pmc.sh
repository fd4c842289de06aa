#!/bin/bash
# counter passes over the f32_split hill climb (tuning aid)
repo=$(cd "$(dirname "$0")/../.." && pwd)
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/hill_trace" -o h -- python "$repo/tools/probes/hill_split_only.py" > "$out/hill_trace.log" 2>&1
grep -E "hill|split_planes" "$out/hill_trace/h_kernel_stats.csv" | cut -c1-160
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d "$out/hill_pmc_$tag" -o h -- python "$repo/tools/probes/hill_split_only.py" > "$out/hill_pmc_$tag.log" 2>&1
  python - "$out/hill_pmc_$tag" <<'P'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if "ms_hill_planes" in r["Kernel_Name"] or "ms_hill_split" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:28s} per dispatch {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")
P
done
