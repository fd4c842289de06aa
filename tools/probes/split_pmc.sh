#!/bin/bash
# counter passes over the late-round f32_split kernels (3x3 convolution, hill climb): MfmaUtil / LdsUtil / LDSBankConflict / VALUBusy
repo=$(cd "$(dirname "$0")/../.." && pwd)
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in MfmaUtil LdsUtil LDSBankConflict VALUBusy; do
  rocprofv3 --pmc $c --output-format csv -d "$out/split_pmc_conv_$c" -o c -- python "$repo/tools/probes/conv3_split_time.py" > "$out/split_pmc_conv_$c.log" 2>&1
  rocprofv3 --pmc $c --output-format csv -d "$out/split_pmc_hill_$c" -o h -- python "$repo/tools/probes/hill_split_only.py" > "$out/split_pmc_hill_$c.log" 2>&1
done
python - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(dict)
for d in glob.glob(out + "/split_pmc_*_*/"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        v = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            v[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), xs in v.items():
            agg[k][c] = sum(xs) / len(xs)
for k, cs in sorted(agg.items()):
    if any(s in k for s in ("conv3x3", "ms_hill", "gn_apply", "ms_split")):
        print(k, {c: round(x, 1) for c, x in sorted(cs.items())})
P
