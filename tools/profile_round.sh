#!/bin/bash
# Round profile on the GPU box: kernel trace + separate PMC passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil, LDSBankConflict, LdsUtil).
# Usage (through gpurun): bash tools/profile_round.sh <tag>; outputs under gpurun_out/<tag>_*; summarise with tools/summarize_profile.py
tag=${1:-r02}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -o "$tag" -- python "$repo/bench.py" --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-extras --min-seconds 0 > "$out/${tag}_trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil LDSBankConflict LdsUtil; do
  rocprofv3 --pmc $c --output-format csv -d "$out/${tag}_pmc_$c" -o "$tag" -- python "$repo/bench.py" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --min-seconds 0 > "$out/${tag}_pmc_$c.log" 2>&1
done
ls "$out" | grep "^${tag}_"
