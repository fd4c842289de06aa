#!/bin/bash
# Round profile on the GPU box: kernel trace + separate PMC passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil, LDSBankConflict, LdsUtil, raw MFMA counters).
# Usage (through gpurun): bash tools/profile_round.sh <tag> [precision]; outputs under gpurun_out/<tag>_*; summarise with
# tools/summarize_profile.py (precision: f32 (default), bf16, f16, f32_split -- the plan bench.py runs under the profiler)
tag=${1:-r03}
prec=${2:-f32}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-bf16-leg --min-seconds 0 --precision $prec"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -o "$tag" -- python "$repo/bench.py" $args > "$out/${tag}_trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil LDSBankConflict LdsUtil; do
  rocprofv3 --pmc $c --output-format csv -d "$out/${tag}_pmc_$c" -o "$tag" -- python "$repo/bench.py" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --no-bf16-leg --min-seconds 0 --precision $prec > "$out/${tag}_pmc_$c.log" 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$out/${tag}_pmc_mfma_raw" -o "$tag" -- python "$repo/bench.py" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --no-bf16-leg --min-seconds 0 --precision $prec > "$out/${tag}_pmc_mfma_raw.log" 2>&1
ls "$out" | grep "^${tag}_"
