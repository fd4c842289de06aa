#!/bin/bash
# round 4: kernel trace of the bf16 plan (one batch of 8 at 640x480, eager launches) -> gpurun_out/r04b_trace
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
args="--steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-bf16-leg --min-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04b_trace" -o r04b -- python "$repo/bench.py" $args --precision bf16 > "$out/r04b_trace.log" 2>&1
ls "$out"/r04b_trace
