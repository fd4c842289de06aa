#!/bin/bash
# round 3 (late): kernel traces of the f32_split and bf16 plans + the mean-shift unit in both forms
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
args="--steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-bf16-leg --min-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03s_trace" -o r03s -- python "$repo/bench.py" $args --precision f32_split > "$out/r03s_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03b_trace" -o r03b -- python "$repo/bench.py" $args --precision bf16 > "$out/r03b_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03m_trace" -o r03m -- python "$repo/tools/probes/meanshift_time.py" > "$out/r03m_trace.log" 2>&1
ls "$out"/r03s_trace "$out"/r03b_trace "$out"/r03m_trace
