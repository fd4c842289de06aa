#!/bin/bash
# round 3 (late): the round's rocprof passes on the final code + a graph-replay trace of one batch in flight
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
cd "$repo"
bash tools/profile_round.sh r03 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/r03_gtrace" -o g -- python "$repo/bench.py" --steps 20 --warmup 3 --inflight 1 --no-cpu-baseline --no-extras --no-bf16-leg --min-seconds 0 > "$out/r03_gtrace.log" 2>&1
ls "$out/r03_gtrace"
