#!/bin/bash
# round 3, GPU call: full GPU test suite, the default bench line, the round's rocprof passes
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
cd "$repo"
python -m pytest tests -m gpu -q > "$out/r3_gpu_all2.log" 2>&1
tail -4 "$out/r3_gpu_all2.log"
python bench.py > "$out/r3_bench2.json" 2> "$out/r3_bench2.err"
tail -c 300 "$out/r3_bench2.json"
bash tools/profile_round.sh r03 | tail -3
