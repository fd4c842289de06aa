#!/bin/bash
# round 3: full GPU test suite + the default bench line
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
cd "$repo"
python -m pytest tests -m gpu -q > "$out/r3_gpu_all3.log" 2>&1
tail -4 "$out/r3_gpu_all3.log"
python bench.py > "$out/r3_bench3.json" 2> "$out/r3_bench3.err"
tail -c 200 "$out/r3_bench3.json"; tail -3 "$out/r3_bench3.err"
