#!/bin/bash
# round 3 (late): GPU tests touched by the f32_split work + the f32_split bench entries
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
cd "$repo"
python -m pytest tests/test_gpu_configs.py tests/test_gpu_modules.py -q -x -k "split or config1 or pixel_decoder or mean_shift" > "$out/r3_gpu_split.log" 2>&1
tail -4 "$out/r3_gpu_split.log"
python bench.py --precision f32_split --no-cpu-baseline --no-extras --no-bf16-leg > "$out/r3_bench_split.json" 2> "$out/r3_bench_split.err"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r3_bench_split.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("one_batch_in_flight"))
P
tail -3 "$out/r3_bench_split.err"
