#!/bin/bash
# round 3, GPU call 1: parity at the real configs, the default bench line, raw SQ counters for the mask step
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd "$repo"
python -m pytest tests/test_gpu_configs.py -x -q -s > "$out/r3_cfg_tests.log" 2>&1
tail -3 "$out/r3_cfg_tests.log"
python bench.py > "$out/r3_bench1.json" 2> "$out/r3_bench1.err"
tail -c 600 "$out/r3_bench1.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$out/r3_pmc_mfma_raw" -o r3 -- python "$repo/bench.py" --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --min-seconds 0 --no-bf16-leg > "$out/r3_pmc_mfma_raw.log" 2>&1
ls "$out/r3_pmc_mfma_raw" | head
