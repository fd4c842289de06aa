#!/bin/bash
# pipelined-mode sweep (tuning aid): batches in flight x hardware queues
cd "$(dirname "$0")/../.."
for q in 8 16; do
  for n in 3 4 5 6 8; do
    GPU_MAX_HW_QUEUES=$q python bench.py --inflight $n --no-extras --no-cpu-baseline --no-bf16-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('queues $q inflight $n:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
  done
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
