#!/bin/bash
# encoder block (bf16 / split plans) with parts switched off (tuning aid; run on the GPU box): ES_EXP 0 as shipped, 1 no LDS fragment reads, 2 no MFMAs, 3 no stage barriers / DMA waits
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DES_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/enc_block_split.hip -o /tmp/es_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v enc_block_split)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/es_$e.o -o $L
  echo "== ES_EXP=$e"
  timeout 200 python -u tools/probes/enc_lp_time.py 2>&1 | grep -v amdgpu
done
cp /tmp/ship.so $L
