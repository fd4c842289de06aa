#!/bin/bash
# the fused K/V + attention kernel (csrc/attention.hip, hs_attn_fkv_kernel) with parts switched off (tuning aid; run on the GPU box):
# FK_EXP 0 as shipped, 1 no constant loads, 2 no mask loads, 3 no x loads, 4 no exponentials, 5 no P V MFMAs
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in ${@:-0 1 2 3 4 5}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFK_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/attention.hip -o /tmp/fk_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v "attention.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fk_$e.o -o $L
  echo "== FK_EXP=$e"
  timeout 200 python -u tools/probes/fkv_time.py fused-only 2>&1 | grep "fused K/V"
done
cp /tmp/ship.so $L
