#!/bin/bash
# K/V projection with parts switched off (tuning aid; run on the GPU box): KP_EXP 0 as shipped, 1 no stores, 2 no MFMAs, 3 stores as whole lines (wrong places, same bytes)
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DKP_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/kv_proj.hip -o /tmp/kv_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v kv_proj)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/kv_$e.o -o $L
  echo "== KP_EXP=$e"
  timeout 200 python -u tools/probes/kv_multi_time.py 2>&1 | grep -v amdgpu | grep "dense"
done
cp /tmp/ship.so $L
