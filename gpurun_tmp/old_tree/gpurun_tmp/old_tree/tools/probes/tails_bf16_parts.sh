#!/bin/bash
# decoder tails (csrc/dec_chain.hip) with parts switched off (tuning aid; run on the GPU box): DC_EXP 0 as shipped, 1 no MFMAs, 2 no weight loads
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in 0 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDC_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/dec_chain.hip -o /tmp/dc_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v dec_chain)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/dc_$e.o -o $L
  echo "== DC_EXP=$e"
  timeout 200 python -u tools/probes/tails_bf16_time.py 2>&1 | grep -v amdgpu
done
cp /tmp/ship.so $L
