#!/bin/bash
# bf16-pipe input projections (csrc/conv_in.hip, conv_in_lp_stream) with parts switched off (tuning aid; run on the GPU box): CL_EXP 0 as
# shipped, 1 no MFMAs, 2 no weight loads, 3 no hi / lo split, 4 no stores
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in 0 1 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCL_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/conv_in.hip -o /tmp/cl_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v conv_in)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/cl_$e.o -o $L
  echo "== CL_EXP=$e"
  timeout 200 python -u tools/probes/conv_in_time.py 2>&1 | grep "lp=True"
done
cp /tmp/ship.so $L
