#!/bin/bash
# the folded mask-convolution kernel (csrc/mask_conv.hip) with parts switched off (tuning aid; run on the GPU box):
# MC_EXP is a bit mask: 0 as shipped, 1 no MFMAs, 2 no F fragment reads in the loop, 4 no x loads in the loop, 8 no prologue conversion
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in ${@:-0 1 2 4 8 14 15}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMC_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/mask_conv.hip -o /tmp/mc_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v "mask_conv.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mc_$e.o -o $L
  echo "== MC_EXP=$e"
  timeout 200 python -u tools/probes/mask_conv_time.py 2>&1 | grep "Q=100 bits\|Q= 20\|Q= 16"
done
cp /tmp/ship.so $L
