#!/bin/bash
# bf16-plan encoder block (csrc/enc_lp.hip) with parts switched off (tuning aid; run on the GPU box): EH_EXP 0 as shipped, 1 no LDS fragment
# reads, 2 no MFMAs, 3 no stores
cd "$(dirname "$0")/../.."
L=unseenobjectswithmeanshift_amd/libmsm_hip.so
cp $L /tmp/ship.so
for e in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DEH_EXP=$e -c unseenobjectswithmeanshift_amd/csrc/enc_lp.hip -o /tmp/eh_$e.o 2>/dev/null
  objs=$(ls unseenobjectswithmeanshift_amd/build/*.o | grep -v enc_lp)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/eh_$e.o -o $L
  echo "== EH_EXP=$e"
  timeout 200 python -u tools/probes/enc_lp_time.py 2>&1 | grep "enc_block hm"
done
cp /tmp/ship.so $L
