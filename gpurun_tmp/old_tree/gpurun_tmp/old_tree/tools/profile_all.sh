#!/bin/bash
# Round profiles on the GPU box, summarised there (the raw traces exceed what gpurun copies back): kernel trace + PMC passes of the three
# plans (tools/profile_round.sh), the UCN step traces (tools/ucn_step.py), the summaries (tools/summarize_profile.py /
# summarize_ucn_profile.py) written to profiles/ and copied to gpurun_out/prof_out/.   bash tools/profile_all.sh <round tag, e.g. r05>
tag=${1:-r05}
repo=$(cd "$(dirname "$0")/.." && pwd)
cd "$repo"
mkdir -p gpurun_out/prof_out
for pl in f32 bf16 f16; do
  t=${tag}$([ $pl = f32 ] || echo _$pl)
  bash tools/profile_round.sh raw_$pl $pl > /dev/null 2>&1
  o=gpurun_out/raw_$pl
  python tools/summarize_profile.py ${o}_trace $t ${o}_pmc_FETCH_SIZE ${o}_pmc_WRITE_SIZE ${o}_pmc_MfmaUtil ${o}_pmc_LDSBankConflict ${o}_pmc_LdsUtil ${o}_pmc_mfma_raw > gpurun_out/prof_out/summarize_$pl.log 2>&1
  tail -2 ${o}_trace.log > gpurun_out/prof_out/trace_$pl.log
  rm -rf gpurun_out/raw_${pl}_*
done
( cd /tmp && export TMPDIR=/tmp
  for pl in bf16 f16 f32; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/raw_ucn_$pl" -o $tag -- python "$repo/tools/ucn_step.py" --precision $pl --steps 10 > "$repo/gpurun_out/prof_out/ucn_$pl.log" 2>&1
  done )
python tools/summarize_ucn_profile.py gpurun_out/raw_ucn_bf16 $tag bf16 gpurun_out/raw_ucn_f16:f16 gpurun_out/raw_ucn_f32:f32 > /dev/null 2>> gpurun_out/prof_out/summarize_ucn.log
rm -rf gpurun_out/raw_ucn_*
cp profiles/${tag}_* profiles/${tag}*.json profiles/step_traffic*.json profiles/mask_step_traffic.json gpurun_out/prof_out/ 2>/dev/null
ls gpurun_out/prof_out | wc -l
