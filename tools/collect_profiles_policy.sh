#!/bin/bash
# Profile collection for the POLICY-IN-THE-LOOP path (round tag RT, default r5) (run on the GPU box from the repo root), one methodology:
# each workload is tools/policy_profile_run.py (W warm-up + K timed chunks of pw_policy_rollout, HIP-event bracket) under
#   rocprofv3 --kernel-trace --stats                                   (average duration of the timed dispatches)
#   rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA
#   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
#   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES
#   rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE                      (separate passes: MI355X guide)
# (counter passes carry no trace flags: gpurun refuses the mix) and tools/summarize_policy_prof.py turns each into
# gpurun_out/prof_${RT}p/summaries/${RT}_policy_<tag>_summary.json, copied into profiles/.
# Usage: [RT=r5] tools/collect_profiles_policy.sh [tag ...]   (default: c2 tag ref n12 n24 n48)
set -o pipefail
R=$PWD
RT=${RT:-r5}
O=$R/gpurun_out/prof_${RT}p
mkdir -p $O/summaries
cd /tmp; export TMPDIR=/tmp
# tag | kernel family
CONFIGS=("c2|pw_policy_rollout3_kernel" "tag|pw_policy_rollout_tag_kernel" "ref|pw_policy_rollout_ref_kernel" "n12|pw_policy_rollout3_kernel" "n24|pw_policy_rollout" "n48|pw_policy_rollout")
DEFAULT=" c2 tag ref n12 n24 n48 "
want=" $* "
for c in "${CONFIGS[@]}"; do
  IFS='|' read -r tag kern <<< "$c"
  if [ $# -gt 0 ]; then [[ "$want" != *" $tag "* ]] && continue; else [[ "$DEFAULT" != *" $tag "* ]] && continue; fi
  D=$O/$tag; rm -rf $D; mkdir -p $D
  B="python3 $R/tools/policy_profile_run.py --workload $tag"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $B --steps 10 --warmup 3 > $D/line.json 2> $D/trace.err || { echo "$tag: trace pass failed" >&2; tail -3 $D/trace.err >&2; continue; }
  i=0
  for ctrs in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" \
              "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d $D/pmc$i -- $B --steps 4 --warmup 1 > /dev/null 2> $D/pmc$i.err || echo "$tag: pmc pass $i ($ctrs) failed" >&2
  done
  ( cd $R && python3 tools/summarize_policy_prof.py --tag ${RT}_policy_$tag --dir $D --kernel "$kern" --out $O/summaries > $D/summary.txt 2>&1 ) || { echo "$tag: summarize failed" >&2; tail -5 $D/summary.txt >&2; }
  find $D -name '*_kernel_trace.csv' -size +1M -delete; find $D -name '*_counter_collection.csv' -size +1M -delete
  echo "$tag done: $(grep -o '"timed_avg_ns": [0-9.]*' $O/summaries/${RT}_policy_${tag}_summary.json) $(grep -o '"mfma_busy_share": [0-9.]*' $O/summaries/${RT}_policy_${tag}_summary.json)" >&2
done
du -sh $O >&2
