#!/bin/bash
# Profile collection of the ENVIRONMENT kernels (round tag RT, default r5), ONE methodology for every BASELINE configuration (run on the GPU box from the repo root):
# each configuration is the bench.py command itself (1 bench step = 1 pw_rollout launch of --chunk steps, HIP-event
# bracket, multi-slot output ring) under
#   rocprofv3 --kernel-trace --stats                  (average duration of the timed dispatches)
#   rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE     (separate passes, no trace flags: MI355X guide; gpurun refuses the mix)
#   rocprofv3 --pmc SQ_* (two passes)                 (VALU issue share)
# and tools/summarize_prof.py turns each into gpurun_out/prof_$RT/summaries/${RT}_<tag>_summary.json (+ _kernel_stats.csv),
# which are copied into profiles/.  Every summary records the hash of the kernel sources it was collected from
# (multiagent_rl_amd/build_native.py kernel_source_hash): bench.py refuses to quote a summary of other kernel code.
# Usage: [RT=r5] tools/collect_profiles_env.sh [tag ...]   (default: all)
set -o pipefail
R=$PWD
RT=${RT:-r5}
O=$R/gpurun_out/prof_$RT
mkdir -p $O/summaries
export PW_BENCH_NO_POLICY=1
cd /tmp; export TMPDIR=/tmp
# tag | bench arguments | kernel family of the dominant kernel
CONFIGS=(
  "c2_b4096|--envs 4096 --agents 6 --chunk 1000|pw_spread_quad_kernel"
  "c3_b8192|--scenario simple_tag --envs 8192 --chunk 1000|pw_tag_"
  "c5_n3|--envs 4096 --agents 3 --chunk 1000|pw_spread_"
  "c5_n12|--envs 4096 --agents 12 --chunk 1000|pw_spread_"
  "c5_n24|--envs 4096 --agents 24 --chunk 500|pw_spread_"
  "c5_n48|--envs 4096 --agents 48 --chunk 200|pw_spread_"
  "b65536_n6|--envs 65536 --agents 6 --chunk 100|pw_spread_"
)
want=" $* "
for c in "${CONFIGS[@]}"; do
  IFS='|' read -r tag bargs kern <<< "$c"
  if [ $# -gt 0 ] && [[ "$want" != *" $tag "* ]]; then continue; fi
  D=$O/$tag; rm -rf $D; mkdir -p $D
  B="python3 $R/bench.py --no-cpu-baseline $bargs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $B --steps 10 --warmup 3 > $D/bench_line.json 2> $D/trace.err || { echo "$tag: trace pass failed" >&2; tail -3 $D/trace.err >&2; continue; }
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $D/pmc_$ctr -- $B --steps 4 --warmup 1 > /dev/null 2> $D/pmc_$ctr.err || echo "$tag: $ctr pass failed" >&2
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $D/sq1 -- $B --steps 4 --warmup 1 > /dev/null 2> $D/sq1.err || echo "$tag: sq1 failed" >&2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $D/sq2 -- $B --steps 4 --warmup 1 > /dev/null 2> $D/sq2.err || echo "$tag: sq2 failed" >&2
  ( cd $R && python3 tools/summarize_prof.py --tag ${RT}_$tag --stats $D/trace --fetch $D/pmc_FETCH_SIZE --write $D/pmc_WRITE_SIZE \
      --kernel "$kern" --bench $D/bench_line.json --sq $D/sq1 $D/sq2 --out $O/summaries > $D/summary.txt 2>&1 ) || { echo "$tag: summarize failed" >&2; tail -5 $D/summary.txt >&2; }
  # keep what travels back small: the per-dispatch traces and counter dumps are summarised above
  find $D -name '*_kernel_trace.csv' -size +1M -delete; find $D -name '*_counter_collection.csv' -size +1M -delete
  echo "$tag done: $(grep -o '"timed_avg_ns": [0-9.]*' $O/summaries/${RT}_${tag}_summary.json) $(grep -o '"traffic_bytes_per_env_step": [0-9.]*' $O/summaries/${RT}_${tag}_summary.json)" >&2
done
du -sh $O >&2
