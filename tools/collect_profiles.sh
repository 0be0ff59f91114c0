#!/bin/bash
# Round-2 profile collection (run on the GPU box from the repo root): kernel trace + HBM counters + SQ counters of
# the bench workload, and the same counters for the two store-bound regimes.  PMC passes are separate runs with
# no trace flags (MI355X guide; gpurun refuses the combination).  Output: gpurun_out/prof_r2/*, summarised by
# tools/summarize_prof.py and tools/pmc_table.py into profiles/r2_*.
set -o pipefail
R=$PWD
O=$R/gpurun_out/prof_r2
mkdir -p $O
export PW_BENCH_NO_POLICY=1
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 20 --warmup 5 > $O/bench_line.json 2> $O/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $B --steps 4 --warmup 1 > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- $B --steps 4 --warmup 1 > /dev/null 2> $O/sq1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/sq2 -- $B --steps 4 --warmup 1 > /dev/null 2> $O/sq2.err
echo "bench passes done" >&2
# regimes: B = 65536 at N = 6, and N = 48 at B = 4096 (tools/sweep.py: 100-step launches)
for reg in "65536 6" "4096 48"; do
  set -- $reg
  S="python3 $R/tools/sweep.py --envs $1 --agents $2 --chunk 100 --steps 300 --ring 100"
  tag=B$1_N$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- $S > $O/${tag}_sweep.txt 2> /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/${tag}_$c -- $S > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/${tag}_sq1 -- $S > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/${tag}_sq2 -- $S > /dev/null 2>&1
  echo "regime $tag done" >&2
done
cd $R
python3 tools/pmc_table.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq1 $O/sq2 --kernel pw_spread > $O/bench_counters.txt
for tag in B65536_N6 B4096_N48; do
  { cat $O/${tag}_sweep.txt | grep -v amdgpu; python3 tools/pmc_table.py $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE $O/${tag}_sq1 $O/${tag}_sq2 --kernel pw_spread;
    f=$(find $O/${tag}_trace -name '*_kernel_stats.csv' | head -1); head -4 $f; } > $O/${tag}_counters.txt
done
find $O -name '*_kernel_stats.csv' | head -3 >&2
# keep what goes back small: drop the per-dispatch traces of the regime runs
find $O -name '*_kernel_trace.csv' -path '*_N*' -delete
du -sh $O >&2
