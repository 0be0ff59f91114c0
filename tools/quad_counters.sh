#!/bin/bash
# SQ counter passes for the bench workload's kernel (run on the GPU box from the repo root); output: gpurun_out/quadpmc/table.txt
set -o pipefail
R=$PWD
O=$R/gpurun_out/quadpmc
rm -rf $O; mkdir -p $O
export PW_BENCH_NO_POLICY=1
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $B > /dev/null 2> $O/p$i.err || echo "pass $i failed: $set" >&2
done
cd $R
python3 tools/pmc_table.py $O/p* --kernel pw_spread > $O/table.txt 2>&1
cat $O/table.txt
find $O -name '*.csv' -size +2M -delete
