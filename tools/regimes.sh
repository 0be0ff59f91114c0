#!/bin/bash
# Regime evidence for DESIGN.md / profiles/ (run on the GPU box from the repo root):
#   per-segment shader-cycle stamps (tools/stamp_probe.bin) and throughput (tools/sweep.py) of the rollout kernels
#   in the regimes VERDICT r1 asked to back with numbers: B = 65536 at N = 6, and N = 48 at B = 4096.
set -o pipefail
OUT=${1:-gpurun_out/r2_regimes.txt}
{
echo "== stamps: duo kernel, C2 (B=4096, N=6)";           tools/stamp_probe.bin 4096 6
echo "== stamps: stream kernel, C2";                       PWORLD_NO_DUO=1 tools/stamp_probe.bin 4096 6
echo "== stamps: stream kernel, B=65536, N=6";             PWORLD_NO_DUO=1 tools/stamp_probe.bin 65536 6
echo "== stamps: duo kernel, B=65536, N=6";                tools/stamp_probe.bin 65536 6
echo "== stamps: stream kernel, B=4096, N=48";             PWORLD_NO_DUO=1 tools/stamp_probe.bin 4096 48
echo "== stamps: duo kernel, B=4096, N=48";                tools/stamp_probe.bin 4096 48
echo "== stamps: stream kernel, B=4096, N=24";             PWORLD_NO_DUO=1 tools/stamp_probe.bin 4096 24
echo "== sweep"
python tools/sweep.py --envs 4096,65536 --agents 6 --chunk 100 --steps 1000 --ring 100
python tools/sweep.py --envs 4096 --agents 3,12,24,48 --chunk 100 --steps 1000 --ring 100
PWORLD_NO_DUO=1 python tools/sweep.py --envs 4096,65536 --agents 6,48 --chunk 100 --steps 1000 --ring 100
} > $OUT 2>&1
