#!/bin/bash
# one-launch policy rollout, us/step per kernel form and N (B = 4096): python tools/policy_rollout_probe.py per case
set -e
for N in ${NS:-2 3 6 7 10 12 16 24}; do for F in 1 2 3; do
  if [ $F = 1 ] && [ $N -gt 16 ]; then continue; fi
  python tools/policy_rollout_probe.py --agents $N --form $F --steps 500 2>/dev/null
done; done
