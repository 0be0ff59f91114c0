#!/bin/bash
# A/B of the policy-in-the-loop rollout of two TREES in one GPU session: the tree exported to _ab/head (its own library) against the working
# tree, interleaved repeats of tools/policy_profile_run.py per workload; prints us per batched step.   tools/ab_policy_trees.sh "n16 n24 n30" 2
R=$PWD
for rep in $(seq 1 ${2:-2}); do
  for w in ${1:-n16 n24 n30}; do
    for side in old new; do
      D=$R; [ $side = old ] && D=$R/_ab/head
      v=$(cd $D && timeout -k 10 120 python3 tools/policy_profile_run.py --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 | grep -o '"us_per_step": [0-9.]*' | cut -d' ' -f2)
      echo "$side $w $v"
    done
  done
done
