#!/bin/bash
# A/B of the environment kernels of two trees in ONE GPU session: the round-3 final tree (c58c133,
# exported to _ab/c58c133 with its own libpworld.so and its own bench.py) against the working tree, interleaved,
# REPS repeats each of the driver-style bench command for the workloads the driver saw move between rounds 3 and 4
# (B = 65536 at N = 6: -11.6 %, N = 24 at B = 4096: -4.7 %) and the headline.  Prints env-steps/s per run, then medians.
# Run on the GPU box from the repo root:  tools/ab_env_regression.sh > gpurun_out/ab_env.txt
set -o pipefail
R=$PWD
OLD=${OLD:-$R/_ab/c58c133}
REPS=${REPS:-5}
export PW_BENCH_NO_POLICY=1
O=$R/gpurun_out/ab_env
mkdir -p $O
if [ -n "$AB_CASES" ]; then IFS=';' read -r -a CASES <<< "$AB_CASES"; else
CASES=(
  "b65536_n6|--envs 65536 --agents 6 --chunk 100"
  "c5_n24|--envs 4096 --agents 24 --chunk 500"
  "c2|--envs 4096 --agents 6 --chunk 1000"
)
fi
echo "# side case rep env-steps/s launch_ms kernel"
for rep in $(seq 1 $REPS); do
  for c in "${CASES[@]}"; do
    IFS='|' read -r tag bargs <<< "$c"
    for side in old new; do
      D=$R; [ $side = old ] && D=$OLD
      ( cd $D && python3 bench.py --no-cpu-baseline $bargs --steps 10 --warmup 3 ) > $O/${side}_${tag}_$rep.json 2> $O/${side}_${tag}_$rep.err \
        || { echo "$side $tag $rep FAILED" ; tail -3 $O/${side}_${tag}_$rep.err; continue; }
      python3 - "$O/${side}_${tag}_$rep.json" $side $tag $rep <<'EOF'
import json, sys
l = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith('{')][-1])
print(sys.argv[2], sys.argv[3], sys.argv[4], '%.4g' % l['value'], '%.4f' % l['roofline']['launch_ms'], l['roofline']['kernel'], flush=True)
EOF
    done
  done
done
python3 - $O $REPS <<'EOF'
import glob, json, os, statistics, sys
o = sys.argv[1]
print('# medians (env-steps/s), min .. max, new / old')
for tag in sorted(set(os.path.basename(f).split('_', 1)[1].rsplit('_', 1)[0] for f in glob.glob(os.path.join(o, 'old_*.json')))):
    v = {}
    for side in ('old', 'new'):
        xs = []
        for f in sorted(glob.glob(os.path.join(o, '%s_%s_*.json' % (side, tag)))):
            try:
                xs.append(json.loads([x for x in open(f).read().splitlines() if x.startswith('{')][-1])['value'])
            except Exception:
                pass
        v[side] = xs
    if v['old'] and v['new']:
        mo, mn = statistics.median(v['old']), statistics.median(v['new'])
        print('%-10s old %.4g (%.4g .. %.4g)   new %.4g (%.4g .. %.4g)   new/old %.4f'
              % (tag, mo, min(v['old']), max(v['old']), mn, min(v['new']), max(v['new']), mn / mo))
EOF
# the driver's own view: those figures come from other_configs INSIDE the default run (after the headline and the policy extra)
unset PW_BENCH_NO_POLICY
[ -n "$AB_NO_FULL" ] && exit 0
for rep in 1 2; do
  for side in old new; do
    D=$R; [ $side = old ] && D=$OLD
    ( cd $D && python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 ) > $O/full_${side}_$rep.json 2> $O/full_${side}_$rep.err || echo "full $side $rep FAILED"
    python3 - $O/full_${side}_$rep.json $side $rep <<'EOF'
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith('{')][-1])
    print('full', sys.argv[2], sys.argv[3], 'headline %.4g' % l['value'], ' '.join('%s=%.4g' % (o['config'].split(',')[0].replace(' ', '_')[-14:] + ('_B65536' if '65536' in o['config'] else ''), o['value']) for o in l.get('other_configs', []) if isinstance(o, dict) and 'value' in o),
          'policy %.4g' % ((l.get('policy_in_loop') or {}).get('value') or 0), flush=True)
except Exception as e:
    print('full', sys.argv[2], sys.argv[3], 'unreadable', repr(e))
EOF
  done
done
