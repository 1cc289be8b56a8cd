#!/bin/bash
# round 4: opts.rescue_from re-swept on the minimal workloads, after round 3's second tries of the dual   (GPU box)
cd $GRAFT_REPO_ROOT
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k"; do
  for rf in 16 24 32 40 48 64; do
    timeout 300 python bench.py $w --opt rescue_from=$rf --no-cpu-baseline --pmc off --no-overlap --no-f64-ab --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w rescue_from=$rf', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"
  done
done
