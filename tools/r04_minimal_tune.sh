#!/bin/bash
# round 4: the schedule of four-correspondence problems re-tuned on the kernel that queues its survivors (needs the -DCVXQ_TAIL_EXPERIMENTS build:
# there opts.lane_iters sets the length of their first phase)   GPU box
cd $GRAFT_REPO_ROOT
export CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_tailexp.so
run() { timeout 300 python bench.py $1 $2 --no-cpu-baseline --pmc off --no-overlap --no-f64-ab --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 | $2 |', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"; }
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k"; do
  for li in 12 16 20 24 28 32; do run "$w" "--opt lane_iters=$li"; done
  for rf in 24 28 32 40; do run "$w" "--opt rescue_from=$rf"; done
  for li in 16 20; do for rf in 24 32; do run "$w" "--opt lane_iters=$li --opt rescue_from=$rf"; done; done
  for fc in 5 7 9; do run "$w" "--opt first_check=$fc"; done
done
