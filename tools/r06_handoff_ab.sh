#!/bin/bash
# round 6 (GPU box): with the eigen-gradient step in the wave-per-problem phase -- when should the quad phase hand over, and should the resumed
# problem make its attempt at once?  P = product library, E = -DCVXW_RESUME_EARLY_CHECK=1 build; lane_iters = the hand-over iteration.
cd $GRAFT_REPO_ROOT
n=${1:-2}; out=${2:-gpurun_out/r06/handoff_ab.txt}
mkdir -p $(dirname $out); : > $out
one() { tag=$1; lib=$2; shift 2
  CVXPNPL_AMD_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['solver']
print('$tag', '$*', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), 'mixed', round((d.get('value_mixed') or 0)/1e6,2), 'iters mean/max', round(s.get('mean_iters'),4), s.get('max_iters_seen'), s['status_hist'])" >> $out
}
for i in $(seq $n); do for args in "--workload pnp_n10_10k" "--workload pnp_n10_10k --seed 1" "--workload pnp_n10_10k --seed 3" "--workload pnp_n10_10k --batch 16000" "--workload pnp_n10_10k --batch 5000"; do
  for li in 0 5 6; do
    one "P li=$li" $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so $args --opt lane_iters=$li
    one "E li=$li" $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_early.so $args --opt lane_iters=$li
  done
done; done
cat $out
