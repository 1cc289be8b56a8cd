#!/bin/bash
# round 6 (GPU box): the eigen-gradient step made AT the hand-over (the quad phase parks its failed dual): A = the library before
# (tools/diag/libcvxpnpl_r06a.so), B = the product library; alternating runs.
cd $GRAFT_REPO_ROOT
n=${1:-3}; out=${2:-gpurun_out/r06/handover_refine_ab.txt}
mkdir -p $(dirname $out); : > $out
one() { tag=$1; lib=$2; shift 2
  CVXPNPL_AMD_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['solver']
print('$tag', '$*', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), 'mixed', round((d.get('value_mixed') or 0)/1e6,2), 'iters mean/max', round(s.get('mean_iters'),4), s.get('max_iters_seen'), s['status_hist'])" >> $out
}
for args in "--workload pnp_n10_10k" "--workload pnp_n10_10k --seed 1" "--workload pnp_n10_10k --seed 3" "--workload pnp_n10_10k --seed 7" "--workload pnp_n10_10k --batch 16000" "--workload pnp_n10_10k --batch 5000" "--workload pnp_n10_125k --steps 20" "--workload pnp_n4_50k --steps 10" "--workload ransac_n4_50k --steps 10" "--opt variant=1 --batch 50000 --steps 10"; do
  for i in $(seq $n); do
    one A $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_r06a.so $args
    one B $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so $args
  done
done
cat $out
