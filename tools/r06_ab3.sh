#!/bin/bash
# round 6 (GPU box): three-way same-box A/B of the eigen-gradient step of the dual: N = library built without it
# (tools/diag/libcvxpnpl_norefine.so, -DCVX_DUAL_REFINE_COMPILED=0), R0 = product library with opts.dual_refine = 0, R1 = product default.
cd $GRAFT_REPO_ROOT
n=${1:-2}; out=${2:-gpurun_out/r06/refine_ab3.txt}
mkdir -p $(dirname $out); : > $out
one() { # tag lib refine args...
  tag=$1; lib=$2; rf=$3; shift 3
  CVXPNPL_AMD_LIB=$lib timeout 300 python bench.py "$@" --opt dual_refine=$rf --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['solver']
print('$tag', '$*', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), 'mixed', round((d.get('value_mixed') or 0)/1e6,2), 'iters mean/max', round(s.get('mean_iters'),4), s.get('max_iters_seen'), s['status_hist'])" >> $out
}
run() {
  for i in $(seq $n); do
    one N  $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_norefine.so 0 "$@"
    one R0 $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so 0 "$@"
    one R1 $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so 1 "$@"
  done
}
run --workload pnp_n10_10k
run --workload pnp_n10_10k --seed 1
run --workload pnp_n10_10k --seed 3
run --workload pnp_n10_125k --steps 20
run --workload pnpl_5p5l_100k --steps 20
run --workload pnp_n10_10k --batch 2000
run --workload pnp_n10_10k --batch 16000
run --workload pnp_n4_50k --steps 10
run --workload ransac_n4_50k --steps 10
run --workload pnp_n10_125k --batch 1000000 --steps 5
cat $out
