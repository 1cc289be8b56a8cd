#!/bin/bash
# round 5 (GPU box): same-box A/B of the third-order reciprocal root / reciprocal (cvx::rsqrt_, cvx::rcp: one step after the hardware seed instead of two Newton steps) against the build before
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/rsq_c3_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in "" "--batch 16000" "--batch 2000" "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--opt variant=1 --batch 50000" "--workload pnp_n10_125k --batch 1000000 --steps 10 --warmup 2"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_before.so before "$w"
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after "$w"
  done
done
cat $O

