#!/bin/bash
# round 5 (GPU box): the lane kernels in a translation unit of their own with the two-Newton refinement (lane_kernel.hip, CVX_REFINE_NEWTON2) against
# the single-unit build with the third-order refinement everywhere (13e1a842) and the build before that change (9c0fd8fb)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/lane_tu_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--workload pnp_n10_125k --batch 1000000 --steps 10 --warmup 2" "" "--workload pnp_n10_125k --batch 32000"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_before.so newton2_one_unit "$w"
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_c3all.so third_order_one_unit "$w"
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so two_units "$w"
  done
done
cat $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
