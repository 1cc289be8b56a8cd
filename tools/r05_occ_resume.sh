#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/occ_resume_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'])" >> $O; }
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--opt variant=1 --batch 50000" "--workload pnp_scal --n 5 --batch 100000"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so occ2 "$w"
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_occ3.so occ3 "$w"
  done
done
cat $O
