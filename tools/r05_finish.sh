#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/quad_finish_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_prev.so before "$w"
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after "$w"
  done
done
cat $O
timeout 900 python -m pytest tests/test_gpu_full_configs.py tests/test_precision_modes.py tests/test_gpu_rescue_and_dist.py tests/test_ipm_quad.py tests/test_capi_exports.py -m gpu -q 2>&1 | tail -6
bash tools/kseq.sh --workload pnp_n4_50k --precision mixed 2>&1 | head -9
