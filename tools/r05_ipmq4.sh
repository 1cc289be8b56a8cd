#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_ipm_quad.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -10
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r05/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/pytest_gpu.log
tail -12 gpurun_out/r05/pytest_gpu.log
for w in pnp_n4_50k ransac_n4_50k "pnp_scal --n 6 --batch 125000" "pnp_scal --n 5 --batch 100000"; do
 for lib in $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_fused.so; do
  CVXPNPL_AMD_LIB=$lib timeout 600 python bench.py --workload $w --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$(basename $lib)', '$w', 'value', round(d['value']/1e6,2), d['dtype'], 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])"
 done
done 2>&1 | tee gpurun_out/r05/ipmq_ab.txt
