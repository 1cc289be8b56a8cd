#!/bin/bash
# round 5 (GPU box): the four-per-wavefront interior-point kernel: its own test, the tests of the path, the workloads that use it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_ipm_quad.py -m gpu -q -x > gpurun_out/r05/ipmq_test.log 2>&1; echo "ipmq rc=$?"; tail -25 gpurun_out/r05/ipmq_test.log
timeout 900 python -m pytest tests/test_gpu_rescue_and_dist.py tests/test_rc_variant.py tests/test_gpu_full_configs.py -m gpu -q > gpurun_out/r05/ipmq_path.log 2>&1; echo "path rc=$?"; tail -25 gpurun_out/r05/ipmq_path.log
for w in pnp_n4_50k ransac_n4_50k; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$w', 'value', round(d['value']/1e6,2), d['dtype'], 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])"
done
