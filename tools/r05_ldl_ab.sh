#!/bin/bash
# round 5 (GPU box): same-box A/B of the certificate's LDL^T of the quad kernel (rows in lanes + row_newbcast) against the build before it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/quad_ldl_rows_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3))" >> $O; }
for i in 1 2 3; do
  run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_prev.so before ""
  run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after ""
done
for w in "--batch 16000" "--batch 5000" "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--opt variant=1 --batch 50000"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_prev.so before "$w"
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after "$w"
  done
done
cat $O
cd tools/microbench && ./lat_probe > $GRAFT_REPO_ROOT/gpurun_out/r05/lat_probe.txt; cat $GRAFT_REPO_ROOT/gpurun_out/r05/lat_probe.txt
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_precision_modes.py tests/test_ipm_quad.py -m gpu -q 2>&1 | tail -3
