#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/rc_f64_quad_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --opt variant=1 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3))" >> $O; }
for w in "--batch 50000" "--batch 10000" "--batch 125000"; do
  run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_prev.so before "$w"
  run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after "$w"
done
cat $O
timeout 900 python -m pytest tests/test_rc_variant.py tests/test_precision_modes.py tests/test_scs_compat.py -m gpu -q 2>&1 | tail -3
