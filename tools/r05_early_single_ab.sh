#!/bin/bash
# round 5 (GPU box): same-box A/B of the early hand-over of a wavefront's LAST open problem (quad kernel: one problem open after an attempt -> wave-per-problem at once) against the shipped build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/early_single_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in "" "--seed 1" "--seed 2" "--seed 3" "--batch 16000" "--batch 5000" "--batch 3000" "--batch 19000" "--workload pnpl_5p5l_100k --batch 10000" "--opt variant=1 --batch 10000" "--sigma 5 --batch 10000"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so before "$w"
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_early1.so after "$w"
  done
done
cat $O
CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_early1.so timeout 900 python -m pytest tests -m gpu -x -q -k "parity or precision or host_build or hybrid or layout" 2>&1 | tail -3
