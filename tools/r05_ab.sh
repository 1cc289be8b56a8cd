#!/bin/bash
# round 5 (GPU box): same-box A/B of the shipped library against the round-4 kernels (tools/diag/libcvxpnpl_r04.so, built from the commit before
# the sweep ordering changed): alternating runs, both precisions of the default workload, the four-point and config-5 workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/ab_r04_vs_r05.txt
: > $O
run() { # lib label, bench args
  CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'value', round(d['value']/1e6,2), d['dtype'], 'ms', round(d['ms_per_step'],4), 'median', round(d.get('median_ms_per_step',0),4), 'mixed', round(d.get('value_mixed',0)/1e6,2), 'two streams', round((d.get('overlapped') or {}).get('value',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'], 'sweeps', round(d['solver']['mean_jacobi_sweeps'],3))" >> $O
}
for i in 1 2 3; do
  run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_r04.so r04 ""
  run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so r05 ""
done
for w in pnp_n4_50k ransac_n4_50k "pnp_n10_10k --batch 16000" pnp_n10_125k; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_r04.so r04 "--workload $w --no-overlap"
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so r05 "--workload $w --no-overlap"
  done
done
cat $O
# one-round geometry (verdict item 5): experiment build, layout 9 = six iterations without certificate code at three wavefronts per SIMD, every
# problem parked and queued, first attempt by the resume kernel behind the launch -- against the shipped schedule, mixed precision (the
# experiment kernel has the single-precision instantiation only)
O2=gpurun_out/r05/one_round.txt
: > $O2
for b in 5000 10000 16000; do for i in 1 2; do
  for v in "shipped $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so 0 -1" "one_round_6 $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_exp.so 9 6" "one_round_5 $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_exp.so 9 5" "one_round_7 $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_exp.so 9 7"; do
    set -- $v
    CVXPNPL_AMD_LIB=$2 timeout 600 python bench.py --batch $b --layout $3 --opt lane_iters=$4 --precision mixed --no-f64-ab --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', 'batch', $b, 'M poses/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O2
  done
done; done
cat $O2
