#!/bin/bash
# round 5 (GPU box): the survivors-queued first phase of four-correspondence problems with single-precision sweeps at two instead of three wavefronts per SIMD
# (-DCVXPNPL_MINQ_OCC=2: 256 registers, no scratch access in its loops; the shipped kernel: 168 registers, 232 B) -- and the quad kernel's phase clocks in both precision modes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/minq_occ_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--workload pnp_n4_50k --batch 10000"; do
  for i in 1 2 3; do
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so occ3 "$w"
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_minq2.so occ2 "$w"
  done
done
cat $O
P=gpurun_out/r05/quad_phases.jsonl; : > $P
for m in "" f64; do for b in 4 10000; do CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_phases.so python tools/quad_phases.py $b $m >> $P 2>/dev/null; done; done
cat $P
