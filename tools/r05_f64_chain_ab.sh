#!/bin/bash
# round 5 (GPU box): same-box A/B of the shortened float64 rotation-parameter chain (halved quantities, third-order step for the second root: cvx::rsqrt_c3) against the build before it; quad phase clocks; float64 parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/f64_chain_ab.txt; : > $O
run() { CVXPNPL_AMD_LIB=$1 timeout 600 python bench.py $3 --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$2', '$3', 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for i in 1 2 3; do
  run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_qroles.so before ""
  run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after ""
done
for w in "--batch 16000" "--batch 2000" "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--opt variant=1 --batch 50000" "--workload pnp_n10_125k --batch 1000000 --steps 10 --warmup 2"; do
  for i in 1 2; do
    run $GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_qroles.so before "$w"
    run $GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_amd.so after "$w"
  done
done
cat $O

P=gpurun_out/r05/quad_phases_chain.jsonl; : > $P
for m in "" f64; do for b in 4 10000; do CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_phases.so python tools/quad_phases.py $b $m >> $P 2>/dev/null; done; done
cat $P | cut -c1-700
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or precision or host_build or full_configs" 2>&1 | tail -3
timeout 600 python tools/fuzz_parity.py 32 256 f64 2>&1 | tail -1
