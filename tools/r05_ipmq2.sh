#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/ipmq_time.py > gpurun_out/r05/ipmq_time.jsonl 2>&1; cat gpurun_out/r05/ipmq_time.jsonl
echo "--- split (shipped) pnp_n4_50k"; bash tools/ktime.sh --workload pnp_n4_50k --pmc off --no-transfer --no-f64-ab --precision mixed
echo "--- fused pnp_n4_50k"; CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_fused.so bash tools/ktime.sh --workload pnp_n4_50k --pmc off --no-transfer --no-f64-ab --precision mixed
echo "--- split ransac"; bash tools/ktime.sh --workload ransac_n4_50k --pmc off --no-transfer --no-f64-ab --precision mixed
