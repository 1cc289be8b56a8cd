#!/bin/bash
# round 5 (GPU box): the last, thinly filled round of a quad launch as one-problem wavefronts (CVXPNPL_HYBRID_TAIL = largest such remainder) against the shipped schedule
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/hybrid_tail.txt; : > $O
run() { CVXPNPL_HYBRID_TAIL=$1 timeout 600 python bench.py --batch $2 --no-cpu-baseline --pmc off --no-transfer --no-overlap --seed $3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('tail_max', $1, 'batch', $2, 'seed', $3, 'f64', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for b in 8500 9000 10000 11000 12000 13000 14000; do
  for seed in 42 43; do
    run 0 $b $seed
    run 6000 $b $seed
  done
done
cat $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_precision_modes.py -m gpu -q 2>&1 | tail -2
