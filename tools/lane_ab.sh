#!/bin/bash
# A/B on one box: the lane-hybrid schedule with the register-budgeted first phase (layout 1: solve_lane2_kernel) against the general
# scalar core (layout 10: solve_lane_kernel), alternating.   usage (GPU box): tools/lane_ab.sh [repeats]
cd $GRAFT_REPO_ROOT
n=${1:-2}
for w in "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--workload pnp_n10_125k --batch 1000000 --steps 10 --warmup 2" "--workload pnp_n10_125k --batch 32000" "--workload pnp_n4_50k"; do
  for i in $(seq $n); do for lay in 1 10; do
    timeout 300 python bench.py $w --layout $lay --no-cpu-baseline --pmc off --no-f64-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('layout $lay', '$w', 'ms', round(d['roofline']['mean_launch_ms'],4), 'M/s', round(d['value']/1e6,2), '2-stream', round((d.get('overlapped') or {}).get('value',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])"
  done; done
done
