#!/bin/bash
# round 5 (GPU box): hand-off point of the quad schedule re-swept with every sweep in float64 (the headline mode since this round; 7 was tuned with single-precision sweeps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/handoff_f64_sweep.txt; : > $O
run() { timeout 600 python bench.py --precision f64 --no-f64-ab --no-cpu-baseline --pmc off --no-transfer --no-overlap $1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), 'median', round(d.get('median_ms_per_step',0),4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for rep in 1 2; do
for h in 5 6 7 8 9; do run "--opt lane_iters=$h"; done
run "--opt lane_iters=7 --opt first_check=6"
run "--opt lane_iters=8 --opt first_check=6"
run "--opt lane_iters=6 --opt first_check=4"
run "--batch 16000 --opt lane_iters=7"
run "--batch 16000 --opt lane_iters=9"
run "--batch 16000 --opt lane_iters=6"
done
cat $O
