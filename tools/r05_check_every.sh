#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/check_every_sweep.txt; : > $O
run() { timeout 600 python bench.py --workload $1 --precision mixed --no-f64-ab --no-cpu-baseline --pmc off --no-transfer --no-overlap $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', '$2', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in pnp_n4_50k ransac_n4_50k; do
  for o in "" "--opt check_every=3" "--opt check_every=4" "--opt first_check=15" "--opt first_check=19" "--opt first_check=21 --opt check_every=3"; do run $w "$o"; done
done
cat $O
