#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/direct_handoff.txt; : > $O
run() { timeout 600 python bench.py --workload $1 --precision mixed --no-f64-ab --no-cpu-baseline --pmc off --no-transfer --no-overlap $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', '$2', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for w in pnp_n4_50k ransac_n4_50k; do
  run $w ""
  run $w "--opt check_every=3"
  run $w "--opt check_every=4"
  run $w "--opt first_check=19"
  run $w "--opt first_check=21"
  run $w "--opt rescue_from=28"
  run $w "--opt rescue_from=36"
  run $w "--opt rescue_from=40"
done
cat $O
timeout 900 python -m pytest tests/test_gpu_rescue_and_dist.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -5
bash tools/kseq.sh --workload pnp_n4_50k --precision mixed 2>&1 | head -8
