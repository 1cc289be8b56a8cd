#!/bin/bash
# round 6 (GPU box): opts.dual_refine 0 / 1, same library, alternating runs.  usage: tools/r06_refine_ab.sh [repeats] [out]
cd $GRAFT_REPO_ROOT
n=${1:-3}; out=${2:-gpurun_out/r06/refine_ab.txt}
mkdir -p $(dirname $out); : > $out
run() { # args...
  for i in $(seq $n); do for v in 0 1; do
    timeout 300 python bench.py "$@" --opt dual_refine=$v --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['solver']
print('refine=$v', '$*', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), 'other', round((d.get('value_mixed') or d.get('value_all_f64') or 0)/1e6,2), 'iters mean/max', s.get('mean_iters'), s.get('max_iters_seen'), s['status_hist'])" >> $out
  done; done
}
run --workload pnp_n10_10k
run --workload pnp_n10_10k --seed 1
run --workload pnp_n10_10k --seed 3
run --workload pnp_n10_125k --steps 20
run --workload pnpl_5p5l_100k --steps 20
run --workload pnp_n10_10k --batch 2000
run --workload pnp_n10_10k --batch 16000
run --workload pnp_n4_50k --steps 10
run --workload ransac_n4_50k --steps 10
cat $out
