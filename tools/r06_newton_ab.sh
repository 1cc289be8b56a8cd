#!/bin/bash
# round 6 (GPU box): opts.dual_refine = 1 (the eigen-gradient step: the product before) against 2 (+ the barrier Newton solve), with the phases
# it is made in (+4: resume phase behind a lane phase, +8: fresh wave-per-problem solves); one library, alternating runs.
cd $GRAFT_REPO_ROOT
n=${1:-2}; out=${2:-gpurun_out/r06/newton_ab.txt}
mkdir -p $(dirname $out); : > $out
one() { tag=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['solver']
print('$tag', '$*', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), 'mixed', round((d.get('value_mixed') or 0)/1e6,2), 'iters mean/max', round(s.get('mean_iters'),4), s.get('max_iters_seen'), s['status_hist'])" >> $out
}
run() { # workload args ... -- modes
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  for i in $(seq $n); do for m in "$@"; do one "refine=$m" "${args[@]}" --opt dual_refine=$m; done; done
}
run --workload pnp_n10_125k --steps 20 -- 1 2
run --workload pnpl_5p5l_100k --steps 20 -- 1 2
run --workload pnp_n10_125k --batch 1000000 --steps 5 --warmup 2 -- 1 2
run --workload pnp_n10_125k --batch 30000 --steps 20 -- 1 2
run --workload pnp_n4_50k --steps 10 -- 1 2
run --workload ransac_n4_50k --steps 10 -- 1 2
run --workload pnp_n10_10k -- 1 2
cat $out
