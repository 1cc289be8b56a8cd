#!/bin/bash
# round 6 (GPU box): quad against lane-hybrid by launch size on the round's kernels, both precision modes -- is AUTO's crossover still 20 000?
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06/crossover.txt; mkdir -p gpurun_out/r06; : > $out
for b in 16000 20000 24000 28000 32000 40000; do for rep in 1 2; do for lay in 3 1; do
  python bench.py --batch $b --layout $lay --steps 40 --warmup 5 --no-cpu-baseline --pmc off --no-transfer --no-overlap --seed $((42+rep)) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b layout $lay seed $((42+rep))', 'ms', round(d['ms_per_step'],4), 'f64 M/s', round(d['value']/1e6,2), 'mixed', round((d.get('value_mixed') or 0)/1e6,2), 'max_it', d['solver']['max_iters_seen'])" >> $out
done; done; done
cat $out
