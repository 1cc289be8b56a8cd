#!/bin/bash
# quad vs lane-hybrid (register-budgeted first phase) by launch size, alternating on one box: the data behind the AUTO crossover
cd $GRAFT_REPO_ROOT
for b in 6000 8000 10000 12000 14000 16000 20000 24000 32000; do for rep in 1 2; do for lay in 3 1; do
  python bench.py --batch $b --layout $lay --steps 40 --warmup 5 --no-cpu-baseline --pmc off --no-f64-ab --seed $((42+rep)) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b layout $lay seed $((42+rep))', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), '2-stream', round((d.get('overlapped') or {}).get('value',0)/1e6,2), 'max_it', d['solver']['max_iters_seen'])"
done; done; done
