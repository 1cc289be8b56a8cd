#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --opt variant=1 --steps 12 --warmup 2 --no-cpu-baseline --pmc off --no-f64-ab --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"; }
for rf in 48 64 80 96; do for li in 28 36 44; do run "--batch 50000 --layout 3 --opt first_check=11 --opt lane_iters=$li --opt rescue_from=$rf"; done; done
run "--batch 50000 --layout 3 --opt first_check=15 --opt lane_iters=36 --opt rescue_from=64"
run "--batch 50000 --layout 3 --opt first_check=11 --opt check_every=3 --opt lane_iters=36 --opt rescue_from=64"
run "--batch 125000 --layout 3 --opt first_check=11 --opt lane_iters=36 --opt rescue_from=64"
