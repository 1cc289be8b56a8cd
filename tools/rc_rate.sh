#!/bin/bash
# The 16-equality variant (rc): launch time by layout and first-attempt schedule.  GPU box: tools/rc_rate.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --opt variant=1 --steps 20 --warmup 3 --no-cpu-baseline --pmc off --no-f64-ab --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"; }
run "--layout 2"
run "--layout 3"
for fc in 7 9 11 13 15; do run "--layout 3 --opt first_check=$fc"; done
for li in 16 20 36; do run "--layout 3 --opt first_check=11 --opt lane_iters=$li"; done
run "--layout 2 --opt first_check=11"
run "--batch 50000 --layout 3 --opt first_check=11"
run "--batch 50000 --layout 2 --opt first_check=11"
