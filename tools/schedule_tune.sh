#!/bin/bash
# certificate-attempt schedules averaged over problem sets (GPU box): tools/schedule_tune.sh "<bench args>" fc:ce ...
args=$1; shift
for cfg in "$@"; do
  fc=${cfg%%:*}; ce=${cfg##*:}
  tot=0; line=""
  for seed in 42 43 44 45 46 47 48 49; do
    v=$(python bench.py $args --seed $seed --opt first_check=$fc --opt check_every=$ce --no-cpu-baseline --no-overlap --steps 40 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f'%(d['value']/1e6))")
    line="$line $v"; tot=$(python -c "print($tot+$v)")
  done
  echo "first_check $fc check_every $ce [$args]: mean $(python -c "print('%.2f'%($tot/8))") |$line"
done
