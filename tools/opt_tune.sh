#!/bin/bash
# solver options averaged over 8 problem sets (GPU box): tools/opt_tune.sh "<bench args>" "name=v name=v" "name=v" ...
args=$1; shift
for cfg in "$@"; do
  o=""; for kv in $cfg; do o="$o --opt $kv"; done
  tot=0; line=""
  for seed in 42 43 44 45 46 47 48 49; do
    v=$(python bench.py $args --seed $seed $o --no-cpu-baseline --no-overlap --steps 40 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f'%(d['value']/1e6))")
    line="$line $v"; tot=$(python -c "print($tot+$v)")
  done
  echo "[$args] $cfg: mean $(python -c "print('%.2f'%($tot/8))") |$line"
done
