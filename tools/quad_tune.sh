#!/bin/bash
for b in 16000 24000 32000; do
  line="batch $b:"
  for li in 4 5 6 8; do
    v=$(timeout 120 python bench.py --no-cpu-baseline --no-overlap --steps 20 --warmup 3 --layout 3 --batch $b --opt lane_iters=$li 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f'%(d['value']/1e6))")
    line="$line li$li ${v}M"
  done
  echo "$line"
done
