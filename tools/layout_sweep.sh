#!/bin/bash
# poses/s of every layout over batch sizes (GPU box, repo root)
for b in 2000 5000 10000 16000 24000 32000 50000 125000; do
  line="batch $b:"
  for lay in 2 1 3; do
    v=$(timeout 120 python bench.py --no-cpu-baseline --no-overlap --steps 20 --warmup 3 --layout $lay --batch $b 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f'%(d['value']/1e6))")
    line="$line layout$lay ${v}M"
  done
  echo "$line"
done
