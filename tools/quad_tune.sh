#!/bin/bash
# hand-off iteration of the quad schedule by launch size (GPU box, repo root): tools/quad_tune.sh
for n in ${SIZES:-4000 6000 8000 10000 16000 24000 32000 50000}; do
  line="batch $n:"
  for li in ${LIS:-4 5 6 7 8}; do
    v=$(timeout 120 python bench.py --batch $n --layout 3 --opt lane_iters=$li --no-cpu-baseline --no-overlap --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f'%(d['value']/1e6))")
    line="$line li$li ${v}M"
  done
  w=$(timeout 120 python bench.py --batch $n --layout 2 --no-cpu-baseline --no-overlap --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f'%(d['value']/1e6))")
  l=$(timeout 120 python bench.py --batch $n --layout 1 --no-cpu-baseline --no-overlap --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f'%(d['value']/1e6))")
  echo "$line | wave ${w}M lane ${l}M"
done
