#!/bin/bash
# diagnostics: pure per-iteration cost of each layout (no certification until the last iteration)
# usage (on the GPU box, repo root): tools/iter_cost.sh
for lay in 1 2; do for mi in 10 30; do
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --layout $lay --batch 8192 --opt max_iters=$mi --opt first_check=100000 > /tmp/o.json
  python - <<PY
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print("layout $lay max_iters $mi: %.3f ms/step  sweeps/problem %.1f"%(d["ms_per_step"], d["solver"]["mean_jacobi_sweeps"]))
PY
done; done
