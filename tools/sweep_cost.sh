#!/bin/bash
# diagnostics: marginal cost of one Jacobi sweep and of one iteration in the wave layout
# (fixed sweep counts: jacobi_tol=0; no certification until the last iteration)
for sw in 4 8; do for mi in 10 20; do
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --layout 2 --batch 10000 --opt max_iters=$mi --opt first_check=100000 --opt jacobi_tol=0 --opt jacobi_sweeps=$sw > /tmp/o.json
  python - <<PY
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print("wave layout sweeps/eig $sw max_iters $mi: %.3f ms/step  sweeps/problem %.1f iters %.1f"%(d["ms_per_step"], d["solver"]["mean_jacobi_sweeps"], d["solver"]["mean_iters"]))
PY
done; done
