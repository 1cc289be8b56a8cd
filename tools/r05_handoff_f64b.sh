#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/handoff_f64_sweep2.txt; : > $O
run() { timeout 600 python bench.py --precision f64 --no-f64-ab --no-cpu-baseline --pmc off --no-transfer --no-overlap $1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])" >> $O; }
for b in 3000 5000 8000 10000 12000 16000 19000; do
 for seed in 42 43 44; do
  for c in "--opt lane_iters=7 --opt first_check=5" "--opt lane_iters=8 --opt first_check=6" "--opt lane_iters=9 --opt first_check=6" "--opt lane_iters=9 --opt first_check=7" "--opt lane_iters=10 --opt first_check=6"; do
    run "--batch $b --seed $seed $c"
  done
 done
done
python - <<'PY'
import re, collections
rows = [l.split() for l in open("gpurun_out/r05/handoff_f64_sweep2.txt")]
acc = collections.defaultdict(list)
for r in rows:
    b = int(r[1]); cfg = (r[5].split("=")[1], r[7].split("=")[1]); v = float(r[9])
    acc[(b, cfg)].append(v)
for b in sorted({k[0] for k in acc}):
    base = sum(acc[(b, ("7", "5"))]) / 3
    print(b, {"/".join(c): round(sum(v) / len(v) / base, 3) for (bb, c), v in sorted(acc.items()) if bb == b}, "base M/s", round(base, 1))
PY
