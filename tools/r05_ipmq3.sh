#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/ipmq_clock.py 2>&1 | tail -6 | tee gpurun_out/r05/ipmq_clock.jsonl
python tools/ipmq_time.py 2>&1 | grep problems | tee gpurun_out/r05/ipmq_time.jsonl
python -m pytest tests/test_ipm_quad.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -20
for w in pnp_n4_50k ransac_n4_50k; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$w', 'value', round(d['value']/1e6,2), d['dtype'], 'ms', round(d['ms_per_step'],4), 'mixed', round(d.get('value_mixed',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])"
done
