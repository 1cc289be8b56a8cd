#!/bin/bash
# round 4: the GPU test-suite + smoke + the headline bench lines of the current build   (GPU box; tools/r04_gpu_suite.sh [tag])
cd $GRAFT_REPO_ROOT
tag=${1:-r04_suite}
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -15 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
for w in "" "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k"; do
  timeout 600 python bench.py $w --no-cpu-baseline --pmc off 2> $out/bench_err.log | tee -a $out/bench_lines.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$w', 'default M/s', round(d['value']/1e6,2), 'all_f64', round(d.get('value_all_f64',0)/1e6,2), 'median ms', d.get('median_ms_per_step'), 'transfer M/s', round((d.get('transfer_inclusive') or {}).get('value',0)/1e6,2), d['solver']['status_hist'])"
done
