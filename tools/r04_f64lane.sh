#!/bin/bash
# round 4: float64 lane phase -- precision A/B tests, then default vs all-f64 at the sizes of configs 3 / 4   (GPU box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_f64lane; mkdir -p $out
timeout 900 python -m pytest tests/test_precision_modes.py -m gpu -x -q > $out/pytest_precision.log 2>&1; tail -3 $out/pytest_precision.log
for w in "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--workload pnp_n10_125k --batch 1000000 --steps 10 --warmup 2"; do
  for i in 1 2; do
  timeout 300 python bench.py $w --no-cpu-baseline --pmc off --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'default M/s', round(d['value']/1e6,2), 'all_f64 M/s', round(d['value_all_f64']/1e6,2), d['all_f64'].get('status_equal_to_default_frac'), d['all_f64'].get('max_rot_diff_vs_default_rad'), d['solver']['status_hist'])"
  done
done | tee $out/bench.txt
