cd $GRAFT_REPO_ROOT
for w in "" "--batch 2000" "--batch 32000" "--workload pnp_n10_125k" "--workload pnp_scal --n 8" "--workload pnp_n10000_1k"; do for i in 1 2; do for ds in 0 0.015; do
  timeout 300 python bench.py $w --opt dual_shift=$ds --no-cpu-baseline --pmc off --no-f64-ab --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ds $ds', '$w', 'ms', round(d['ms_per_step'],4), 'M/s', round(d['value']/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])"
done; done; done
