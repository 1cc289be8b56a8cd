cd $GRAFT_REPO_ROOT
for b in 10000 12000 16000 19000; do for rep in 1 2; do for li in 0 5 6; do
  CVXQ_LATE_ITERS=$li python bench.py --batch $b --no-cpu-baseline --pmc off --no-f64-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b late hand-off $li', 'ms', round(d['roofline']['mean_launch_ms'],4), 'M/s', round(d['value']/1e6,2), '2-stream', round((d.get('overlapped') or {}).get('value',0)/1e6,2), 'iters', round(d['solver']['mean_iters'],3), d['solver']['max_iters_seen'])"
done; done; done
