#!/bin/bash
# usage (GPU box): tools/ab_run.sh "<bench args>" [repeats]     -- alternates lib A / lib B (tools/ab_build.sh)
cd $GRAFT_REPO_ROOT
args=$1; n=${2:-3}
for i in $(seq $n); do for v in A B; do
  CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_$v.so timeout 300 python bench.py $args --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$args', round(d['roofline']['mean_launch_ms'],4), round(d['value']/1e6,2), round((d.get('overlapped') or {}).get('value',0)/1e6,2), d['solver']['status_hist'])"
done; done
