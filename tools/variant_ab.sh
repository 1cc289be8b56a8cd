#!/bin/bash
# alternate bench runs of library builds on one box:  tools/variant_ab.sh "<bench args>" repeats lib1.so lib2.so ...   (GPU box; "-" = the in-tree library)
cd $GRAFT_REPO_ROOT
args=$1; n=$2; shift 2
for i in $(seq $n); do for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset CVXPNPL_AMD_LIB; else export CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/$lib; fi
  timeout 300 python bench.py $args --no-cpu-baseline --pmc off --no-overlap --no-f64-ab --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s' % '$lib', '$args', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"
done; done
