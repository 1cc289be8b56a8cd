#!/bin/bash
# A/B of library builds: tools/variant_bench.sh lib1.so lib2.so ...  (paths relative to the repo root)
for lib in "$@"; do
  line="$lib:"
  for cfg in "--batch 10000" "--batch 24000" "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k"; do
    v=$(CVXPNPL_AMD_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --no-overlap --steps 30 --warmup 3 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f'%(d['value']/1e6))")
    line="$line [$cfg] ${v}M"
  done
  echo "$line"
done
