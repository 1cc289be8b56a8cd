#!/bin/bash
# Wavefronts per SIMD of the queue-driven wave-per-problem kernels (resume_wave_kernel, rescue_wave_kernel): builds with
# -DCVXW_OCC_RESUME / -DCVXW_OCC_RESCUE, timed alternately on one box.   usage: tools/occ_variants.sh build | run [repeats]
root=$(cd $(dirname $0)/.. && pwd)
variants="R2S2:-DCVXW_OCC_RESUME=2,-DCVXW_OCC_RESCUE=2 R1S1:-DCVXW_OCC_RESUME=1,-DCVXW_OCC_RESCUE=1 R2S1:-DCVXW_OCC_RESUME=2,-DCVXW_OCC_RESCUE=1 R1S2:-DCVXW_OCC_RESUME=1,-DCVXW_OCC_RESCUE=2"
if [ "$1" = build ]; then
  mkdir -p $root/tools/diag
  for v in $variants; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -enable-ipra=0 $flags -o $root/tools/diag/libcvxpnpl_$name.so $root/cvxpnpl_amd/csrc/cvxpnpl_hip.hip $root/cvxpnpl_amd/csrc/lane_kernel.hip $root/cvxpnpl_amd/csrc/host_recover.cpp &
  done
  wait; ls -la $root/tools/diag
else
  cd $GRAFT_REPO_ROOT
  for w in "--workload pnp_n4_50k" "--workload pnp_n10_125k" "--workload pnpl_5p5l_100k" "--workload pnp_scal --n 6" "--opt variant=1 --batch 50000"; do
  for i in $(seq ${2:-2}); do for v in $variants; do
    name=${v%%:*}
    CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_$name.so timeout 300 python bench.py $w --no-cpu-baseline --pmc off --no-f64-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '$w', 'ms', round(d['roofline']['mean_launch_ms'],4), 'M/s', round(d['value']/1e6,2), '2-stream', round((d.get('overlapped') or {}).get('value',0)/1e6,2), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],3))"
  done; done; done
  for v in $variants; do name=${v%%:*}; echo $name config5; CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_$name.so python tools/config5_sweep.py 2>/dev/null | head -2 | cut -c1-250; done
fi
