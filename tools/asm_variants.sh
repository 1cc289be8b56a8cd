#!/bin/bash
# Build variants of the blocked assembly (ring depth CVXA_STAGES, cache policy CVXA_AUX of the LDS copies) and time them alternately on
# one box.   usage: tools/asm_variants.sh build   (here)   |   tools/asm_variants.sh run [repeats]   (GPU box)
root=$(cd $(dirname $0)/.. && pwd)
variants="S4_A2:-DCVXA_STAGES=4,-DCVXA_AUX=2 S2_A2:-DCVXA_STAGES=2,-DCVXA_AUX=2 S4_A0:-DCVXA_STAGES=4,-DCVXA_AUX=0 S8_A2:-DCVXA_STAGES=8,-DCVXA_AUX=2"
if [ "$1" = build ]; then
  mkdir -p $root/tools/diag
  for v in $variants; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -enable-ipra=0 $flags -o $root/tools/diag/libcvxpnpl_$name.so $root/cvxpnpl_amd/csrc/cvxpnpl_hip.hip $root/cvxpnpl_amd/csrc/lane_kernel.hip $root/cvxpnpl_amd/csrc/host_recover.cpp &
  done
  wait; ls -la $root/tools/diag
else
  cd $GRAFT_REPO_ROOT
  for w in ${3:-"--workload pnp_n10000_1k"}; do w=$(echo $w | tr ',' ' ')
  for i in $(seq ${2:-2}); do for v in $variants; do
    name=${v%%:*}
    CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_$name.so timeout 300 python bench.py $w --no-cpu-baseline --pmc off --no-f64-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', '$w', 'ms', round(r['mean_launch_ms'],4), 'TB/s', round(r['achieved']/1e3,3), 'value', round(d['value']/1e6,2))"
  done; done; done
fi
