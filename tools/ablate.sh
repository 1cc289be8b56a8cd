#!/bin/bash
# timing ablations of the wave kernel (container: build; GPU box: run).  usage: tools/ablate.sh build | run
src=cvxpnpl_amd/csrc
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -DCVXW_STOP_AFTER_ASSEMBLY -o tools/microbench/libcvxpnpl_amd_asm.so $src/cvxpnpl_hip.hip $src/lane_kernel.hip $src/host_recover.cpp
else
  for lib in tools/microbench/libcvxpnpl_amd_asm.so cvxpnpl_amd/libcvxpnpl_amd.so; do
    CVXPNPL_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-overlap --layout 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '%.3f ms'%d['ms_per_step'])"
  done
fi
