#!/bin/bash
# build variants of the lane microbenchmark:  tools/microbench/lane_build.sh name "flags" [name "flags" ...]   (container)
cd $(dirname $0)
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -I ../../cvxpnpl_amd/csrc -Rpass-analysis=kernel-resource-usage $2 -o lane_bench_$1.so lane_bench.hip 2>&1 | grep -E "Scratch|VGPRs Spill|error" | tr '\n' ' '; echo " <- $1"
  shift 2
done
