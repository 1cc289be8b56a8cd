#!/bin/bash
# Experiment build of the library (container or GPU box): everything the product build leaves out -- layouts 9-13 (quad iterations only /
# the lane schedule on the general scalar core / the tail experiments of round 4) and the general solve_lane_kernel -- behind
# -DCVXPNPL_EXPERIMENTS, into tools/diag/libcvxpnpl_exp.so.  Use: CVXPNPL_AMD_LIB=$PWD/tools/diag/libcvxpnpl_exp.so python bench.py --layout 9 ...
# Extra -D switches (e.g. -DCVXW_SPLIT_IPM) may be appended.
cd "$(dirname "$0")/../.."
mkdir -p tools/diag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -enable-ipra=0 -DCVXPNPL_EXPERIMENTS "$@" \
      -o tools/diag/libcvxpnpl_exp.so cvxpnpl_amd/csrc/cvxpnpl_hip.hip cvxpnpl_amd/csrc/lane_kernel.hip cvxpnpl_amd/csrc/host_recover.cpp
