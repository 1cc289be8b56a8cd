#!/bin/bash
# A/B harness: build cvxpnpl_amd/libcvxpnpl_A.so from a git revision (default HEAD) and libcvxpnpl_B.so from the working tree;
# tools/ab_run.sh then alternates bench runs of the two on the GPU box (box-to-box variation is several per cent: only
# same-box comparisons mean anything).   usage: tools/ab_build.sh [rev]
rev=${1:-HEAD}
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
mkdir -p $root/tools/diag
mkdir -p $tmp/cvxpnpl_amd/csrc $tmp/include
for f in $(git -C $root ls-tree --name-only $rev cvxpnpl_amd/csrc/); do git -C $root show $rev:$f > $tmp/$f; done
git -C $root show $rev:include/cvxpnpl_amd.h > $tmp/include/cvxpnpl_amd.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -enable-ipra=0 -o $root/tools/diag/libcvxpnpl_A.so $tmp/cvxpnpl_amd/csrc/cvxpnpl_hip.hip $tmp/cvxpnpl_amd/csrc/lane_kernel.hip $tmp/cvxpnpl_amd/csrc/host_recover.cpp &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -enable-ipra=0 -o $root/tools/diag/libcvxpnpl_B.so $root/cvxpnpl_amd/csrc/cvxpnpl_hip.hip $root/cvxpnpl_amd/csrc/lane_kernel.hip $root/cvxpnpl_amd/csrc/host_recover.cpp &
wait
rm -rf $tmp
ls -la $root/tools/diag/libcvxpnpl_A.so $root/tools/diag/libcvxpnpl_B.so
