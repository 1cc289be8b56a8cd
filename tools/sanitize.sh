#!/bin/bash
# The CPU restatement of the device algorithm under AddressSanitizer + UndefinedBehaviorSanitizer (the kernels are
# wave-local, the risk is indexing -- and solver_core.h / lane_core.h / ipm_core.h / recover_core.h are the SAME source the kernels
# instantiate).  Builds tests/hostsim/hostsim.cpp with g++ -O1 -g -fsanitize=address,undefined into a scratch directory, preloads both
# runtimes into python and runs the host-side tests of the device algorithm against that build (CVXPNPL_HOSTSIM_LIB).
#   usage (container, repo root):  tools/sanitize.sh [pytest args]        exit code = pytest's; any sanitizer report fails the run
set -e
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
trap 'rm -rf $tmp' EXIT
g++ -O1 -g -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -o $tmp/libhostsim_san.so $root/tests/hostsim/hostsim.cpp
asan=$(g++ -print-file-name=libasan.so)
ubsan=$(g++ -print-file-name=libubsan.so)
cd $root
# detect_leaks=0: python itself leaks at exit; everything else is fatal
CVXPNPL_HOSTSIM_LIB=$tmp/libhostsim_san.so LD_PRELOAD="$asan $ubsan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 OMP_NUM_THREADS=4 \
  python -m pytest tests/test_device_algorithm_hostsim.py tests/test_dual_retry.py -x -q -m "not gpu" -p no:cacheprovider "$@"
