#!/bin/bash
# round 5 (GPU box): parity beyond the standing campaigns -- other draws of configurations and problems (seed=7, seed=11), both precision modes, the minimal configurations,
# and the GPU suite twice more (flakiness check of its measured thresholds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python tools/fuzz_parity.py 128 256 seed=7 > gpurun_out/r05/fuzz_parity_seed7.txt 2>&1; tail -1 gpurun_out/r05/fuzz_parity_seed7.txt
timeout 900 python tools/fuzz_parity.py 64 256 f64 seed=7 > gpurun_out/r05/fuzz_parity_seed7_f64.txt 2>&1; tail -1 gpurun_out/r05/fuzz_parity_seed7_f64.txt
timeout 900 python tools/fuzz_parity.py 64 256 minimal seed=11 > gpurun_out/r05/fuzz_parity_seed11_minimal.txt 2>&1; tail -1 gpurun_out/r05/fuzz_parity_seed11_minimal.txt
timeout 900 python tools/fuzz_parity.py 32 256 minimal f64 seed=11 > gpurun_out/r05/fuzz_parity_seed11_minimal_f64.txt 2>&1; tail -1 gpurun_out/r05/fuzz_parity_seed11_minimal_f64.txt

