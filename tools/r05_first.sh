#!/bin/bash
# round 5 (GPU box): GPU suite, smoke, the default bench line, the Jacobi ordering microbenchmark, then the A/B against the round-4 kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r05/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/pytest_gpu.log
tail -15 gpurun_out/r05/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.log 2>&1; tail -2 gpurun_out/r05/smoke.log
timeout 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r05/bench_default.json
bash tools/microbench/eig16x_run.sh > gpurun_out/r05/eig16x.log 2>&1
bash tools/r05_ab.sh > gpurun_out/r05/ab.log 2>&1
cat gpurun_out/r05/ab_r04_vs_r05.txt gpurun_out/r05/one_round.txt
