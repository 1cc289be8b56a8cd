set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests/test_large_n.py tests/test_gpu_parity.py -m gpu -q -k "large or blocked or edge" --durations=5 > gpurun_out/r02e/pytest.log 2>&1; tail -15 gpurun_out/r02e/pytest.log
timeout 400 python bench.py --workload pnp_n10000_1k --steps 20 > gpurun_out/r02e/bench_n10000.json 2> gpurun_out/r02e/bench_n10000.err; cat gpurun_out/r02e/bench_n10000.json; tail -3 gpurun_out/r02e/bench_n10000.err
