set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -30 gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err; echo "bench rc=$?"
cat gpurun_out/r02a/bench_default.json
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r02a/bench_2rank.json 2> gpurun_out/r02a/bench_2rank.err; echo "bench2 rc=$?"
cat gpurun_out/r02a/bench_2rank.json; tail -5 gpurun_out/r02a/bench_2rank.err
timeout 300 python tools/iters_hist.py config5 2500 > gpurun_out/r02a/iters_config5.json 2>&1
timeout 300 python tools/iters_hist.py 10000 2500 > gpurun_out/r02a/iters_10k.json 2>&1
cat gpurun_out/r02a/iters_config5.json gpurun_out/r02a/iters_10k.json
