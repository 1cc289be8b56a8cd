cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_tests
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r04_tests/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_tests/pytest_gpu.log
tail -20 gpurun_out/r04_tests/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_tests/smoke.log 2>&1; tail -2 gpurun_out/r04_tests/smoke.log
bash tools/profile_round.sh r04 > gpurun_out/r04_tests/profile_round.log 2>&1; tail -5 gpurun_out/r04_tests/profile_round.log
