cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02m/pytest.log 2>&1; tail -4 gpurun_out/r02m/pytest.log
bash tools/ab_run.sh "--steps 50" 3
bash tools/ab_run.sh "--workload pnp_n10_125k --steps 20" 2
bash tools/ab_run.sh "--workload pnpl_5p5l_100k --steps 20" 2
