cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p; timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02p/pytest.log 2>&1; tail -4 gpurun_out/r02p/pytest.log
for seed in 42 1; do bash tools/ab_run.sh "--steps 50 --seed $seed" 1; done
for seed in 42; do bash tools/ab_run.sh "--batch 24000 --steps 30 --seed $seed" 1; done
bash tools/ab_run.sh "--workload pnp_n10_125k --steps 20" 1
bash tools/ab_run.sh "--workload pnpl_5p5l_100k --steps 20" 1
