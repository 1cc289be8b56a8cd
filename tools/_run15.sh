cd $GRAFT_REPO_ROOT
for seed in 1 2 3 4 5 6; do bash tools/ab_run.sh "--batch 24000 --steps 30 --seed $seed" 1; done
for seed in 1 2 3 4 5 6; do bash tools/ab_run.sh "--steps 50 --seed $seed" 1; done
