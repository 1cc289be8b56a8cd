#!/bin/bash
# round 4: split interior-point path -- oracle tests of the path, then the workloads it serves and the judged one   (GPU box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_ipm; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_rescue_and_dist.py tests/test_rc_variant.py tests/test_gpu_full_configs.py -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--opt variant=1 --batch 50000" "--workload pnp_scal --n 6" ""; do
  for i in 1 2; do
  timeout 300 python bench.py $w --no-cpu-baseline --pmc off --no-overlap --no-f64-ab --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$w', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"
  done
done | tee $out/bench.txt
python tools/config5_sweep.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['eps'], d['max_iters'], round(d['poses_per_s']/1e6,2), 'M/s cert', round(d['certified'],5), 'best', d['best_inliers'])" | tee $out/config5.txt
