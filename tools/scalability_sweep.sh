#!/bin/bash
# The reference's scalability experiment (benchmarks/scalability/pnp.py:26-40: N = 4...10 and 200...10 000 points per problem) for this
# solver: ms per pose, poses/s, and -- where the blocked assembly is the dominant kernel -- its fraction of the HBM peak; plus the
# crossover between the in-kernel assembly and the blocked one (cvxpnpl_amd.api.LARGE_N).  GPU box:  tools/scalability_sweep.sh > scalability.jsonl
cd $GRAFT_REPO_ROOT
row() { python bench.py --workload pnp_scal --n $1 --steps 20 --warmup 3 --no-cpu-baseline --no-overlap --no-f64-ab --pmc off $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; c = d['config']
print(json.dumps({'n': c['n_points'], 'problems_per_step': c['problems_per_gpu_per_step'], 'path': '$3', 'ms_per_step': round(d['ms_per_step'], 5),
                  'us_per_pose': round(1e3 * d['ms_per_step'] / c['problems_per_gpu_per_step'], 5), 'poses_per_s': round(d['value']), 'points_per_s': round(d['value'] * c['n_points']),
                  'dominant_kernel_ms': round(r['mean_launch_ms'], 5), 'hbm_GBps': round(r['achieved'], 1), 'hbm_frac': round(r['frac'], 4),
                  'kernel': r['kernel'][:60], 'certified_frac': d['solver']['certified_frac'], 'mean_iters': round(d['solver']['mean_iters'], 2)}))"; }
for n in 4 5 6 7 8 9 10; do row $n "" auto; done
for n in 200 715 1231 1747 2263 2778 3294 3810 4326 4842 5357 5873 6389 6905 7421 7936 8452 8968 9484 10000; do row $n "" auto; done   # np.linspace(200, 10000, 20, dtype=int)
for n in 32 64 96 128 160 192 256 384; do row $n "--blocked 0" in_kernel; row $n "--blocked 1" blocked; done
