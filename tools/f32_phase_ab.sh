#!/bin/bash
# Round-3 advisor: the long single-precision first phases (four-point problems: 24 quad iterations, rc variant: 36-48) against float64 sweeps:
# certified fraction, statuses, pose agreement (bench.py's all_f64 block).   GPU box: tools/f32_phase_ab.sh > profiles/r04/f32_phase_ab.txt
cd $GRAFT_REPO_ROOT
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k" "--opt variant=1 --batch 50000" "--opt variant=1 --workload pnp_n4_50k" "--workload pnp_scal --n 5" "--workload pnp_scal --n 6"; do
  timeout 600 python bench.py $w --no-cpu-baseline --pmc off --no-overlap --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); a=d['all_f64']
print('%-45s' % '$w', 'default (f32 sweeps while young): certified %.6f  M/s %.2f | float64 sweeps: certified %.6f  M/s %.2f | statuses equal %.6f  max rotation difference of poses certified in both %.2e rad | mean iterations %.3f / %.3f' % (d['solver']['certified_frac'], d['value']/1e6, a['certified_frac'], a['value']/1e6, a['status_equal_to_default_frac'], a['max_rot_diff_vs_default_rad'], d['solver']['mean_iters'], a['mean_iters']))"
done
