cd $GRAFT_REPO_ROOT
export CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_tailexp.so
run() { timeout 300 python bench.py $1 $2 --no-cpu-baseline --pmc off --no-overlap --no-f64-ab --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 | $2 |', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"; }
for w in "--workload pnp_n4_50k" "--workload ransac_n4_50k"; do
  for fc in 17 19 21 23; do run "$w" "--opt first_check=$fc --opt check_every=2"; done
  for fc in 19 21; do run "$w" "--opt first_check=$fc --opt check_every=3"; done
  run "$w" "--opt first_check=17 --opt check_every=2 --opt rescue_from=40"
  run "$w" "--opt first_check=21 --opt check_every=2 --opt rescue_from=40"
  run "$w" "--opt first_check=15 --opt check_every=2 --opt lane_iters=20"
  run "$w" "--opt first_check=13 --opt check_every=2 --opt lane_iters=16"
done
