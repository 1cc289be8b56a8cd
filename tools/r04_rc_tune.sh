cd $GRAFT_REPO_ROOT
export CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_tailexp.so
run() { timeout 300 python bench.py $1 $2 --no-cpu-baseline --pmc off --no-overlap --no-f64-ab --no-transfer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 | $2 |', 'M/s', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],4), d['solver']['status_hist'], 'iters', round(d['solver']['mean_iters'],2), d['solver']['max_iters_seen'])"; }
for w in "--opt variant=1 --batch 50000" "--opt variant=1 --batch 10000"; do
  for lay in 0 11; do for fc in 11 15 19 23; do run "$w" "--layout $lay --opt first_check=$fc"; done; done
done
