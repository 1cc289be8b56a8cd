cd $GRAFT_REPO_ROOT
for v in A B A B; do CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_$v.so python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4))"; done
