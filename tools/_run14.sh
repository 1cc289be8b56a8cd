cd $GRAFT_REPO_ROOT
bash tools/ab_run.sh "--steps 50" 3
bash tools/ab_run.sh "--batch 24000 --steps 30" 2
bash tools/ab_run.sh "--workload pnp_n10_125k --steps 20" 2
bash tools/ab_run.sh "--batch 2000 --steps 50" 2
for v in A B; do CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_$v.so timeout 300 python bench.py --no-cpu-baseline --no-overlap --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v traffic', r['traffic'], r['traffic_detail']['fetch_bytes'], r['traffic_detail']['write_bytes'])"; done
for v in A B; do CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_$v.so timeout 300 python bench.py --workload pnp_n10_125k --no-cpu-baseline --no-overlap --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v traffic125k', r['traffic'], r['traffic_detail']['fetch_bytes'], r['traffic_detail']['write_bytes'])"; done
mkdir -p gpurun_out/r02n; CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_B.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02n/pytest.log 2>&1; tail -3 gpurun_out/r02n/pytest.log
