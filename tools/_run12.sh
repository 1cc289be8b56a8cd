cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -q -x > gpurun_out/r02k/pytest.log 2>&1; tail -3 gpurun_out/r02k/pytest.log
for i in 1 2; do
timeout 300 python bench.py --workload pnp_n10_125k --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('125k', r['mean_launch_ms'], d['value'], d['overlapped']['value'])"
timeout 200 python bench.py --workload pnpl_5p5l_100k --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pnpl100k', d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
done
