set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hybrid or queue or planar or hip_vs_oracle" > gpurun_out/r02j/pytest.log 2>&1; tail -4 gpurun_out/r02j/pytest.log
timeout 300 python bench.py --workload pnp_n10_125k --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('125k', r['mean_launch_ms'], d['value'], d['overlapped']['value'], r['traffic'], json.dumps(r.get('traffic_detail',{}).get('by_kernel')))"
timeout 200 python bench.py --workload pnpl_5p5l_100k --no-cpu-baseline --pmc off --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pnpl100k', d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
timeout 200 python bench.py --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
timeout 200 python bench.py --batch 2000 --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2k wave', d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
