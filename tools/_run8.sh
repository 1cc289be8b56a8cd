set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02h/pytest.log 2>&1; tail -8 gpurun_out/r02h/pytest.log
timeout 400 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r02h/bench_default.json 2>gpurun_out/r02h/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02h/bench_default.json'))
r=d['roofline']; print('default', d['value'], d['overlapped']['value'], r['mean_launch_ms'], 'traffic', r['traffic'], r.get('valu'))
print(json.dumps(r.get('traffic_detail',{}).get('by_kernel')))
PY
for b in 8192 16384 24000; do timeout 200 python bench.py --batch $b --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"; done
timeout 300 python bench.py --workload pnp_n10_125k --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('125k', r['mean_launch_ms'], d['value'], d['overlapped']['value'], r['traffic'], json.dumps(r.get('traffic_detail',{}).get('by_kernel')))"
timeout 200 python bench.py --workload pnpl_5p5l_100k --no-cpu-baseline --pmc off --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pnpl100k', d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
