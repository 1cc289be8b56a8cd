set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r02d/pytest.log 2>&1; tail -12 gpurun_out/r02d/pytest.log
timeout 300 python tools/iters_hist.py config5 2500 > gpurun_out/r02d/iters_config5.json 2>&1; cat gpurun_out/r02d/iters_config5.json
timeout 300 python tools/planar_timing.py > gpurun_out/r02d/planar_timing.jsonl 2>&1; tail -3 gpurun_out/r02d/planar_timing.jsonl
timeout 200 python bench.py --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
timeout 200 python bench.py --workload pnp_n10_125k --no-cpu-baseline --pmc off --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
