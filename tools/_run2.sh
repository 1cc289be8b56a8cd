set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log
tail -40 gpurun_out/r02b/pytest.log
for lay in 9 8; do for li in 3 5 10; do
  echo "== layout $lay lane_iters $li"
  timeout 120 python bench.py --layout $lay --opt lane_iters=$li --no-cpu-baseline --no-overlap --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['mean_launch_ms'], d['value'])"
done; done
echo "== normal"
timeout 120 python bench.py --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['mean_launch_ms'], d['value'], d['overlapped']['value'])"
