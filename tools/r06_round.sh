#!/bin/bash
# round 6, final build (GPU box): the GPU suite, smoke(), the examples, the profiling round (rocprofv3 kernel stats + PMC + bench lines + fuzz
# slices: tools/profile_round.sh r06), the large parity campaigns in both precision modes, the RANSAC frame's kernels, the graph-replay gap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_tests
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r06_tests/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_tests/pytest_gpu.log
tail -4 gpurun_out/r06_tests/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_tests/smoke.log 2>&1; tail -2 gpurun_out/r06_tests/smoke.log
for e in pnp pnl pnpl pnp_batch ransac; do echo "== python examples/$e.py"; python examples/$e.py 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r06_tests/examples.log 2>&1
bash tools/profile_round.sh r06 > gpurun_out/r06_tests/profile_round.log 2>&1; tail -3 gpurun_out/r06_tests/profile_round.log
cp gpurun_out/r06_tests/*.log gpurun_out/r06/ 2>/dev/null
timeout 900 python tools/fuzz_parity.py 128 256 > gpurun_out/r06/fuzz_parity_large.txt 2>&1; tail -1 gpurun_out/r06/fuzz_parity_large.txt
timeout 900 python tools/fuzz_parity.py 64 256 f64 > gpurun_out/r06/fuzz_parity_f64.txt 2>&1; tail -1 gpurun_out/r06/fuzz_parity_f64.txt
bash tools/r06_ransac_kernels.sh > /dev/null 2>&1; tail -1 gpurun_out/r06/ransac/frame_kernels.txt
python tools/r06_ransac_frames.py 60 > gpurun_out/r06/ransac_frames.txt 2>/dev/null; python tools/r06_ransac_frames.py 60 f64 >> gpurun_out/r06/ransac_frames.txt 2>/dev/null; cat gpurun_out/r06/ransac_frames.txt
bash tools/r06_graph_gap.sh > /dev/null 2>&1; cat gpurun_out/r06/graph_gap/graph_gap.txt
python tools/r06_refine_cost.py 2>/dev/null | grep problem > gpurun_out/r06/refine_cost.txt
for s in 1 3 7; do python bench.py --seed $s --no-cpu-baseline --pmc off --no-transfer --no-overlap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('seed $s', round(d['value']/1e6,2), round(d['value_mixed']/1e6,2), d['solver']['max_iters_seen'])"; done > gpurun_out/r06/bench_seeds.txt; cat gpurun_out/r06/bench_seeds.txt
ls gpurun_out/r06 | wc -l
