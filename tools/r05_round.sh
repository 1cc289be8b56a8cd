#!/bin/bash
# round 5, final build (GPU box): the GPU suite, smoke(), the profiling round (rocprofv3 kernel stats + PMC + bench lines + fuzz slices),
# the large parity campaigns in both precision modes, the reference's accuracy and scalability grids, the interior-point kernel on its own
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_tests
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r05_tests/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_tests/pytest_gpu.log
tail -8 gpurun_out/r05_tests/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_tests/smoke.log 2>&1; tail -2 gpurun_out/r05_tests/smoke.log
bash tools/profile_round.sh r05 > gpurun_out/r05_tests/profile_round.log 2>&1; tail -3 gpurun_out/r05_tests/profile_round.log
python tools/ipmq_time.py > gpurun_out/r05/ipm_quad_time.jsonl 2>/dev/null
python tools/ipmq_clock.py > gpurun_out/r05/ipm_quad_clock.jsonl 2>/dev/null
timeout 900 python tools/fuzz_parity.py 128 256 > gpurun_out/r05/fuzz_parity_large.txt 2>&1; tail -1 gpurun_out/r05/fuzz_parity_large.txt
timeout 900 python tools/fuzz_parity.py 64 256 f64 > gpurun_out/r05/fuzz_parity_f64.txt 2>&1; tail -1 gpurun_out/r05/fuzz_parity_f64.txt
mkdir -p gpurun_out/r05/accuracy
timeout 1200 python tools/accuracy_sweep.py --out-dir gpurun_out/r05/accuracy > gpurun_out/r05/accuracy/accuracy.md 2>gpurun_out/r05/accuracy/err.log; tail -3 gpurun_out/r05/accuracy/accuracy.md
timeout 1500 bash tools/scalability_sweep.sh > gpurun_out/r05/scalability.jsonl 2>/dev/null; wc -l gpurun_out/r05/scalability.jsonl
bash tools/kseq.sh --workload pnp_n4_50k --precision mixed > gpurun_out/r05/kseq_n4_50k.txt 2>&1
bash tools/kseq.sh --workload ransac_n4_50k --precision mixed 2>&1 | head -8 > gpurun_out/r05/kseq_ransac.txt
if [ -f tools/diag/libcvxpnpl_phases.so ]; then # shader cycles per phase of the quad kernel (-DCVXQ_PHASES build of the same sources), one wavefront alone and the judged launch, both precision modes
  P=gpurun_out/r05/quad_phases.jsonl; : > $P
  for m in "" f64; do for b in 4 10000; do CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/tools/diag/libcvxpnpl_phases.so python tools/quad_phases.py $b $m >> $P 2>/dev/null; done; done
fi
