#!/bin/bash
# round 5 (GPU box): the whole GPU suite (no -x) + smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r05/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/pytest_gpu.log
tail -25 gpurun_out/r05/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.log 2>&1; tail -2 gpurun_out/r05/smoke.log
