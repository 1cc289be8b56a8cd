set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "device_rank" > gpurun_out/r02g/pytest.log 2>&1; tail -25 gpurun_out/r02g/pytest.log
