set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests/test_device_toolkit.py -m gpu -q > gpurun_out/r02f/pytest.log 2>&1; tail -25 gpurun_out/r02f/pytest.log
