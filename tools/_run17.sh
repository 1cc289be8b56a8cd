cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hipgraph or planar_batch" > gpurun_out/r02q/pytest.log 2>&1; tail -15 gpurun_out/r02q/pytest.log
