set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "uncertified" > gpurun_out/r02c/pytest.log 2>&1; tail -5 gpurun_out/r02c/pytest.log
CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_timeline.so timeout 300 python tools/timeline.py 10000 > gpurun_out/r02c/timeline_10k.json 2> gpurun_out/r02c/timeline.err; cat gpurun_out/r02c/timeline_10k.json; tail -3 gpurun_out/r02c/timeline.err
CVXPNPL_AMD_LIB=$GRAFT_REPO_ROOT/cvxpnpl_amd/libcvxpnpl_timeline.so timeout 300 python tools/timeline.py 8192 > gpurun_out/r02c/timeline_8k.json 2>> gpurun_out/r02c/timeline.err; cat gpurun_out/r02c/timeline_8k.json
