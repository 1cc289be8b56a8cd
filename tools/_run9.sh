set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02i/t125k -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload pnp_n10_125k --steps 20 --warmup 3 --no-cpu-baseline --no-overlap --pmc off > $GRAFT_REPO_ROOT/gpurun_out/r02i/t125k.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r02i/t125k -name "*kernel_stats.csv" -exec cat {} \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02i/tpnpl -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload pnpl_5p5l_100k --steps 20 --warmup 3 --no-cpu-baseline --no-overlap --pmc off > $GRAFT_REPO_ROOT/gpurun_out/r02i/tpnpl.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r02i/tpnpl -name "*kernel_stats.csv" -exec cat {} \;
