#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run from the repo root):
#   tools/profile_round.sh r01
# Writes gpurun_out/<tag>/ : kernel-trace stats (csv) of the default bench command, and two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass, MI355X guide).
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- $CMD > $out/trace_bench.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o fetch -- $CMD > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o write -- $CMD > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -o sq -- $CMD > $out/pmc_sq.log 2>&1
find $out -name "*.csv" | head -40
