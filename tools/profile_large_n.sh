#!/bin/bash
# The large-N part of tools/profile_round.sh on its own (after a change to assemble_kernel.h): trace + PMC passes of pnp_n10000_1k,
# its bench line, the scalability sweep and the routing crossover.   GPU box:  tools/profile_large_n.sh r03
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
name=large_n
CMD="python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap --pmc off --workload pnp_n10000_1k"
rm -rf $out/$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/trace -o trace -- $CMD > $out/$name.trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/pmc_fetch -o fetch -- $CMD --pmc-child > $out/$name.fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/pmc_write -o write -- $CMD --pmc-child > $out/$name.write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/$name/pmc_sq -o sq -- $CMD > $out/$name.sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/$name/pmc_sq2 -o sq2 -- $CMD > $out/$name.sq2.log 2>&1
cd $root
python bench.py --workload pnp_n10000_1k --steps 20 > $out/bench_n10000_1k.json 2>/dev/null
python tools/large_n_crossover.py > $out/large_n_crossover.jsonl 2>/dev/null
tools/scalability_sweep.sh > $out/scalability.jsonl 2>/dev/null
