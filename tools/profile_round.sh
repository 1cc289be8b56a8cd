#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run from the repo root):
#   tools/profile_round.sh r04
# Writes gpurun_out/<tag>/ ; `python tools/collect_profiles.py <tag>` (container) then condenses it into
# profiles/<tag>/ and profiles/pmc_traffic.json.  Counter passes are separate runs with no tracing
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, MI355X guide).
tag=${1:-r05}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
rm -rf $out; mkdir -p $out
sha256sum $root/cvxpnpl_amd/libcvxpnpl_amd.so | cut -c1-16 > $out/lib_sha16.txt   # the build that is profiled (collect_profiles.py stamps it)
export TMPDIR=/tmp
cd /tmp
prof() { # name, bench args...
  name=$1; shift
  CMD="python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap --pmc off $@"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/trace -o trace -- $CMD > $out/$name.trace.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/pmc_fetch -o fetch -- $CMD --pmc-child > $out/$name.fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/pmc_write -o write -- $CMD --pmc-child > $out/$name.write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/$name/pmc_sq -o sq -- $CMD > $out/$name.sq.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/$name/pmc_sq2 -o sq2 -- $CMD > $out/$name.sq2.log 2>&1
}
prof default
prof quad_16k --batch 16000
prof hybrid_125k --workload pnp_n10_125k
prof hybrid_125k_f64 --workload pnp_n10_125k --opt f32_sweeps_until=0
prof pnpl_100k --workload pnpl_5p5l_100k
prof pnpl_100k_f64 --workload pnpl_5p5l_100k --opt f32_sweeps_until=0
prof ransac --workload ransac_n4_50k
prof large_n --workload pnp_n10000_1k
prof minimal_50k --workload pnp_n4_50k
cd $root
# bench lines of the same build (full default run incl. both CPU baselines and the in-run PMC passes)
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time
python bench.py --batch 16000 --no-cpu-baseline > $out/bench_quad_16k.json 2>/dev/null
python bench.py --workload pnp_n10_125k --no-cpu-baseline > $out/bench_125k.json 2>/dev/null
python bench.py --workload pnpl_5p5l_100k --no-cpu-baseline > $out/bench_pnpl_100k.json 2>/dev/null
python bench.py --workload pnp_n10000_1k --steps 20 > $out/bench_n10000_1k.json 2>/dev/null
python bench.py --workload pnp_n10_125k --batch 1000000 --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_1m.json 2>/dev/null
python bench.py --workload pnp_n4_50k --no-cpu-baseline > $out/bench_n4_50k.json 2>/dev/null
python bench.py --workload ransac_n4_50k --no-cpu-baseline > $out/bench_ransac_n4_50k.json 2>/dev/null
python bench.py --opt variant=1 --batch 50000 --no-cpu-baseline --pmc off > $out/bench_rc_50k.json 2>/dev/null
python bench.py --force-dist --steps 20 --warmup 3 --no-cpu-baseline --pmc off > $out/bench_force_dist_1rank_rccl.json 2>/dev/null
python bench.py --gpus 2 --steps 20 --warmup 3 > $out/bench_2ranks_one_device.json 2>/dev/null
python tools/config5_sweep.py > $out/config5_sweep.jsonl 2>/dev/null
python tools/planar_timing.py > $out/planar_timing.jsonl 2>/dev/null
python tools/iters_hist.py config5 2500 > $out/iters_config5.json 2>/dev/null
python tools/fuzz_parity.py > $out/fuzz_parity.txt 2>&1
python tools/fuzz_hard.py > $out/fuzz_hard.txt 2>&1
ls $out
