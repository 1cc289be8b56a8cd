#!/bin/bash
# SQ counters per kernel of one bench configuration (GPU box, repo root): tools/pmc_sq.sh [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap $@"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/a -o a -- $CMD > $out/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $out/b -o b -- $CMD > $out/b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- $CMD > $out/t.log 2>&1
python - <<PY
import csv, glob, collections
for pat in ("$out/a/**/*counter_collection.csv", "$out/b/**/*counter_collection.csv"):
    for f in glob.glob(pat, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:28]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if "rocclr" in k: continue
            print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
for f in glob.glob("$out/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:40], r["Calls"], r["AverageNs"])
PY
