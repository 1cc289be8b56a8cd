#!/bin/bash
# instruction-cache behaviour of the solve kernels (run on the GPU box from the repo root)
out=$GRAFT_REPO_ROOT/gpurun_out/icache
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap $@"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $out/ic -o ic -- $CMD > $out/ic.log 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/sq -o sq -- $CMD > $out/sq.log 2>&1
python - <<PY
import csv, glob, collections
for pat in ("$out/ic/**/*counter_collection.csv", "$out/sq/**/*counter_collection.csv"):
    for f in glob.glob(pat, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k, {c: v / n[(k, c)] for c, v in d.items()})
PY
