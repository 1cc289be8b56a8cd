#!/bin/bash
# per-kernel average durations of one bench configuration (GPU box, repo root): tools/ktime.sh [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/ktime
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap "$@" > $out/t.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$out/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("   %-40s calls %s avg %.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
