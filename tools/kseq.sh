#!/bin/bash
# the kernels of ONE step of a bench configuration in launch order, with durations (GPU box, repo root): tools/kseq.sh [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/kseq
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/t -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-overlap --pmc off --no-transfer --no-f64-ab "$@" > $out/t.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last complete step: from the last but one launch of the first solver kernel
first = [i for i, n in enumerate(names) if ("solve_quad_kernel" in n and "solve_quad_kernel<4" not in n) or "solve_lane2_kernel" in n or "solve_wave_kernel" in n]
if len(first) >= 2:
    a, b = first[-2], first[-1]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        print("   +%8.1f us  %-44s %8.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:44], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("   step: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
