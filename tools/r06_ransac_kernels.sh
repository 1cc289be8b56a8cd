#!/bin/bash
# round 6 (GPU box): kernels of a RANSAC frame (rocprofv3 --kernel-trace --stats of tools/r06_ransac_frames.py) -> gpurun_out/r06/ransac/
out=$GRAFT_REPO_ROOT/gpurun_out/r06/ransac
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- python $GRAFT_REPO_ROOT/tools/r06_ransac_frames.py 20 > $out/run.log 2>&1
grep FRAMES $out/run.log
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$out/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = [i for i, n in enumerate(names) if "sample_sets_kernel" in n]
a, b = first[-2], first[-1]
t0 = int(rows[a]["Start_Timestamp"])
with open("$out/frame_kernels.txt", "w") as fo:
    for r in rows[a:b]:
        line = "   +%8.1f us  %-60s %8.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print(line); fo.write(line + "\n")
    tail = "   frame: %d kernels, %.1f us from the first kernel's start to the next frame's" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
    print(tail); fo.write(tail + "\n")
PY
for f in $(find $out/t -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
rm -rf $out/t
