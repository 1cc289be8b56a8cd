#!/bin/bash
# round 6 (GPU box): idle gap between consecutive steps, plain launches vs hipGraph replays (kernel trace of tools/r06_graph_gap.py)
out=$GRAFT_REPO_ROOT/gpurun_out/r06/graph_gap
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
python $GRAFT_REPO_ROOT/tools/r06_graph_gap.py 2>/dev/null | grep GRAPH | sed 's/^/untraced: /' > $out/graph_gap.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/t -o t -- python $GRAFT_REPO_ROOT/tools/r06_graph_gap.py > $out/run.log 2>&1
grep GRAPH $out/run.log | sed 's/^/traced:   /' >> $out/graph_gap.txt
python - >> $out/graph_gap.txt <<PY
import csv, glob
rows = []
for f in glob.glob("$out/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "solve_quad_kernel" in r["Kernel_Name"]]
# steps: [first kernel index .. next first kernel index)
steps = []
for a, b in zip(first, first[1:] + [len(rows)]):
    ks = [r for r in rows[a:b] if "solve_quad" in r["Kernel_Name"] or "wave_kernel" in r["Kernel_Name"]]
    steps.append((int(ks[0]["Start_Timestamp"]), int(ks[-1]["End_Timestamp"]), len(ks)))
gaps = [(steps[i + 1][0] - steps[i][1]) / 1e3 for i in range(len(steps) - 1)]
durs = [(s[1] - s[0]) / 1e3 for s in steps]
# the script runs 5 + 40 plain steps, 1 capture-free warm-up is not traced as kernels, then 3 + 40 replays
n = len(steps)
plain_g, rep_g = gaps[5:44], gaps[-39:]
import statistics as st
print("steps traced:", n, " kernels per step:", sorted(set(s[2] for s in steps)))
print("plain launches: median gap between steps %.1f us (p90 %.1f), median step (first start -> last end) %.1f us" % (st.median(plain_g), sorted(plain_g)[int(0.9 * len(plain_g))], st.median(durs[5:45])))
print("graph replays:  median gap between steps %.1f us (p90 %.1f), median step %.1f us" % (st.median(rep_g), sorted(rep_g)[int(0.9 * len(rep_g))], st.median(durs[-40:])))
PY
cat $out/graph_gap.txt
rm -rf $out/t
