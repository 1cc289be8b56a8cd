#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>/ and profiles/pmc_traffic.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
WORKLOAD_KEY = {"default": "pnp_n10_10k:10000", "quad_24k": "pnp_n10_10k:24000", "quad_16k": "pnp_n10_10k:16000", "hybrid_125k": "pnp_n10_125k:125000",
                "pnpl_100k": "pnpl_5p5l_100k:100000", "hybrid_125k_f64": "pnp_n10_125k:125000:f64", "pnpl_100k_f64": "pnpl_5p5l_100k:100000:f64", "ransac": "ransac_n4_50k:50000", "large_n": "pnp_n10000_1k:1000", "minimal_50k": "pnp_n4_50k:50000"}


def counters(run):
    """per-kernel means over launches of every counter in gpurun_out/<tag>/<run>/pmc_*"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(src, run, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            if "rocclr" in k:
                continue
            per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if x in r}
        for (k, _), d in per.items():
            for c, v in d.items():
                acc[k][c].append(v)
    out = {}
    for k, d in acc.items():
        out[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in sorted(d.items())}
        out[k]["_kernel"] = meta.get(k, {})
    return out


import hashlib

_sha_file = os.path.join(src, "lib_sha16.txt")  # written on the GPU box by profile_round.sh: the build that was profiled
LIB_SHA = open(_sha_file).read().strip() if os.path.exists(_sha_file) else hashlib.sha256(open(os.path.join(ROOT, "cvxpnpl_amd", "libcvxpnpl_amd.so"), "rb").read()).hexdigest()[:16]
traffic = {}
for run in WORKLOAD_KEY:
    if not os.path.isdir(os.path.join(src, run)):
        continue
    os.makedirs(os.path.join(dst, run), exist_ok=True)
    for f in glob.glob(os.path.join(src, run, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, run, "kernel_stats.csv"))
    c = counters(run)
    json.dump(c, open(os.path.join(dst, run, "pmc_summary.json"), "w"), indent=1)
    # the kernels of one step; the known-size copies of the same passes calibrate the byte counters (bench.py does the same)
    step = {k: v for k, v in c.items() if any(t in k for t in ("solve_", "resume_", "rescue_", "assemble_", "ipm_quad_"))}  # (score_kernel: in pmc_summary.json, not part of a solve step)
    cal = [v for k, v in c.items() if "calibration_copy_kernel<8>" in k]
    nbytes = float(64 << 20)
    ff = cal[0]["FETCH_SIZE"]["mean_per_launch"] * 1024 / nbytes if cal and "FETCH_SIZE" in cal[0] else 1.0
    wf = cal[0]["WRITE_SIZE"]["mean_per_launch"] * 1024 / nbytes if cal and "WRITE_SIZE" in cal[0] else 1.0
    fetch = sum(v.get("FETCH_SIZE", {}).get("mean_per_launch", 0.0) for v in step.values()) * 1024 / ff
    write = sum(v.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) for v in step.values()) * 1024 / wf
    tot = lambda name: sum(v.get(name, {}).get("mean_per_launch", 0.0) for v in step.values())  # noqa: E731
    traffic[WORKLOAD_KEY[run]] = {
        "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
        "valu_insts_per_launch": tot("SQ_INSTS_VALU"), "salu_insts_per_launch": tot("SQ_INSTS_SALU"), "lds_insts_per_launch": tot("SQ_INSTS_LDS"),
        "calibration": {"fetch_reported_over_true": ff, "write_reported_over_true": wf, "on": "64 MiB copy, 8 bytes per lane, same passes"},
        "lib_sha16": LIB_SHA,
        "source": f"profiles/{tag}/{run}/pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB*1024 divided by what the "
                  "same passes report for a known 64 MiB copy, summed over the kernels of one step"}
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "*.jsonl")) + glob.glob(os.path.join(src, "*.txt")) + glob.glob(os.path.join(src, "*.json")):
    shutil.copy(f, dst)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
