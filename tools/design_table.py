#!/usr/bin/env python3
"""(container) The workload table of DESIGN.md section 6 from the bench lines of a round: python tools/design_table.py r06"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
rows = (("`pnp_n10_10k` (judged)", "bench_default"), ("16 000 PnP", "bench_quad_16k"), ("`pnp_n10_125k` (config 4's shard)", "bench_125k"),
        ("`pnpl_5p5l_100k` (config 3)", "bench_pnpl_100k"), ("1 M PnP, one launch", "bench_1m"), ("`pnp_n4_50k`", "bench_n4_50k"),
        ("`ransac_n4_50k` (config 5)", "bench_ransac_n4_50k"), ("`rc` variant, 50 k", "bench_rc_50k"), ("N = 10⁴ points × 1 000", "bench_n10000_1k"))
for name, f in rows:
    d = json.load(open(f"profiles/{tag}/{f}.json"))
    r = d["roofline"]
    b = d["config"]["problems_per_gpu_per_step"]
    tr = (r.get("traffic") or 0) / (r["algorithmic_bytes_per_problem"] * b)
    fl = (r.get("flops") or {}).get("frac")
    va = (r.get("valu") or {}).get("issue_frac")
    print(f"| {name} | **{d['value'] / 1e6:.1f} M** | {(d.get('value_mixed') or 0) / 1e6:.1f} M | {d['ms_per_step']:.4f} | {tr:.2f}× | "
          f"{('%.3f' % fl) if fl else '—'} | {('%.2f' % va) if va else '—'} | {d['solver']['mean_iters']:.2f} / {d['solver']['max_iters_seen']} |")
d = json.load(open(f"profiles/{tag}/bench_default.json"))
print("judged: frac", d["roofline"]["frac"], "achieved GB/s", d["roofline"]["achieved"], "traffic", d["roofline"]["traffic"], "latency_b1", d["latency_b1_us"]["median"], d["latency_b1_us"]["p90"],
      "overlapped", d["overlapped"]["value"] / 1e6, "graph", d["graph_replay"]["value"] / 1e6, "transfer", d["transfer_inclusive"]["value"] / 1e6,
      d["transfer_inclusive"]["one_stream"]["value"] / 1e6, d["transfer_inclusive"]["two_streams"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["pass_ms_p10_p90"],
      d["cpu_baseline"]["single_core"]["value"], d["cpu_baseline_reference_path_port"]["value"], "flops", d["roofline"]["flops"]["achieved"], d["roofline"]["flops"]["flops_per_pose"])
try:
    print("ransac frames/s", json.load(open(f"profiles/{tag}/bench_ransac_n4_50k.json"))["ransac_frame"]["frames_per_s"])
except Exception as e:  # noqa: BLE001
    print("ransac", e)
