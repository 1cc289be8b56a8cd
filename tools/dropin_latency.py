#!/usr/bin/env python3
"""Latency of the drop-in single-problem API (cvxpnpl_amd.pnp / pnl / pnpl), host call to host result (GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth

d = synth.make_pnpl(1, 6, 6, 1.0, seed=3)
calls = {
    "pnp  (6 points)": lambda: ca.pnp(d["pts_2d"][0], d["pts_3d"][0], d["K"]),
    "pnl  (6 lines)": lambda: ca.pnl(d["line_2d"][0], d["line_3d"][0], d["K"]),
    "pnpl (6 + 6)": lambda: ca.pnpl(d["pts_2d"][0], d["line_2d"][0], d["pts_3d"][0], d["line_3d"][0], d["K"]),
}
for name, f in calls.items():
    for _ in range(20):
        f()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        poses = f()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"{name}: median {np.median(ts):.0f} us, p90 {np.percentile(ts, 90):.0f} us, {len(poses)} pose(s)")
