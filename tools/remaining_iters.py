#!/usr/bin/env python3
"""What a first-order solve that is still open after R iterations has ahead of it (no interior-point path: rescue_from = 0):
per workload and R, how many problems get there and how many more iterations they need (median, 90 %, share above the ~75
iterations an interior-point solve costs).  The data behind the default opts.rescue_from.  GPU box: python tools/remaining_iters.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

sets = {
    "pnp_n10_1M": lambda: synth.make_pnp(1_000_000, 10, 2.0, seed=44),
    "pnpl_5p5l_100k": lambda: synth.make_pnpl(100_000, 5, 5, 2.0, seed=45),
    "pnp_n8_100k": lambda: synth.make_pnp(100_000, 8, 2.0, seed=47),
    "pnp_n7_100k": lambda: synth.make_pnp(100_000, 7, 2.0, seed=48),
    "pnp_n6_100k": lambda: synth.make_pnp(100_000, 6, 2.0, seed=49),
    "pnp_n5_100k": lambda: synth.make_pnp(100_000, 5, 2.0, seed=50),
    "pnp_n4_100k": lambda: synth.make_pnp(100_000, 4, 2.0, seed=51),
    "config5_50k": lambda: synth.make_ransac(50_000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46),
}
for name, make in sets.items():
    d = make()
    tt = lambda k: torch.as_tensor(d[k], device="cuda") if k in d and d[k] is not None and len(d[k]) else None  # noqa: E731
    n_l = d["line_3d"].shape[1] if "line_3d" in d and d["line_3d"] is not None and d["line_3d"].ndim == 4 else 0
    if n_l:
        res = ca.pnpl_batch(tt("pts_2d"), tt("line_2d"), tt("pts_3d"), tt("line_3d"), tt("K"), max_iters=2500, rescue_from=0)
    else:
        res = ca.pnp_batch(tt("pts_2d"), tt("pts_3d"), tt("K"), max_iters=2500, rescue_from=0)
    it = res.iters.cpu().numpy()
    row = {"workload": name, "n": int(it.size), "iters_max": int(it.max())}
    for R in (32, 48, 64, 96):
        m = it > R
        rem = it[m] - R
        row[f"R{R}"] = {"reached": int(m.sum()), "share": float(m.mean()),
                        "remaining_median": float(np.median(rem)) if m.any() else None,
                        "remaining_p90": float(np.percentile(rem, 90)) if m.any() else None,
                        "share_of_those_above_75_more": float((rem > 75).mean()) if m.any() else None}
    print(json.dumps(row))
    del d, res
