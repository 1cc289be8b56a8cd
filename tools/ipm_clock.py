#!/usr/bin/env python3
"""Where the time of a rescued problem goes (GPU box; needs the diagnostic build libcvxpnpl_ipmclock.so = -DCVXW_IPM_CLOCK):
100 MHz ticks of (assembly + exit, interior-point solve, first-order solve after it) per rescued problem."""
import json
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVXPNPL_AMD_LIB"] = os.path.join(root, "tools", "diag", "libcvxpnpl_ipmclock.so")
sys.path.insert(0, root)
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "planar":
    d = synth.make_pnp(10000, 10, 0.0, seed=1)
    d["pts_3d"][:, :, 2] = 0.0
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"])
else:
    d = synth.make_pnp(10_000, 4, 2.0, seed=3)
p2, p3, K = (torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
for R in (40, 120):
    res = ca.pnp_batch(p2, p3, K, max_iters=2500, rescue_from=R)
    it = res.iters.cpu().numpy()
    cost = res.cost.cpu().numpy()
    t = res.t.cpu().numpy()
    Rm = res.R.cpu().numpy().reshape(-1, 9)
    m = it > R
    ipm = cost[m, 0]
    nit = np.round((ipm - np.floor(ipm)) * 1e3)
    print(json.dumps({"rescue_from": R, "rescued": int(m.sum()), "us_before": float(np.median(t[m, 0]) / 100), "us_ipm": float(np.median(np.floor(ipm)) / 100),
                      "ipm_iters_median": float(np.median(nit)), "us_per_ipm_iter": float(np.median(np.floor(ipm) / np.maximum(nit, 1)) / 100),
                      "us_after": float(np.median(cost[m, 1]) / 100), "iters_after_median": float(np.median(it[m] - R - nit)),
                      "us_by_stage(cholS,inv,schur,cholM,regs,rhs+solve+dS,mul+dZ,steps,rest)": [round(float(v) / 100, 1) for v in np.median(Rm[m], axis=0)]}))
