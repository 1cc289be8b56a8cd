#!/usr/bin/env python
"""solve_lane4_kernel (four lanes per problem) against solve_lane2_kernel (one lane per problem) on the same inputs, then launch times
by size for quad / lane4 / lane.  GPU box: python tools/lane4_check.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
for n_p, n_l, sig, B in [(10, 0, 2.0, 5003), (5, 5, 1.0, 3001), (4, 0, 1.0, 1000), (10, 0, 0.0, 777)]:
    d = synth.make_pnpl(B, n_p, n_l, sig, seed=9)
    tt = lambda x: torch.as_tensor(x, device=dev)  # noqa: E731
    a = (tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None, tt(d["line_3d"]) if n_l else None, tt(d["K"]))
    r1 = {k: v.cpu().numpy() for k, v in ca.pnpl_batch(*a, layout=1, want_Z=True).items()}
    r4 = {k: v.cpu().numpy() for k, v in ca.pnpl_batch(*a, layout=5, want_Z=True).items()}
    same = r1["status"] == r4["status"]
    both = (r1["status"] == 0) & (r4["status"] == 0)
    print(json.dumps({"case": [n_p, n_l, sig, B], "status_equal": float(same.mean()), "hist1": np.bincount(r1["status"], minlength=5).tolist(),
                      "hist4": np.bincount(r4["status"], minlength=5).tolist(), "max_rot_diff": float(synth.geodesic(r1["R"], r4["R"])[both].max()),
                      "max_t_diff": float(np.abs(r1["t"] - r4["t"])[both].max()), "iters_equal": float((r1["iters"] == r4["iters"]).mean()),
                      "sweeps": [float(r1["work"][:, 1].mean()), float(r4["work"][:, 1].mean())]}))
