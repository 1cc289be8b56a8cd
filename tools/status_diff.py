#!/usr/bin/env python3
"""Problems whose status differs between layouts (diagnostics, GPU box): python tools/status_diff.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
import oracle as orc  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
for (n_p, n_l, sigma, seed) in ((5, 0, 0.0, 5005), (5, 0, 0.5, 5010), (9, 1, 2.0, 5012), (0, 5, 2.0, 5023), (0, 4, 2.0, 5019)):
    d = synth.make_pnpl(192, n_p, n_l, sigma, seed=seed)
    tt = lambda x: torch.as_tensor(x, device=dev)  # noqa: E731
    res = {}
    for name, layout in (("wave", 2), ("quad", 3), ("lane", 1)):
        r = ca.pnpl_batch(tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                          tt(d["line_3d"]) if n_l else None, tt(d["K"]), layout=layout)
        res[name] = {k: v.cpu().numpy() for k, v in r.items()}
    o = orc.pnpl_batch(d["pts_2d"] if n_p else None, d["line_2d"] if n_l else None, d["pts_3d"] if n_p else None,
                       d["line_3d"] if n_l else None, d["K"], eps=1e-11, max_iters=200000)
    diff = np.where((res["wave"]["status"] != res["quad"]["status"]) | (res["wave"]["status"] != res["lane"]["status"]))[0]
    for i in diff:
        print(f"n_p {n_p} n_l {n_l} sigma {sigma} problem {i}: oracle n_poses {o['n_poses'][i]}",
              " ".join(f"{k}: st {res[k]['status'][i]} it {res[k]['iters'][i]} rank {res[k]['work'][i, 0]} cost {res[k]['cost'][i, 0]:.3e}/{res[k]['cost'][i, 1]:.3e}" for k in res),
              f"geo(wave,quad) {synth.geodesic(res['wave']['R'][i:i+1], res['quad']['R'][i:i+1])[0]:.2e}")
