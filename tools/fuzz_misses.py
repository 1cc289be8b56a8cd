#!/usr/bin/env python3
"""Companion of fuzz_parity.py: the problems the GPU does NOT certify, next to the oracle's verdict (GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
import oracle as orc  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

ncfg, nprob = 40, 192
rs = np.random.RandomState(2026)
dev = torch.device("cuda:0")
for c in range(ncfg):
    kind = rs.choice(["pnp", "pnl", "pnpl"])
    n_p = int(rs.randint(4, 25)) if kind != "pnl" else 0
    n_l = int(rs.randint(4, 13)) if kind == "pnl" else (int(rs.randint(1, 9)) if kind == "pnpl" else 0)
    if kind == "pnpl":
        n_p = int(rs.randint(2, 13))
    sigma = float(rs.choice([0.0, 0.5, 1.0, 2.0, 5.0]))
    d = synth.make_pnpl(nprob, n_p, n_l, sigma, seed=5000 + c)
    tt = lambda x: torch.as_tensor(x, device=dev)  # noqa: E731
    r = ca.pnpl_batch(tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                      tt(d["line_3d"]) if n_l else None, tt(d["K"]), want_Z=True)
    st = r.status.cpu().numpy()
    idx = np.where(st != 0)[0]
    if not len(idx):
        continue
    sel = lambda a: a[idx] if a is not None and len(a) else None  # noqa: E731
    o = orc.pnpl_batch(sel(d["pts_2d"]) if n_p else None, sel(d["line_2d"]) if n_l else None, sel(d["pts_3d"]) if n_p else None,
                       sel(d["line_3d"]) if n_l else None, d["K"], eps=1e-11, max_iters=400000)
    for k, i in enumerate(idx):
        Z = r.Z[i].cpu().numpy()
        M = np.zeros((10, 10)); M[np.triu_indices(10)] = Z; M = M + M.T - np.diag(np.diag(M))
        ev = np.linalg.eigvalsh(M)[::-1]
        geo = synth.geodesic(r.R[i:i + 1].cpu().numpy(), o["R"][k:k + 1, 0])[0]
        print(f"cfg {c} {kind} n_p {n_p} n_l {n_l} sigma {sigma} problem {i}: gpu status {st[i]} iters {int(r.iters[i])} eig(Z) {ev[0]:.3f} {ev[1]:.3f} {ev[2]:.3f} | "
              f"oracle n_poses {o['n_poses'][k]} converged {bool(o['converged'][k]) if 'converged' in o else '?'} geo(gpu R, oracle pose 0) {geo:.1e}")
