#!/usr/bin/env python3
"""Latency of the seam calls solve_relaxation / solve_relaxation_rc (one problem as A, B; the harness's unit for the rc ablation) next to pnp (GPU box)."""
import time, numpy as np, sys
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth, api
d = synth.make_pnpl(1, 8, 0, 1.0, seed=3)
Bt, Qt = api.assemble_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"])
q = Qt[0].cpu().numpy(); B = Bt[0].cpu().numpy().reshape(3, 9)
Q = np.zeros((9, 9)); iu = np.triu_indices(9); Q[iu] = q; Q = Q + Q.T - np.diag(np.diag(Q))
w, V = np.linalg.eigh(Q); A = (V * np.sqrt(np.maximum(w, 0))).T
for name, f in (("solve_relaxation", lambda: ca.solve_relaxation(A, B)), ("solve_relaxation_rc", lambda: ca.solve_relaxation_rc(A, B)), ("pnp", lambda: ca.pnp(d["pts_2d"][0], d["pts_3d"][0], d["K"]))):
    for _ in range(20): f()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); p = f(); ts.append(time.perf_counter() - t0)
    print(name, "median %.0f us p90 %.0f us" % (np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6), len(p), "pose(s)", np.round(p[0][1], 4))
