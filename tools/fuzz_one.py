#!/usr/bin/env python3
"""One configuration of a tools/fuzz_parity.py campaign again, with the problems that did not certify spelled out (GPU box):
python tools/fuzz_one.py <seed> <cfg> [problems]: status, iterations, eigenvalues of the returned Z, the oracle's answer for the same problem."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
import oracle as orc  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
nprob = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rs = np.random.RandomState(seed)
for c in range(want + 1):  # (the draws of tools/fuzz_parity.py, non-minimal campaign)
    kind = rs.choice(["pnp", "pnl", "pnpl"])
    n_p = int(rs.randint(4, 25)) if kind != "pnl" else 0
    n_l = int(rs.randint(4, 13)) if kind == "pnl" else (int(rs.randint(1, 9)) if kind == "pnpl" else 0)
    if kind == "pnpl":
        n_p = int(rs.randint(2, 13))
    sigma = float(rs.choice([0.0, 0.5, 1.0, 2.0, 5.0]))
d = synth.make_pnpl(nprob, n_p, n_l, sigma, seed=5000 + want + (0 if seed == 2026 else 100000 + seed * 1000))
tt = lambda x: torch.as_tensor(x, device="cuda:0")  # noqa: E731
print(f"seed {seed} cfg {want}: {kind} n_p {n_p} n_l {n_l} sigma {sigma}")
for name, layout in (("wave", 2), ("quad", 3), ("lane", 1)):
    r = ca.pnpl_batch(tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                      tt(d["line_3d"]) if n_l else None, tt(d["K"]), layout=layout, want_Z=True)
    st = r.status.cpu().numpy()
    for i in np.where(st != 0)[0]:
        sel = lambda a: a[i:i + 1] if a is not None and len(a) else None  # noqa: E731
        o = orc.pnpl_batch(sel(d["pts_2d"]) if n_p else None, sel(d["line_2d"]) if n_l else None, sel(d["pts_3d"]) if n_p else None,
                           sel(d["line_3d"]) if n_l else None, d["K"], eps=1e-11, max_iters=400000)
        Z = r.Z[i].cpu().numpy()
        M = np.zeros((10, 10)); M[np.triu_indices(10)] = Z; M = M + M.T - np.diag(np.diag(M))
        ev = np.linalg.eigvalsh(M)[::-1]
        geo = synth.geodesic(r.R[i:i + 1].cpu().numpy(), o["R"][:, 0])[0]
        print(f"  {name}: problem {i} status {st[i]} iters {int(r.iters[i])} eig(Z) {ev[0]:.4f} {ev[1]:.2e} {ev[2]:.2e}; oracle: n_poses {int(o['n_poses'][0])}"
              f" status {int(o['status'][0]) if 'status' in o else '-'}; rotation vs the oracle's first pose {geo:.2e} rad")
