#!/usr/bin/env python3
"""Iteration / status histogram of a workload (GPU box): python tools/iters_hist.py [workload] [max_iters]
Diagnostics: where the tail of a launch comes from."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config5"
max_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
dev = torch.device("cuda:0")
if wl == "config5":
    d = synth.make_ransac(50_000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
    n_p = 4
else:
    d = synth.make_pnp(int(wl), 10, 2.0, seed=42)
    n_p = 10
p2, p3, K = (torch.as_tensor(d[k], device=dev) for k in ("pts_2d", "pts_3d", "K"))
for _ in range(2):
    res = ca.pnp_batch(p2, p3, K, max_iters=max_iters)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = ca.pnp_batch(p2, p3, K, max_iters=max_iters)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
it = res.iters.cpu().numpy()
st = res.status.cpu().numpy()
q = [50, 90, 99, 99.9, 99.99, 100]
out = {"workload": wl, "n": len(it), "ms": 1e3 * dt, "rate": len(it) / dt, "status": np.bincount(st, minlength=5).tolist(),
       "iters_pct": {str(p): float(np.percentile(it, p)) for p in q}, "mean_iters": float(it.mean()),
       "iters_by_status": {str(s): {"n": int((st == s).sum()), "mean": float(it[st == s].mean()) if (st == s).any() else None,
                                    "max": int(it[st == s].max()) if (st == s).any() else None,
                                    "sum_frac": float(it[st == s].sum() / it.sum())} for s in range(5)},
       "n_over": {str(k): int((it > k).sum()) for k in (10, 20, 50, 100, 200, 500, 1000, 2000)}}
print(json.dumps(out))
