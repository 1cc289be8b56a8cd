"""Throughput on planar scenes (rank-2, two-fold ambiguous): the slow path of the solver."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth

for batch, sigma in ((10000, 0.0), (10000, 1.0)):
    d = synth.make_pnp(batch, 10, 0.0, seed=1)
    d["pts_3d"][:, :, 2] = 0.0
    rs = np.random.RandomState(5)
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"]) + rs.normal(scale=sigma, size=d["pts_2d"].shape)
    dev = {k: torch.as_tensor(v, device="cuda") for k, v in d.items() if k in ("pts_2d", "pts_3d", "K")}
    for layout in (1, 2, 3, 0):
        ca.pnp_batch(dev["pts_2d"], dev["pts_3d"], dev["K"], layout=layout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ca.pnp_batch(dev["pts_2d"], dev["pts_3d"], dev["K"], layout=layout)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        it = res.iters.cpu().numpy()
        st = np.bincount(res.status.cpu().numpy(), minlength=5).tolist()
        print(json.dumps({"workload": "planar_pnp_n10", "batch": batch, "sigma_px": sigma, "layout": layout, "ms": dt * 1e3,
                          "problems_per_s": batch / dt, "status": st, "iters_mean": float(it.mean()),
                          "iters_p50": float(np.median(it)), "iters_p99": float(np.percentile(it, 99))}))
