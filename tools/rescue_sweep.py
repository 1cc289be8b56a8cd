#!/usr/bin/env python3
"""opts.rescue_from (first-order iterations before the interior-point path takes a problem over) against launch time, on the
workloads with slow problems (GPU box): python tools/rescue_sweep.py [values...]      0 = no interior-point path"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

vals = [int(v) for v in sys.argv[1:]] or [0, 32, 48, 64, 96, 128, 192, 256]
dev = torch.device("cuda:0")


def planar(sigma):
    d = synth.make_pnp(10000, 10, 0.0, seed=1)
    d["pts_3d"][:, :, 2] = 0.0
    rs = np.random.RandomState(5)
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"]) + rs.normal(scale=sigma, size=d["pts_2d"].shape)
    return d


sets = {
    "config5_50k": synth.make_ransac(50_000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46),
    "config5_2500": synth.make_ransac(2_500, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=47),
    "pnp_n4_10k_s2": synth.make_pnp(10_000, 4, 2.0, seed=3),
    "pnp_n5_10k_s2": synth.make_pnp(10_000, 5, 2.0, seed=4),
    "pnp_n6_10k_s5": synth.make_pnp(10_000, 6, 5.0, seed=5),
    "planar_s0": planar(0.0),
    "planar_s1": planar(1.0),
    "pnp_n10_10k": synth.make_pnp(10_000, 10, 2.0, seed=42),
    "pnp_n10_125k": synth.make_pnp(125_000, 10, 2.0, seed=43),
    "pnp_n10_1M": synth.make_pnp(1_000_000, 10, 2.0, seed=44),
}
for name, d in sets.items():
    p2, p3, K = (torch.as_tensor(d[k], device=dev) for k in ("pts_2d", "pts_3d", "K"))
    row = {}
    for v in vals:
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = ca.pnp_batch(p2, p3, K, max_iters=2500, rescue_from=v)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        it = res.iters.cpu().numpy()
        st = np.bincount(res.status.cpu().numpy(), minlength=5).tolist()
        row[str(v)] = {"ms": round(1e3 * min(ts[1:]), 3), "status": st[:3], "iters_max": int(it.max()), "iters_mean": round(float(it.mean()), 2)}
    print(json.dumps({"workload": name, "n": int(p2.shape[0]), "by_rescue_from": row}))
