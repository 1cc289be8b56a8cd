#!/usr/bin/env python3
"""Time of the four-per-wavefront interior-point kernel alone (GPU box): cvxpnpl_ipm_batch on the costs of four-point problems."""
import json
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

for n in (4096, 8192, 10000, 16384, 40000):
    d = synth.make_pnp(n, 4, 2.0, seed=3)
    p2, p3, K = (torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
    Bt, Qt = ca.assemble_batch(p2, None, p3, None, K)
    for variant in (0, 1):
        Z, S, gap, it = ca.ipm_batch(Qt, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            Z, S, gap, it = ca.ipm_batch(Qt, variant=variant)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        itn = it.cpu().numpy() & 255
        print(json.dumps({"problems": n, "variant": variant, "ms": round(float(np.median(ts)), 4), "us_per_problem_slot": round(float(np.median(ts)) * 1e3 / max(1, -(-n // 8192)), 1),
                          "iters_mean": round(float(itn.mean()), 2), "iters_max": int(itn.max()), "gap_median": float(np.median(gap.cpu().numpy())), "gap_max": float(gap.max())}))
