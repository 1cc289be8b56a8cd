#!/usr/bin/env python3
"""One hard problem alone on the device (the slowest of 10 k four-point problems): launch-to-result time of a batch of one, without
and with the interior-point path (opts.rescue_from).  GPU box: python tools/hard_single.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

d = synth.make_pnp(10_000, 4, 2.0, seed=3)
p2, p3, K = (torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
it = ca.pnp_batch(p2, p3, K, max_iters=2500, rescue_from=0).iters.cpu().numpy()
for idx in np.argsort(-it)[:3]:
    q2, q3 = p2[idx:idx + 1].contiguous(), p3[idx:idx + 1].contiguous()
    row = {"problem": int(idx)}
    ref = None
    for rf in (0, 96, 32):
        for _ in range(3):
            r = ca.pnp_batch(q2, q3, K, max_iters=2500, rescue_from=rf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r = ca.pnp_batch(q2, q3, K, max_iters=2500, rescue_from=rf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        R = r.R.cpu().numpy()[0]
        ref = R if ref is None else ref
        row[f"rescue_from_{rf}"] = {"us": round(dt * 1e6, 1), "iters": int(r.iters[0]), "status": int(r.status[0]),
                                    "rot_diff_vs_off_rad": float(synth.geodesic(R[None], ref[None])[0])}
    print(json.dumps(row))
