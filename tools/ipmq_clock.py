#!/usr/bin/env python3
"""Where the cycles of the four-per-wavefront interior-point solve go (GPU box; needs tools/diag/libcvxpnpl_ipmqclock.so = -DCVXI_CLOCK):
shader-clock cycles per stage, per wavefront, at one and at two wavefronts per SIMD."""
import json
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVXPNPL_AMD_LIB"] = os.path.join(root, "tools", "diag", "libcvxpnpl_ipmqclock.so")
sys.path.insert(0, root)
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

names = ["ldl S + inverse", "schur", "ldl schur", "rhs + solves", "dS, dZ", "step tests", "second-order term", "update"]
for n in (4, 4096, 8192):
    d = synth.make_pnp(n, 4, 2.0, seed=3)
    p2, p3, K = (torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
    Bt, Qt = ca.assemble_batch(p2, None, p3, None, K)
    for variant in (0, 1):
        Z, S, gap, it = ca.ipm_batch(Qt, variant=variant)
        torch.cuda.synchronize()
        rec = S.cpu().numpy().reshape(n, 100)[::4, :10]   # one record per wavefront
        clk, calls = rec[:, :8], rec[:, 8:]
        itw = (it.cpu().numpy() & 255).reshape(-1, 4).max(axis=1) if n % 4 == 0 else (it.cpu().numpy() & 255)[:1]
        tot = clk.sum(axis=1)
        print(json.dumps({"problems": n, "variant": variant, "wave_iters_mean": round(float(itw.mean()), 2), "cycles_per_wave": round(float(tot.mean())),
                          "cycles_per_wave_iteration": round(float(tot.mean() / max(1.0, itw.mean() + 1))),
                          "step_test_calls_per_iteration(predictor, corrector)": [round(float(calls[:, 0].mean() / max(1.0, itw.mean() + 1)), 2), round(float(calls[:, 1].mean() / max(1.0, itw.mean() + 1)), 2)], "share": {nm: round(float(clk[:, i].mean() / tot.mean()), 3) for i, nm in enumerate(names)}}))
