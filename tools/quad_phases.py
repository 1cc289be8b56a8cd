#!/usr/bin/env python3
"""Shader cycles per phase of the quad kernel (diagnostics; GPU box):
    hipcc ... -DCVXQ_PHASES -o tools/diag/libcvxpnpl_phases.so ; CVXPNPL_AMD_LIB=tools/diag/libcvxpnpl_phases.so python tools/quad_phases.py [batch]
batch = 4 is ONE wavefront alone on the chip: the latency regime of the stragglers that end every launch."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = {"f32_sweeps_until": 0} if "f64" in sys.argv[2:] else {}   # (f64: every sweep in float64, bench.py's headline mode)
dev = torch.device("cuda:0")
d = synth.make_pnp(max(batch, 4), 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(d[k][:batch], device=dev) for k in ("pts_2d", "pts_3d", "K"))
K = torch.as_tensor(d["K"], device=dev)
for _ in range(3):
    res = ca.pnp_batch(p2, p3, K, layout=3, **kw)
torch.cuda.synchronize()
c = res.cost.cpu().numpy().reshape(-1)
w = res.work.cpu().numpy().reshape(-1)
nw = batch // 4
P = np.stack([c[8 * i: 8 * i + 8] for i in range(nw)])
its = np.array([w[8 * i] for i in range(nw)])
sw = np.array([w[8 * i + 1] for i in range(nw)])
names = ["g_build", "jacobi", "wp", "check_top", "polish", "dual", "ldl_out", "proj_update"]
tot = P.sum(1)
out = {"batch": batch, "mode": "f64" if kw else "mixed", "waves": nw, "mean_iters_of_wave": float(its.mean()), "mean_sweeps": float(sw.mean()),
       "cycles_per_wave_total_median": float(np.median(tot)),
       "cycles_per_wave_median": {n: float(np.median(P[:, k])) for k, n in enumerate(names)},
       "share": {n: float(P[:, k].sum() / tot.sum()) for k, n in enumerate(names)},
       "per_iteration_cycles": {n: float(np.median(P[:, k] / np.maximum(its, 1))) for k, n in enumerate(names)}}
print(json.dumps(out))
