#!/usr/bin/env python3
"""round 6 (GPU box): what one barrier Newton solve of the dual costs a wavefront on its chain.  64 ten-point problems of the 125 000-problem set
that certify after 7 iterations with it and after 9 or more without, solved as one lane-schedule launch (one wavefront in the lane phase, then
one wavefront per problem in the resume phase: the launch lasts as long as one chain) with opts.dual_refine = 1 and 2."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _solve  # noqa: E402

from cvxpnpl_amd import pnp_batch, synth  # noqa: E402

dev = torch.device("cuda:0")
d = synth.make_pnpl(125000, 10, 0, 2.0, seed=42)
a = _solve(dev, d, 10, 0, dual_refine=1)
b = _solve(dev, d, 10, 0, dual_refine=2)
pick = np.flatnonzero((b["iters"] == 7) & (a["iters"] >= 9))[:64]
print("problems", len(pick), "iterations without", np.bincount(a["iters"][pick])[7:].tolist(), "(from 7)")
p2, p3 = (torch.as_tensor(d[k][pick], device=dev).contiguous() for k in ("pts_2d", "pts_3d"))
K = torch.as_tensor(d["K"], device=dev)
for mode in (1, 2, 1, 2):
    for _ in range(5):
        r = pnp_batch(p2, p3, K, layout=1, dual_refine=mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        r = pnp_batch(p2, p3, K, layout=1, dual_refine=mode)
    e1.record()
    torch.cuda.synchronize()
    print(f"dual_refine={mode}: {1e3 * e0.elapsed_time(e1) / 50:8.1f} us per launch of 64 problems, iterations {np.bincount(r.iters.cpu().numpy())[6:].tolist()} (from 6)")
