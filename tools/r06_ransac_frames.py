#!/usr/bin/env python3
"""Round 6 (GPU box): N RANSAC frames of BASELINE config 5 (one scene of 100 correspondences, 30 % clutter, 50 000 hypotheses) through
cvxpnpl_amd.ransac.ransac_pnp; prints frames/s.  Under `rocprofv3 --kernel-trace --stats` (tools/r06_ransac_kernels.sh) the trace gives the
kernels of a frame."""
import sys
import time

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvxpnpl_amd import ransac, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
f64 = len(sys.argv) > 2 and sys.argv[2] == "f64"   # every Jacobi sweep in float64 (the reference's precision) for the hypotheses' solves
dev = torch.device("cuda:0")
d = synth.make_ransac(1, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
x, X, K = (torch.as_tensor(d[k], device=dev) for k in ("scene_2d", "scene_3d", "K"))
kw = {"f32_sweeps_until": 0} if f64 else {}
for i in range(5):
    fr = ransac.ransac_pnp(x, X, K, n_hyp=50_000, thresh=2.0, seed=100 + i, eps=1e-9, max_iters=2500, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    fr = ransac.ransac_pnp(x, X, K, n_hyp=50_000, thresh=2.0, seed=i, eps=1e-9, max_iters=2500, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"FRAMES {'float64 sweeps' if f64 else 'default sweeps'}: {n} frames, {1.0 / dt:.1f} frames/s, {1e3 * dt:.3f} ms per frame; last frame: {fr['n_inliers']} inliers, status {fr['status']}, "
      f"{fr['n_certified']} certified hypotheses, rotation error {float(synth.geodesic(fr['R'].cpu().numpy()[None], d['R_gt'][None])[0]):.2e} rad")
