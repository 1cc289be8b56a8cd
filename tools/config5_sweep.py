#!/usr/bin/env python3
"""BASELINE config 5: RANSAC-style minimal (N=4) PnP hypotheses with 30 % outliers, tolerance sweep.

One scene of 100 correspondences (30 % of the 2D points replaced by uniform clutter), 50 000 random
4-subsets, each solved as PnP N=4 on the GPU; hypotheses are scored by reprojection inliers
(< 2 px) over the whole scene.  Sweeps the solver budget (max_iters) and tolerance (eps) and prints
one JSON line per setting: poses/s, fraction certified / rank>1 / uncertified, inliers of the best
hypothesis and its rotation error against the ground truth.
Usage (GPU box, repo root):  python tools/config5_sweep.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402


def score(R, t, K, X, x, thresh=2.0):
    """inlier counts of every hypothesis: R [H,3,3], t [H,3]; scene X [M,3], x [M,2]."""
    Xc = torch.einsum("hij,mj->hmi", R, X) + t[:, None, :]
    uv = torch.einsum("ij,hmj->hmi", K, Xc)
    uv = uv[..., :2] / uv[..., 2:3]
    err = torch.linalg.norm(uv - x[None], dim=-1)
    ok = (err < thresh) & (Xc[..., 2] > 0)
    return ok.sum(1)


def main():
    dev = torch.device("cuda:0")
    d = synth.make_ransac(50_000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
    p2 = torch.as_tensor(d["pts_2d"], device=dev)
    p3 = torch.as_tensor(d["pts_3d"], device=dev)
    K = torch.as_tensor(d["K"], device=dev)
    X = torch.as_tensor(d["scene_3d"], device=dev)
    x = torch.as_tensor(d["scene_2d"], device=dev)
    n_inl_true = int(d["inlier"].sum())
    for eps in (1e-3, 1e-6, 1e-9):
        for max_iters in (20, 50, 100, 300, 1000, 2500):
            ca.pnp_batch(p2, p3, K, eps=eps, max_iters=max_iters)  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                res = ca.pnp_batch(p2, p3, K, eps=eps, max_iters=max_iters)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            st = res.status
            usable = (st == 0) | (st == 2)
            inl = score(torch.nan_to_num(res.R), torch.nan_to_num(res.t), K, X, x)
            inl = torch.where(usable, inl, torch.zeros_like(inl))
            best = int(torch.argmax(inl))
            Rb = res.R[best].cpu().numpy()
            out = {
                "eps": eps, "max_iters": max_iters, "poses_per_s": 50_000 / dt, "ms": 1e3 * dt,
                "certified": float((st == 0).float().mean()), "rank_gt1": float((st == 1).float().mean()),
                "uncertified": float(((st == 2) | (st == 4)).float().mean()),
                "mean_iters": float(res.iters.float().mean()),
                "best_inliers": int(inl[best]), "true_inliers": n_inl_true,
                "best_rot_err_rad": float(synth.geodesic(Rb, d["R_gt"])),
                "hyp_with_80pct_inliers": int((inl >= 0.8 * n_inl_true).sum()),
            }
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
