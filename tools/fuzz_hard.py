#!/usr/bin/env python3
"""Second parity campaign, unfriendly inputs (GPU box, repo root):  python tools/fuzz_hard.py [problems_per_config]
Scene scale 1e-2 .. 1e2, world origin far from the scene, quasi-planar and quasi-collinear scenes, per-problem
intrinsics, heavy noise, a few gross outliers, minimal sets.  For every configuration and layout:
  cert      fraction CERTIFIED
  agree     certified poses vs the oracle's converged single-pose solve (count beyond 1e-6 rad / relative 1e-6 in t)
  miss      oracle converged to one pose with a certifiably tight relaxation but the GPU did not certify
  lay       problems whose status differs from the wave layout's
Diagnostics / evidence tool (uses the oracle: not product code)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
import oracle as orc  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
rs = np.random.RandomState(77)


def scene(kind, n_p, n_l, sigma, seed):
    d = synth.make_pnpl(nprob, n_p, n_l, 0.0, seed=seed)
    r = np.random.RandomState(seed + 1)
    P = np.concatenate([d["pts_3d"], d["line_3d"].reshape(nprob, 2 * n_l, 3)], axis=1)
    R, t = d["R_gt"], d["t_gt"].copy()
    K = d["K"]
    if kind == "scale":  # whole geometry scaled by s: same images
        s = 10.0 ** r.uniform(-2, 2, (nprob, 1, 1))
        P, t = P * s, t * s[:, 0]
    elif kind == "offset":  # world origin 1e3 scene sizes away
        c = r.normal(size=(nprob, 1, 3)) * 1e3
        P = P + c
        t = t - np.einsum("bij,bj->bi", R, c[:, 0])
    elif kind == "quasiplanar":
        P = P * np.array([1.0, 1.0, 10.0 ** r.uniform(-4, -1)])
    elif kind == "quasicollinear":
        P = P * np.array([1.0, 10.0 ** r.uniform(-3, -1), 10.0 ** r.uniform(-3, -1)])
    elif kind == "perK":
        f = r.uniform(300, 3000, (nprob, 1))
        K = np.tile(np.eye(3), (nprob, 1, 1))
        K[:, 0, 0], K[:, 1, 1] = f[:, 0], f[:, 0] * r.uniform(0.9, 1.1, nprob)
        K[:, 0, 2], K[:, 1, 2] = r.uniform(200, 1000, nprob), r.uniform(200, 800, nprob)
        K[:, 0, 1] = r.uniform(-2, 2, nprob)
    Xc = np.einsum("bij,bnj->bni", R, P) + t[:, None, :]
    uvw = np.einsum("bij,bnj->bni", K, Xc) if K.ndim == 3 else np.einsum("ij,bnj->bni", K, Xc)
    x = uvw[..., :2] / uvw[..., 2:3]
    x = x + r.normal(scale=sigma, size=x.shape) if sigma > 0 else x
    if kind == "outliers" and n_p >= 8:
        x[:, :2] += r.normal(scale=80.0, size=x[:, :2].shape)
    return {"pts_2d": np.ascontiguousarray(x[:, :n_p]), "pts_3d": np.ascontiguousarray(P[:, :n_p]),
            "line_2d": np.ascontiguousarray(x[:, n_p:].reshape(nprob, n_l, 2, 2)),
            "line_3d": np.ascontiguousarray(P[:, n_p:].reshape(nprob, n_l, 2, 3)), "K": K, "R_gt": R, "t_gt": t}


CONFIGS = [("scale", 10, 0, 1.0), ("scale", 5, 5, 1.0), ("offset", 10, 0, 1.0), ("offset", 0, 8, 1.0), ("quasiplanar", 10, 0, 1.0),
           ("quasiplanar", 6, 4, 0.5), ("quasicollinear", 12, 0, 1.0), ("perK", 10, 0, 2.0), ("perK", 4, 4, 1.0), ("noise", 10, 0, 20.0),
           ("noise", 6, 6, 10.0), ("outliers", 12, 0, 1.0), ("minimal", 4, 0, 0.5), ("minimal", 3, 0, 0.0), ("minimal", 0, 3, 0.0),
           ("minimal", 0, 4, 1.0), ("minimal", 2, 2, 1.0)]
t0 = time.time()
tot = bad_tot = miss_tot = lay_tot = 0
for c, (kind, n_p, n_l, sigma) in enumerate(CONFIGS):
    d = scene(kind, n_p, n_l, sigma, 9000 + c)
    tt = lambda x: torch.as_tensor(x, device=dev)  # noqa: E731
    o = orc.pnpl_batch(d["pts_2d"] if n_p else None, d["line_2d"] if n_l else None, d["pts_3d"] if n_p else None,
                       d["line_3d"] if n_l else None, d["K"], eps=1e-11, max_iters=200000)
    line = f"{kind:14s} n_p {n_p:2d} n_l {n_l:2d} sigma {sigma:4.1f}: oracle 1-pose {np.mean(o['n_poses'] == 1):.2f} |"
    ref_st = None
    for name, layout in (("wave", 2), ("quad", 3), ("lane", 1), ("penta", 4)):
        r = ca.pnpl_batch(tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                          tt(d["line_3d"]) if n_l else None, tt(d["K"]), layout=layout)
        st = r.status.cpu().numpy()
        R, t = r.R.cpu().numpy(), r.t.cpu().numpy()
        one = o["n_poses"] == 1
        ok = (st == 0) & one
        geo = synth.geodesic(R, o["R"][:, 0])
        te = np.linalg.norm(t - o["t"][:, 0], axis=1) / np.maximum(np.linalg.norm(o["t"][:, 0], axis=1), 1e-300)
        bad = int(((geo > 1e-6) | (te > 1e-6))[ok].sum())
        miss = int(((st != 0) & one).sum())
        if ref_st is None:
            ref_st = st
        lay = int((st != ref_st).sum())
        tot += nprob; bad_tot += bad; miss_tot += miss; lay_tot += lay
        line += f" {name} cert {np.mean(st == 0):.3f} agree-bad {bad} (max {geo[ok].max() if ok.any() else 0:.1e}) miss {miss} lay {lay} |"
    print(line, flush=True)
print(f"summary: {tot} solves, {bad_tot} certified poses beyond 1e-6 of the oracle, {miss_tot} oracle-single-pose problems not certified, "
      f"{lay_tot} status differences between layouts; {time.time() - t0:.0f} s")
