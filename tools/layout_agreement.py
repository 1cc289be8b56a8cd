"""Large-sample agreement of the three layouts (GPU box): statuses, iteration counts and poses."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth

def planar(n, seed):
    d = synth.make_pnp(n, 10, 0.0, seed=seed)
    d["pts_3d"][:, :, 2] = 0.0
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"]) + np.random.RandomState(seed).normal(scale=0.5, size=d["pts_2d"].shape)
    return d

sets = {"pnp10_200k": synth.make_pnp(200000, 10, 2.0, seed=5), "pnpl_100k": synth.make_pnpl(100000, 5, 5, 1.5, seed=6),
        "pnl8_50k": synth.make_pnpl(50000, 0, 8, 1.0, seed=7), "pnp4_30k": synth.make_pnp(30000, 4, 1.0, seed=8),
        "pnp6_noisy_50k": synth.make_pnp(50000, 6, 5.0, seed=9), "planar_20k": planar(20000, 10)}
for name, d in sets.items():
    args = (d.get("pts_2d"), d.get("line_2d"), d.get("pts_3d"), d.get("line_3d"), d["K"])
    args = tuple(None if a is None or (hasattr(a, "size") and a.size == 0) else a for a in args)
    ref = ca.pnpl_batch(*args, layout=2, max_iters=400)
    sr = ref.status.cpu().numpy()
    for layout in (1, 3):
        r = ca.pnpl_batch(*args, layout=layout, max_iters=400)
        s = r.status.cpu().numpy()
        both = (s == 0) & (sr == 0)
        dR = (r.R - ref.R).abs().flatten(1).max(1).values.cpu().numpy()
        dt = (r.t - ref.t).abs().max(1).values.cpu().numpy()
        dit = (r.iters - ref.iters).abs().cpu().numpy()
        nanbad = int((torch.isnan(r.R).flatten(1).any(1).cpu().numpy() & (s != 3)).sum())
        print("%-15s layout %d: status mismatch %6d (%.4f%%)  certified both %.4f  max dR %.1e max dt %.1e  mean|d it| %.4f max %d  NaN-with-good-status %d  hist %s"
              % (name, layout, (s != sr).sum(), 100.0 * (s != sr).mean(), both.mean(), dR[both].max() if both.any() else 0, dt[both].max() if both.any() else 0,
                 dit[both].mean() if both.any() else 0, dit[both].max() if both.any() else 0, nanbad, np.bincount(s, minlength=5).tolist()))
