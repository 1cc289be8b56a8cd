"""Planar scenes in a general world frame (random plane, random offset): status, iterations, both poses."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth


if __name__ == "__main__":
    for sigma in (0.0, 0.5):
        d = synth.make_planar_pnp(4000, 10, sigma, seed=3)
        for layout in (2, 1, 3):
            res = ca.pnp_batch(d["pts_2d"], d["pts_3d"], d["K"], want_Z=True, layout=layout)
            st, it = res.status.cpu().numpy(), res.iters.cpu().numpy()
            Bt, Qt = ca.assemble_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"])
            R, t, cnt = ca.recover_multi_batch(res, Bt, Qt)
            err = np.array([min(synth.geodesic(R[i, k], d["R_gt"][i]) + np.linalg.norm(t[i, k] - d["t_gt"][i]) for k in range(max(cnt[i], 1))) for i in range(0, 4000, 10)])
            one = synth.geodesic(res.R.cpu().numpy(), d["R_gt"])
            print("sigma %.1f layout %d: status %s iters p50 %d p90 %d mean %.1f | poses %s best-pose err median %.1e max %.1e | returned R is gt for %.2f"
                  % (sigma, layout, np.bincount(st, minlength=5).tolist(), np.percentile(it, 50), np.percentile(it, 90), it.mean(),
                     np.bincount(cnt, minlength=5).tolist(), np.median(err), err.max(), np.mean(one < (1e-6 if sigma == 0 else 0.2))))
