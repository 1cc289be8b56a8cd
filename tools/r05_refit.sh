#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, torch, numpy as np
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth
from cvxpnpl_amd.ransac import ransac_pnp
d = synth.make_ransac(50_000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
x, X, K = (torch.as_tensor(d[k], device="cuda") for k in ("scene_2d", "scene_3d", "K"))
for rounds in (1, 2, 1, 2):
    for _ in range(3): out = ransac_pnp(x, X, K, n_hyp=50_000, thresh=2.0, max_iters=2500, eps=1e-9, seed=1, refit_rounds=rounds)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): out = ransac_pnp(x, X, K, n_hyp=50_000, thresh=2.0, max_iters=2500, eps=1e-9, seed=i, refit_rounds=rounds)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("refit_rounds", rounds, "ms/frame", round(1e3 * dt, 3), "fps", round(1 / dt, 1), "inliers", out["n_inliers"], "status", out["status"], "rot err", synth.geodesic(out["R"].cpu().numpy(), d["R_gt"]))
PY
