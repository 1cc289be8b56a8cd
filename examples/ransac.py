"""BASELINE config 5 as a caller sees it: one scene of 100 correspondences, 30 % of the pixels replaced by clutter, 50 000
four-point hypotheses sampled, solved, scored and refitted on the device (cvxpnpl_amd.ransac.ransac_pnp; not in the reference)."""
import numpy as np
import torch

import _scene  # noqa: F401
from cvxpnpl_amd import ransac, synth

d = synth.make_ransac(1, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)  # (the scene only: ransac_pnp draws its own subsets)
fr = ransac.ransac_pnp(d["scene_2d"], d["scene_3d"], d["K"], n_hyp=50_000, thresh=2.0, seed=1, device=torch.device("cuda:0"))
R = fr["R"].cpu().numpy()
inl = fr["inliers"].cpu().numpy().astype(bool)
truth = d["inlier"]
gap = float(synth.geodesic(R[None], d["R_gt"][None])[0])
print(f"{fr['n_inliers']} inliers of {len(inl)} correspondences ({int(truth.sum())} true ones), {fr['n_certified']} of {fr['n_hyp']} hypotheses "
      f"certified, rotation error {gap:.2e} rad")
assert inl.sum() >= 0.9 * truth.sum() and (inl & ~truth).sum() <= 2 and gap < 5e-3, (inl.sum(), (inl & ~truth).sum(), gap)
