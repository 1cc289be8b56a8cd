"""RANSAC on top of the batched solver (BASELINE config 5).

The reference has no RANSAC; this is the natural consumer of tens of thousands of minimal
hypotheses per frame: sample 4-subsets, solve them all in one launch, score every hypothesis by
reprojection inliers over the whole scene, refit the best consensus set with one more (N = #inliers)
solve.  Sampling, the solves and the scoring are the HIP path (cvxpnpl_sample_minimal_sets, cvxpnpl_solve_batch,
cvxpnpl_score_hypotheses); torch takes the arg-max.
"""
from typing import Optional

import torch

from .api import pnp_batch, sample_minimal_sets, score_hypotheses


def reprojection_inliers(R: torch.Tensor, t: torch.Tensor, K: torch.Tensor, pts_3d: torch.Tensor, pts_2d: torch.Tensor,
                         thresh: float = 2.0) -> torch.Tensor:
    """[H, M] bool: correspondence m is an inlier of hypothesis h (reprojection error < thresh px, in front
    of the camera).  R [H,3,3], t [H,3], scene pts_3d [M,3], pts_2d [M,2].  (HIP scoring kernel.)"""
    return score_hypotheses(R, t, K, pts_2d, pts_3d, thresh, want_mask=True)[1].bool()


def ransac_pnp(pts_2d, pts_3d, K, n_hyp: int = 4096, thresh: float = 2.0, max_iters: int = 100, eps: float = 1e-6,
               seed: Optional[int] = 0, refit: bool = True, device=None):
    """Robust PnP for one scene with outliers.

    pts_2d [M,2], pts_3d [M,3] (numpy or torch), K [3,3].  Returns dict with R [3,3], t [3],
    inliers [M] bool, n_inliers, status of the final solve, n_certified hypotheses.
    """
    if device is None:  # like pnpl_batch: the device of a CUDA input, else the current device
        for a in (pts_3d, pts_2d, K):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                device = a.device
                break
        else:
            device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    x = torch.as_tensor(pts_2d, dtype=torch.float64, device=device)
    X = torch.as_tensor(pts_3d, dtype=torch.float64, device=device)
    Kd = torch.as_tensor(K, dtype=torch.float64, device=device)
    M = X.shape[0]
    # n_hyp random 4-subsets without replacement, gathered into the solve's inputs by one kernel (cvxpnpl_sample_minimal_sets; round 3 drew
    # them with torch.rand().topk(4): 0.32 ms of a 3.5 ms frame at 50 000 hypotheses)
    if seed is None:
        seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
    x4, X4 = sample_minimal_sets(x, X, n_hyp, 4, seed)
    res = pnp_batch(x4, X4, Kd, eps=eps, max_iters=max_iters)
    score = score_hypotheses(res.R, res.t, Kd, x, X, thresh, status=res.status, usable=(0, 2))
    # From here on the host needs a few integers (the size of the consensus set is the N of the refit, a host argument of the solve); each
    # read-back is a synchronisation of ~40 us, so they are batched: ONE per stage instead of one per number.
    best = torch.argmax(score).reshape(1)                      # stays on the device
    R, t = res.R.index_select(0, best)[0], res.t.index_select(0, best)[0]
    mask = reprojection_inliers(R[None], t[None], Kd, X, x, thresh)[0]
    head = torch.stack([res.status.index_select(0, best)[0].to(torch.int64), mask.sum(), (res.status == 0).sum()]).cpu()   # sync 1
    final_status, n_inl, n_cert = int(head[0]), int(head[1]), int(head[2])
    if refit and n_inl >= 4:
        for _ in range(2):  # refit on the consensus set, re-evaluate it once
            sel = torch.argsort((~mask).to(torch.int8), stable=True)[:n_inl]   # the inliers' indices, in order, without a host round trip
            fit = pnp_batch(x.index_select(0, sel)[None], X.index_select(0, sel)[None], Kd, eps=1e-9, max_iters=2500)
            new = reprojection_inliers(fit.R, fit.t, Kd, X, x, thresh)[0]
            st_new = torch.stack([fit.status[0].to(torch.int64), new.sum()]).cpu()   # sync 2 (3)
            if int(st_new[0]) not in (0, 2):
                break
            R, t, final_status = fit.R[0], fit.t[0], int(st_new[0])
            if int(st_new[1]) <= n_inl:
                break
            mask, n_inl = new, int(st_new[1])
    return {"R": R, "t": t, "inliers": mask, "n_inliers": n_inl, "status": final_status, "n_certified": n_cert, "n_hyp": n_hyp}
