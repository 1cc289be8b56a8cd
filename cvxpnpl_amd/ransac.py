"""RANSAC on top of the batched solver (BASELINE config 5).

The reference has no RANSAC; this is the natural consumer of tens of thousands of minimal
hypotheses per frame: sample 4-subsets, solve them all in one launch, score every hypothesis by
reprojection inliers over the whole scene, refit the best consensus set (assembled on the device from scene + inlier mask,
solved at the cost seam: no host round trip inside a frame).  Sampling, the solves and the scoring are the HIP path (cvxpnpl_sample_minimal_sets, cvxpnpl_solve_batch,
cvxpnpl_score_hypotheses, cvxpnpl_select_best, cvxpnpl_assemble_subsets, cvxpnpl_solve_cost_batch, cvxpnpl_refit_update): no torch kernel in a frame.
"""
from typing import Optional

import torch

from .api import assemble_subsets, pnp_batch, refit_update, sample_minimal_sets, score_hypotheses, select_best, solve_cost_batch


def reprojection_inliers(R: torch.Tensor, t: torch.Tensor, K: torch.Tensor, pts_3d: torch.Tensor, pts_2d: torch.Tensor,
                         thresh: float = 2.0) -> torch.Tensor:
    """[H, M] bool: correspondence m is an inlier of hypothesis h (reprojection error < thresh px, in front
    of the camera).  R [H,3,3], t [H,3], scene pts_3d [M,3], pts_2d [M,2].  (HIP scoring kernel.)"""
    return score_hypotheses(R, t, K, pts_2d, pts_3d, thresh, want_mask=True)[1].bool()


def ransac_pnp(pts_2d, pts_3d, K, n_hyp: int = 4096, thresh: float = 2.0, max_iters: int = 100, eps: float = 1e-6,
               seed: Optional[int] = 0, refit: bool = True, device=None, refit_rounds: int = 1, **solver_opts):
    """Robust PnP for one scene with outliers.

    pts_2d [M,2], pts_3d [M,3] (numpy or torch), K [3,3].  Returns dict with R [3,3], t [3],
    inliers [M] bool, n_inliers, status of the final solve, n_certified hypotheses.  refit_rounds: refits of the consensus set (each one
    assembly + solve + scoring launch; no host synchronisation inside a frame, whatever the count).
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
    res = pnp_batch(x4, X4, Kd, eps=eps, max_iters=max_iters, **solver_opts)  # (solver_opts: e.g. f32_sweeps_until=0, every sweep in float64)
    score = score_hypotheses(res.R, res.t, Kd, x, X, thresh, status=res.status, usable=(0, 2))
    # From here on everything stays on the device until the one read-back at the end, and (round 6) nothing of it is a torch kernel:
    # cvxpnpl_select_best takes the arg-max (lowest index on a tie), gathers the winner's pose and scores it for its inlier MASK; the refit
    # assembles straight from scene + mask (cvxpnpl_assemble_subsets -- the size of the set never leaves the device), solves at the cost
    # seam, and cvxpnpl_refit_update takes the refitted pose, with its own mask and count, when it is usable and keeps at least the
    # consensus it was fitted to.  (Round 5: ~40 small torch kernels around torch.argmax / torch.where, 0.35 ms of a 2.2 ms frame.)
    R, t, head, mask = select_best(score, res.R, res.t, res.status, Kd, x, X, thresh)
    if refit:
        for _ in range(max(1, int(refit_rounds))):  # refit on the consensus set (a second round re-fits the set the first one found)
            Bt, Qt, cnt = assemble_subsets(x, X, Kd, mask)
            fit = solve_cost_batch(Qt, Bt, eps=1e-9, max_iters=2500, device=device)
            refit_update(fit, cnt, Kd, x, X, thresh, R, t, head, mask)
    h = head.cpu()   # the frame's one synchronisation
    return {"R": R[0], "t": t[0], "inliers": mask[0].bool(), "n_inliers": int(h[1]), "status": int(h[0]), "n_certified": int(h[3]), "n_hyp": n_hyp,
            "best_index": int(h[2])}
