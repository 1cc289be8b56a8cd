"""Accuracy metrics of the reference's benchmark toolkit, batched.

* pose_errors      angular error [deg] and relative translation error, the two numbers of every accuracy plot of
                   the reference (benchmarks/toolkit/suites/suite.py:22-34: angle of R_gt^-1 R, |t - t_gt| / |t_gt|);
* disambiguate     which of up to four returned poses a benchmark scores (suite.py:90-110: the pose whose
                   reprojection of a few support points is closest to the ground truth's).

numpy on the host: these run once per experiment on the gathered results, not on the hot path.
"""
from typing import Tuple

import numpy as np


def rotation_angle(R: np.ndarray) -> np.ndarray:
    """Rotation angle [rad] of [...,3,3] matrices, after projecting them onto SO(3)/O(3) (suite.py:8-14)."""
    R = np.asarray(R, dtype=np.float64)
    U, _, Vh = np.linalg.svd(R)
    Q = U @ Vh
    c = 0.5 * (np.trace(Q, axis1=-2, axis2=-1) - 1.0)
    return np.arccos(np.clip(c, -1.0, 1.0))


def pose_errors(R_gt: np.ndarray, t_gt: np.ndarray, R: np.ndarray, t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(angular error [deg], relative translation error) of estimates R [B,3,3], t [B,3] against the ground
    truth (suite.py:22-34).  NaN estimates give NaN errors."""
    R_gt, R = np.asarray(R_gt, dtype=np.float64), np.asarray(R, dtype=np.float64)
    t_gt, t = np.asarray(t_gt, dtype=np.float64), np.asarray(t, dtype=np.float64)
    bad = ~np.isfinite(R).all(axis=(-2, -1))
    Rs = np.where(bad[..., None, None], np.eye(3), R)
    rel = np.linalg.solve(R_gt, Rs)  # R_gt^-1 R, as the reference does (not R_gt^T R: R_gt is taken as given)
    ang = np.degrees(rotation_angle(rel))
    ang = np.where(bad, np.nan, ang)
    trans = np.linalg.norm(t - t_gt, axis=-1) / np.linalg.norm(t_gt, axis=-1)
    return ang, trans


def _project(P, K, R, t):
    X = np.einsum("...ij,nj->...ni", R, P) + t[..., None, :]
    uvw = np.einsum("ij,...nj->...ni", K, X)
    return uvw[..., :2] / uvw[..., 2:3]


def disambiguate(R_all: np.ndarray, t_all: np.ndarray, n_poses: np.ndarray, K: np.ndarray, R_gt: np.ndarray, t_gt: np.ndarray,
                 n_support: int = 20, seed: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Pick, per problem, the candidate the reference's harness would score (suite.py:90-110).

    R_all [B,4,3,3], t_all [B,4,3], n_poses [B] (outputs of recover_multi_batch; entries beyond n_poses are
    ignored), K [3,3], ground truth R_gt [B,3,3], t_gt [B,3].  Returns (R [B,3,3], t [B,3], index [B]); problems
    with n_poses <= 0 get NaN and index -1."""
    R_all, t_all = np.asarray(R_all, dtype=np.float64), np.asarray(t_all, dtype=np.float64)
    n_poses = np.asarray(n_poses)
    B, C = R_all.shape[0], R_all.shape[1]
    S = np.random.RandomState(seed).random_sample((n_support, 3)) - 0.5
    ref = _project(S, K, np.asarray(R_gt, dtype=np.float64), np.asarray(t_gt, dtype=np.float64))  # [B,n,2]
    with np.errstate(all="ignore"):
        est = _project(S, K, R_all, t_all)  # [B,C,n,2]
        err = np.linalg.norm(est - ref[:, None], axis=-1).sum(-1)  # [B,C]
    valid = (np.arange(C)[None, :] < np.maximum(n_poses, 0)[:, None]) & np.isfinite(err)
    err = np.where(valid, err, np.inf)
    idx = np.argmin(err, axis=1)
    none = ~valid.any(axis=1)
    R = R_all[np.arange(B), idx].copy()
    t = t_all[np.arange(B), idx].copy()
    R[none], t[none] = np.nan, np.nan
    return R, t, np.where(none, -1, idx)


# ----------------------------------------------------------------------------------------------------- on the device
def _dev_f64(x, device):
    import torch

    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64))
    return x.to(device=device, dtype=torch.float64).contiguous()


def pose_errors_device(R_gt, t_gt, R, t):
    """pose_errors on the device (cvxpnpl_pose_errors, one HIP launch): (angular error [deg], relative translation error)
    as float64 device tensors [B].  Inputs: device tensors (or anything torch.as_tensor takes)."""
    import ctypes as C

    import torch

    from . import _lib

    dev = R.device if isinstance(R, torch.Tensor) and R.is_cuda else torch.device("cuda", torch.cuda.current_device())
    Rg, tg, Rd, td = (_dev_f64(x, dev) for x in (R_gt, t_gt, R, t))
    B = Rd.shape[0]
    ang = torch.empty((B,), dtype=torch.float64, device=dev)
    trans = torch.empty((B,), dtype=torch.float64, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    with torch.cuda.device(dev):
        rc = _lib.lib().cvxpnpl_pose_errors(B, p(Rg), p(tg), p(Rd), p(td), p(ang), p(trans), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_pose_errors failed ({rc}): {_lib.last_error()}")
    return ang, trans


def disambiguate_device(R_all, t_all, n_poses, K, R_gt, t_gt, n_support: int = 20, seed: int = 0):
    """disambiguate on the device (cvxpnpl_disambiguate): the same support points as the numpy version (same seed), the
    same choice.  Returns device tensors (R [B,3,3], t [B,3], index [B] int32)."""
    import ctypes as C

    import torch

    from . import _lib

    dev = R_gt.device if isinstance(R_gt, torch.Tensor) and R_gt.is_cuda else torch.device("cuda", torch.cuda.current_device())
    Ra, ta, Kd, Rg, tg = (_dev_f64(x, dev) for x in (R_all, t_all, K, R_gt, t_gt))
    npz = torch.as_tensor(np.asarray(n_poses) if not isinstance(n_poses, torch.Tensor) else n_poses).to(device=dev, dtype=torch.int32).contiguous()
    S = torch.as_tensor(np.random.RandomState(seed).random_sample((n_support, 3)) - 0.5, device=dev)
    B = Ra.shape[0]
    R = torch.empty((B, 3, 3), dtype=torch.float64, device=dev)
    t = torch.empty((B, 3), dtype=torch.float64, device=dev)
    idx = torch.empty((B,), dtype=torch.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    with torch.cuda.device(dev):
        rc = _lib.lib().cvxpnpl_disambiguate(B, p(Ra), p(ta), p(npz), p(Kd), p(Rg), p(tg), p(S), n_support, p(R), p(t), p(idx),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_disambiguate failed ({rc}): {_lib.last_error()}")
    return R, t, idx
