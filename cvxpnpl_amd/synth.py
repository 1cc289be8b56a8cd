"""Synthetic absolute-pose problems, following the reference's benchmark generator.

Spec (reference benchmarks/toolkit/suites/synth.py:27-42, :49-55, :277-346 and
suite.py:17-19): Kinect intrinsics, rotation = Rodrigues(2 pi U * normalised(U(-.5,.5)^3)),
t = [U-.5, U-.5, 1.6 U + .6], 3D points 0.6 (U[0,1)^3 - .5), pixels = K (R X + t)
dehomogenised, plus N(0, sigma^2) pixel noise.  Lines are consecutive point pairs.
Vectorised over the batch with a seeded numpy RandomState (the legacy MT19937 stream the
reference seeds with np.random.seed) -- the distributions are the reference's, the draw
ORDER is batch-major rather than per-problem.
"""
import numpy as np

K_KINECT = np.array([[572.41140, 0.0, 325.26110], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]])
LENGTH = 0.6


def random_poses(rs, batch):
    axis = rs.random_sample((batch, 3)) - 0.5
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = 2.0 * np.pi * rs.random_sample(batch)
    kx, ky, kz = axis.T
    zero = np.zeros(batch)
    Kx = np.stack([np.stack([zero, -kz, ky], 1), np.stack([kz, zero, -kx], 1), np.stack([-ky, kx, zero], 1)], 1)
    s, c = np.sin(ang)[:, None, None], np.cos(ang)[:, None, None]
    R = np.eye(3)[None] + s * Kx + (1.0 - c) * (Kx @ Kx)
    t = np.concatenate([rs.random_sample((batch, 2)) - 0.5, 1.6 * rs.random_sample((batch, 1)) + 0.6], 1)
    return R, t


def project(P, K, R, t):
    """P [B,n,3] -> pixels [B,n,2]  (suite.py:17-19)."""
    x = (P @ np.swapaxes(R, 1, 2) + t[:, None, :]) @ K.T
    return x[..., :2] / x[..., 2:3]


def make_pnpl(batch, n_p, n_l, sigma=0.0, seed=42, K=K_KINECT):
    """Returns dict with pts_2d [B,n_p,2], pts_3d [B,n_p,3], line_2d [B,n_l,2,2], line_3d [B,n_l,2,3], R_gt, t_gt, K."""
    rs = np.random.RandomState(seed)
    R, t = random_poses(rs, batch)
    P = LENGTH * (rs.random_sample((batch, n_p + 2 * n_l, 3)) - 0.5)
    x = project(P, K, R, t)
    if sigma > 0:
        x = x + rs.normal(scale=sigma, size=x.shape)
    return {
        "pts_2d": np.ascontiguousarray(x[:, :n_p]), "pts_3d": np.ascontiguousarray(P[:, :n_p]),
        "line_2d": np.ascontiguousarray(x[:, n_p:].reshape(batch, n_l, 2, 2)),
        "line_3d": np.ascontiguousarray(P[:, n_p:].reshape(batch, n_l, 2, 3)),
        "R_gt": R, "t_gt": t, "K": np.array(K, dtype=np.float64),
    }


def make_pnp(batch, n, sigma=0.0, seed=42, K=K_KINECT):
    return make_pnpl(batch, n, 0, sigma, seed, K)


def make_ransac(n_hyp, n_corr=100, outlier_frac=0.3, sigma=0.0, seed=46, K=K_KINECT, width=640, height=480):
    """BASELINE config 5: one scene of n_corr correspondences, a fraction of the 2D points
    replaced by uniform clutter, n_hyp random minimal (N=4) subsets."""
    rs = np.random.RandomState(seed)
    R, t = random_poses(rs, 1)
    P = LENGTH * (rs.random_sample((1, n_corr, 3)) - 0.5)
    x = project(P, K, R, t)[0]
    if sigma > 0:
        x = x + rs.normal(scale=sigma, size=x.shape)
    n_out = int(round(outlier_frac * n_corr))
    out_idx = rs.choice(n_corr, n_out, replace=False)
    x[out_idx] = rs.random_sample((n_out, 2)) * np.array([width, height])
    inlier = np.ones(n_corr, bool)
    inlier[out_idx] = False
    idx = np.stack([rs.choice(n_corr, 4, replace=False) for _ in range(n_hyp)])
    return {"pts_2d": np.ascontiguousarray(x[idx]), "pts_3d": np.ascontiguousarray(P[0][idx]), "idx": idx,
            "scene_2d": x, "scene_3d": P[0], "inlier": inlier, "R_gt": R[0], "t_gt": t[0], "K": np.array(K, dtype=np.float64)}


def geodesic(Ra, Rb):
    """Batched rotation angle of Ra^T Rb, accurate at tiny angles."""
    D = np.swapaxes(Ra, -1, -2) @ Rb
    s = 0.5 * np.stack([D[..., 2, 1] - D[..., 1, 2], D[..., 0, 2] - D[..., 2, 0], D[..., 1, 0] - D[..., 0, 1]], -1)
    return np.arctan2(np.linalg.norm(s, axis=-1), 0.5 * (np.trace(D, axis1=-2, axis2=-1) - 1.0))


def make_planar_pnp(batch, n_p, sigma=0.0, seed=42, general=True, K=K_KINECT):
    """PnP problems whose 3D points lie in one plane: Z = 0 in the world frame (general=False) or a random
    plane per problem (general=True: the Z = 0 scene moved by a random rigid transform).  Same dict as make_pnp."""
    d = make_pnp(batch, n_p, 0.0, seed=seed, K=K)
    d["pts_3d"][:, :, 2] = 0.0
    rs = np.random.RandomState(seed + 1)
    if general:
        G, _ = np.linalg.qr(rs.normal(size=(batch, 3, 3)))
        G[np.linalg.det(G) < 0, :, 0] *= -1
        c = rs.normal(size=(batch, 3))
        P = np.einsum("bij,bnj->bni", G, d["pts_3d"]) + c[:, None, :]
        R = np.einsum("bij,bkj->bik", d["R_gt"], G)  # R_gt G^T
        t = d["t_gt"] - np.einsum("bij,bj->bi", R, c)
        d["pts_3d"], d["R_gt"], d["t_gt"] = P, R, t
    d["pts_2d"] = project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"])
    if sigma > 0:
        d["pts_2d"] = d["pts_2d"] + rs.normal(scale=sigma, size=d["pts_2d"].shape)
    return d
