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


def example_pnp():
    """The single problem of the reference's examples/pnp.py:5-26 (six points of the 0.6 cube from the legacy generator seeded with 42,
    integer toy intrinsics, a literal pose with 8 digits): the reference's own unit of timing is one such call
    (benchmarks/toolkit/suites/suite.py:75-85).  Returns pts_2d [6,2], pts_3d [6,3], K [3,3] (int), R_gt, t_gt."""
    rs = np.random.RandomState(42)
    X = LENGTH * (rs.random_sample((6, 3)) - 0.5)
    K = np.array([[160, 0, 320], [0, 120, 240], [0, 0, 1]])
    R = np.array([[-0.48048015, 0.1391384, -0.86589799], [-0.0333282, -0.98951829, -0.14050899], [-0.8763721, -0.03865296, 0.48008113]])
    t = np.array([-0.10266772, 0.25450789, 1.70391109])
    h = (X @ R.T + t) @ K.T
    return {"pts_2d": h[:, :2] / h[:, 2:], "pts_3d": X, "K": K, "R_gt": R, "t_gt": t}


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


# ----------------------------------------------------------------------------------------------------- on the device
def device_pnpl(batch, n_p, n_l, sigma=0.0, seed=42, K=K_KINECT, device=None):
    """The same problems generated ON THE DEVICE (cvxpnpl_synth_batch: one HIP launch, counter-based Philox4x32-10):
    no host generation, no H2D copy -- what accuracy sweeps over millions of problems use.  Returns a dict of float64
    device tensors with the keys of make_pnpl (pts_2d, pts_3d, line_2d, line_3d, R_gt, t_gt, K).  The distributions
    are the reference's (see the module docstring); the stream is Philox, restated in numpy by philox_pnpl below."""
    import ctypes as C

    import torch

    from . import _lib

    if not torch.cuda.is_available():
        raise RuntimeError("device_pnpl needs a ROCm GPU")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    Kd = torch.as_tensor(np.ascontiguousarray(K, dtype=np.float64), device=dev)
    mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)  # noqa: E731
    out = {"pts_2d": mk(batch, n_p, 2), "pts_3d": mk(batch, n_p, 3), "line_2d": mk(batch, n_l, 2, 2), "line_3d": mk(batch, n_l, 2, 3),
           "R_gt": mk(batch, 3, 3), "t_gt": mk(batch, 3), "K": Kd}
    p = lambda t: C.c_void_p(t.data_ptr()) if t.numel() else C.c_void_p(0)  # noqa: E731
    with torch.cuda.device(dev):
        rc = _lib.lib().cvxpnpl_synth_batch(batch, n_p, n_l, float(sigma), int(seed) & 0xFFFFFFFFFFFFFFFF, p(Kd), p(out["pts_2d"]), p(out["pts_3d"]),
                                            p(out["line_2d"]), p(out["line_3d"]), p(out["R_gt"]), p(out["t_gt"]),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_synth_batch failed ({rc}): {_lib.last_error()}")
    return out


def _philox4x32(c, k0, k1):
    """Philox4x32-10 on uint32 arrays c [..., 4]; keys scalars.  numpy restatement of synth_kernel.h's generator."""
    c = [np.asarray(c[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return c


def _u53(a, b):
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def philox_minimal_sets(n_hyp, n_corr, k=4, seed=0):
    """numpy restatement of cvxpnpl_sample_minimal_sets (score_kernel.h: partial Fisher-Yates on the Philox stream with counter
    (hypothesis, 0xFFFFFFFE, draw / 4)): idx [n_hyp, k] -- the checker of the device sampler."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    h = np.arange(n_hyp, dtype=np.uint64)
    idx = np.zeros((n_hyp, k), dtype=np.int64)
    perm = np.tile(np.arange(n_corr, dtype=np.int64), (n_hyp, 1))
    rows = np.arange(n_hyp)
    w = None
    for j in range(k):
        if j % 4 == 0:
            c = np.stack([h & np.uint64(0xFFFFFFFF), h >> np.uint64(32), np.full(n_hyp, 0xFFFFFFFE, np.uint64), np.full(n_hyp, j // 4, np.uint64)], axis=-1)
            w = _philox4x32(c, k0, k1)
        r = j + ((w[j % 4] * np.uint64(n_corr - j)) >> np.uint64(32)).astype(np.int64)
        idx[:, j] = perm[rows, r]
        perm[rows, r] = perm[rows, j]
    return idx


def philox_pnpl(batch, n_p, n_l, sigma=0.0, seed=42, K=K_KINECT):
    """numpy restatement of cvxpnpl_synth_batch (same counters, same arithmetic up to libm rounding): the checker of
    the device generator."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    b = np.arange(batch, dtype=np.uint64)

    def draw(rec, j, shape):
        c = np.zeros(shape + (4,), dtype=np.uint64)
        c[..., 0] = (b & np.uint64(0xFFFFFFFF)).reshape((-1,) + (1,) * (len(shape) - 1))
        c[..., 1] = (b >> np.uint64(32)).reshape((-1,) + (1,) * (len(shape) - 1))
        c[..., 2] = rec
        c[..., 3] = j
        return _philox4x32(c, k0, k1)

    w = [draw(0xFFFFFFFF, j, (batch,)) for j in range(4)]
    axis = np.stack([_u53(w[0][0], w[0][1]) - 0.5, _u53(w[0][2], w[0][3]) - 0.5, _u53(w[1][0], w[1][1]) - 0.5], 1)
    axis = axis * (1.0 / np.sqrt((axis * axis).sum(1)))[:, None]
    ang = 2.0 * np.pi * _u53(w[1][2], w[1][3])
    kx, ky, kz = axis.T
    zero = np.zeros(batch)
    Kx = np.stack([np.stack([zero, -kz, ky], 1), np.stack([kz, zero, -kx], 1), np.stack([-ky, kx, zero], 1)], 1)
    R = np.eye(3)[None] + np.sin(ang)[:, None, None] * Kx + (1.0 - np.cos(ang))[:, None, None] * (Kx @ Kx)
    t = np.stack([_u53(w[2][0], w[2][1]) - 0.5, _u53(w[2][2], w[2][3]) - 0.5, 1.6 * _u53(w[3][0], w[3][1]) + 0.6], 1)
    nrec = n_p + 2 * n_l
    rec = np.arange(nrec, dtype=np.uint64)[None, :].repeat(batch, 0)
    v = [draw(rec, j, (batch, nrec)) for j in range(3)]
    P = LENGTH * np.stack([_u53(v[0][0], v[0][1]) - 0.5, _u53(v[0][2], v[0][3]) - 0.5, _u53(v[1][0], v[1][1]) - 0.5], -1)
    x = project(P, np.asarray(K, dtype=np.float64), R, t)
    if sigma > 0:
        rad = np.sqrt(-2.0 * np.log(1.0 - _u53(v[1][2], v[1][3])))
        th = 2.0 * np.pi * _u53(v[2][0], v[2][1])
        x = x + sigma * np.stack([rad * np.cos(th), rad * np.sin(th)], -1)
    return {"pts_2d": np.ascontiguousarray(x[:, :n_p]), "pts_3d": np.ascontiguousarray(P[:, :n_p]),
            "line_2d": np.ascontiguousarray(x[:, n_p:].reshape(batch, n_l, 2, 2)),
            "line_3d": np.ascontiguousarray(P[:, n_p:].reshape(batch, n_l, 2, 3)), "R_gt": R, "t_gt": t, "K": np.array(K, dtype=np.float64)}
