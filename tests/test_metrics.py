"""Accuracy metrics / pose disambiguation of the benchmark toolkit (cvxpnpl_amd.metrics), CPU."""
import numpy as np

from cvxpnpl_amd import metrics, synth


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def test_pose_errors_known_values():
    rs = np.random.RandomState(0)
    R_gt, t_gt = synth.random_poses(rs, 5)
    angs = np.radians([0.0, 0.5, 3.0, 45.0, 179.0])
    R = np.stack([R_gt[i] @ _rot(rs.randn(3), angs[i]) for i in range(5)])
    t = t_gt * np.array([1.0, 1.01, 0.9, 2.0, 1.0])[:, None]
    ang, tr = metrics.pose_errors(R_gt, t_gt, R, t)
    np.testing.assert_allclose(ang, np.degrees(angs), atol=1e-6)  # acos near 0 loses half the digits
    np.testing.assert_allclose(tr, [0.0, 0.01, 0.1, 1.0, 0.0], atol=1e-12)
    assert np.allclose(ang, np.degrees(synth.geodesic(R, R_gt)), atol=1e-6)
    R[2] = np.nan
    ang, _ = metrics.pose_errors(R_gt, t_gt, R, t)
    assert np.isnan(ang[2]) and np.isfinite(np.delete(ang, 2)).all()


def test_disambiguate_picks_the_true_pose_among_candidates():
    rs = np.random.RandomState(1)
    B = 16
    R_gt, t_gt = synth.random_poses(rs, B)
    R_all = np.full((B, 4, 3, 3), np.nan)
    t_all = np.full((B, 4, 3), np.nan)
    n = rs.choice([0, 1, 2, 4], B)
    n[0], n[1] = 0, -1
    where = np.full(B, -1)
    for b in range(B):
        for c in range(max(n[b], 0)):
            R_all[b, c] = R_gt[b] @ _rot(rs.randn(3), rs.uniform(0.3, 2.0))
            t_all[b, c] = t_gt[b] * rs.uniform(0.5, 1.5)
        if n[b] > 0:
            where[b] = rs.randint(n[b])
            R_all[b, where[b]] = R_gt[b] @ _rot(rs.randn(3), 1e-4)
            t_all[b, where[b]] = t_gt[b] * (1 + 1e-5)
    # a decoy beyond n_poses must be ignored
    R_all[2, 3], t_all[2, 3] = R_gt[2], t_gt[2]
    if n[2] == 4:
        n[2] = 2
        where[2] = 0
        R_all[2, 0], t_all[2, 0] = R_gt[2] @ _rot([1, 0, 0], 1e-4), t_gt[2]
    R, t, idx = metrics.disambiguate(R_all, t_all, n, synth.K_KINECT, R_gt, t_gt)
    assert (idx == where).all()
    has = n > 0
    ang, tr = metrics.pose_errors(R_gt[has], t_gt[has], R[has], t[has])
    assert ang.max() < 0.01 and tr.max() < 1e-4
    assert np.isnan(R[~has]).all() and (idx[~has] == -1).all()
