"""Shared set-up of the example callers: the toy camera and the two literal poses the reference's examples use
(reference examples/pnp.py:12-26, pnl.py:15-26, pnpl.py:16-27 -- 8-digit literals, i.e. data), and a projection helper.

Every example builds its scene from numpy's legacy global generator seeded with 42, like the reference's scripts do, so the
numbers printed here can be laid beside theirs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

K_TOY = np.array([[160, 0, 320], [0, 120, 240], [0, 0, 1]])  # integer intrinsics on purpose: the drop-in must accept them

POSE_A = (np.array([[-0.48048015, 0.1391384, -0.86589799],
                    [-0.0333282, -0.98951829, -0.14050899],
                    [-0.8763721, -0.03865296, 0.48008113]]),
          np.array([-0.10266772, 0.25450789, 1.70391109]))
POSE_B = (np.array([[0.89802142, -0.41500101, 0.14605372],
                    [0.24509948, 0.7476071, 0.61725997],
                    [-0.36535431, -0.51851499, 0.77308372]]),
          np.array([-0.0767557, 0.13917375, 1.9708239]))


def cube_points(*shape):
    """points of the cube of side 0.6 about the origin, shape (..., 3)"""
    return 0.6 * (np.random.random(shape + (3,)) - 0.5)


def to_pixels(X, R, t, K=K_TOY):
    """world points (n, 3) -> pixels (n, 2) of the camera x = K (R X + t)"""
    h = (X @ R.T + t) @ K.T
    return h[:, :2] / h[:, 2:]


def rotation_gap(R, R_true):
    """geodesic distance on SO(3), radians (the atan2 form: the arccos of a trace cannot resolve angles below ~1e-8, and the
    8-digit literals are orthogonal to ~1e-8 only)"""
    D = R_true.T @ R
    s = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    return float(np.arctan2(np.linalg.norm(s), 0.5 * (np.trace(D) - 1.0)))


def report(poses, R_true, t_true, tol=1e-6):
    """prints what the reference's examples print and checks the known answer (the literals carry 8 digits)"""
    R, t = poses[0]
    print("Nr of possible poses:", len(poses))
    print("R (ground truth):", R_true, "R (estimate):", R, sep="\n")
    print("t (ground truth):", t_true)
    print("t (estimate):", t)
    gap, dt = rotation_gap(R, R_true), float(np.linalg.norm(t - t_true) / np.linalg.norm(t_true))
    print(f"rotation off by {gap:.2e} rad, translation by {dt:.2e} (relative)")
    assert len(poses) == 1 and gap < tol and dt < tol, (len(poses), gap, dt)
