"""One PnP problem through the drop-in: six points, noise-free (call pattern of the reference's examples/pnp.py:30-41).

    python examples/pnp.py        (needs the GPU: cvxpnpl_amd has no CPU fallback)
"""
import numpy as np

from _scene import K_TOY, POSE_A, cube_points, report, to_pixels

from cvxpnpl_amd import pnp

np.random.seed(42)
X = cube_points(6)
R_true, t_true = POSE_A
poses = pnp(pts_2d=to_pixels(X, R_true, t_true), pts_3d=X, K=K_TOY)  # over-determined: one pose comes back
report(poses, R_true, t_true)
