"""Points and lines together: four of each (reference examples/pnpl.py:29-47)."""
import numpy as np

from _scene import K_TOY, POSE_B, cube_points, report, to_pixels

from cvxpnpl_amd import pnpl

np.random.seed(42)
X = cube_points(4)
segments = cube_points(4, 2)
R_true, t_true = POSE_B
px = to_pixels(np.vstack((X, segments.reshape(-1, 3))), R_true, t_true)
poses = pnpl(pts_2d=px[:4], line_2d=px[4:].reshape(-1, 2, 2), pts_3d=X, line_3d=segments, K=K_TOY)
report(poses, R_true, t_true)
