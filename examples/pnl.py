"""One PnL problem through the drop-in: six 3D segments seen as six 2D segments (reference examples/pnl.py:28-40).

Lines are given by two points each, arrays laid out (line, endpoint, coordinate).
"""
import numpy as np

from _scene import K_TOY, POSE_B, cube_points, report, to_pixels

from cvxpnpl_amd import pnl

np.random.seed(42)
segments = cube_points(6, 2)
R_true, t_true = POSE_B
seen = to_pixels(segments.reshape(-1, 3), R_true, t_true).reshape(-1, 2, 2)
poses = pnl(line_2d=seen, line_3d=segments, K=K_TOY)
report(poses, R_true, t_true)
