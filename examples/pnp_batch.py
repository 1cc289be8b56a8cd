"""What the library is for: ten thousand independent PnP problems in one launch (BASELINE config 2).

Inputs and outputs stay on the device; the statuses say which poses carry a certificate of global optimality of the relaxation
(`0 <= cost - dobj <= eps`, the statement cvxpnpl.py:516-519 checks).
"""
import numpy as np
import torch

import _scene  # noqa: F401  (puts the repository root on sys.path)
from cvxpnpl_amd import pnp_batch, synth

dev = torch.device("cuda:0")
d = synth.make_pnp(10_000, 10, sigma=0.0, seed=42)  # the reference's generator (Kinect intrinsics, benchmarks/toolkit/suites/synth.py)
res = pnp_batch(torch.as_tensor(d["pts_2d"], device=dev), torch.as_tensor(d["pts_3d"], device=dev), torch.as_tensor(d["K"], device=dev))
torch.cuda.synchronize()
status = res.status.cpu().numpy()
gap = synth.geodesic(res.R.cpu().numpy(), d["R_gt"])
dt = np.linalg.norm(res.t.cpu().numpy() - d["t_gt"], axis=1) / np.linalg.norm(d["t_gt"], axis=1)
print(f"{len(status)} problems, {(status == 0).sum()} certified, mean iterations {res.iters.float().mean().item():.2f}")
print(f"noise-free: worst rotation error {gap[status == 0].max():.2e} rad, worst relative translation error {dt[status == 0].max():.2e}")
assert (status == 0).all() and gap.max() < 1e-6 and dt.max() < 1e-6
