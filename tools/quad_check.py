"""Quick agreement check of the quad layout against the wave layout (GPU box)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth

for name, d in (("pnp10", synth.make_pnp(10000, 10, 2.0, seed=42)), ("pnp4", synth.make_pnp(3000, 4, 1.0, seed=7)),
                ("pnpl", synth.make_pnpl(5001, 5, 5, 2.0, seed=3))):
    args = (d.get("pts_2d"), d.get("line_2d"), d.get("pts_3d"), d.get("line_3d"), d["K"])
    ref = ca.pnpl_batch(*args, layout=2, want_Z=True)
    for li in (6, 3, 4):
        out = ca.pnpl_batch(*args, layout=3, lane_iters=li, want_Z=True)
        torch.cuda.synchronize()
        st_r, st_o = ref.status.cpu().numpy(), out.status.cpu().numpy()
        ok = (st_r == 0) & (st_o == 0)
        dR = (ref.R - out.R).abs().flatten(1).max(1).values.cpu().numpy()
        dt = (ref.t - out.t).abs().max(1).values.cpu().numpy()
        dc = (ref.cost[:, 0] - out.cost[:, 0]).abs().cpu().numpy()
        dZ = (ref.Z - out.Z).abs().max(1).values.cpu().numpy()
        print(name, "lane_iters", li, "status wave", np.bincount(st_r, minlength=5), "quad", np.bincount(st_o, minlength=5),
              "iters mean %.3f vs %.3f" % (ref.iters.double().mean(), out.iters.double().mean()),
              "max dR %.2e dt %.2e dcost %.2e dZ %.2e" % (dR[ok].max(), dt[ok].max(), dc[ok].max(), dZ[ok].max()),
              "status mismatches", int((st_r != st_o).sum()))
