import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth
d = synth.make_pnpl(3000, 5, 5, 1.0, seed=31)
args = (d["pts_2d"], d["line_2d"], d["pts_3d"], d["line_3d"], d["K"])
ref = ca.pnpl_batch(*args, layout=2)
out = []
for layout, li in ((1, 3), (1, 10), (1, 0), (3, 6)):
    r = ca.pnpl_batch(*args, layout=layout, lane_iters=li)
    out.append("L%d/li%d d_it %.3f" % (layout, li, (r.iters - ref.iters).abs().double().mean().item()))
big = synth.make_pnp(125000, 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(big[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
for _ in range(3): ca.pnp_batch(p2, p3, K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ca.pnp_batch(p2, p3, K)
torch.cuda.synchronize()
print(" | ".join(out), "| 125k: %.1fM/s" % (125000 * 20 / (time.perf_counter() - t0) / 1e6))
