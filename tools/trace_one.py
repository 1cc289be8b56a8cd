import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth
d = synth.make_pnpl(3000, 5, 5, 1.0, seed=31)
i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 1
li = int(sys.argv[3]) if len(sys.argv) > 3 else 0
args = [d[k][i:i+1] for k in ("pts_2d", "line_2d", "pts_3d", "line_3d")] + [d["K"]]
r = ca.pnpl_batch(*args, layout=layout, lane_iters=li)
torch.cuda.synchronize()
print("layout", layout, "li", li, "iters", r.iters.item(), "status", r.status.item(), "sweeps", r.work[0,1].item())
