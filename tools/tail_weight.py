"""How much of a launch is the straggler tail?  Times the default batch, then the same number of
problems drawn only from those that finish within N iterations."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth

def timed(p2, p3, K, layout, reps=30):
    for _ in range(3): ca.pnp_batch(p2, p3, K, layout=layout)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = ca.pnp_batch(p2, p3, K, layout=layout)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r

d = synth.make_pnp(40000, 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
_, r = timed(p2, p3, K, 2, reps=1)
it = r.iters.cpu().numpy()
for layout in (2,):
    ms, _ = timed(p2[:10000], p3[:10000], K, layout)
    print("layout", layout, "first 10k as is: %.3f ms, max iters %d" % (ms, it[:10000].max()))
    for cap in (12, 8, 6, 5, 4, 3):
        idx = np.where(it <= cap)[0][:10000]
        if len(idx) < 10000: idx = np.resize(idx, 10000)
        ii = torch.as_tensor(idx, device="cuda")
        ms, rr = timed(p2[ii].contiguous(), p3[ii].contiguous(), K, layout)
        print("  only problems with iters <= %2d: %.3f ms (mean iters %.2f)" % (cap, ms, rr.iters.double().mean()))
