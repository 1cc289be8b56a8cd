"""Per-iteration latency of a lone wavefront: the slowest problems of a batch, solved alone."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth
d = synth.make_pnp(125000, 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K"))
r = ca.pnp_batch(p2, p3, K, layout=2)
it = r.iters.cpu().numpy()
order = np.argsort(-it)[:8]
print("slowest:", it[order].tolist())
for idx in order[:3]:
    q2, q3 = p2[idx:idx + 1].repeat(64, 1, 1).contiguous(), p3[idx:idx + 1].repeat(64, 1, 1).contiguous()
    for layout, kw in ((2, {}), (3, {"lane_iters": 6}), (1, {"lane_iters": 5})):
        for _ in range(3): rr = ca.pnp_batch(q2, q3, K, layout=layout, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): rr = ca.pnp_batch(q2, q3, K, layout=layout, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        n = int(rr.iters[0])
        print("problem %d layout %d: %d iterations, %.1f us per launch, %.2f us per iteration, sweeps %d" % (idx, layout, n, dt * 1e6, dt * 1e6 / n, int(rr.work[0, 1])))
