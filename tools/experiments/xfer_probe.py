import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth
dev = torch.device("cuda:0")
for batch in (10000, 125000):
    d = synth.make_pnpl(batch, 10, 0, 2.0, seed=42)
    h = torch.from_numpy(np.concatenate([d["pts_2d"].ravel(), d["pts_3d"].ravel()])).pin_memory()
    dd = torch.empty_like(h, device=dev)
    out_d = torch.empty((batch, 13), dtype=torch.float64, device=dev); out_h = torch.empty((batch, 13), dtype=torch.float64).pin_memory()
    K = torch.as_tensor(d["K"], device=dev)
    p2 = dd[: batch * 20].view(batch, 10, 2); p3 = dd[batch * 20:].view(batch, 10, 3)
    def t(f, n=30):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(batch, "H2D only ms", round(t(lambda: dd.copy_(h, non_blocking=True)), 4), "D2H only", round(t(lambda: out_h.copy_(out_d, non_blocking=True)), 4),
          "solve only", round(t(lambda: ca.pnp_batch(p2, p3, K)), 4),
          "serial all", round(t(lambda: (dd.copy_(h, non_blocking=True), ca.pnp_batch(p2, p3, K), out_h.copy_(out_d, non_blocking=True))), 4), flush=True)
