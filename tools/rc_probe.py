import sys, numpy as np, torch
sys.path.insert(0, '.')
import cvxpnpl_amd as ca
from cvxpnpl_amd import synth
dev = torch.device('cuda:0')
d = synth.make_pnp(50000, 10, 2.0, seed=42)
Bt, Qt = ca.assemble_batch(torch.as_tensor(d['pts_2d'], device=dev), None, torch.as_tensor(d['pts_3d'], device=dev), None, d['K'])
for rf in (96, 0):
    r = ca.solve_cost_batch(Qt, Bt, variant=1, layout=2, rescue_from=rf, want_Z=True)
    it = r.iters.cpu().numpy(); st = r.status.cpu().numpy()
    print('rescue_from', rf, 'status', np.bincount(st, minlength=5), 'iters pct', np.percentile(it, [50, 90, 99, 99.9, 99.99, 100]))
    idx = np.argsort(-it)[:12]
    print([(int(i), int(it[i]), int(st[i])) for i in idx])
    if rf == 96:
        slow = idx[:6]
        Z = r.Z.cpu().numpy()
        for i in slow:
            M = np.zeros((10, 10)); k = 0
            for a in range(10):
                for b in range(a, 10):
                    M[a, b] = M[b, a] = Z[i, k]; k += 1
            print(int(i), 'eig(Z)', np.round(np.linalg.eigvalsh(M)[-4:], 4), 'cost', r.cost[i].cpu().numpy())
