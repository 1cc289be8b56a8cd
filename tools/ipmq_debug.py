import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from test_ipm_quad import _costs
from hostsim import ipm_solve
import cvxpnpl_amd as ca
Q45, Qs = _costs(1001, 4, 2.0, 100 + 1001)
for variant in (0, 1):
    base = None
    for shift in (0,):
        Qr = np.roll(Q45, shift, axis=0)
        Z, S, gap, it = [x.cpu().numpy() for x in ca.ipm_batch(torch.as_tensor(Qr, device="cuda"), variant=variant)]
        gap = np.roll(gap, -shift); it = np.roll(it, -shift); why = it >> 8; it = it & 255
        bad = np.flatnonzero(gap > 1e-6)
        print("variant", variant, "shift", shift, "bad problems", bad, "their groups", (bad + shift) % 4, "iters", it[bad], "why", why[bad], "gaps", gap[bad], "why hist", np.bincount(why, minlength=6))
    # a batch made of one bad problem only, in every position
    Zh, Sh, gaph, ith = ipm_solve(Qs, variant)
    b0 = 575 if variant == 0 else 509
    Qo = np.repeat(Q45[b0:b0 + 1], 8, axis=0)
    Z, S, gap, it = [x.cpu().numpy() for x in ca.ipm_batch(torch.as_tensor(Qo, device="cuda"), variant=variant)]
    print("   problem", b0, "alone x8: iters", it & 255, "why", it >> 8, "gaps", gap, "host", ith[b0], gaph[b0])
