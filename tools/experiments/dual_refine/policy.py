import numpy as np, sys, ctypes as C
from collect import L, run
sys.path.insert(0, "/root/repo")
from cvxpnpl_amd import synth
L.dr_policy.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
L.dr_tried.restype = C.c_long; L.dr_ok.restype = C.c_long
def go(d, n_p, n_l, pol, **kw):
    L.dr_policy(*pol)
    st, it, R, out = run(d, n_p, n_l, **kw)
    return st, it, R, L.dr_tried(), L.dr_ok()
if __name__ == "__main__":
    wl = sys.argv[1]
    B = int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
    if wl == "pnp10": d = synth.make_pnpl(B, 10, 0, 2.0, seed=seed); n_p, n_l = 10, 0
    elif wl == "pnpl": d = synth.make_pnpl(B, 5, 5, 2.0, seed=seed); n_p, n_l = 5, 5
    elif wl == "pnp6": d = synth.make_pnpl(B, 6, 0, 2.0, seed=seed); n_p, n_l = 6, 0
    elif wl == "pnp4": d = synth.make_pnpl(B, 4, 0, 2.0, seed=seed); n_p, n_l = 4, 0
    elif wl == "pnp8": d = synth.make_pnpl(B, 8, 0, 2.0, seed=seed); n_p, n_l = 8, 0
    ref = None
    pols = [("baseline (D-shift from 2nd attempt)", (0, 0.005, 2, 2.0, 0, 1, 1)),
            ("D then grad, from 2nd attempt", (1, 0.005, 2, 2.0, 0, 1, 1)),
            ("grad only, from 2nd attempt", (1, 0.005, 2, 2.0, 0, 0, 1)),
            ("grad only, every attempt", (1, 0.005, 2, 2.0, 1, 0, 1)),
            ("grad only 2 steps, every attempt", (1, 0.005, 2, 2.0, 1, 0, 2)),
            ("grad sigma .01, every attempt", (1, 0.01, 2, 2.0, 1, 0, 1)),
            ("grad nit 3, every attempt", (1, 0.005, 3, 2.0, 1, 0, 1)),
            ]
    for name, pol in pols:
        st, it, R, tried, okc = go(d, n_p, n_l, pol)
        if ref is None: ref = (st, R)
        both = (st == 0) & (ref[0] == 0)
        dR = np.abs(R - ref[1])[both].max()
        h = np.bincount(it, minlength=20)
        print(f"{name:40s} cert {int((st==0).sum())} mean {it.mean():.3f} p99.9 {np.percentile(it,99.9):.0f} max {it.max()} n>=7 {int((it>=7).sum())} n>=9 {int((it>=9).sum())} n>=12 {int((it>=12).sum())} refine ok/tried {okc}/{tried} max|dR| {dR:.1e}")
