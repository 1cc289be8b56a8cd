import ctypes as C, os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from hostsim import Opts
from cvxpnpl_amd import synth
L = C.CDLL("/tmp/exp/libdump.so")
dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
def run(d, n_p, n_l=0, **kw):
    o = Opts(); L.dr_default_opts(C.byref(o))
    for k, v in kw.items(): setattr(o, k, v)
    B = len(d["pts_3d"])
    st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); R = np.zeros((B, 9))
    a = [np.ascontiguousarray(d[k], dtype=np.float64) for k in ("pts_2d", "pts_3d", "line_2d", "line_3d", "K")]
    P = lambda x: x.ctypes.data_as(dp)
    n = L.dr_solve_batch(B, n_p, P(a[0]), P(a[1]), n_l, P(a[2]) if n_l else None, P(a[3]) if n_l else None, P(a[4]), C.byref(o), st.ctypes.data_as(ip), it.ctypes.data_as(ip), P(R))
    out = np.zeros((n, 223)); L.dr_get(P(out))
    return st, it, R, out
if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
    d = synth.make_pnpl(B, n, 0, 2.0, seed=seed)
    st, it, R, out = run(d, n)
    print("cert", (st == 0).sum(), "mean it", it.mean(), "max", it.max(), "hist", np.bincount(it)[:16], "records", len(out), "failed", int((out[:, 2] == 0).sum()))
    np.savez(f"/tmp/exp/dump_{B}_{n}_{seed}.npz", st=st, it=it, R=R, out=out)
