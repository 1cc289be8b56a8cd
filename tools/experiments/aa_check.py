"""Driver of tools/experiments/aa_hostsim.cpp: iteration histograms of the judged problem set with / without Anderson acceleration."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hostsim import Opts  # noqa: E402

from cvxpnpl_amd import synth  # noqa: E402

L = C.CDLL("/tmp/libaa.so")
L.aa_config.argtypes = [C.c_int, C.c_int, C.c_double]
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def run(d, n, m, frm, safe, **kw):
    o = Opts()
    L.aa_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    L.aa_config(m, frm, safe)
    B = len(d["pts_3d"])
    st = np.zeros(B, np.int32)
    it = np.zeros(B, np.int32)
    a = [np.ascontiguousarray(d[k], dtype=np.float64) for k in ("pts_2d", "pts_3d", "K")]
    L.aa_solve_batch(B, n, a[0].ctypes.data_as(dp), a[1].ctypes.data_as(dp), a[2].ctypes.data_as(dp), C.byref(o), st.ctypes.data_as(ip), it.ctypes.data_as(ip))
    return st, it


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sig = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    d = synth.make_pnpl(B, n, 0, sig, seed=42)
    for sched in ({}, {"first_check": 3, "check_every": 1}):
        for (m, frm, safe) in [(0, 0, 0), (1, 3, 10), (2, 3, 10), (2, 4, 10), (3, 3, 10), (2, 3, 3), (1, 5, 10), (2, 6, 10)]:
            st, it = run(d, n, m, frm, safe, **sched)
            print(sched, "m", m, "from", frm, "safe", safe, "cert", int((st == 0).sum()), "mean", round(it.mean(), 3), "p99", np.percentile(it, 99), "p99.9", np.percentile(it, 99.9),
                  "max", it.max(), "n>=10", int((it >= 10).sum()), "n>=8", int((it >= 8).sum()))
