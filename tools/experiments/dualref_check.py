"""Driver of tools/experiments/dualref_hostsim.cpp: iteration histograms of the judged problem set with dual-only refinement cycles."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hostsim import Opts  # noqa: E402

from cvxpnpl_amd import synth  # noqa: E402

L = C.CDLL("/tmp/libdr.so")
L.dr_config.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int]
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def run(d, n, cycles, over, margin, frm=0, **kw):
    o = Opts()
    L.dr_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    L.dr_config(cycles, over, margin, frm)
    B = len(d["pts_3d"])
    st = np.zeros(B, np.int32)
    it = np.zeros(B, np.int32)
    R = np.zeros((B, 9))
    stats = np.zeros(2, np.int64)
    a = [np.ascontiguousarray(d[k], dtype=np.float64) for k in ("pts_2d", "pts_3d", "K")]
    L.dr_solve_batch(B, n, a[0].ctypes.data_as(dp), a[1].ctypes.data_as(dp), a[2].ctypes.data_as(dp), C.byref(o), st.ctypes.data_as(ip), it.ctypes.data_as(ip),
                     R.ctypes.data_as(dp), stats.ctypes.data_as(C.POINTER(C.c_long)))
    return st, it, R, stats


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sig = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    d = synth.make_pnpl(B, n, 0, sig, seed=42)
    ref = None
    for sched in ({}, {"first_check": 4, "check_every": 1}):
        for (cyc, over, margin, frm) in [(0, 1.0, 0.0, 0), (1, -1.0, 0.02, 0), (1, -1.0, 0.015, 0), (1, -1.0, 0.01, 0), (1, -1.0, -2, 0), (1, -1.0, -3, 0), (1, 1.0, 0.02, 0)]:
            st, it, R, stats = run(d, n, cyc, over, margin, frm, **sched)
            if ref is None:
                ref = R
            dR = np.abs(R - ref).max()
            print(sched, "cycles", cyc, "over", over, "margin", margin, "from", frm, "cert", int((st == 0).sum()), "mean", round(it.mean(), 3), "p99", np.percentile(it, 99), "p99.9", np.percentile(it, 99.9),
                  "max", it.max(), "n>=8", int((it >= 8).sum()), "n>=12", int((it >= 12).sum()), "cycles used/certifying", stats.tolist(), "max|dR| vs baseline", f"{dR:.1e}")
