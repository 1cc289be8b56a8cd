"""Driver of tools/experiments/retry_policy_hostsim.cpp"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hostsim import Opts  # noqa: E402

from cvxpnpl_amd import synth  # noqa: E402

L = C.CDLL("/tmp/librp.so")
L.rp_config.argtypes = [C.c_int, C.c_double]
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sig = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
d = synth.make_pnpl(B, n, 0, sig, seed=42)
a = [np.ascontiguousarray(d[k], dtype=np.float64) for k in ("pts_2d", "pts_3d", "K")]
for sched in ({}, {"first_check": 6, "check_every": 2}):
    for (pol, k, shift) in [(0, 0, 0.0), (0, 0, 0.015), (1, 0, 0.015), (4, 0, 0.015), (2, 1.0, 0.015), (2, 3.0, 0.015), (2, 10.0, 0.015), (2, 30.0, 0.015), (3, 3.0, 0.015), (3, 10.0, 0.015)]:
        o = Opts()
        L.rp_default_opts(C.byref(o))
        o.dual_shift = shift
        for kk, v in sched.items():
            setattr(o, kk, v)
        L.rp_config(pol, k)
        st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); stats = np.zeros(2, np.int64)
        L.rp_solve_batch(B, n, a[0].ctypes.data_as(dp), a[1].ctypes.data_as(dp), a[2].ctypes.data_as(dp), C.byref(o), st.ctypes.data_as(ip), it.ctypes.data_as(ip), stats.ctypes.data_as(C.POINTER(C.c_long)))
        print(sched, "policy", pol, "k", k, "shift", shift, "mean", round(it.mean(), 4), "p99.9", np.percentile(it, 99.9), "max", it.max(), "n>=8", int((it >= 8).sum()), "n>=12", int((it >= 12).sum()), "LDLs spent / passed", stats.tolist())
