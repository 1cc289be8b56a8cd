#!/usr/bin/env python3
"""tools/microbench/consumer_probe.hip driver (GPU box):  python tools/microbench/consumer_probe.py [batch] [consumer grid]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxpnpl_amd import synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
cgrid = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
d = synth.make_pnpl(batch, 10, 0, 2.0, seed=42)
tt = lambda x: torch.as_tensor(x, device=dev).contiguous()  # noqa: E731
p2, p3, K = tt(d["pts_2d"]), tt(d["pts_3d"]), tt(d["K"])
ptr = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "consumer_probe.so"))
grid = (batch + 63) // 64
z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device=dev)  # noqa: E731
R, t, cost = z(batch, 9), z(batch, 3), z(batch, 2)
st, it = z(batch, dt=torch.int32), z(batch, dt=torch.int32)
work = z(batch, 2, dt=torch.int32)
ctr, ent, ws = z(128, dt=torch.int32), z(batch + 64, dt=torch.int32), z(batch, 56)
tend, t0, t1, ns = z(grid, dt=torch.int64), z(cgrid, dt=torch.int64), z(cgrid, dt=torch.int64), z(cgrid, dt=torch.int32)
L.consumer_probe_run.argtypes = [C.c_int64, C.c_int] + [C.c_void_p] * 16 + [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
ref = None
for mode, name in ((0, "lane kernel, then consumers (one stream)"), (1, "consumers on a second stream, gated behind the lane kernel's last dispatch"), (0, "one stream again")):
    st.fill_(-7)
    ms = C.c_float()
    rc = L.consumer_probe_run(batch, 10, ptr(p2), ptr(p3), ptr(K), ptr(R), ptr(t), ptr(st), ptr(it), ptr(cost), ptr(work), ptr(ctr), ptr(ent), ptr(ws), ptr(tend),
                              ptr(t0), ptr(t1), ptr(ns), cgrid, mode, 10, C.byref(ms))
    torch.cuda.synchronize()
    s_ = st.cpu().numpy()
    c = ctr.cpu().numpy()
    a1 = tend.cpu().numpy().astype(np.float64) / 100
    b0 = t0.cpu().numpy().astype(np.float64) / 100
    b1 = t1.cpu().numpy().astype(np.float64) / 100
    n = ns.cpu().numpy()
    act = b0 > 0
    base = a1.min()
    line = (f"{name}: rc {rc}  {ms.value * 1e3:.1f} us per step | status hist {np.bincount(s_ + 7, minlength=12)[7:12].tolist()} unfinished {(s_ == -7).sum()} | pushed {c[32]} claimed {c[64]} "
            f"| lane blocks end: first {0:.0f} last {a1.max() - base:.0f} us | consumers that solved: {act.sum()} (max {n.max()} each), first claim at {b0[act].min() - base if act.any() else -1:.0f}, "
            f"median {np.median(b0[act]) - base if act.any() else -1:.0f}, last exit {b1.max() - base:.0f}")
    if ref is None:
        ref = (s_.copy(), R.cpu().numpy().copy())
    else:
        line += f" | statuses equal {np.mean(s_ == ref[0]):.6f}, max |dR| {np.abs(R.cpu().numpy() - ref[1]).max():.2e}"
    print(line, flush=True)
