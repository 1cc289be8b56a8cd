#!/usr/bin/env python3
"""tools/microbench/overlap_probe.hip driver (GPU box):  python tools/microbench/overlap_probe.py [batch] [nb] [spin_us]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxpnpl_amd import synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
spin_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
dev = torch.device("cuda:0")
d = synth.make_pnpl(batch, 10, 0, 2.0, seed=42)
tt = lambda x: torch.as_tensor(x, device=dev).contiguous()  # noqa: E731
p2, p3, K = tt(d["pts_2d"]), tt(d["pts_3d"]), tt(d["K"])
ptr = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "overlap_probe.so"))
R = torch.zeros((batch, 9), dtype=torch.float64, device=dev); t = torch.zeros((batch, 3), dtype=torch.float64, device=dev)
st = torch.zeros(batch, dtype=torch.int32, device=dev); qc = torch.zeros(64, dtype=torch.int32, device=dev)
qe = torch.zeros(batch + 64, dtype=torch.int32, device=dev); ws = torch.zeros((batch, 56), dtype=torch.float64, device=dev)
grid = (batch + 63) // 64
ta0 = torch.zeros(grid, dtype=torch.int64, device=dev); ta1 = torch.zeros_like(ta0)
tb0 = torch.zeros(nb, dtype=torch.int64, device=dev); tb1 = torch.zeros_like(tb0)
sink = torch.zeros(8, dtype=torch.float64, device=dev)
L.overlap_probe_run.argtypes = [C.c_int64, C.c_int] + [C.c_void_p] * 13 + [C.c_int, C.c_int, C.c_int, C.c_void_p]
for mode, name in ((0, "B normal priority"), (1, "B low priority"), (2, "A high, B low")):
    rc = L.overlap_probe_run(batch, 10, ptr(p2), ptr(p3), ptr(K), ptr(R), ptr(t), ptr(st), ptr(qc), ptr(qe), ptr(ws), ptr(ta0), ptr(ta1), ptr(tb0), ptr(tb1),
                             nb, int(spin_us * 100), mode, ptr(sink))
    torch.cuda.synchronize()
    a0, a1, b0, b1 = (x.cpu().numpy().astype(np.float64) / 100.0 for x in (ta0, ta1, tb0, tb1))  # us (100 MHz clock)
    z = a0.min()
    print(f"{name}: rc {rc}  A: first start 0, last start {a0.max() - z:.0f}, ends p50 {np.median(a1) - z:.0f} max {a1.max() - z:.0f} us | "
          f"B starts: min {b0.min() - z:.0f} p10 {np.percentile(b0, 10) - z:.0f} p50 {np.median(b0) - z:.0f} p90 {np.percentile(b0, 90) - z:.0f} max {b0.max() - z:.0f}; "
          f"B started before A's last block started: {(b0 < a0.max()).sum()} of {nb}; both done at {max(a1.max(), b1.max()) - z:.0f} us", flush=True)
