#!/usr/bin/env python3
"""Round 6 (GPU box): what ONE eigen-gradient step of the dual costs on a chain.  Problems of the judged set that certify after 5 iterations
with the step and after 7 without it (host build of the device algorithm), each alone on the device (64 copies, wave layout): launch time
with opts.dual_refine = 0 / 1.  T(0) = 7 iterations + 2 attempts, T(1) = 5 iterations + 1 attempt + 1 step."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostsim  # noqa: E402

import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

d = synth.make_pnp(10000, 10, 2.0, seed=42)
h0 = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts(dual_refine=0))
h1 = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts(dual_refine=1))
pick = np.where((h0["iters"] == 7) & (h1["iters"] == 5))[0][:6]
easy = np.where((h0["iters"] == 5) & (h1["iters"] == 5))[0][:2]
dev = torch.device("cuda:0")
K = torch.as_tensor(d["K"], device=dev)
for label, idxs in (("rescued (7 -> 5)", pick), ("certifies at 5 either way", easy)):
    for i in idxs:
        p2 = torch.as_tensor(np.repeat(d["pts_2d"][i:i + 1], 64, 0), device=dev)
        p3 = torch.as_tensor(np.repeat(d["pts_3d"][i:i + 1], 64, 0), device=dev)
        row = []
        for f64 in (0, 1):
            for rf in (0, 1):
                kw = dict(layout=2, dual_refine=rf)
                if f64:
                    kw["f32_sweeps_until"] = 0
                for _ in range(5):
                    r = ca.pnp_batch(p2, p3, K, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for _ in range(30):
                    e0.record()
                    r = ca.pnp_batch(p2, p3, K, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                row.append((f64, rf, float(np.median(ts)), int(r.iters[0].item()), int(r.status[0].item())))
        print(label, "problem", int(i), " ".join(f"[f64={a} refine={b}: {t:.1f} us, {it} its, st {s}]" for a, b, t, it, s in row))
