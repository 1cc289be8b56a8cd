#!/usr/bin/env python3
"""Randomised parity campaign: HIP path (every layout) vs the CPU oracle over many problem shapes and noise
levels (GPU box, repo root):  python tools/fuzz_parity.py [n_configs] [problems_per_config] [f64] [minimal] [seed=N]
("minimal": four to six correspondences only -- the configurations whose slow problems go through the interior-point path, csrc/ipm_quad.h;
 "f64": every layout with opts.f32_sweeps_until = 0 -- the float64 instantiations, round 4: cvxl::lane_phase_f64 among them)
One line per configuration + a summary; certified GPU poses are compared with the oracle's converged solve
(rotation geodesic, relative translation).  Diagnostics / evidence tool (uses the oracle: not product code)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
import oracle as orc  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 192
extra = {"f32_sweeps_until": 0} if "f64" in sys.argv[3:] else {}
minimal = "minimal" in sys.argv[3:]
seed = next((int(a.split('=')[1]) for a in sys.argv[3:] if a.startswith('seed=')), 2026)   # (seed=N: another draw of configurations and problems)
rs = np.random.RandomState(seed)
dev = torch.device("cuda:0")
worst = {"rot": 0.0, "t": 0.0}
tot = cert = cmp_ = mism = 0
t0 = time.time()
for c in range(ncfg):
    kind = rs.choice(["pnp", "pnl", "pnpl"])
    n_p = int(rs.randint(4, 25)) if kind != "pnl" else 0
    n_l = int(rs.randint(4, 13)) if kind == "pnl" else (int(rs.randint(1, 9)) if kind == "pnpl" else 0)
    if kind == "pnpl":
        n_p = int(rs.randint(2, 13))
    if minimal:  # at most six correspondences (a line counts as one)
        if kind == "pnp":
            n_p = int(rs.randint(4, 7))
        elif kind == "pnl":
            n_l = int(rs.randint(4, 7))
        else:
            n_p = int(rs.randint(2, 4)); n_l = int(rs.randint(2, 4))
    sigma = float(rs.choice([0.0, 0.5, 1.0, 2.0, 5.0]))
    d = synth.make_pnpl(nprob, n_p, n_l, sigma, seed=5000 + c + (0 if seed == 2026 else 100000 + seed * 1000))
    tt = lambda x: torch.as_tensor(x, device=dev)  # noqa: E731
    o = orc.pnpl_batch(d["pts_2d"] if n_p else None, d["line_2d"] if n_l else None, d["pts_3d"] if n_p else None,
                       d["line_3d"] if n_l else None, d["K"], eps=1e-11, max_iters=200000)
    line = f"cfg {c:2d} {kind:4s} n_p {n_p:2d} n_l {n_l:2d} sigma {sigma:3.1f}:"
    for name, layout in (("wave", 2), ("quad", 3), ("lane", 1), ("penta", 4)):
        r = ca.pnpl_batch(tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                          tt(d["line_3d"]) if n_l else None, tt(d["K"]), layout=layout, **extra)
        st = r.status.cpu().numpy()
        R, t = r.R.cpu().numpy(), r.t.cpu().numpy()
        ok = (st == 0) & (o["n_poses"] == 1)
        geo = synth.geodesic(R, o["R"][:, 0])
        te = np.linalg.norm(t - o["t"][:, 0], axis=1) / np.linalg.norm(o["t"][:, 0], axis=1)
        g, e = (geo[ok].max(), te[ok].max()) if ok.any() else (0.0, 0.0)
        bad = int(((geo > 1e-6) | (te > 1e-6))[ok].sum())
        worst["rot"], worst["t"] = max(worst["rot"], g), max(worst["t"], e)
        tot += nprob; cert += int((st == 0).sum()); cmp_ += int(ok.sum()); mism += bad
        line += f" {name} cert {np.mean(st == 0):.3f} rot {g:.1e} t {e:.1e}" + (f" MISMATCH {bad}" if bad else "")
    print(line, flush=True)
print(("float64 sweeps (f32_sweeps_until = 0) -- " if extra else "") + ("minimal configurations (4-6 correspondences) -- " if minimal else "") + f"summary: {ncfg} configurations x {nprob} problems x 4 layouts = {tot} solves, {cert} certified, {cmp_} compared with a "
      f"converged single-pose oracle solve, {mism} beyond 1e-6; worst rotation {worst['rot']:.2e} rad, worst relative "
      f"translation {worst['t']:.2e}; {time.time() - t0:.0f} s")
