#!/usr/bin/env python3
"""round 6 (GPU box): the barrier Newton solve of the dual (opts.dual_refine & 3 == 2) on the device -- statuses, iteration counts and poses
against dual_refine = 0 / 1 on the quad, wave and lane schedules; the certificate statement on every certified problem."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _solve, geodesic_np  # noqa: E402

from cvxpnpl_amd import synth  # noqa: E402

dev = torch.device("cuda:0")


def hist(it):
    return {int(k): int(v) for k, v in zip(*np.unique(it, return_counts=True))}


for name, d, n_p, n_l, kw, modes in (
        ("pnp10 10k quad", synth.make_pnpl(10000, 10, 0, 2.0, seed=42), 10, 0, {}, (0, 1, 2)),
        ("pnp10 10k quad seed 1", synth.make_pnpl(10000, 10, 0, 2.0, seed=1), 10, 0, {}, (0, 1, 2)),
        ("pnp10 2k wave", synth.make_pnpl(2000, 10, 0, 2.0, seed=42), 10, 0, {}, (0, 1, 2)),
        ("pnp10 125k lane", synth.make_pnpl(125000, 10, 0, 2.0, seed=42), 10, 0, {}, (0, 1, 2)),
        ("pnpl 5+5 100k lane", synth.make_pnpl(100000, 5, 5, 2.0, seed=42), 5, 5, {}, (0, 1, 2)),
        ("pnp6 30k", synth.make_pnpl(30000, 6, 0, 2.0, seed=42), 6, 0, {}, (0, 1, 2)),
        ("pnp4 20k", synth.make_pnpl(20000, 4, 0, 1.0, seed=42), 4, 0, {}, (0, 1, 2))):
    ref = None
    for m in modes:
        r = _solve(dev, d, n_p, n_l, dual_refine=m, **kw)
        if ref is None:
            ref = r
        both = (r["status"] == 0) & (ref["status"] == 0)
        diff = np.flatnonzero(both & (r["iters"] != ref["iters"]))
        geo = max([geodesic_np(r["R"][i], ref["R"][i]) for i in diff[:2000]], default=0.0)
        gap = (r["cost"][:, 0] - r["cost"][:, 1])[r["status"] == 0]
        h = hist(r["iters"])
        print(f"{name:24s} dual_refine={m:2d} status {np.bincount(r['status'], minlength=5).tolist()} mean iters {r['iters'].mean():.4f} max {r['iters'].max():4d} "
              f"{h if len(h) < 12 else ''} differing {len(diff)} max geodesic {geo:.1e} gap [{gap.min():.1e}, {gap.max():.1e}] nan poses {int(np.isnan(r['R']).any(axis=(1, 2)).sum())}", flush=True)
