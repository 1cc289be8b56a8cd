#!/usr/bin/env python
"""In-kernel assembly (cvxpnpl_solve_batch: the solve kernel streams its problem's correspondences) against the blocked assembly
(cvxpnpl_assemble_large_batch + cvxpnpl_solve_cost_batch) by problem size and batch: the data behind cvxpnpl_amd.api's routing rule.
GPU box:  python tools/large_n_crossover.py > large_n_crossover.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import api, synth  # noqa: E402

dev = torch.device("cuda:0")


def timed(f, reps):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for batch in (1, 16, 256, 1000, 4000, 16000, 50000):
    for n in (96, 192, 384, 768, 1536, 3072):
        if batch * n > 4e7:
            continue
        d = synth.device_pnpl(batch, n, 0, sigma=2.0, seed=3, device=dev)
        a = (d["pts_2d"], None, d["pts_3d"], None, d["K"])
        old = (api.LARGE_N, api.LARGE_N_MANY)
        reps = 20 if batch * n < 4e6 else 8
        api.LARGE_N = api.LARGE_N_MANY = 1 << 30
        t_in = timed(lambda: ca.pnpl_batch(*a), reps)
        api.LARGE_N = api.LARGE_N_MANY = 1
        t_bl = timed(lambda: ca.pnpl_batch(*a), reps)
        api.LARGE_N, api.LARGE_N_MANY = old
        print(json.dumps({"batch": batch, "n": n, "in_kernel_ms": round(1e3 * t_in, 4), "blocked_ms": round(1e3 * t_bl, 4),
                          "blocked_over_in_kernel": round(t_bl / t_in, 3)}), flush=True)
