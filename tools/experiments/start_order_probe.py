#!/usr/bin/env python3
"""Round-4 verdict item 4(b): what would a perfect start-order predictor be worth on the judged launch (10 000 PnP, N = 10)?
The problems of the bench set are permuted on the host using the iteration counts of a previous launch of the SAME set (an oracle no real
predictor can beat): (1) slow problems first (they start in the first round of wavefronts), (2) slow first AND at most one slow problem
per wavefront (spread: four consecutive problems share a wavefront), (3) slow problems last (the worst case), against (0) the set as it is
and (4) a random permutation.  Prints ms per launch (HIP events around 50 back-to-back launches, 3 repeats).   GPU box, repo root."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import _lib, synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
dev = torch.device("cuda:0")
d = synth.make_pnpl(batch, 10, 0, 2.0, seed=42)
res = ca.pnp_batch(torch.as_tensor(d["pts_2d"], device=dev), torch.as_tensor(d["pts_3d"], device=dev), torch.as_tensor(d["K"], device=dev))
it = res.iters.cpu().numpy()
slow_first = np.argsort(-it, kind="stable")
nslow = int((it > 7).sum())
# spread: the k-th slowest problem goes to wavefront k (slot 0), the rest fill the other slots in order
spread = np.empty(batch, dtype=np.int64)
nw = (batch + 3) // 4
order = slow_first
pos = np.concatenate([np.arange(nw) * 4 + s for s in range(4)])
pos = pos[pos < batch]
spread[pos] = order[: len(pos)]
perms = {"as is": np.arange(batch), "slow first": slow_first, "slow first, one per wavefront": spread, "slow last": slow_first[::-1].copy(),
         "random": np.random.RandomState(1).permutation(batch)}
L = _lib.lib()
print(f"{batch} problems, {nslow} with more than 7 iterations (max {it.max()})")
for name, p in perms.items():
    p2 = torch.as_tensor(np.ascontiguousarray(d["pts_2d"][p]), device=dev)
    p3 = torch.as_tensor(np.ascontiguousarray(d["pts_3d"][p]), device=dev)
    K = torch.as_tensor(d["K"], device=dev)
    ms_all = []
    for rep in range(3):
        for _ in range(5):
            r = ca.pnp_batch(p2, p3, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            r = ca.pnp_batch(p2, p3, K)
        e1.record()
        torch.cuda.synchronize()
        ms_all.append(e0.elapsed_time(e1) / 50)
    ok = bool((r.iters.cpu().numpy() == it[p]).all())
    print(f"{name:32s} ms per launch {min(ms_all):.4f} (of {', '.join('%.4f' % m for m in ms_all)})  same iteration counts: {ok}", flush=True)
