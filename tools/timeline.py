#!/usr/bin/env python3
"""Wavefront timeline of the quad kernel (diagnostics; GPU box):
    hipcc ... -DCVXQ_TIMELINE -o tools/diag/libcvxpnpl_timeline.so ;  CVXPNPL_AMD_LIB=tools/diag/libcvxpnpl_timeline.so python tools/timeline.py [batch]
Every wavefront stamps the shader clock at its start, after the assembly, at the end of the quad loop and at its end;
prints when waves start and end relative to the first start, their durations, and how busy the chip is over time."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxpnpl_amd as ca  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 3
li = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # hand-off iteration of the quad schedule (-1: default)
kw = {"f32_sweeps_until": 0} if "f64" in sys.argv[4:] else {}  # (f64: every sweep in float64, bench.py's headline mode)
dev = torch.device("cuda:0")
d = synth.make_pnp(batch, 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(d[k], device=dev) for k in ("pts_2d", "pts_3d", "K"))
for _ in range(3):
    res = ca.pnp_batch(p2, p3, K, layout=layout, lane_iters=li, **kw)
torch.cuda.synchronize()
pit = res.iters.cpu().numpy()
c = res.cost.cpu().numpy().reshape(-1)
w = res.work.cpu().numpy().reshape(-1)
npw = 5 if layout == 4 else 4  # problems per wavefront: quad 4, penta 5
nw = (batch + npw - 1) // npw
st_ = 2 * npw
T = np.stack([c[st_ * i: st_ * i + 4] for i in range(nw) if st_ * i + 4 <= len(c)])
its = np.array([w[st_ * i + 1] for i in range(len(T))])
hw = np.array([w[st_ * i] for i in range(len(T))]).astype(np.uint32)
xcc = np.array([w[st_ * i + 2] for i in range(len(T))]).astype(np.int64)
T -= T[:, 0].min()
clk = 100e6  # s_memrealtime: the 100 MHz reference clock, common to the whole device
us = T / clk * 1e6
out = {"batch": batch, "lane_iters": li, "waves": len(T), "span_us": float(us[:, 3].max()),
       "start_us_pct": {str(p): float(np.percentile(us[:, 0], p)) for p in (1, 25, 50, 75, 90, 99, 100)},
       "end_us_pct": {str(p): float(np.percentile(us[:, 3], p)) for p in (1, 25, 50, 75, 90, 99, 100)},
       "dur_us_pct": {str(p): float(np.percentile(us[:, 3] - us[:, 0], p)) for p in (1, 25, 50, 75, 90, 99, 100)},
       "assembly_us_median": float(np.median(us[:, 1] - us[:, 0])), "quadloop_us_median": float(np.median(us[:, 2] - us[:, 1])),
       "tail_us_max": float((us[:, 3] - us[:, 2]).max()), "n_with_tail": int(((us[:, 3] - us[:, 2]) > 1.0).sum()),
       "iters_hist": np.bincount(its.clip(0, 20)).tolist(),
       "slowest_waves": [{"block": int(i), "start": float(us[i, 0]), "quad_end": float(us[i, 2]), "end": float(us[i, 3]), "it": int(its[i]), "iters": pit[npw * i: npw * i + npw].tolist()} for i in np.argsort(-us[:, 3])[:12]]}
first = np.where(us[:, 0] < 10.0)[0]  # wavefronts of the first round
out["slowest_first_round"] = [{"block": int(i), "start": float(us[i, 0]), "quad_end": float(us[i, 2]), "end": float(us[i, 3]), "iters": pit[npw * i: npw * i + npw].tolist()}
                              for i in first[np.argsort(-us[first, 3])[:8]]]
# resident waves over time
grid = np.linspace(0, us[:, 3].max(), 41)
out["resident_waves"] = [int(((us[:, 0] <= g) & (us[:, 3] > g)).sum()) for g in grid]
out["grid_us"] = [round(float(g), 1) for g in grid]
# per SIMD occupancy: hw id bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] (gfx9)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
key = (se.astype(np.int64) * 16 + cu) * 4 + simd
key = key * 8 + xcc
out["waves_per_xcc"] = np.bincount(xcc).tolist()
out["distinct_simd_keys_seen"] = int(len(np.unique(key)))
out["waves_per_simd_key_hist"] = np.bincount(np.bincount(key)).tolist()
print(json.dumps(out))
