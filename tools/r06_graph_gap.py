#!/usr/bin/env python3
"""Round 6 (GPU box): why does a hipGraph replay of the judged step lose to plain launches (BENCH_r05: 47.6 vs 49.3 M poses/s)?
Runs K plain steps, then K replays of the captured step, on one stream; under `rocprofv3 --kernel-trace` (tools/r06_graph_gap.sh) the trace
gives the idle gap between the last kernel of a step and the first kernel of the next, plain vs replay.  Wall-clock figures are printed too."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvxpnpl_amd import _lib, synth  # noqa: E402

K_STEPS = 40
dev = torch.device("cuda:0")
d = synth.make_pnp(10000, 10, 2.0, seed=42)
p2, p3, K = (torch.as_tensor(d[k], device=dev) for k in ("pts_2d", "pts_3d", "K"))
B = 10000
R = torch.empty((B, 3, 3), dtype=torch.float64, device=dev); t = torch.empty((B, 3), dtype=torch.float64, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
L = _lib.lib()
opts = _lib.default_opts(f32_sweeps_until=0)
ptr = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
gs = torch.cuda.Stream(dev)


def step(stream):
    rc = L.cvxpnpl_solve_batch(B, 10, ptr(p2), ptr(p3), 0, None, None, ptr(K), 0, C.byref(opts), ptr(R), ptr(t), ptr(st), ptr(it), None, None, None,
                               C.c_void_p(stream.cuda_stream))
    assert rc == 0, _lib.last_error()


with torch.cuda.stream(gs):
    for _ in range(5):
        step(gs)
gs.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(gs):
    for _ in range(K_STEPS):
        step(gs)
gs.synchronize()
plain = (time.perf_counter() - t0) / K_STEPS
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=gs):
    step(torch.cuda.current_stream(dev))
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
# marker between the two regions for the trace: a tiny torch kernel
torch.zeros(1, device=dev).add_(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K_STEPS):
    g.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / K_STEPS
print(f"GRAPH plain launches {1e3 * plain:.4f} ms per step ({B / plain / 1e6:.2f} M poses/s), graph replay {1e3 * rep:.4f} ms per step ({B / rep / 1e6:.2f} M poses/s)")
