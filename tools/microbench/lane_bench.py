#!/usr/bin/env python3
"""Time variants of the lane phase side by side (GPU box, repo root):
    python tools/microbench/lane_bench.py tools/microbench/lane_bench_*.so [--batch 125000] [--reps 20]
Every variant runs the same 125 k PnP N = 10 problems (bench.py's pnp_n10_125k set); prints ms per launch, certified / parked
counts, mean sweeps, and the largest rotation difference of the certified poses against the first variant."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxpnpl_amd import synth  # noqa: E402


def main():
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 125_000
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    dev = torch.device("cuda:0")
    d = synth.make_pnpl(batch, 10, 0, 2.0, seed=42)
    tt = lambda x: torch.as_tensor(x, device=dev).contiguous()  # noqa: E731
    p2, p3, K = tt(d["pts_2d"]), tt(d["pts_3d"]), tt(d["K"])
    ptr = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    ref = None
    for path in libs:
        L = C.CDLL(os.path.abspath(path))
        f64 = "f64" in os.path.basename(path)
        R = torch.zeros((batch, 3, 3), dtype=torch.float64, device=dev)
        t = torch.zeros((batch, 3), dtype=torch.float64, device=dev)
        st, it, sw = (torch.zeros(batch, dtype=torch.int32, device=dev) for _ in range(3))
        qc = torch.zeros(64, dtype=torch.int32, device=dev)
        qe = torch.zeros(batch + 64, dtype=torch.int32, device=dev)
        ws = torch.zeros((batch, 56), dtype=torch.float64, device=dev)
        ms = C.c_float()
        L.lane_bench_run.argtypes = [C.c_int64, C.c_int] + [C.c_void_p] * 11 + [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        rc = L.lane_bench_run(batch, 10, ptr(p2), ptr(p3), ptr(K), ptr(R), ptr(t), ptr(st), ptr(it), ptr(sw), ptr(qc), ptr(qe), ptr(ws), 6,
                              0 if f64 else 64, reps, C.byref(ms))
        torch.cuda.synchronize()
        s = st.cpu().numpy()
        Rn = R.cpu().numpy()
        line = f"{os.path.basename(path):40s} rc {rc}  {ms.value * 1e3:8.1f} us   certified {(s == 0).sum()}  parked {(s == -1).sum()}  other {((s != 0) & (s != -1)).sum()}  mean sweeps {sw.float().mean().item():.3f}"
        if ref is None:
            ref = (s, Rn)
        else:
            both = (s == 0) & (ref[0] == 0)
            line += f"  status== {np.mean(s == ref[0]):.5f}  max dR {synth.geodesic(Rn[both], ref[1][both]).max():.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
