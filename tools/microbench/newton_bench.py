#!/usr/bin/env python3
"""Driver of newton_bench.hip: `--build` compiles it; without it (GPU box) cvxw::coop_newton runs on the recorded failed duals, one wavefront
each -- alone on the device and all at once -- and the outcome is held against the host statement cvx::dual_newton (tests/hostsim)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "newton_bench.so")


def build(extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-I", os.path.join(ROOT, "cvxpnpl_amd", "csrc"),
           "-Rpass-analysis=kernel-resource-usage", *extra, "-o", SO, os.path.join(HERE, "newton_bench.hip")]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    for ln in r.stderr.splitlines():
        if "error" in ln or any(k in ln for k in ("VGPRs:", "ScratchSize", "Occupancy", "VGPRs Spill", "LDS Size")):
            print(ln.split("remark:")[-1].split("[-R")[0].strip())
    print("rc", r.returncode)


def main():
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim
    rec = np.load(os.path.join(HERE, "newton_records.npy"))
    n = len(rec)
    S, R, delta, lam = (np.ascontiguousarray(x) for x in (rec[:, :55], rec[:, 55:64], rec[:, 64], rec[:, 65]))
    host = np.array([hostsim.dual_newton(S[i], R[i], delta[i], lam[i]) for i in range(n)])
    L = C.CDLL(SO)
    L.newton_bench_run.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    d = [torch.as_tensor(x, device="cuda") for x in (S, R, delta, lam)]
    L.newton_bench_clocks.argtypes = [C.c_void_p, C.c_int]
    for alone in (0, 1):
        out = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
        L.newton_bench_clocks(None, 1)
        rc = L.newton_bench_run(n, *(x.data_ptr() for x in d), out.data_ptr(), alone)
        if not alone:
            clk = np.zeros(8 * 512, np.int64)
            L.newton_bench_clocks(clk.ctypes.data, 0)
            clk = clk.reshape(512, 8)[:n]
            one = host[:, 3] == 1
            print("cycles by stage, one-step problems (median): setup", np.median(clk[one, 0]), "start", np.median(clk[one, 1]), "Hessian", np.median(clk[one, 2]),
                  "15x15", np.median(clk[one, 3]), "line search", np.median(clk[one, 4]), "LDL test", np.median(clk[one, 5]))
        o = out.cpu().numpy()
        okd, okh = o[:, 0] > 0, host[:, 0] > 0
        cyc = o[:, 2]
        for st in (1, 2, 3):
            m = okh & (host[:, 3] == st)
            if m.any():
                print(f"{'alone' if alone else 'all at once':12s} host steps {st}: {int(m.sum()):4d} problems, device certified {int((okd & m).sum()):4d}, "
                      f"cycles median {np.median(cyc[m]):9.0f} = {np.median(cyc[m]) / 2400:6.1f} us at 2.4 GHz (min {cyc[m].min():.0f}, max {cyc[m].max():.0f})")
        print(f"{'alone' if alone else 'all at once':12s} rc {rc}: certified on the device {int(okd.sum())} / host {int(okh.sum())} of {n}; disagreeing {int((okd != okh).sum())}; "
              f"|zSz| max {np.abs(o[okd, 1]).max():.1e}", flush=True)


if __name__ == "__main__":
    build(sys.argv[2:]) if "--build" in sys.argv else main()
