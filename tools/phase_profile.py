"""Per-phase shader-clock profile of the wave-per-problem kernel.

Build (container):   python tools/phase_profile.py --build
Run (GPU box):       python tools/phase_profile.py [--batch 10000] [--layout 2]
Uses a separate -DCVXW_PROFILE build of the library (tools/microbench/libcvxpnpl_amd_prof.so);
the product library carries no instrumentation.
"""
import argparse
import numpy as np
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "tools", "microbench", "libcvxpnpl_amd_prof.so")
PHASES = ["assemble", "eig_setup", "jacobi", "wp_build", "top_select", "polish", "dual", "check_tail(+twin)", "dr_update", "output",
          "  polish.polar", "  polish.newton", "  polish.final", "  dual.hint", "  dual.mbuild", "  dual.ldl1", "  dual.backsub",
          "  dual.range", "  dual.ldl2"]
COUNTS = {20: "newton_iters", 21: "polish_calls", 22: "dual_calls"}


def build():
    src = os.path.join(ROOT, "cvxpnpl_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-DCVXW_PROFILE", "-o", PROF, os.path.join(src, "cvxpnpl_hip.hip"), os.path.join(src, "lane_kernel.hip"), os.path.join(src, "host_recover.cpp")]
    subprocess.check_call(cmd)
    print(PROF)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--batch", type=int, default=10000)
    ap.add_argument("--layout", type=int, default=2)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--straggler", type=int, default=-1, help="profile 64 copies of problem IDX of the 125 k bench set (a lone slow problem)")
    args = ap.parse_args()
    if args.build:
        return build()
    import torch
    from cvxpnpl_amd import _lib, synth
    _lib.LIB_PATH = PROF
    import cvxpnpl_amd as ca
    L = _lib.lib()
    d = synth.make_pnp(args.batch, args.n, args.sigma, seed=42)
    if args.straggler >= 0:
        big = synth.make_pnp(125000, 10, 2.0, seed=42)
        d = {"pts_2d": np.repeat(big["pts_2d"][args.straggler:args.straggler + 1], 64, 0), "pts_3d": np.repeat(big["pts_3d"][args.straggler:args.straggler + 1], 64, 0), "K": big["K"]}
        args.batch = 64
    dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("pts_2d", "pts_3d", "K")}
    buf = (C.c_ulonglong * 32)()
    import time
    for rep in range(3):
        L.cvxpnpl_debug_phase_cycles(buf, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ca.pnp_batch(dev["pts_2d"], dev["pts_3d"], dev["K"], layout=args.layout)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    L.cvxpnpl_debug_phase_cycles(buf, 0)
    cyc = list(buf)
    waves = cyc[31]
    tot = sum(cyc[:10])
    out = {"batch": args.batch, "layout": args.layout, "waves": waves, "mean_iters": float(res.iters.double().mean()),
           "cycles_per_wave_total": tot / max(waves, 1), "wall_ms_instrumented": wall * 1e3,
           "counts_per_wave": {v: cyc[k] / max(waves, 1) for k, v in COUNTS.items()}}
    print(json.dumps(out))
    for k, name in enumerate(PHASES):
        print("%-22s %10.0f  %.3f" % (name, cyc[k] / max(waves, 1), cyc[k] / max(tot, 1)))


if __name__ == "__main__":
    main()
