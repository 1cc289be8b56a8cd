#!/usr/bin/env python
"""The accuracy experiment of the reference's paper figures for THIS solver, entirely on the device
(SURVEY.md section 8(f) row 2 -- what the device toolkit was built for).

Grids of benchmarks/synth/pnp.py:23, pnl.py:25, pnpl.py:23: n_elements in {4, 6, 8, 10, 12} x pixel noise sigma in
{0, 1, 2}; per cell `--problems` (default 100 000) synthetic problems from the reference's generator distributions
(cvxpnpl_synth_batch), solved with the reference's defaults (eps 1e-9, max_iters 2500), every rank > 1 solution expanded
into its 2 / 4 poses (cvxpnpl_recover_multi_device) and disambiguated with 20 support points the way the harness does
(suite.py:90-110, cvxpnpl_disambiguate), errors by suite.py:22-33 (cvxpnpl_pose_errors): angular error in degrees,
translation error in per cent of |t_gt|.  PnPL: the reference draws the point / line split per problem (synth.py:323);
here it is fixed at n_p = n // 2 points, n - n_p lines per cell (a batch has one shape).

GPU box:  python tools/accuracy_sweep.py [--problems N] [--out-dir profiles/r03]
Writes accuracy_{pnp,pnl,pnpl}.json (one record per cell) and prints a markdown table."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cell(kind, n, sigma, B, seed, dev):
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import api, metrics, synth

    n_p, n_l = {"pnp": (n, 0), "pnl": (0, n), "pnpl": (n // 2, n - n // 2)}[kind]
    d = synth.device_pnpl(B, n_p, n_l, sigma=sigma, seed=seed, device=dev)
    a = (d["pts_2d"] if n_p else None, d["line_2d"] if n_l else None, d["pts_3d"] if n_p else None, d["line_3d"] if n_l else None, d["K"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ca.pnpl_batch(*a, want_Z=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = res.status
    R, t = res.R.clone(), res.t.clone()
    multi = st == 1
    n_multi = int(multi.sum())
    if n_multi:
        Bt, Qt = api.assemble_batch(*a, device=dev)
        Rm, tm, cnt = api.recover_multi_device(res, Bt, Qt)
        Rs, ts, idx = metrics.disambiguate_device(Rm, tm, cnt, d["K"], d["R_gt"], d["t_gt"], n_support=20, seed=seed)
        ok = multi & (idx >= 0)
        R[ok], t[ok] = Rs[ok], ts[ok]
    ang, tr = metrics.pose_errors_device(d["R_gt"], d["t_gt"], R, t)
    ang, tr, stn = ang.cpu().numpy(), 100.0 * tr.cpu().numpy(), st.cpu().numpy()
    fin = np.isfinite(ang) & np.isfinite(tr)
    q = lambda x, p: float(np.percentile(x[fin], p)) if fin.any() else None  # noqa: E731
    return {"kind": kind, "n_elements": n, "n_points": n_p, "n_lines": n_l, "sigma_px": sigma, "problems": B, "seed": seed,
            "ang_deg": {"median": q(ang, 50), "mean": float(ang[fin].mean()), "p25": q(ang, 25), "p75": q(ang, 75), "p99": q(ang, 99)},
            "trans_pct": {"median": q(tr, 50), "mean": float(tr[fin].mean()), "p25": q(tr, 25), "p75": q(tr, 75), "p99": q(tr, 99)},
            "failed_frac": float(1.0 - fin.mean()), "status_hist": np.bincount(stn, minlength=5).tolist(),
            "certified_frac": float((stn == 0).mean()), "rank_gt1_frac": float((stn == 1).mean()),
            "mean_iters": float(res.iters.float().mean().item()), "solve_ms": 1e3 * dt, "poses_per_s": B / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=100_000)
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out", "accuracy"))
    ap.add_argument("--kinds", default="pnp,pnl,pnpl")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda:0")
    os.makedirs(args.out_dir, exist_ok=True)
    from cvxpnpl_amd import _lib

    import hashlib

    lib_hash = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    for kind in args.kinds.split(","):
        rows = []
        for n in (4, 6, 8, 10, 12):           # benchmarks/synth/pnp.py:23
            for sigma in (0.0, 1.0, 2.0):
                cell(kind, n, sigma, 2048, 1, dev)  # warm-up of this shape (workspace, code objects)
                r = cell(kind, n, sigma, args.problems, 42 + 100 * n + int(sigma), dev)
                r["lib_sha16"] = lib_hash
                rows.append(r)
                print(json.dumps(r), file=sys.stderr, flush=True)
        with open(os.path.join(args.out_dir, f"accuracy_{kind}.json"), "w") as f:
            json.dump(rows, f, indent=1)
        print(f"\n### {kind}: median (mean) angular error [deg] | median (mean) translation error [%] | certified | rank>1 | M poses/s")
        print("| n | sigma 0 | sigma 1 | sigma 2 |")
        print("|---|---|---|---|")
        for n in (4, 6, 8, 10, 12):
            cs = [r for r in rows if r["n_elements"] == n]
            f = lambda r: (f"{r['ang_deg']['median']:.2e} ({r['ang_deg']['mean']:.2e}) \\| {r['trans_pct']['median']:.2e} ({r['trans_pct']['mean']:.2e}) "  # noqa: E731
                           f"\\| {100 * r['certified_frac']:.2f} % \\| {100 * r['rank_gt1_frac']:.2f} % \\| {r['poses_per_s'] / 1e6:.1f}")
            print(f"| {n} | " + " | ".join(f(r) for r in cs) + " |")


if __name__ == "__main__":
    main()
