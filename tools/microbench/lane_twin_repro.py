#!/usr/bin/env python3
"""Driver of lane_twin_repro.hip (see there).  `--build` (container): compiles the variants into tools/microbench/lane_twin_*.so;
without it (GPU box): runs each variant on three problem sets -- ten points, planar scenes (the twin branch fires on every problem), four
points -- and holds status / iterations / pose / Z against the g++ host build of the same header (tests/hostsim)."""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = {
    "O3_w1": ["-O3", "-DREPRO_WAVES=1"], "O3_w2": ["-O3", "-DREPRO_WAVES=2"], "O3_w4": ["-O3", "-DREPRO_WAVES=4"],
    "O3_w1_noipra": ["-O3", "-DREPRO_WAVES=1", "-mllvm", "-enable-ipra=0"], "O1_w1": ["-O1", "-DREPRO_WAVES=1"], "O2_w2": ["-O2", "-DREPRO_WAVES=2"],
}


def build():
    procs = []
    for name, flags in VARIANTS.items():
        out = os.path.join(HERE, f"lane_twin_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage"] + flags + \
              ["-o", out, os.path.join(HERE, "lane_twin_repro.hip")]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, p in procs:
        err = p.communicate()[1]
        res = [ln.split("remark:")[1].split("[-R")[0].strip() for ln in err.splitlines() if any(k in ln for k in ("ScratchSize", "VGPRs Spill", "Occupancy"))]
        print(name, "rc", p.returncode, "| twin kernel:", "; ".join(res[:3]))


def main():
    import torch
    import hostsim
    from cvxpnpl_amd import synth

    dev = torch.device("cuda:0")
    sets = {"pnp10": synth.make_pnp(2048, 10, 2.0, seed=42), "planar10": synth.make_planar_pnp(1024, 10, 1.0, seed=5), "pnp4": synth.make_pnp(1024, 4, 1.0, seed=9)}
    for f in sorted(glob.glob(os.path.join(HERE, "lane_twin_*.so"))):
        L = C.CDLL(f)
        L.repro_run.argtypes = [C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_void_p]
        for sname, d in sets.items():
            B, n = d["pts_3d"].shape[:2]
            p2, p3, K = (torch.as_tensor(d[k], device=dev).contiguous() for k in ("pts_2d", "pts_3d", "K"))
            for twin in (1, 0):
                for f64 in (1, 0):
                    R = torch.full((B, 9), float("nan"), dtype=torch.float64, device=dev)
                    Z = torch.full((B, 55), float("nan"), dtype=torch.float64, device=dev)
                    st = torch.full((B,), -7, dtype=torch.int32, device=dev)
                    it = torch.zeros((B,), dtype=torch.int32, device=dev)
                    rc = L.repro_run(twin, B, n, p2.data_ptr(), p3.data_ptr(), K.data_ptr(), 300, f64, R.data_ptr(), st.data_ptr(), it.data_ptr(), Z.data_ptr(), None)
                    torch.cuda.synchronize()
                    h = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"],
                                            opts=hostsim.default_opts(max_iters=300, rescue_from=0, f32_sweeps_until=0 if f64 else 64), want_Z=True)
                    Rg, Zg, sg, ig = R.cpu().numpy().reshape(B, 3, 3), Z.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
                    nanR = int(np.isnan(Rg).any(axis=(1, 2)).sum()) - int(np.isnan(h["R"]).any(axis=(1, 2)).sum())
                    nanZ = int(np.isnan(Zg).any(axis=1).sum()) - int(np.isnan(h["Z"]).any(axis=1).sum())
                    same_st = float((sg == h["status"]).mean()) if twin else float("nan")
                    both = (sg == 0) & (h["status"] == 0)
                    geo = float(synth.geodesic(Rg, h["R"])[both].max()) if both.any() and twin else float("nan")
                    print(f"{os.path.basename(f):28s} {sname:9s} twin={twin} f64={f64} rc={rc} NaN poses beyond the host's {nanR:4d}  NaN Z {nanZ:4d}  status 3 {int((sg == 3).sum()):4d} "
                          f"(host {int((h['status'] == 3).sum())})  unwritten {int((sg == -7).sum())}  status == host {same_st:.4f}  iters == host {float((ig == h['iters']).mean()):.4f}  "
                          f"max geodesic certified-by-both {geo:.1e}", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
