#!/usr/bin/env python3
"""Driver of lane2x_sweep.hip: `--build` compiles it (the two-lane kernel for 1, 2 and 3 wavefronts per SIMD) and prints the compiler's
resource report; without it (GPU box) both geometries run 4 float64 sweeps on 125 000 / 32 768 random symmetric matrices of the iterate's
scale, the squared column norms are checked against numpy's eigenvalues, and the time per launch is printed."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    for w in (1, 2, 3):
        out = os.path.join(HERE, f"lane2x_sweep_w{w}.so")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", f"-DB_WAVES={w}",
                            "-Rpass-analysis=kernel-resource-usage", "-o", out, os.path.join(HERE, "lane2x_sweep.hip")], stderr=subprocess.PIPE, text=True)
        cur = None
        for ln in r.stderr.splitlines():
            if "Function Name" in ln:
                cur = "one lane per problem " if "one_lane" in ln else "two lanes per problem"
            for k in ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "VGPRs Spill"):
                if k in ln and cur:
                    print(f"B_WAVES={w} {cur}: {ln.split('remark:')[1].split('[-R')[0].strip()}")
        print("rc", r.returncode)


def main():
    import torch
    rs = np.random.RandomState(3)
    for batch in (125000, 32768):
        A = rs.standard_normal((batch, 10, 10))
        S = (A + A.transpose(0, 2, 1)) * 0.3   # symmetric, eigenvalues of either sign: like the iterate W
        iu = np.triu_indices(10)
        W55 = np.ascontiguousarray(S[:, iu[0], iu[1]])
        fro = np.sqrt((S ** 2).sum(axis=(1, 2)))
        sigma = 1.5 * fro + 1e-300
        lam = np.linalg.eigvalsh(S) + sigma[:, None]
        ref = np.sort(lam ** 2, axis=1)
        dW = torch.as_tensor(W55, device="cuda")
        for w in (1, 2, 3):
            L = C.CDLL(os.path.join(HERE, f"lane2x_sweep_w{w}.so"))
            L.lane2x_run.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
            for which, name in ((0, "one lane per problem "), (1, f"two lanes per problem (compiled for {w} wavefronts per SIMD)")):
                if which == 0 and w != 1:
                    continue
                for sweeps in (4, 8):
                    out = torch.zeros((batch, 10), dtype=torch.float64, device="cuda")
                    ms = C.c_float()
                    rc = L.lane2x_run(which, batch, dW.data_ptr(), sweeps, out.data_ptr(), 20, C.byref(ms))
                    torch.cuda.synchronize()
                    got = np.sort(out.cpu().numpy(), axis=1)
                    err = np.abs(got - ref).max() / ref.max()
                    print(f"{batch:7d} problems, {sweeps} sweeps, {name}: {1e3 * ms.value:8.1f} us per launch, {1e3 * ms.value / sweeps:7.1f} us per sweep, "
                          f"rc {rc}, max relative error of the squared eigenvalues {err:.1e}", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
