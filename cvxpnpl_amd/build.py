"""Build the HIP shared library in-tree:  python -m cvxpnpl_amd.build"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cvxpnpl_hip.hip")
HOST_SRC = os.path.join(HERE, "csrc", "host_recover.cpp")
LANE_SRC = os.path.join(HERE, "csrc", "lane_kernel.hip")  # solve_lane2_kernel: a translation unit of its own (see its header)
OUT = os.path.join(HERE, "libcvxpnpl_amd.so")
DEPS = [SRC, HOST_SRC, LANE_SRC, os.path.join(HERE, "csrc", "batch_args.h"), os.path.join(HERE, "csrc", "solver_core.h"), os.path.join(HERE, "csrc", "problem_io.h"),
        os.path.join(HERE, "csrc", "wave_kernel.h"), os.path.join(HERE, "csrc", "quad_kernel.h"), os.path.join(HERE, "csrc", "score_kernel.h"), os.path.join(HERE, "csrc", "assemble_kernel.h"), os.path.join(HERE, "csrc", "synth_kernel.h"), os.path.join(HERE, "csrc", "recover_core.h"), os.path.join(HERE, "csrc", "recover_kernel.h"),
        os.path.join(HERE, "csrc", "ipm_core.h"), os.path.join(HERE, "csrc", "ipm_wave.h"), os.path.join(HERE, "csrc", "ipm_quad.h"), os.path.join(HERE, "csrc", "lane_core.h"),
        os.path.join(os.path.dirname(HERE), "include", "cvxpnpl_amd.h")]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    deps = [d for d in DEPS if os.path.exists(d)]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    srcs = [SRC, LANE_SRC] + ([HOST_SRC] if os.path.exists(HOST_SRC) else [])
    # -enable-ipra=0: the one non-inlined device function (cvxw::coop_ipm) is called from the rescue kernel only; with
    # inter-procedural register allocation the CALLER's first-order loop around it came out 40 % slower (profiles/r02/ipm_clock.jsonl)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-mllvm", "-enable-ipra=0", "-o", OUT] + srcs
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
