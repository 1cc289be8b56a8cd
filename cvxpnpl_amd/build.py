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


RESOURCES = os.path.join(HERE, "libcvxpnpl_amd.resources.txt")  # the compiler's kernel-resource remarks of the build that made OUT


def compile_cmd(out=OUT):
    srcs = [SRC, LANE_SRC] + ([HOST_SRC] if os.path.exists(HOST_SRC) else [])
    # -enable-ipra=0: the one non-inlined device function (cvxw::coop_ipm) is called from the rescue kernel only; with
    # inter-procedural register allocation the CALLER's first-order loop around it came out 40 % slower (profiles/r02/ipm_clock.jsonl)
    # -Rpass-analysis=kernel-resource-usage: registers / scratch / occupancy / LDS of every kernel, kept beside the library
    # (RESOURCES) and held against tests/golden/kernel_resources.json by tests/test_kernel_resources.py
    return [hipcc(), "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
            "-mllvm", "-enable-ipra=0", "-o", out] + srcs


def build(force=False, verbose=False):
    deps = [d for d in DEPS if os.path.exists(d)]
    fresh = os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps)
    if not force and fresh and os.path.exists(RESOURCES) and os.path.getmtime(RESOURCES) >= os.path.getmtime(OUT):
        return OUT
    cmd = compile_cmd()
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-8000:])
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(RESOURCES, "w") as f:
        f.write(r.stderr)
    if verbose:
        sys.stderr.write(r.stderr)
    return OUT


def kernel_resources(path=RESOURCES):
    """{demangled kernel name: {"vgpr", "agpr", "scratch", "occupancy", "sgpr_spill", "vgpr_spill", "lds"}} from the remarks of a build"""
    import re

    rows, cur = [], None
    for line in open(path):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    out = {}
    for r, d in zip(rows, names):
        d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "")
        out[d] = {"vgpr": int(r.get("VGPRs", -1)), "agpr": int(r.get("AGPRs", -1)), "scratch": int(r.get("ScratchSize [bytes/lane]", -1)),
                  "occupancy": int(r.get("Occupancy [waves/SIMD]", -1)), "sgpr_spill": int(r.get("SGPRs Spill", -1)),
                  "vgpr_spill": int(r.get("VGPRs Spill", -1)), "lds": int(r.get("LDS Size [bytes/block]", -1))}
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
