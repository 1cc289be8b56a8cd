"""Kernel resources of the shipped build against a committed table (round-5 verdict, weak items 4-5).

Several kernels owe their speed to a register allocation that one innocent edit or a compiler bump can lose without any test noticing:
`csrc/lane_kernel.hip` is a translation unit of its own only to keep one (CVX_REFINE_NEWTON2), `build.py` passes `-mllvm -enable-ipra=0`
for another, CVXW_ROLES hides loop invariants behind empty asm.  The compiler reports registers, scratch, occupancy and LDS of every kernel
(-Rpass-analysis=kernel-resource-usage); `cvxpnpl_amd/build.py` keeps those remarks beside the library, and this test holds them against
`tests/golden/kernel_resources.json` (rewritten on purpose with `python tools/resource_table.py --write-golden` when a change is meant).

No GPU needed: hipcc cross-compiles.  When the library is up to date the test only parses a file; a stale library is rebuilt first (minutes).
Checked by hand when the test was written: without CVX_REFINE_NEWTON2 in lane_kernel.hip, and without -enable-ipra=0, the table differs
beyond the tolerances below (the numbers are in the assertion messages of test_the_guards_are_live).
"""
import json
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
SCRATCH_SLACK = 16  # bytes per lane
REG_SLACK = 8       # VGPRs / AGPRs: an edit may move a few; occupancy (exact) is what they must not cross


def _compare(table, golden, only=None):
    problems = []
    for name, g in golden.items():
        if only and not any(o in name for o in only):
            continue
        r = table.get(name)
        if r is None:
            problems.append(f"{name}: kernel missing from the build")
            continue
        if r["occupancy"] != g["occupancy"]:
            problems.append(f"{name}: occupancy {r['occupancy']} waves/SIMD, table says {g['occupancy']}")
        if r["lds"] != g["lds"]:
            problems.append(f"{name}: LDS {r['lds']} B/block, table says {g['lds']}")
        if r["scratch"] > g["scratch"] + SCRATCH_SLACK:
            problems.append(f"{name}: scratch {r['scratch']} B/lane, table says {g['scratch']} (+{SCRATCH_SLACK} allowed)")
        for k in ("vgpr", "agpr"):
            if r[k] > g[k] + REG_SLACK:
                problems.append(f"{name}: {k} {r[k]}, table says {g[k]} (+{REG_SLACK} allowed)")
    if not only:
        for name in table:
            if name not in golden:
                problems.append(f"{name}: new kernel, not in the table (python tools/resource_table.py --write-golden)")
    return problems


def test_shipped_kernels_match_the_committed_resource_table():
    from cvxpnpl_amd import build as b

    b.build()
    assert os.path.exists(b.RESOURCES) and os.path.getmtime(b.RESOURCES) >= os.path.getmtime(b.OUT) - 1
    table = b.kernel_resources()
    golden = json.load(open(GOLDEN))
    assert len(table) >= 30, len(table)
    problems = _compare(table, golden)
    assert not problems, "\n".join(problems)
    # The general scalar core on lanes (cvx::solve_sdp<TWIN = false> as a kernel of its own: solve_lane_kernel) is NOT part of the shipped
    # library: run beyond the six iterations it is specified for it produces NaN iterates in -O1 builds (tools/microbench/lane_twin_repro.*,
    # profiles/r06/lane_twin_repro.txt) -- it exists in experiment builds only.
    assert not [k for k in table if "solve_lane_kernel" in k], [k for k in table if "solve_lane_kernel" in k]


@pytest.mark.slow
def test_the_guards_are_live():
    """The table really guards what it is meant to: the lane kernels' translation unit compiled WITHOUT its CVX_REFINE_NEWTON2 fails the
    comparison (one file, ~1 min).  Not in the default run (-m slow)."""
    from cvxpnpl_amd import build as b

    src = open(b.LANE_SRC).read()
    assert "#define CVX_REFINE_NEWTON2" in src
    with tempfile.TemporaryDirectory() as tmp:
        alt = os.path.join(os.path.dirname(b.LANE_SRC), "_lane_kernel_without_newton2.hip")
        try:
            open(alt, "w").write(src.replace("#define CVX_REFINE_NEWTON2", "// (removed by the test) CVX_REFINE_NEWTON2"))
            r = subprocess.run([b.hipcc(), "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                                "-mllvm", "-enable-ipra=0", "-c", "-o", os.path.join(tmp, "l.o"), alt], stderr=subprocess.PIPE, text=True)
        finally:
            if os.path.exists(alt):
                os.remove(alt)
        assert r.returncode == 0, r.stderr[-2000:]
        rem = os.path.join(tmp, "rem.txt")
        open(rem, "w").write(r.stderr)
        table = b.kernel_resources(rem)
    problems = _compare(table, json.load(open(GOLDEN)), only=("solve_lane2_kernel",))
    assert problems, "the lane kernels compile to the same resources without CVX_REFINE_NEWTON2: the separate translation unit is no longer needed"
