""""run the CPU restatement under ASan/UBSan".  tools/sanitize.sh builds the host statement of the device
algorithm (tests/hostsim/hostsim.cpp over cvxpnpl_amd/csrc/{solver_core,lane_core,ipm_core,problem_io}.h -- the source the kernels
instantiate) with -fsanitize=address,undefined and runs the host-side tests of that algorithm against it; any report aborts."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.slow
def test_host_build_of_the_device_algorithm_is_clean_under_asan_and_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    env = dict(os.environ)
    env.pop("CVXPNPL_HOSTSIM_LIB", None)
    # a representative slice (the whole of both files takes several minutes under the sanitizers): the scalar solve end to end incl.
    # planar and degenerate inputs, both lane-phase restatements, the interior-point core, the dual's second tries
    sel = "solves_noise_free or lane_core or planar or dual_retry or interior or degenerate or ipm"
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize.sh"), "-k", sel], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
