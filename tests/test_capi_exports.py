"""The C-ABI library loads, exports every symbol include/cvxpnpl_amd.h declares, and its
host-side entry points work -- no GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from cvxpnpl_amd import _lib, build

    build.build()
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from cvxpnpl_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "cvxpnpl_amd.h")).read()
    declared = set(re.findall(r"\b(cvxpnpl_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None


def test_default_opts_match_reference_defaults(L):
    from cvxpnpl_amd import _lib

    o = _lib.default_opts()
    assert o.eps == 1e-9 and o.max_iters == 2500  # cvxpnpl.py:527-528
    with pytest.raises(TypeError):
        _lib.default_opts(nonsense=1)


def test_bad_arguments_are_rejected_without_gpu(L):
    from cvxpnpl_amd import _lib

    o = _lib.default_opts()
    rc = L.cvxpnpl_solve_batch(4, 0, None, None, 0, None, None, None, 0, C.byref(o), None, None, None, None, None, None, None, None)
    assert rc == -1 and b"bad arguments" in L.cvxpnpl_last_error()
    assert b"gfx950" in L.cvxpnpl_version()


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cvxpnpl_amd

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cvxpnpl_amd.pnp(np.zeros((6, 2)), np.zeros((6, 3)), np.eye(3))


def test_product_does_not_import_oracle():
    """The shipped package must not reach into oracle/ or the test-only host build."""
    pkg = os.path.join(ROOT, "cvxpnpl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "import hostsim" not in src, f
                assert "oracle.h" not in src and "liboracle" not in src and "libhostsim" not in src, f


def test_recover_multi_host_path_matches_reference(L, golden):
    """cvxpnpl_recover_multi (host C++) against the reference's rank-2 / rank-4 outputs."""
    from cvxpnpl_amd.api import recover_multi

    B = golden["g3_pnp_B"]
    for tag, n, tol in (("r1", 1, 1e-10), ("r2", 2, 1e-7), ("r4", 4, 1e-6)):
        poses = recover_multi(golden[f"g6_{tag}_x"], B)
        Rg, tg = golden[f"g6_{tag}_R"], golden[f"g6_{tag}_t"]
        assert len(poses) == n == len(Rg)
        used = set()
        for R, t in poses:
            d = [np.abs(R - Rg[k]).max() + np.abs(t - tg[k]).max() if k not in used else np.inf for k in range(n)]
            k = int(np.argmin(d))
            assert d[k] < tol, (tag, d)
            used.add(k)
    with pytest.raises(NotImplementedError):
        recover_multi(np.full(55, np.nan), B)
