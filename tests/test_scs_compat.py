"""The `scs.solve`-shaped seam (cvxpnpl_amd/scs_compat.py): the reference's one native call (cvxpnpl.py:485-492, :517;
benchmarks/toolkit/methods/rc.py:90-96) answered by the HIP solver.

CPU tests pin the module's own restatement of the static problem data to the golden dump of the reference's `_A`, `_b`, `_A_rc`
(tests/golden, written by importing the reference) and its unpacking of `c = vech(Q, 2)`.  The -m gpu tests feed every cost vector the
reference itself produced (g3_*_c, e2e_*_c of both golden files) through the seam and compare `x`, `dobj` and the poses the
reference's post-processing -- restated in the oracle -- recovers from the returned `x` with what the reference returned.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONES = {"z": 22, "l": 0, "q": [], "ep": 0, "s": [10]}          # cvxpnpl.py:13 (SCS 3 keys)
CONES_RC = {"f": 16, "l": 0, "q": [], "ep": 0, "s": [10]}       # rc.py:92 (SCS 2 keys)


@pytest.fixture(scope="module")
def grc():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors_rc.npz"), allow_pickle=False))


def _csc(A):
    import scipy.sparse as sp

    return sp.csc_matrix(A)


# ------------------------------------------------------------------------------------------------ CPU: data handling
def test_static_data_is_the_references(golden, grc):
    from cvxpnpl_amd import _lib, scs_compat as sc

    A, b = sc.static_data(_lib.VARIANT_FULL)
    assert A.shape == (77, 55) and np.array_equal(A, golden["g4_A"]) and np.array_equal(b, golden["g4_b"])
    A, b = sc.static_data(_lib.VARIANT_RC)
    assert A.shape == (71, 55) and np.array_equal(A, grc["rc_A"]) and np.array_equal(b, grc["rc_b"])
    assert int(golden["g4_cone_zero"]) == 22 and list(golden["g4_cone_s"]) == [10]
    assert int(grc["cone_f"]) == 16 and list(np.atleast_1d(grc["cone_s"])) == [10]


def test_cost_unpacking_matches_vech_of_the_reference(golden, grc):
    from cvxpnpl_amd import scs_compat as sc
    from cvxpnpl_amd.api import pack_cost

    for k in ("pnp", "pnl", "pnpl"):
        A = golden[f"g3_{k}_A"]
        np.testing.assert_allclose(sc.cost_from_c(golden[f"g3_{k}_c"]), pack_cost(A.T @ A), rtol=0, atol=1e-13 * np.abs(A.T @ A).max())
    for i in range(int(grc["e2e_count"])):
        A = grc[f"e2e_{i}_A"]
        np.testing.assert_allclose(sc.cost_from_c(grc[f"e2e_{i}_c"]), pack_cost(A.T @ A), rtol=0, atol=1e-13 * np.abs(A.T @ A).max())
    c = golden["g3_pnp_c"].copy()
    c[9] = 1e-3  # entry (9, 0): a cost on the homogenising column is outside the family
    with pytest.raises(ValueError):
        sc.cost_from_c(c)
    batch = np.stack([golden["g3_pnp_c"], golden["g3_pnl_c"]])
    assert sc.cost_from_c(batch).shape == (2, 45)


def test_foreign_problem_data_is_refused(golden, grc):
    """not a general conic solver: anything but the two static sets raises before a GPU is touched"""
    from cvxpnpl_amd import _lib, scs_compat as sc

    good = {"A": _csc(golden["g4_A"]), "b": golden["g4_b"], "c": golden["g3_pnp_c"]}
    assert sc._identify(good, CONES) == _lib.VARIANT_FULL
    assert sc._identify(good, {"f": 22, "l": 0, "q": [], "ep": 0, "s": [10]}) == _lib.VARIANT_FULL  # SCS 2 key (cvxpnpl.py:16)
    assert sc._identify({"A": _csc(grc["rc_A"]), "b": grc["rc_b"]}, CONES_RC) == _lib.VARIANT_RC
    A = golden["g4_A"].copy()
    A[3, 7] = 0.5
    with pytest.raises(ValueError):
        sc._identify({"A": _csc(A), "b": golden["g4_b"]}, CONES)
    with pytest.raises(ValueError):
        sc._identify({"A": _csc(golden["g4_A"]), "b": 2 * golden["g4_b"]}, CONES)
    with pytest.raises(ValueError):
        sc._identify(good, {"z": 22, "l": 3, "q": [], "ep": 0, "s": [10]})
    with pytest.raises(ValueError):
        sc._identify(good, {"z": 22, "l": 0, "q": [], "ep": 0, "s": [9]})
    with pytest.raises(ValueError):
        sc._identify({"A": _csc(grc["rc_A"]), "b": grc["rc_b"]}, CONES)
    with pytest.raises(TypeError):
        sc.solve(good, CONES, epsilon=1e-9)


@pytest.mark.skipif(not os.path.exists("/root/reference/cvxpnpl.py"), reason="the reference tree only exists in the build container")
def test_reference_imports_over_the_seam_and_its_data_is_accepted():
    """sys.modules['scs'] = cvxpnpl_amd.scs_compat: the unmodified reference imports, picks the SCS-3 names from __version__, and the
    static data it would pass on every call is the data the seam accepts"""
    import subprocess

    code = (
        "import sys; sys.dont_write_bytecode = True; sys.path.insert(0, %r)\n"
        "import cvxpnpl_amd.scs_compat as sc\n"
        "sys.modules['scs'] = sc\n"
        "sys.path.insert(0, '/root/reference')\n"
        "import cvxpnpl\n"
        "assert cvxpnpl._CONES == {'z': 22, 'l': 0, 'q': [], 'ep': 0, 's': [10]} or 'z' in cvxpnpl._CONES\n"
        "assert sc._identify({'A': cvxpnpl._A, 'b': cvxpnpl._b}, cvxpnpl._CONES) == 0\n"
        "assert cvxpnpl._scs_kwarg_map['eps'] == 'eps_abs'\n"
        "print('ok')\n" % ROOT
    )
    out = subprocess.run([sys.executable, "-B", "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ GPU: the reference's own inputs
@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _AB(orc, g, i):
    """A, B of golden problem i, by the oracle's restatement of the reference's assembly (pinned to G1/G2/G3 in test_oracle_golden)"""
    K = g["K_kinect"]
    p2, p3, l2, l3 = g[f"e2e_{i}_pts2d"], g[f"e2e_{i}_pts3d"], g[f"e2e_{i}_line2d"], g[f"e2e_{i}_line3d"]
    Cs, Ns = [], []
    if len(p3):
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(p2, p3, K)
        Cs += [c1, c2, c3]
        Ns += [n1, n2, n3]
    if len(l3):
        Cl, Nl = orc.line_constraints(l2, l3, K)
        Cs.append(Cl)
        Ns.append(Nl)
    B, A = orc.eliminate(np.vstack(Cs), np.vstack(Ns))
    return A, B


TOL_X = 1e-6      # |x - x_ref|_inf, Z entries are O(1): both are the rank-1 optimum z z^T of a tight relaxation
TOL_ROT = 1e-6    # rad, north_star's tolerance
TOL_T = 1e-6


@pytest.mark.gpu
def test_seam_on_the_references_own_solver_inputs(gpu, golden, orc):
    """every (c -> x, dobj) pair the golden file holds for the 22-equality problem: through scs_compat.solve exactly as
    cvxpnpl.py:485-489 calls it, then the reference's post-processing (oracle restatement of :492-520) on the returned x"""
    from conftest import geodesic
    from cvxpnpl_amd import scs_compat as sc

    data = {"A": _csc(golden["g4_A"]), "b": golden["g4_b"]}
    kw = {"eps_abs": float(golden["g3_kw_eps_abs"]), "max_iters": int(golden["g3_kw_max_iters"]), "verbose": False}
    for i in range(int(golden["e2e_count"])):
        c = golden[f"e2e_{i}_c"]
        r = sc.solve(dict(data, c=c), CONES, **kw)
        x, info = r["x"], r["info"]
        assert x.shape == (55,) and info["status_val"] == 1 and info["cvxpnpl_status"] == 0
        assert np.abs(x - golden[f"e2e_{i}_x"]).max() < TOL_X
        # the dual bound: certified below the primal value of the returned x, within eps of it, and equal to the reference run's dobj
        # to the accuracy that run reached (its SCS stand-in stops at a residual, not at a gap)
        tr = c[[0, 10, 19, 27, 34, 40, 45, 49, 52]].sum()
        assert info["dobj"] <= info["pobj"] + 1e-12 * tr and info["pobj"] - info["dobj"] <= kw["eps_abs"] + 1e-12 * tr
        assert abs(info["dobj"] - float(golden[f"e2e_{i}_dobj"])) < 1e-6 * max(1.0, tr)
        A, B = _AB(orc, golden, i)
        poses, st, rank = orc.recover(x, info["dobj"], A, B, eps=kw["eps_abs"])
        assert rank == 1 and len(poses) == 1
        assert geodesic(poses[0][0], golden[f"e2e_{i}_R"]) < TOL_ROT
        assert np.abs(poses[0][1] - golden[f"e2e_{i}_t"]).max() < TOL_T
        assert st == 0  # the reference's certificate check |‖Ar‖² − dobj| <= eps (cvxpnpl.py:516-519) passes on (x, dobj)
    # the three example problems' captured costs (G3): solved, feasible for the static equalities, rank 1
    for k in ("pnp", "pnl", "pnpl"):
        r = sc.solve(dict(data, c=golden[f"g3_{k}_c"]), CONES, **kw)
        x = r["x"]
        assert r["info"]["cvxpnpl_status"] == 0
        assert np.abs(golden["g4_A"][:22] @ x - golden["g4_b"][:22]).max() < 1e-9
        poses, st, rank = orc.recover(x, r["info"]["dobj"], golden[f"g3_{k}_A"], golden[f"g3_{k}_B"], eps=kw["eps_abs"])
        assert rank == 1 and st == 0
        assert geodesic(poses[0][0], golden[f"ex_{k}_R"]) < TOL_ROT


@pytest.mark.gpu
def test_seam_rc_variant_on_the_references_own_inputs(gpu, grc, orc):
    """benchmarks/toolkit/methods/rc.py:90-96: 16 equalities, SCS-2 keywords"""
    from conftest import geodesic
    from cvxpnpl_amd import scs_compat as sc

    data = {"A": _csc(grc["rc_A"]), "b": grc["rc_b"]}
    for i in range(int(grc["e2e_count"])):
        r = sc.solve(dict(data, c=grc[f"e2e_{i}_c"]), CONES_RC, verbose=False, eps=float(grc["kw_eps"]), max_iters=int(grc["kw_max_iters"]))
        x = r["x"]
        assert np.abs(x - grc[f"e2e_{i}_x"]).max() < TOL_X
        # rc.py:97-131 on the returned x: eigh, rank-1 ratio, SVD projection, t = -B r (no certificate check in the rc variant)
        Z = orc.vech10_inv(x)
        vals, vecs = np.linalg.eigh(Z)
        assert (vals > 1e-3).sum() == 1
        rv = vecs[:-1, -1] / vecs[-1, -1]
        U, _, Vh = np.linalg.svd(rv.reshape(3, 3).T)
        R = U @ Vh
        t = -grc[f"e2e_{i}_B"] @ R.ravel("F")
        assert geodesic(R, grc[f"e2e_{i}_R"]) < TOL_ROT and np.abs(t - grc[f"e2e_{i}_t"]).max() < TOL_T


@pytest.mark.gpu
def test_seam_batched_equals_single_calls(gpu, golden):
    from cvxpnpl_amd import scs_compat as sc

    cs = np.stack([golden[f"e2e_{i}_c"] for i in range(int(golden["e2e_count"]))] + [golden[f"g3_{k}_c"] for k in ("pnp", "pnl", "pnpl")])
    rb = sc.solve_batch(cs, eps=1e-9, max_iters=2500)
    assert rb["x"].shape == (len(cs), 55) and (rb["status"] == 0).all()
    data = {"A": _csc(golden["g4_A"]), "b": golden["g4_b"]}
    for i, c in enumerate(cs):
        r1 = sc.solve(dict(data, c=c), CONES, eps_abs=1e-9, max_iters=2500)
        assert np.abs(r1["x"] - rb["x"][i]).max() < 1e-9 and abs(r1["info"]["dobj"] - rb["dobj"][i]) < 1e-9 * max(1.0, abs(rb["dobj"][i]))


@pytest.mark.gpu
def test_seam_reports_an_uncertified_exit_so_that_the_reference_warns(gpu, golden):
    """an iteration cap below what a certificate needs: an iterate, no certificate.  dobj must then make |cost - dobj| > eps fire
    (cvxpnpl.py:516-519); a NaN would pass that check silently."""
    from cvxpnpl_amd import scs_compat as sc

    data = {"A": _csc(golden["g4_A"]), "b": golden["g4_b"]}
    seen = 0
    for i in range(int(golden["e2e_count"])):
        for cap in (2, 3):
            r = sc.solve(dict(data, c=golden[f"e2e_{i}_c"]), CONES, eps_abs=1e-9, max_iters=cap)
            if r["info"]["cvxpnpl_status"] == 0:
                continue  # (an easy problem can certify at the cap: the last iteration always makes an attempt)
            seen += 1
            assert r["info"]["status_val"] == 2 and r["info"]["iter"] <= cap
            if np.isfinite(r["x"]).all():
                assert not np.isnan(r["info"]["dobj"]) and abs(r["info"]["pobj"] - r["info"]["dobj"]) > 1e-9
    assert seen >= 1
