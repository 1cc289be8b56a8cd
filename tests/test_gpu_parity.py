"""GPU parity: the HIP path, called through the C ABI (cvxpnpl_amd.api -> ctypes ->
libcvxpnpl_amd.so), against the CPU oracle, the reference's golden vectors and
size-independent properties.  Run with -m gpu on an MI355X.

Tolerances (float64 path): rotation geodesic <= 1e-6 rad and relative translation <= 1e-6
against the oracle / ground truth (the north-star tolerance); in practice ~1e-9 (the
oracle's own convergence) and ~1e-14 against noise-free ground truth."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_ROT = 1e-6
TOL_T = 1e-6


@pytest.fixture(scope="module")
def gpu():
    import torch

    from cvxpnpl_amd import _lib

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.lib()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def _solve(gpu, d, n_p, n_l, **kw):
    import torch

    import cvxpnpl_amd as ca

    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    res = ca.pnpl_batch(tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                        tt(d["line_3d"]) if n_l else None, tt(d["K"]), **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in res.items()}


CASES = [(10, 0, 0.0, 256), (10, 0, 2.0, 256), (5, 5, 0.0, 128), (5, 5, 1.0, 128), (0, 6, 1.0, 64), (6, 0, 1.0, 64), (4, 0, 1.0, 64),
         (20, 9, 1.0, 67)]  # the last one: 38 records = more than one staging chunk in every layout, ragged batch
LAYOUTS = {"lane": 1, "wave": 2, "quad": 3, "penta": 4}  # CVXPNPL_LAYOUT_* (penta: the quad schedule with 12 lanes per problem)


_ORACLE_CACHE = {}


def _oracle_case(orc, d, n_p, n_l, key):
    """the oracle's converged solve of EVERY problem of a case (shared by the three layouts)"""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = orc.pnpl_batch(d["pts_2d"] if n_p else None, d["line_2d"] if n_l else None, d["pts_3d"] if n_p else None,
                                            d["line_3d"] if n_l else None, d["K"], eps=1e-11, max_iters=200000)
    return _ORACLE_CACHE[key]


def _reference_A_B(orc, d, i, n_p, n_l):
    """A (m x 9), B (3 x 9) of problem i the way cvxpnpl.pnp / pnl / pnpl build them (cvxpnpl.py:545-549, :577-580, :619-624)"""
    Cs, Ns = [], []
    if n_p:
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], d["K"] if d["K"].ndim == 2 else d["K"][i])
        Cs += [c1, c2, c3]
        Ns += [n1, n2, n3]
    if n_l:
        cl, nl = orc.line_constraints(d["line_2d"][i], d["line_3d"][i], d["K"] if d["K"].ndim == 2 else d["K"][i])
        Cs.append(cl)
        Ns.append(nl)
    B, A = orc.eliminate(np.vstack(Cs), np.vstack(Ns))
    return A, B


def _check_uncertified_exits(orc, d, r, n_p, n_l, idx):
    """Problems that left WITHOUT a certificate (status 1, 2, 4) follow the reference's recovery (cvxpnpl.py:499-513)
    from the very Z the kernel returned: the oracle's restatement of that recovery, fed with the GPU's Z, must give
    the GPU's pose.  Rank 1 (status 2 / 4): the same single pose, reflections included, to 1e-9.  Rank > 1 (status 1):
    the pose is finite and orthogonal, t = -B r, and the reference-style multi-solution recovery of the product
    (host side, unpolished) returns the oracle's poses."""
    import cvxpnpl_amd as ca

    n_cmp = 0
    for i in idx:
        st = int(r["status"][i])
        A, B = _reference_A_B(orc, d, i, n_p, n_l)
        poses, ost, ork = orc.recover(r["Z"][i], 0.0, A, B)
        R, t = r["R"][i], r["t"][i]
        assert np.isfinite(R).all() and np.isfinite(t).all(), (i, st)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-9
        assert np.abs(t + B @ R.T.reshape(9)).max() < 1e-9 * max(1.0, np.abs(t).max())  # t = -B r, r = vec_colmajor(R)
        if st in (2, 4):
            assert ork == 1 and len(poses) == 1, (i, st, ork)
            assert geodesic_np(R, poses[0][0]) < 1e-9 and np.abs(t - poses[0][1]).max() < 1e-9 * max(1.0, np.abs(t).max()), (i, st)
            assert (np.linalg.det(R) < 0) == (st == 4)
            n_cmp += 1
        else:
            assert st == 1 and ork > 1, (i, st, ork)
            if any(not np.isfinite(Ro).all() for Ro, _ in poses):
                continue  # the reference divides by a ~0 eigenvector entry there
            if ork not in (2, 4):
                # odd rank: the reference pads the basis with the next eigenvector (cvxpnpl.py:231-233), which for a
                # projected iterate lies in an exactly degenerate null space -- its own output then depends on
                # LAPACK's arbitrary choice there (measured: oracle and product both differ from the reference by
                # radians on such Z, and agree with it to 1e-14 for rank 2 and 4).  Rank > 4 is truncated the same way.
                continue
            # the reference divides the basis by the last entry of the top eigenvector (cvxpnpl.py:236); where that entry is
            # small the division amplifies rounding and the reference itself, the oracle and the product (which then pivots on
            # another eigenvector) agree only to ~1e-3 (measured) -- not a parity statement
            vals, vecs = np.linalg.eigh(orc.vech10_inv(r["Z"][i]))
            k = 2 if ork == 2 else 4
            if abs(vecs[9, -1]) < 0.05 * np.abs(vecs[9, -k:]).max():
                continue
            mine = ca.recover_multi(r["Z"][i], B.reshape(27))
            assert len(mine) == len(poses), (i, len(mine), len(poses))
            for Rm, tm in mine:  # (two eigen-solvers, then a quartic: 1e-5 is the conditioning of the rank-4 branch, typical 1e-12)
                assert min(geodesic_np(Rm, Ro) + np.abs(tm - to).max() for Ro, to in poses) < 1e-5, i
            n_cmp += 1
    return n_cmp


def geodesic_np(Ra, Rb):
    from cvxpnpl_amd import synth

    return float(synth.geodesic(Ra[None], Rb[None])[0])


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("n_p,n_l,sigma,batch", CASES)
def test_hip_vs_oracle(gpu, orc, n_p, n_l, sigma, batch, layout):
    """EVERY problem of the case against the oracle's converged solve; the certified fraction is asserted from the
    campaign numbers (profiles/r01/fuzz_parity.txt: 122 843 / 122 880), not merely 'most'."""
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(batch, n_p, n_l, sigma, seed=200 + n_p + 7 * n_l)
    r = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout], want_Z=True)
    o = _oracle_case(orc, d, n_p, n_l, (n_p, n_l, sigma, batch))
    cert = r["status"] == 0
    one = o["n_poses"] == 1
    ok = cert & one
    # minimal sets (n = 4) are often not tight; everything else certifies -- at most one straggler per case
    assert cert.sum() >= (0.5 * batch if n_p + n_l <= 4 else batch - 1), np.bincount(r["status"])
    # a certified pose is THE optimum of the relaxation: the oracle, where it converged to one pose, found the same
    assert (cert & ~one).sum() <= max(1, batch // 50), ((cert & ~one).sum(), batch)
    geo = synth.geodesic(r["R"], o["R"][:, 0])
    terr = np.linalg.norm(r["t"] - o["t"][:, 0], axis=1) / np.linalg.norm(o["t"][:, 0], axis=1)
    assert geo[ok].max() < TOL_ROT and terr[ok].max() < TOL_T, (geo[ok].max(), terr[ok].max())
    c = r["cost"][cert]
    assert (c[:, 0] - c[:, 1] >= -1e-15).all() and (c[:, 0] - c[:, 1] <= 1e-9).all()
    # every exit without a certificate: the reference's recovery of the returned Z
    _check_uncertified_exits(orc, d, r, n_p, n_l, np.where(np.isin(r["status"], (1, 2, 4)))[0][:40])
    assert not np.isin(r["status"], (3,)).any()
    if sigma == 0.0:
        assert cert.mean() > 0.98
        assert synth.geodesic(r["R"], d["R_gt"])[cert].max() < TOL_ROT


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_uncertified_exits_follow_reference_recovery(gpu, orc, layout):
    """cvxpnpl.py:499-513 on the GPU: solves cut short (max_iters 2 .. 12, no certificate yet) must return what the
    reference's recovery makes of the same Z -- rank-1 poses incl. reflections (status 2 / 4) bit-for-bit to 1e-9,
    rank > 1 flagged with a finite candidate pose and the multi-solution recovery matching the oracle's."""
    from cvxpnpl_amd import synth

    seen = {1: 0, 2: 0, 4: 0}
    for n_p, n_l, sigma, seed in ((10, 0, 2.0, 71), (5, 5, 1.0, 72), (4, 0, 1.0, 73), (0, 6, 1.0, 74)):
        d = synth.make_pnpl(96, n_p, n_l, sigma, seed=seed)
        for max_iters in (2, 3, 4, 6, 8, 12):
            # first_check beyond the cap: the only certificate attempt is the one of the last iteration
            r = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout], max_iters=max_iters, first_check=1000, want_Z=True)
            assert (r["iters"] <= max_iters).all()
            idx = np.where(np.isin(r["status"], (1, 2, 4)))[0]
            _check_uncertified_exits(orc, d, r, n_p, n_l, idx[:24])
            for s_ in seen:
                seen[s_] += int((r["status"] == s_).sum())
            assert not (r["status"] == 3).any()
    # both branches exercised.  Rank-1 exits without a certificate are intrinsically rare (a rank-1 iterate almost always
    # certifies: 9 of ~2300 cut-short solves here), reflections rarer still.
    assert seen[1] > 50 and seen[2] + seen[4] >= 5, seen


def test_hip_equals_host_build_of_device_algorithm(gpu):
    """Same source, two compilers: the HIP result equals the g++ host build to rounding."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hostsim
    from cvxpnpl_amd import synth

    d = synth.make_pnp(512, 10, 1.0, seed=77)
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    for name, layout in LAYOUTS.items():
        # first_check = 5 is the host build's schedule; left at its default (0) the library attempts first after 6 iterations in
        # the lane-hybrid layout (checked at the end)
        r = _solve(gpu, d, 10, 0, want_Z=True, layout=layout, first_check=5)
        assert (r["status"] == hs["status"]).mean() > 0.99
        same = (r["status"] == 0) & (hs["status"] == 0)
        assert synth.geodesic(r["R"], hs["R"])[same].max() < 1e-10
        assert np.abs(r["t"] - hs["t"])[same].max() < 1e-10
        assert np.abs(r["Z"] - hs["Z"])[same].max() < 1e-9
        assert np.abs(r["iters"] - hs["iters"])[same].mean() < 0.1, name  # same algorithm, same path
    dflt = {name: _solve(gpu, d, 10, 0, layout=layout) for name, layout in LAYOUTS.items()}
    assert dflt["lane"]["iters"].min() == 6 and dflt["wave"]["iters"].min() == 5 and dflt["quad"]["iters"].min() == 5
    both = (dflt["lane"]["status"] == 0) & (dflt["wave"]["status"] == 0)
    assert both.mean() > 0.99 and synth.geodesic(dflt["lane"]["R"], dflt["wave"]["R"])[both].max() < 1e-9


def test_hybrid_lane_then_wave_schedule(gpu):
    """Lane and quad layouts with hand-off: problems unfinished after `lane_iters` iterations are resumed one
    per wavefront.  Forcing the hand-off early (lane_iters=3, 4) must not change any result."""
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(3000, 5, 5, 1.0, seed=31)
    ref = _solve(gpu, d, 5, 5, layout=LAYOUTS["wave"])
    for layout, li in (("lane", 3), ("lane", 4), ("lane", 5), ("lane", 0), ("quad", 3), ("quad", 6), ("quad", 12), ("penta", 0), ("penta", 4)):
        # (first_check = 5 for all: the iteration counts are compared; by default the lane layout attempts first after 6)
        r = _solve(gpu, d, 5, 5, layout=LAYOUTS[layout], lane_iters=li, first_check=5)
        assert (r["status"] == ref["status"]).mean() > 0.995, (layout, li)
        both = (r["status"] == 0) & (ref["status"] == 0)
        assert both.mean() > 0.99
        # two certified answers agree to the certificate's resolution: the cost gap is <= 1e-9, which on
        # an ill-conditioned (flat) problem leaves ~1e-9 rad of play; typical agreement is 1e-16
        assert synth.geodesic(r["R"], ref["R"])[both].max() < 1e-7, (layout, li)
        assert np.abs(r["t"] - ref["t"])[both].max() < 1e-7
        assert np.median(synth.geodesic(r["R"], ref["R"])[both]) < 1e-14
        assert np.abs(r["iters"][both] - ref["iters"][both]).mean() < 0.5
    # minimal problems: many hand-offs, uncertifiable ones included
    d4 = synth.make_pnp(2000, 4, 1.0, seed=9)
    a = _solve(gpu, d4, 4, 0, layout=LAYOUTS["wave"], max_iters=300)
    for layout in ("lane", "quad", "penta"):
        b = _solve(gpu, d4, 4, 0, layout=LAYOUTS[layout], lane_iters=5 if layout == "lane" else 6, max_iters=300)
        assert (a["status"] == b["status"]).mean() > 0.97, layout
        both = (a["status"] == 0) & (b["status"] == 0)
        assert synth.geodesic(a["R"], b["R"])[both].max() < 1e-7, layout


def test_examples_known_answer_single_problem_api(gpu, golden):
    """BASELINE config 1: examples/pnp.py (and pnl.py / pnpl.py) through the drop-in API."""
    import warnings

    import cvxpnpl_amd as ca
    from conftest import geodesic

    with warnings.catch_warnings():
        warnings.simplefilter("error")  # certified -> the reference would not warn either
        poses = ca.pnp(pts_2d=golden["ex_pnp_pts2d"], pts_3d=golden["ex_pnp_pts3d"], K=golden["ex_pnp_K"].astype(int))
        assert len(poses) == 1
        R, t = poses[0]
        assert R.shape == (3, 3) and t.shape == (3,)
        assert geodesic(R, golden["ex_pnp_R"]) < 1e-6
        assert np.linalg.norm(t - golden["ex_pnp_t"]) / np.linalg.norm(golden["ex_pnp_t"]) < 1e-6
        R, t = ca.pnl(line_2d=golden["ex_pnl_line2d"], line_3d=golden["ex_pnl_line3d"], K=golden["ex_pnl_K"])[0]
        assert geodesic(R, golden["ex_pnl_R"]) < 1e-6
        R, t = ca.pnpl(pts_2d=golden["ex_pnpl_pts2d"], line_2d=golden["ex_pnpl_line2d"], pts_3d=golden["ex_pnpl_pts3d"],
                       line_3d=golden["ex_pnpl_line3d"], K=golden["ex_pnpl_K"])[0]
        assert geodesic(R, golden["ex_pnpl_R"]) < 1e-6
        assert np.linalg.norm(t - golden["ex_pnpl_t"]) / np.linalg.norm(golden["ex_pnpl_t"]) < 1e-6


def test_reference_e2e_vectors(gpu, golden):
    """Poses the reference's own pnp/pnl/pnpl returned (tests/golden, e2e_*)."""
    import cvxpnpl_amd as ca
    from conftest import geodesic

    for i in range(int(golden["e2e_count"])):
        p2, p3 = golden[f"e2e_{i}_pts2d"], golden[f"e2e_{i}_pts3d"]
        l2, l3 = golden[f"e2e_{i}_line2d"], golden[f"e2e_{i}_line3d"]
        poses = ca.pnpl(p2, l2, p3, l3, golden["K_kinect"])
        assert len(poses) == 1
        assert geodesic(poses[0][0], golden[f"e2e_{i}_R"]) < TOL_ROT
        assert np.abs(poses[0][1] - golden[f"e2e_{i}_t"]).max() < TOL_T


def test_full_size_properties_config2_and_3(gpu):
    """BASELINE configs 2 and 3 at full size through size-independent properties:
    noise-free ground truth recovery, certificate validity, feasibility of Z, and
    invariance to the order of the correspondences (the cost is a sum over them)."""
    from cvxpnpl_amd import synth

    for n_p, n_l, batch in ((10, 0, 10_000), (5, 5, 100_000)):
        d = synth.make_pnpl(batch, n_p, n_l, 0.0, seed=42 + n_l)
        r = _solve(gpu, d, n_p, n_l)
        cert = r["status"] == 0
        assert cert.mean() > 0.995, np.bincount(r["status"])
        assert synth.geodesic(r["R"], d["R_gt"])[cert].max() < TOL_ROT
        assert (np.linalg.norm(r["t"] - d["t_gt"], axis=1) / np.linalg.norm(d["t_gt"], axis=1))[cert].max() < TOL_T
        # R is a proper rotation
        assert np.abs(np.linalg.det(r["R"][cert]) - 1).max() < 1e-12
        # permutation invariance
        perm = np.random.RandomState(0).permutation(n_p)
        d2 = dict(d)
        d2["pts_2d"], d2["pts_3d"] = d["pts_2d"][:, perm].copy(), d["pts_3d"][:, perm].copy()
        r2 = _solve(gpu, d2, n_p, n_l)
        both = cert & (r2["status"] == 0)
        assert synth.geodesic(r["R"], r2["R"])[both].max() < 1e-9


def test_per_problem_K_and_int_K(gpu):
    from cvxpnpl_amd import synth

    d = synth.make_pnp(70, 8, 0.0, seed=11)
    Kb = np.repeat(d["K"][None], 70, 0)
    d2 = dict(d)
    d2["K"] = Kb
    # a different camera per problem: scale the pixels of problem 5 and its K consistently
    d3 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in d2.items()}
    S = np.diag([2.0, 0.5, 1.0])
    d3["K"][5] = S @ d3["K"][5]
    d3["pts_2d"][5] = d3["pts_2d"][5] * np.array([2.0, 0.5])
    for layout in LAYOUTS.values():
        r1 = _solve(gpu, d, 8, 0, layout=layout)
        r2 = _solve(gpu, d2, 8, 0, layout=layout)
        assert np.array_equal(r1["R"], r2["R"]) and np.array_equal(r1["t"], r2["t"])
        r3 = _solve(gpu, d3, 8, 0, layout=layout)
        assert (r3["status"] == 0).all() and synth.geodesic(r3["R"], d["R_gt"]).max() < TOL_ROT


def test_edge_cases(gpu):
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    for layout in LAYOUTS.values():
        # degenerate: one point -> NaN pose, status 3 (reference: LinAlgError / NaN sentinel)
        d1 = synth.make_pnp(5, 1, 0.0, seed=1)
        r = _solve(gpu, d1, 1, 0, layout=layout)
        assert (r["status"] == 3).all() and np.isnan(r["R"]).all() and np.isnan(r["t"]).all()
        # NaN in one problem does not leak into its neighbours (same wavefront, same DPP row neighbours)
        d2 = synth.make_pnp(130, 6, 0.0, seed=2)
        d2["pts_2d"][64, 0, 0] = np.nan
        d2["pts_3d"][3, 2, 1] = np.inf
        r = _solve(gpu, d2, 6, 0, layout=layout)
        assert r["status"][64] == 3 and r["status"][3] == 3 and (np.delete(r["status"], [3, 64]) == 0).all()
        assert synth.geodesic(np.delete(r["R"], [3, 64], 0), np.delete(d2["R_gt"], [3, 64], 0)).max() < TOL_ROT
    # ragged batch sizes around the wavefront size, and empty batch
    for layout in LAYOUTS.values():
        for b in (1, 3, 4, 5, 63, 64, 65):
            d = synth.make_pnp(b, 10, 0.0, seed=b)
            r = _solve(gpu, d, 10, 0, layout=layout)
            assert (r["status"] == 0).all() and synth.geodesic(r["R"], d["R_gt"]).max() < TOL_ROT
    res = ca.pnp_batch(torch.zeros((0, 10, 2), device=gpu, dtype=torch.float64), torch.zeros((0, 10, 3), device=gpu, dtype=torch.float64),
                       torch.eye(3, device=gpu, dtype=torch.float64))
    assert res.R.shape == (0, 3, 3)
    # many correspondences through the small-problem kernels themselves (chunked staging in every layout; the API routes
    # such problems to the blocked assembly instead: tests/test_large_n.py): C ABI directly, against the blocked path
    import ctypes as C

    from cvxpnpl_amd import _lib
    d = synth.make_pnp(8, 500, 1.0, seed=4)
    ref = ca.pnp_batch(d["pts_2d"], d["pts_3d"], d["K"])
    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    p2, p3, Kd = tt(d["pts_2d"]), tt(d["pts_3d"]), tt(d["K"])
    for layout in LAYOUTS.values():
        R = torch.empty((8, 3, 3), dtype=torch.float64, device=gpu)
        t = torch.empty((8, 3), dtype=torch.float64, device=gpu)
        st = torch.empty((8,), dtype=torch.int32, device=gpu)
        o = _lib.default_opts(layout=layout)
        rc = _lib.lib().cvxpnpl_solve_batch(8, 500, C.c_void_p(p2.data_ptr()), C.c_void_p(p3.data_ptr()), 0, None, None, C.c_void_p(Kd.data_ptr()), 0,
                                            C.byref(o), C.c_void_p(R.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(st.data_ptr()), None, None, None, None, None)
        torch.cuda.synchronize()
        assert rc == 0 and (st.cpu().numpy() == 0).all()
        assert synth.geodesic(R.cpu().numpy(), ref.R.cpu().numpy()).max() < 1e-9
    # invalid arguments are launch-level errors
    with pytest.raises(ValueError):
        ca.pnp_batch(None, None, np.eye(3))


def test_minimal_problems_rank_gt1_flagged_and_recovered(gpu, orc):
    """Config 5 flavour: N=4 hypotheses.  Non-tight problems are flagged (status 1), their Z
    goes through the host multi-solution recovery; certified ones match the oracle."""
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_ransac(512, outlier_frac=0.3, seed=46)
    r = _solve(gpu, d, 4, 0, max_iters=600, want_Z=True)
    assert set(np.unique(r["status"])) <= {0, 1, 2, 4}
    assert (r["status"] == 0).mean() > 0.4
    idx = np.where(r["status"] == 0)[0][:24]
    o = orc.pnpl_batch(d["pts_2d"][idx], None, d["pts_3d"][idx], None, d["K"], eps=1e-11, max_iters=300000)
    ok = o["n_poses"] == 1
    assert synth.geodesic(r["R"][idx], o["R"][:, 0])[ok].max() < TOL_ROT
    # a flagged problem yields 2 or 4 poses through the drop-in API
    flagged = np.where(r["status"] == 1)[0]
    if len(flagged):
        import warnings

        i = int(flagged[0])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            poses = ca.pnp(d["pts_2d"][i], d["pts_3d"][i], d["K"], max_iters=600)
        assert len(poses) in (2, 4)


def test_ransac_wrapper_config5(gpu):
    """BASELINE config 5 flavour: one scene, 30 % outliers, minimal hypotheses, consensus + refit."""
    import torch

    from cvxpnpl_amd import synth
    from cvxpnpl_amd.ransac import ransac_pnp

    d = synth.make_ransac(1, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
    out = ransac_pnp(d["scene_2d"], d["scene_3d"], d["K"], n_hyp=2048, thresh=2.0, device=gpu)
    truth = torch.as_tensor(d["inlier"], device=gpu)
    assert out["n_inliers"] >= 0.95 * int(truth.sum())
    assert int((out["inliers"] & ~truth).sum()) <= 2  # clutter does not sneak in
    assert synth.geodesic(out["R"].cpu().numpy(), d["R_gt"]) < 2e-3  # 0.5 px noise on 70 inliers
    assert out["status"] == 0


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_pose_reuse_does_not_go_stale(gpu, orc, layout):
    """Regression (tools/fuzz_parity.py): problems with two local minima within the rounding tolerance of the
    pose-reuse shortcut ended uncertified in the wave layout; every layout certifies them and agrees with the
    oracle (see tests/test_device_algorithm_hostsim.py::test_pose_reuse_does_not_go_stale)."""
    from cvxpnpl_amd import synth

    for n_p, n_l, sigma, seed, i in [(9, 1, 2.0, 5012, 100), (5, 0, 0.0, 5005, 91), (0, 5, 2.0, 5023, 18)]:
        d = synth.make_pnpl(192, n_p, n_l, sigma, seed=seed)
        r = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout])
        assert r["status"][i] == 0, (n_p, n_l, r["status"][i], r["iters"][i])
        sl = slice(i, i + 1)
        o = orc.pnpl_batch(d["pts_2d"][sl] if n_p else None, d["line_2d"][sl] if n_l else None, d["pts_3d"][sl] if n_p else None,
                           d["line_3d"][sl] if n_l else None, d["K"], eps=1e-11, max_iters=200000)
        assert o["n_poses"][0] == 1
        assert synth.geodesic(r["R"][sl], o["R"][:, 0])[0] < TOL_ROT


def test_hybrid_queue_counters_alternate_between_launches(gpu):
    """The quad and lane schedules queue parked problems for resume_wave_kernel through two counters that
    alternate between launches on a stream (the resume kernel of one launch zeroes the counter of the next).
    Back-to-back launches of planar (everything queued) and ordinary batches, in both layouts and with
    changing sizes (workspace regrowth), must each reproduce the wave layout's result."""
    from cvxpnpl_amd import synth

    planar = synth.make_planar_pnp(900, 8, 0.5, seed=21, general=True)
    plain = synth.make_pnp(1500, 10, 1.0, seed=22)
    bigger = synth.make_pnp(4100, 6, 1.0, seed=23)
    ref = {id(d): _solve(gpu, d, d["pts_3d"].shape[1], 0, layout=LAYOUTS["wave"], max_iters=300) for d in (planar, plain, bigger)}
    seq = [(planar, "quad"), (plain, "quad"), (planar, "lane"), (planar, "quad"), (bigger, "lane"), (plain, "penta"), (bigger, "quad"),
           (planar, "penta"), (planar, "quad"), (bigger, "penta")]
    for d, layout in seq:
        r = _solve(gpu, d, d["pts_3d"].shape[1], 0, layout=LAYOUTS[layout], max_iters=300)
        w = ref[id(d)]
        same = r["status"] == w["status"]
        assert same.mean() > 0.995, (layout, same.mean())
        both = same & (r["status"] == 0)
        if both.any():
            assert synth.geodesic(r["R"], w["R"])[both].max() < 1e-7
        fl = same & (r["status"] == 1)  # rank > 1: a certified twin pair, or the NaN rounding at the iteration cap -- like the wave layout
        if fl.any():
            assert (np.isnan(r["R"][fl]).any(axis=(1, 2)) == np.isnan(w["R"][fl]).any(axis=(1, 2))).mean() > 0.995


@pytest.mark.gpu
def test_interior_point_rescue_matches_first_order_solve(gpu):
    """opts.rescue_from: the problems still open after that many first-order iterations are finished by the interior-point path
    (cvxw::rescue_wave_kernel).  Same SDP, same rounding / polish / certificate: every problem certified both ways has the same
    pose, nothing certified is lost, the iteration counts are bounded by rescue_from + the ~12 second-order iterations + a
    few, and a launch leaves no problem pending -- in every layout, on minimal problems (slow for the first-order iteration)."""
    from cvxpnpl_amd import synth

    d = synth.make_pnp(3000, 4, 2.0, seed=3)
    ref = _solve(gpu, d, 4, 0, layout=LAYOUTS["wave"], max_iters=2500, rescue_from=0)
    assert ref["iters"].max() > 400
    for layout in ("wave", "quad", "lane", "penta"):
        for rf in (48, 0):
            r = _solve(gpu, d, 4, 0, layout=LAYOUTS[layout], max_iters=2500, rescue_from=rf)
            assert (r["status"] < 5).all() and (r["status"] >= 0).all()
            both = (r["status"] == 0) & (ref["status"] == 0)
            assert both.mean() > 0.97
            assert synth.geodesic(r["R"], ref["R"])[both].max() < 1e-6
            assert (np.linalg.norm(r["t"] - ref["t"], axis=1) / np.linalg.norm(ref["t"], axis=1))[both].max() < 1e-6
            if rf:
                assert (r["status"] == 0).sum() >= (ref["status"] == 0).sum()
                assert r["iters"].max() <= rf + 120, r["iters"].max()
                assert (r["iters"] > rf).sum() > 20  # the path was taken
    # the default options have it on (-1: 32 iterations for problems with at most 6 correspondences, 64 for 7, 128 otherwise)
    r = _solve(gpu, d, 4, 0, max_iters=2500)
    assert r["iters"].max() <= 32 + 120 and (r["iters"] > 32).sum() > 20 and (r["status"] == 0).sum() >= (ref["status"] == 0).sum()
    with pytest.raises(RuntimeError, match="bad options"):
        _solve(gpu, d, 4, 0, rescue_from=-2)
    # lines, points + lines, and the cost seam (cvxpnpl_solve_cost_batch: the problem is re-assembled from Q45 / B27 on the way)
    import torch

    import cvxpnpl_amd as ca

    for n_p, n_l, seed in ((0, 4, 11), (2, 2, 12), (3, 1, 13)):
        dl = synth.make_pnpl(1500, n_p, n_l, 2.0, seed=seed)
        a = _solve(gpu, dl, n_p, n_l, max_iters=2500, rescue_from=0)
        b = _solve(gpu, dl, n_p, n_l, max_iters=2500, rescue_from=40)
        both = (a["status"] == 0) & (b["status"] == 0)
        assert (b["status"] < 5).all() and (b["iters"] > 40).sum() > 5 and both.mean() > 0.9
        assert (b["status"] == 0).sum() >= (a["status"] == 0).sum() - 2
        assert synth.geodesic(a["R"], b["R"])[both].max() < 1e-6
    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    Bt, Qt = ca.assemble_batch(tt(d["pts_2d"]), None, tt(d["pts_3d"]), None, tt(d["K"]))
    for layout in ("wave", "quad", "lane"):
        c = {k: v.cpu().numpy() for k, v in ca.solve_cost_batch(Qt, Bt, max_iters=2500, rescue_from=48, layout=LAYOUTS[layout]).items()}
        both = (c["status"] == 0) & (ref["status"] == 0)
        assert (c["status"] < 5).all() and c["iters"].max() <= 48 + 120 and both.mean() > 0.97
        assert synth.geodesic(c["R"], ref["R"])[both].max() < 1e-6


def test_pack_results_kernel_matches_host_packing(gpu):
    """cvxpnpl_pack_results (the [n,13] records the multi-GPU gather exchanges) is bit-identical to the torch
    packing the gloo tests use on host tensors; ragged size, statuses 0..4, NaN poses kept."""
    import torch

    from cvxpnpl_amd import dist as cd

    g = torch.Generator().manual_seed(3)
    n = 1237
    R = torch.randn((n, 3, 3), generator=g, dtype=torch.float64)
    t = torch.randn((n, 3), generator=g, dtype=torch.float64)
    st = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32)
    R[5] = float("nan")
    host = cd.pack_results(R, t, st)
    devp = cd.pack_results(R.to(gpu), t.to(gpu), st.to(gpu))
    assert devp.shape == (n, cd.PACK) and devp.is_cuda
    a, b = host.numpy(), devp.cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True)
    R2, t2, st2 = cd.unpack_results(devp)
    assert torch.equal(st2.cpu(), st) and torch.equal(t2.cpu(), t)
    assert cd.pack_results(R[:0].to(gpu), t[:0].to(gpu), st[:0].to(gpu)).shape == (0, cd.PACK)


def _score_numpy(R, t, K, x, X, thresh, status=None, usable=(0, 2)):
    """Plain restatement of the inlier rule of include/cvxpnpl_amd.h (cvxpnpl_score_hypotheses); returns
    (mask [H,M], margin [H,M]) where margin is the distance of the decision from its threshold."""
    Xc = np.einsum("hij,mj->hmi", R, X) + t[:, None, :]
    uvw = np.einsum("ij,hmj->hmi", K, Xc)
    with np.errstate(all="ignore"):
        uv = uvw[..., :2] / uvw[..., 2:3]
        err = np.linalg.norm(uv - x[None], axis=-1)
    mask = (err < thresh) & (Xc[..., 2] > 0)
    if status is not None:
        mask &= np.isin(status, usable)[:, None]
    return mask, np.minimum(np.abs(err - thresh), np.abs(Xc[..., 2]))


@pytest.mark.parametrize("n_hyp,n_corr", [(1, 100), (300, 100), (5000, 37), (70, 1300)])
def test_score_hypotheses_kernel(gpu, n_hyp, n_corr):
    """HIP scoring kernel vs a numpy restatement of the same rule: identical masks and counts (decisions
    closer than 1e-9 px to the threshold excepted), statuses filtered, NaN poses score 0, LDS tiling (M > 512)."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_ransac(1, n_corr=n_corr, outlier_frac=0.3, sigma=0.5, seed=46 + n_corr)
    rs = np.random.RandomState(n_hyp)
    # hypotheses: the true pose perturbed by 0 .. 2 degrees / 0 .. 5 % translation, some wild, some NaN
    Rg, tg = d["R_gt"], d["t_gt"]
    w = rs.randn(n_hyp, 3) * rs.uniform(0, 0.03, (n_hyp, 1))
    th = np.linalg.norm(w, axis=1, keepdims=True) + 1e-300
    k = w / th
    Kx = np.zeros((n_hyp, 3, 3))
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0], Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    dR = np.eye(3)[None] + np.sin(th)[..., None] * Kx + (1 - np.cos(th))[..., None] * (Kx @ Kx)
    R = dR @ Rg[None]
    t = tg[None] * (1 + rs.randn(n_hyp, 1) * 0.02)
    status = rs.choice([0, 0, 0, 1, 2, 3, 4], n_hyp).astype(np.int32)
    if n_hyp > 10:
        R[3] = synth.random_poses(rs, 1)[0][0]  # unrelated pose
        R[5, 1, 1] = np.nan
        t[7, 0] = np.nan
        t[9] = -tg  # behind the camera
    cnt, mask = ca.score_hypotheses(torch.as_tensor(R, device=gpu), torch.as_tensor(t, device=gpu), d["K"], d["scene_2d"], d["scene_3d"],
                                    thresh=2.0, status=torch.as_tensor(status, device=gpu), want_mask=True)
    ref, margin = _score_numpy(R, t, d["K"], d["scene_2d"], d["scene_3d"], 2.0, status)
    mask, cnt = mask.cpu().numpy().astype(bool), cnt.cpu().numpy()
    clear = ~(margin < 1e-9) | ~np.isin(status, (0, 2))[:, None]
    assert (mask == ref)[clear | ~np.isfinite(margin)].all()
    assert (cnt == mask.sum(1)).all()
    assert ref.sum() > 0
    if n_hyp > 10:
        assert cnt[5] == 0 and cnt[7] == 0 and cnt[9] == 0
    # without statuses every finite hypothesis is scored
    cnt2 = ca.score_hypotheses(torch.as_tensor(R, device=gpu), torch.as_tensor(t, device=gpu), d["K"], d["scene_2d"], d["scene_3d"], thresh=2.0)
    ref2, _ = _score_numpy(R, t, d["K"], d["scene_2d"], d["scene_3d"], 2.0)
    assert np.abs(cnt2.cpu().numpy() - ref2.sum(1)).max() <= 1


def test_planar_scene_returns_both_poses_through_dropin_api(gpu):
    """A planar scene through cvxpnpl_amd.pnp: two poses, like the reference's rank-2 branch, the true
    one among them."""
    import warnings

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_pnp(4, 8, 0.0, seed=3)
    d["pts_3d"][:, :, 2] = 0.0
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"])
    for i in range(4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            poses = ca.pnp(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        assert len(poses) == 2
        assert min(synth.geodesic(R, d["R_gt"][i]) + np.linalg.norm(t - d["t_gt"][i]) for R, t in poses) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [1, 2, 3, 4])
def test_planar_batch_is_certified_two_fold_in_few_iterations(gpu, orc, layout):
    """Planar scenes are exactly two-fold ambiguous for the relaxation (R and R diag(-1,-1,1) have the
    same cost), so the solution is rank 2 (cvxpnpl.py:509-545).  The parity-even dual correction
    certifies the pair early: status 1 (rank > 1), both poses from the host recovery, and a median
    iteration count an order of magnitude below the uncertified ~130."""
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_pnp(2000, 10, 0.0, seed=11)
    d["pts_3d"][:, :, 2] = 0.0
    rs = np.random.RandomState(2)
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"]) + rs.normal(scale=0.5, size=d["pts_2d"].shape)
    res = ca.pnp_batch(d["pts_2d"], d["pts_3d"], d["K"], want_Z=True, layout=layout)
    status, iters = res.status.cpu().numpy(), res.iters.cpu().numpy()
    assert (status == 1).all()
    assert np.median(iters) <= 20
    # R, t of a rank > 1 exit are never NaN (round 1: 355 of 20 000 planar problems were) and are a proper rotation
    Rn, tn = res.R.cpu().numpy(), res.t.cpu().numpy()
    assert np.isfinite(Rn).all() and np.isfinite(tn).all()
    assert np.abs(Rn @ np.swapaxes(Rn, 1, 2) - np.eye(3)).max() < 1e-9
    # both poses against the oracle's rank-2 branch (cvxpnpl.py:221-343).  The reference divides by the
    # last entry of the top eigenvector (cvxpnpl.py:236), which is ~0 for about half of the exactly
    # degenerate planar spectra (eigenvalues 2, 2): it raises LinAlgError there and the oracle returns
    # NaN; our recovery picks the pivot eigenvector by magnitude and must still contain the true pose.
    Bt, Qt = ca.assemble_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"])
    Z, Bt, Qt = res.Z.cpu().numpy(), Bt.cpu().numpy(), Qt.cpu().numpy()
    n_cmp = 0
    for i in range(0, 2000, 80):
        poses = ca.recover_multi(Z[i], Bt[i], Qt[i])
        assert len(poses) == 2
        assert min(synth.geodesic(R, d["R_gt"][i]) for R, t in poses) < 0.1  # 0.5 px noise on a plane
        ref_poses, _ = orc.pnp(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        if any(np.isnan(R).any() for R, t in ref_poses):
            continue
        n_cmp += 1
        for R, t in poses:
            assert min(synth.geodesic(R, Ro) + np.linalg.norm(t - to) for Ro, to in ref_poses) < 5e-6
    assert n_cmp >= 5
    # the batched host path returns the same poses for the whole batch at once
    Rb, tb, cnt = ca.recover_multi_batch(res, Bt, Qt)
    assert np.isin(cnt, (2, 4)).all() and (cnt == 2).mean() > 0.95  # (a few leave through res_tol with rank 3-4)
    for i in range(0, 2000, 80):
        poses = ca.recover_multi(Z[i], Bt[i], Qt[i])
        assert all(np.array_equal(Rb[i, k], poses[k][0]) and np.array_equal(tb[i, k], poses[k][1]) for k in range(2))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [1, 2, 3, 4])
def test_planar_scene_in_a_general_frame(gpu, layout):
    """A plane that is not Z = 0 (random plane, random offset per problem): the kernels detect the direction the
    cost is blind to and solve in the frame whose third axis it is (the first phase of the hybrid schedules
    hands such problems to the wave kernel at once).  Same outcome as for Z = 0: certified pair after a
    handful of iterations, the true pose among the two recovered ones, everything in the caller's frame."""
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_planar_pnp(1500, 10, 0.0, seed=3, general=True)
    res = ca.pnp_batch(d["pts_2d"], d["pts_3d"], d["K"], want_Z=True, layout=layout)
    status, iters = res.status.cpu().numpy(), res.iters.cpu().numpy()
    assert (status == 1).mean() > 0.995
    assert np.median(iters) <= 20
    Bt, Qt = ca.assemble_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"])
    R, t, cnt = ca.recover_multi_batch(res, Bt, Qt)
    sel = np.where((status == 1) & (cnt == 2))[0][::10]
    assert len(sel) > 100
    err = [min(synth.geodesic(R[i, k], d["R_gt"][i]) + np.linalg.norm(t[i, k] - d["t_gt"][i]) for k in range(2)) for i in sel]
    assert max(err) < 1e-7
    Rone = res.R.cpu().numpy()
    assert max(min(synth.geodesic(Rone[i], R[i, k]) for k in range(2)) for i in sel) < 1e-7  # the returned pose is one of the twins


def test_ransac_default_device_and_workspace_entry_points(gpu):
    """ransac_pnp without `device` (advisor finding: the default used to raise), and the workspace entry points:
    a caller-registered (torch-allocated) workspace gives the same results as the library's own, a too-small one is
    refused, release frees the cached allocation."""
    import ctypes as C

    import torch

    from cvxpnpl_amd import _lib, synth
    from cvxpnpl_amd.ransac import ransac_pnp

    d = synth.make_ransac(1, n_corr=60, outlier_frac=0.3, sigma=0.5, seed=5)
    out = ransac_pnp(d["scene_2d"], d["scene_3d"], d["K"], n_hyp=1024)  # numpy in, no device
    assert out["n_inliers"] >= 0.9 * int(d["inlier"].sum())
    out2 = ransac_pnp(torch.as_tensor(d["scene_2d"], device=gpu), torch.as_tensor(d["scene_3d"], device=gpu), d["K"], n_hyp=1024)
    assert out2["R"].device == gpu
    L = _lib.lib()
    dd = synth.make_planar_pnp(700, 8, 0.5, seed=21, general=True)  # planar: every problem goes through the queue
    stream = C.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    ref = _solve(gpu, dd, 8, 0, layout=LAYOUTS["quad"], max_iters=200)
    need = L.cvxpnpl_workspace_bytes(700)
    assert need > 700 * 56 * 8
    buf = torch.empty(need, dtype=torch.uint8, device=gpu)
    assert L.cvxpnpl_set_workspace(C.c_void_p(buf.data_ptr()), need, stream) == 0
    big = synth.make_pnp(5000, 6, 1.0, seed=3)
    try:
        for layout in ("quad", "lane", "quad"):
            r = _solve(gpu, dd, 8, 0, layout=LAYOUTS[layout], max_iters=200)
            assert (r["status"] == ref["status"]).mean() > 0.995
            # the queue is self-cleaning: its three counters are back at zero and every entry at -1 after each launch
            # (both queues: the resume queue and the rescue queue of the interior-point path -- planar problems need ~140 first-order
            # iterations, so most of them go through it here)
            head = buf[:256].view(torch.int32).cpu().numpy()
            assert (head[:3] == 0).all() and (head[16:19] == 0).all(), (head[:3], head[16:19])
            qbytes = (4 * (700 + 2048) + 255) // 256 * 256
            entries = buf[256:256 + 2 * qbytes].view(torch.int32).cpu().numpy()
            assert (entries == -1).all()
            assert (r["status"] < 5).all()  # nothing left pending
        with pytest.raises(RuntimeError, match="workspace"):
            _solve(gpu, big, 6, 0, layout=LAYOUTS["quad"])
    finally:
        assert L.cvxpnpl_set_workspace(C.c_void_p(0), 0, stream) == 0  # back to the library's own allocation
    r = _solve(gpu, big, 6, 0, layout=LAYOUTS["quad"])
    assert (r["status"] == 0).mean() > 0.99
    assert L.cvxpnpl_release_workspace(stream, 1) == 0
    r2 = _solve(gpu, big, 6, 0, layout=LAYOUTS["quad"])
    assert np.array_equal(r["R"], r2["R"])


def test_device_rank_gt1_recovery(gpu, golden, orc):
    """cvxpnpl_recover_multi_device (cvxpnpl.py:221-343, :156-218 on the GPU): the reference's own outputs for injected
    rank-2 / rank-4 / rank-1 solutions (golden G6, G7: _constraint_ortho_det + _re6q3 + the SVD projection), and the host
    path on HIP-produced Z (minimal problems cut short: ranks 2..6; planar scenes: certified pairs)."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth
    from cvxpnpl_amd.api import BatchResult

    # golden: x -> poses through the reference's _solve_relaxation (tests/golden/make_golden.py, G6)
    tags = ("r1", "r1p", "r2", "r4")
    Z = torch.as_tensor(np.stack([golden[f"g6_{t}_x"] for t in tags]), device=gpu)
    B = torch.as_tensor(np.tile(golden["g3_pnp_B"].reshape(1, 27), (4, 1)), device=gpu)
    res = BatchResult(Z=Z, status=torch.ones(4, dtype=torch.int32, device=gpu))
    R, t, cnt = ca.recover_multi_device(res, B)
    R, t, cnt = R.cpu().numpy(), t.cpu().numpy(), cnt.cpu().numpy()
    for k, tag in enumerate(tags):
        Rg, tg = golden[f"g6_{tag}_R"], golden[f"g6_{tag}_t"]
        assert cnt[k] == len(Rg), (tag, cnt[k])
        for i in range(cnt[k]):
            assert min(geodesic_np(R[k, i], Rg[j]) + np.abs(t[k, i] - tg[j]).max() for j in range(len(Rg))) < 1e-8, tag
    # NaN solution: -1 poses (reference: NaN sentinel)
    res = BatchResult(Z=torch.full((1, 55), float("nan"), dtype=torch.float64, device=gpu), status=torch.ones(1, dtype=torch.int32, device=gpu))
    assert int(ca.recover_multi_device(res, B[:1])[2][0]) == -1
    # HIP-produced Z: device == host, with and without the Newton polish
    for d, n_p, kw in ((synth.make_pnp(600, 4, 1.0, seed=9), 4, dict(max_iters=6, first_check=1000)),
                       (synth.make_planar_pnp(400, 8, 0.5, seed=21, general=True), 8, {})):
        tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
        r = ca.pnp_batch(tt(d["pts_2d"]), tt(d["pts_3d"]), tt(d["K"]), want_Z=True, **kw)
        Bt, Qt = ca.assemble_batch(tt(d["pts_2d"]), None, tt(d["pts_3d"]), None, tt(d["K"]))
        st = r.status.cpu().numpy()
        assert (st == 1).sum() > 100
        for Q in (None, Qt):
            Rh, th, ch = ca.recover_multi_batch(r, Bt, Q)
            Rd, td, cd_ = ca.recover_multi_device(r, Bt, Q)
            Rd, td, cd_ = Rd.cpu().numpy(), td.cpu().numpy(), cd_.cpu().numpy()
            assert np.array_equal(ch, cd_)
            assert (cd_[st != 1] == 0).all() and np.isin(cd_[st == 1], (2, 4, -1)).all()
            # same source, two compilers (fused multiply-adds differ): the pose SETS agree (the root finder may list the four
            # poses of the rank-4 branch in another order); typical 1e-14.  Odd ranks pad the basis with an eigenvector from
            # a degenerate null space (cvxpnpl.py:231-233) where any rounding difference changes the answer -- for the
            # reference itself too (test_uncertified_exits_follow_reference_recovery) -- and are only counted.
            Zn = r.Z.cpu().numpy()
            n_ok = n_bad = 0
            for i in np.where(st == 1)[0]:
                if cd_[i] <= 0:
                    continue
                rank = int((np.linalg.eigvalsh(orc.vech10_inv(Zn[i])) > 1e-3).sum())
                e = max(min(geodesic_np(Rd[i, k], Rh[i, j]) + np.abs(td[i, k] - th[i, j]).max() for j in range(ch[i])) for k in range(cd_[i]))
                if rank in (2, 4):
                    assert e < 1e-6, (i, rank, e)
                    n_ok += 1
                else:
                    n_bad += e > 1e-6
            assert n_ok > 50


def test_solve_is_hipgraph_capturable_and_replayable(gpu):
    """A solve (quad kernel + queue-driven resume kernel, and the lane schedule) captured into a hipGraph and replayed:
    same results every replay.  The resume queue cleans itself inside the launch, so a replay -- which repeats the
    launches with the very same arguments -- finds it as the first run did (round 1's two counters alternating under
    host-side bookkeeping could not be replayed)."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    planar = synth.make_planar_pnp(3000, 8, 0.5, seed=21, general=True)  # every problem goes through the queue
    plain = synth.make_pnp(5000, 10, 2.0, seed=22)
    s = torch.cuda.Stream(gpu)
    for d, layout in ((planar, 3), (plain, 3), (plain, 1)):
        p2, p3, K = (torch.as_tensor(d[k], device=gpu) for k in ("pts_2d", "pts_3d", "K"))
        with torch.cuda.stream(s):
            ref = ca.pnp_batch(p2, p3, K, layout=layout, max_iters=300)  # warm-up on the capture stream: workspace allocated
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            res = ca.pnp_batch(p2, p3, K, layout=layout, max_iters=300)
        for _ in range(3):
            res.R.zero_()
            res.status.fill_(-7)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(res.status, ref.status)
            assert torch.equal(res.R, ref.R) and torch.equal(res.t, ref.t)


def test_single_problem_staging_equals_the_batch_entry_point(gpu):
    """pnp / pnl / pnpl go through one pinned staging buffer each way (api._SingleCtx): bit-identical to the batch entry point on the same problem,
    for every shape, repeated calls and interleaved shapes (the buffers are cached per shape)."""
    from cvxpnpl_amd import api, synth

    for rep in range(2):
        for n_p, n_l, seed in ((6, 0, 1), (0, 6, 2), (5, 5, 3), (10, 0, 4), (4, 0, 5)):
            d = synth.make_pnpl(3, n_p, n_l, 1.0, seed=seed)
            for i in range(3):
                p2 = d["pts_2d"][i:i + 1] if n_p else None
                p3 = d["pts_3d"][i:i + 1] if n_p else None
                l2 = d["line_2d"][i:i + 1] if n_l else None
                l3 = d["line_3d"][i:i + 1] if n_l else None
                a = api._single_fast(p2, l2, p3, l3, d["K"], 1e-9, 2500)
                b = api.pnpl_batch(p2, l2, p3, l3, d["K"], want_Z=True, res_tol=0.0)
                for k in ("R", "t", "status", "iters", "cost", "work", "Z"):
                    assert np.array_equal(a[k].numpy(), b[k].cpu().numpy(), equal_nan=True), (n_p, n_l, i, k)
    with pytest.raises(ValueError):
        api.pnp(np.zeros((5, 2)), np.zeros((6, 3)), np.eye(3))


def test_cost_seam_staging_equals_the_batch_entry_point(gpu):
    """solve_relaxation / solve_relaxation_rc (the reference's private seam, one problem) through the staging buffers: bit-identical to
    cvxpnpl_solve_cost_batch on the same cost, both constraint sets."""
    from cvxpnpl_amd import api, synth

    d = synth.make_pnpl(4, 8, 0, 1.0, seed=11)
    Bt, Qt = api.assemble_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"])
    Bn, Qn = Bt.cpu().numpy(), Qt.cpu().numpy()
    for variant in (0, 1):
        for i in range(4):
            a = api._single_cost_fast(Qn[i], Bn[i].reshape(27), 1e-9, 2500, variant)
            b = api.solve_cost_batch(Qn[i:i + 1], Bn[i:i + 1].reshape(1, 27), want_Z=True, variant=variant, res_tol=0.0)
            for k in ("R", "t", "status", "iters", "cost", "work", "Z"):
                assert np.array_equal(a[k].numpy(), b[k].cpu().numpy(), equal_nan=True), (variant, i, k)
