"""The device algorithm (cvxpnpl_amd/csrc/solver_core.h) stepped on the CPU through the
test-only host build, checked against the oracle and the reference's golden vectors.
No GPU needed; the GPU parity tests (test_gpu_parity.py) run the same checks through the
C ABI on the HIP path."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostsim  # noqa: E402
from cvxpnpl_amd import synth  # noqa: E402

# rotation geodesic / relative translation tolerance vs the oracle (float64 path; the oracle's
# own solve is converged to ~1e-11 residuals)
TOL_ROT = 1e-6
TOL_T = 1e-6


def unpack55(v):
    M = np.zeros((10, 10))
    k = 0
    for i in range(10):
        for j in range(i, 10):
            M[i, j] = M[j, i] = v[k]
            k += 1
    return M


def pack55(M):
    return np.array([M[i, j] for i in range(10) for j in range(i, 10)])


def test_assembly_matches_reference_g3(golden):
    """Gram/Schur assembly == the reference's A^T A and B (captured from the reference)."""
    rc, B, Q = hostsim.assemble(golden["ex_pnp_pts2d"], golden["ex_pnp_pts3d"], None, None, golden["ex_pnp_K"])
    A = golden["g3_pnp_A"]
    assert rc == 0
    np.testing.assert_allclose(B, golden["g3_pnp_B"], atol=1e-11)
    np.testing.assert_allclose(Q, A.T @ A, atol=1e-13)
    rc, B, Q = hostsim.assemble(None, None, golden["ex_pnl_line2d"], golden["ex_pnl_line3d"], golden["ex_pnl_K"])
    A = golden["g3_pnl_A"]
    np.testing.assert_allclose(B, golden["g3_pnl_B"], atol=1e-11)
    np.testing.assert_allclose(Q, A.T @ A, atol=1e-13)
    rc, B, Q = hostsim.assemble(golden["ex_pnpl_pts2d"], golden["ex_pnpl_pts3d"], golden["ex_pnpl_line2d"], golden["ex_pnpl_line3d"],
                                golden["ex_pnpl_K"])
    A = golden["g3_pnpl_A"]
    np.testing.assert_allclose(B, golden["g3_pnpl_B"], atol=1e-11)
    np.testing.assert_allclose(Q, A.T @ A, atol=1e-13)
    # the cost vector handed to scs: c = vech(Q, 2) (cvxpnpl.py:486)
    Q10 = np.zeros((10, 10))
    Q10[:9, :9] = Q
    c = np.array([(1 if i == j else 2) * Q10[i, j] for j in range(10) for i in range(j, 10)])
    np.testing.assert_allclose(c, golden["g3_pnpl_c"], atol=1e-13)


def test_affine_projection_matches_reference_constraints(golden):
    """The closed-form projector == dense projection onto {x : E x = b} built from the
    reference's own _A, _b (first 22 rows), in the Frobenius geometry of symmetric matrices."""
    A, b = golden["g4_A"][:22], golden["g4_b"][:22]
    # reference x = vech (column-major lower) == our packing (row-major upper); weights: off-diagonals count twice
    wgt = np.array([1.0 if i == j else 2.0 for i in range(10) for j in range(i, 10)])
    Aw = A / wgt  # <A_i, Z>_F = sum wgt * a_ij z_ij  -> a = A / wgt
    rs = np.random.RandomState(0)
    for homog in (False, True):
        x = rs.normal(size=55)
        bb = np.zeros(22) if homog else b
        # minimise sum wgt (y - x)^2 s.t. A y = bb
        Wi = 1.0 / wgt
        lam = np.linalg.lstsq(A @ (Wi[:, None] * A.T), A @ x - bb, rcond=None)[0]
        y = x - Wi * (A.T @ lam)
        np.testing.assert_allclose(hostsim.proj_affine(x, homog), y, atol=1e-13)
        np.testing.assert_allclose(A @ hostsim.proj_affine(x, homog), bb, atol=1e-13)


def test_psd_projection_matches_numpy():
    rs = np.random.RandomState(1)
    for trial in range(20):
        M = rs.normal(size=(10, 10))
        M = M + M.T
        if trial % 4 == 0:  # clustered / rank deficient spectra
            q, _ = np.linalg.qr(rs.normal(size=(10, 10)))
            M = (q * np.array([3, 1, 1, 1e-9, 0, 0, -1e-9, -1, -1, -2.0])) @ q.T
        w, v = np.linalg.eigh(M)
        Mp = (v * np.maximum(w, 0)) @ v.T
        Wp, lam, sweeps = hostsim.pospart(pack55(M))
        np.testing.assert_allclose(unpack55(Wp), Mp, atol=2e-14 * max(1.0, np.abs(w).max()))
        np.testing.assert_allclose(np.sort(lam), w, atol=2e-14 * max(1.0, np.abs(w).max()))
        assert sweeps <= 12


CASES = [  # (n_p, n_l, sigma, batch)
    (10, 0, 0.0, 128), (10, 0, 2.0, 128), (5, 5, 0.0, 64), (5, 5, 1.0, 64), (0, 6, 1.0, 48), (6, 0, 1.0, 48),
]


@pytest.mark.parametrize("n_p,n_l,sigma,batch", CASES)
def test_hostsim_vs_oracle(orc, n_p, n_l, sigma, batch):
    d = synth.make_pnpl(batch, n_p, n_l, sigma, seed=100 + n_p + 7 * n_l)
    args = (d["pts_2d"] if n_p else None, d["pts_3d"] if n_p else None, d["line_2d"] if n_l else None, d["line_3d"] if n_l else None)
    hs = hostsim.solve_batch(*args, d["K"])
    nb = min(batch, 32)
    o = orc.pnpl_batch(*(a[:nb] if a is not None else None for a in (args[0], args[2], args[1], args[3])), d["K"],
                       eps=1e-11, max_iters=200000)
    ok = (hs["status"][:nb] == 0) & (o["n_poses"] == 1)
    assert ok.mean() > 0.9
    geo = synth.geodesic(hs["R"][:nb], o["R"][:, 0])
    terr = np.linalg.norm(hs["t"][:nb] - o["t"][:, 0], axis=1) / np.linalg.norm(o["t"][:, 0], axis=1)
    assert geo[ok].max() < TOL_ROT, geo[ok].max()
    assert terr[ok].max() < TOL_T
    # certified problems carry a real certificate: 0 <= cost - dobj <= eps
    c = hs["cost"][hs["status"] == 0]
    assert (c[:, 0] - c[:, 1] >= -1e-15).all() and (c[:, 0] - c[:, 1] <= 1e-9).all()
    if sigma == 0.0:
        assert synth.geodesic(hs["R"], d["R_gt"])[hs["status"] == 0].max() < TOL_ROT
        assert (hs["status"] == 0).mean() > 0.98


def test_examples_known_answer(golden):
    """examples/pnp.py, pnl.py, pnpl.py through the device algorithm (BASELINE config 1)."""
    for name, args in (
        ("pnp", (golden["ex_pnp_pts2d"][None], golden["ex_pnp_pts3d"][None], None, None)),
        ("pnl", (None, None, golden["ex_pnl_line2d"][None], golden["ex_pnl_line3d"][None])),
        ("pnpl", (golden["ex_pnpl_pts2d"][None], golden["ex_pnpl_pts3d"][None], golden["ex_pnpl_line2d"][None], golden["ex_pnpl_line3d"][None])),
    ):
        hs = hostsim.solve_batch(*args, golden[f"ex_{name}_K"])
        assert hs["status"][0] == 0
        assert synth.geodesic(hs["R"][0], golden[f"ex_{name}_R"]) < 1e-6  # literals carry 8 digits
        tg = golden[f"ex_{name}_t"]
        assert np.linalg.norm(hs["t"][0] - tg) / np.linalg.norm(tg) < 1e-6


def test_certified_Z_satisfies_reference_constraints(golden):
    """Z returned for certified problems is feasible for the reference's SDP data."""
    d = synth.make_pnp(16, 10, 1.0, seed=5)
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    A, b = golden["g4_A"][:22], golden["g4_b"][:22]
    for i in range(16):
        assert hs["status"][i] == 0
        np.testing.assert_allclose(A @ hs["Z"][i], b, atol=1e-12)
        assert np.linalg.eigvalsh(unpack55(hs["Z"][i])).min() > -1e-12


def test_uncertifiable_and_degenerate_inputs():
    # minimal N=4 problems: some are not tight -> rank>1 flagged, never "certified" wrongly
    d = synth.make_pnp(96, 4, 1.0, seed=9)
    o = hostsim.default_opts(max_iters=400)
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=o, want_Z=True)
    assert set(np.unique(hs["status"])) <= {0, 1, 2, 4}
    assert (hs["status"] == 0).mean() > 0.5
    nz = hs["status"] != 0
    assert (hs["iters"][nz] <= 400).all()
    # fewer than the 2 bearings needed for N^T N to be invertible: NaN pose (reference: LinAlgError)
    d1 = synth.make_pnp(4, 1, 0.0, seed=1)
    hs = hostsim.solve_batch(d1["pts_2d"], d1["pts_3d"], None, None, d1["K"])
    assert (hs["status"] == 3).all() and np.isnan(hs["R"]).all() and np.isnan(hs["t"]).all()
    # NaN input
    d2 = synth.make_pnp(4, 6, 0.0, seed=2)
    d2["pts_2d"][1, 0, 0] = np.nan
    hs = hostsim.solve_batch(d2["pts_2d"], d2["pts_3d"], None, None, d2["K"])
    assert hs["status"][1] == 3 and np.isnan(hs["R"][1]).all()
    assert (hs["status"][[0, 2, 3]] == 0).all()


def test_planar_scenes_certify_the_two_fold_pair_early():
    """Planar scenes: R and R diag(-1,-1,1) have equal cost, the SDP solution is rank 2
    (cvxpnpl.py:509-545).  The parity-even dual correction (solver_core.h, dual_certificate, symm)
    certifies the pair after a handful of iterations instead of ~130."""
    from cvxpnpl_amd import synth

    d = synth.make_pnp(400, 10, 0.0, seed=1)
    d["pts_3d"][:, :, 2] = 0.0
    rs = np.random.RandomState(5)
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"]) + rs.normal(scale=1.0, size=d["pts_2d"].shape)
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"])
    assert (hs["status"] == 1).all()
    assert np.median(hs["iters"]) <= 20


def _sym_from_row(a):
    """row of the reference's equality block (coefficients over vech) -> symmetric 10x10 with <A, Z>_F = a . vech(Z)"""
    M = np.zeros((10, 10))
    k = 0
    for i in range(10):
        for j in range(i, 10):
            M[i, j] = M[j, i] = a[k] if i == j else a[k] / 2.0
            k += 1
    return M


def test_closed_form_dual_multipliers_match_reference_constraints(golden):
    """dual_lambda: the minimum-norm correction dS in span{A_i} with dS z = rhs, computed through the constant
    10x10 system in the frame of R, equals the dense least-norm solution built from the reference's own
    _A (cvxpnpl.py:387-451, golden G4) -- for arbitrary rotations."""
    A = golden["g4_A"][:22]
    mats = [_sym_from_row(a) for a in A]
    # orthonormal basis of span{A_i} in the Frobenius inner product
    V = np.array([m.reshape(-1) for m in mats]).T  # 100 x 22
    u, sv, _ = np.linalg.svd(V, full_matrices=False)
    basis = [u[:, k].reshape(10, 10) for k in range(22) if sv[k] > 1e-10 * sv[0]]
    assert len(basis) == 21  # the 22 rows have one dependency: row sums and column sums of the diagonal block
    rs = np.random.RandomState(3)
    for trial in range(6):
        Rm, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        if np.linalg.det(Rm) < 0:
            Rm[:, 0] = -Rm[:, 0]
        z = np.concatenate([Rm.T.reshape(-1), [1.0]])  # vec_colmajor(R); 1
        # a consistent right-hand side: rhs = G z with G in the span
        G = sum(c * b for c, b in zip(rs.normal(size=21), basis))
        rhs = G @ z
        # dense: dS = sum c_k B_k, minimise |c| s.t. (sum c_k B_k) z = rhs
        Mz = np.array([b @ z for b in basis]).T  # 10 x 21, rank 7: its left null space is the tangent space of SO(3) at R
        assert np.linalg.matrix_rank(Mz, tol=1e-9) == 7
        c = np.linalg.pinv(Mz, rcond=1e-10) @ rhs
        dS_ref = sum(ck * b for ck, b in zip(c, basis))
        assert np.abs(dS_ref @ z - rhs).max() < 1e-12
        lam = hostsim.dual_lambda(Rm, rhs)
        E = 0.5 * (np.outer(lam, z) + np.outer(z, lam))
        dS = unpack55(pack55(E) - hostsim.proj_affine(pack55(E), True))  # P_range(E) = E - P_null(E)
        np.testing.assert_allclose(dS, dS_ref, atol=1e-12)


def test_reuse_test_is_a_polar_factor_test():
    """rounds_to(v, Rp): true when the polar factor of mat(v[:9] / v[9]) is within ~0.16 rad of Rp."""
    rs = np.random.RandomState(4)

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis)
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx

    for trial in range(50):
        Rp, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        if np.linalg.det(Rp) < 0:
            Rp[:, 0] = -Rp[:, 0]
        ang = rs.uniform(0.0, 0.5)
        P = np.eye(3) * rs.uniform(0.6, 1.0) + 0.05 * np.diag(rs.normal(size=3))  # symmetric positive definite stretch
        M0 = Rp @ rot(rs.normal(size=3), ang) @ P
        s = rs.uniform(0.5, 2.0) * (1 if trial % 2 else -1)
        v = np.concatenate([M0.T.reshape(-1), [1.0]]) * s  # any multiple of [vec(M0); 1]
        ok, d0 = hostsim.rounds_to(v, Rp)
        assert d0 > 0
        if ang < 0.10:
            assert ok
        if ang > 0.25:
            assert not ok


def test_planar_scene_in_a_general_frame_is_solved_in_the_canonical_one():
    """A plane that is not Z = 0: the cost is blind to R n, detected from the partial trace of Qs; the solve runs
    in the frame whose third axis is n (canonicalise_planar) and R, Z come back in the caller's frame: same
    behaviour as for Z = 0 -- certified pair after a handful of iterations, both poses recoverable from Z."""
    from cvxpnpl_amd.api import recover_multi

    d = synth.make_planar_pnp(300, 10, 0.0, seed=3, general=True)
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    assert (hs["status"] == 1).all()
    assert np.median(hs["iters"]) <= 20
    for i in range(0, 300, 10):
        _, B, Q = hostsim.assemble(d["pts_2d"][i], d["pts_3d"][i], None, None, d["K"])
        poses = recover_multi(hs["Z"][i], B, Q[np.triu_indices(9)])
        assert len(poses) in (2, 4)
        assert min(synth.geodesic(R, d["R_gt"][i]) + np.linalg.norm(t - d["t_gt"][i]) for R, t in poses) < 1e-8
        # the pose returned with status 1 is one of the two twins, in the caller's frame
        assert min(synth.geodesic(hs["R"][i], R) for R, t in poses) < 1e-8


STALE_REUSE_CASES = [(9, 1, 2.0, 5012, 100), (5, 0, 0.0, 5005, 91), (0, 5, 2.0, 5023, 18)]


@pytest.mark.parametrize("n_p,n_l,sigma,seed,i", STALE_REUSE_CASES)
def test_pose_reuse_does_not_go_stale(n_p, n_l, sigma, seed, i):
    """Found by tools/fuzz_parity.py: two local minima within the rounding tolerance of cvx::rounds_to.  A pose
    polished at an early check (before a spell in the twin branch) used to be taken over by every later
    check, so the problem ran to res_tol uncertified (status 2 after 218-248 iterations, pose 1e-4 rad off)
    although the iterate had converged to the certifiable optimum.  The shortcut is now bounded (REUSE_MAX)."""
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(192, n_p, n_l, sigma, seed=seed)
    sl = slice(i, i + 1)
    r = hostsim.solve_batch(d["pts_2d"][sl] if n_p else None, d["pts_3d"][sl] if n_p else None, d["line_2d"][sl] if n_l else None,
                            d["line_3d"][sl] if n_l else None, d["K"])
    assert r["status"][0] == 0 and r["iters"][0] < 60
    assert 0 <= r["cost"][0, 0] - r["cost"][0, 1] <= 1e-9


def test_assembly_far_from_the_world_origin(orc):
    """Advisor finding (round 1): the Gram difference C^T C - (N^T C)^T B loses |P|^2 / spread^2 digits when the world
    origin is far from the scene.  The sums are now taken about the problem's first 3D point (exact: the shift goes back
    into B): Q and B against the reference's explicit A = C - N B (cvxpnpl.py:545-549) with the origin 1e3 scene sizes
    away, and the certified pose against the oracle."""
    import hostsim
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(12, 6, 3, 0.5, seed=17)
    c = np.random.RandomState(2).normal(size=(12, 1, 3)) * 600.0
    d["pts_3d"] = d["pts_3d"] + c
    d["line_3d"] = d["line_3d"] + c[:, None]
    worst_q = worst_b = 0.0
    for i in range(12):
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        cl, nl = orc.line_constraints(d["line_2d"][i], d["line_3d"][i], d["K"])
        B, A = orc.eliminate(np.vstack((c1, c2, c3, cl)), np.vstack((n1, n2, n3, nl)))
        rc, Bh, Qh = hostsim.assemble(d["pts_2d"][i], d["pts_3d"][i], d["line_2d"][i], d["line_3d"][i], d["K"])
        assert rc == 0
        Q = A.T @ A
        worst_q = max(worst_q, np.abs(Qh - Q).max() / np.abs(Q).max())
        worst_b = max(worst_b, np.abs(Bh - B).max() / np.abs(B).max())
    # (the reference's own A^T A carries the cancellation of C - N B at this offset: ~1e-10 relative)
    assert worst_q < 1e-8 and worst_b < 1e-9, (worst_q, worst_b)
    h = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], d["line_2d"], d["line_3d"], d["K"])
    o = orc.pnpl_batch(d["pts_2d"], d["line_2d"], d["pts_3d"], d["line_3d"], d["K"], eps=1e-11, max_iters=200000)
    ok = (h["status"] == 0) & (o["n_poses"] == 1)
    assert ok.sum() >= 10
    assert synth.geodesic(h["R"], o["R"][:, 0])[ok].max() < 1e-6


def test_interior_point_path_examples_and_first_order_agreement(golden):
    """The interior-point path (csrc/ipm_core.h: the scalar statement of what cvxw::coop_ipm runs for the problems still open
    after opts.rescue_from iterations) reproduces the reference's three examples and agrees with the first-order solve -- it is
    the same SDP, and its solution goes through the same rounding, polish and certificate."""
    for name, args in (
        ("pnp", (golden["ex_pnp_pts2d"][None], golden["ex_pnp_pts3d"][None], None, None)),
        ("pnl", (None, None, golden["ex_pnl_line2d"][None], golden["ex_pnl_line3d"][None])),
        ("pnpl", (golden["ex_pnpl_pts2d"][None], golden["ex_pnpl_pts3d"][None], golden["ex_pnpl_line2d"][None], golden["ex_pnpl_line3d"][None])),
    ):
        ip = hostsim.ipm_batch(*args, golden[f"ex_{name}_K"])
        assert ip["status"][0] == 0 and ip["iters"][0] <= 20
        assert synth.geodesic(ip["R"][0], golden[f"ex_{name}_R"]) < 1e-6
        tg = golden[f"ex_{name}_t"]
        assert np.linalg.norm(ip["t"][0] - tg) / np.linalg.norm(tg) < 1e-6
    # minimal problems: where the first-order iteration is slow (hundreds of iterations) the interior-point path is not
    d = synth.make_pnp(300, 4, 2.0, seed=3)
    fo = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    ip = hostsim.ipm_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    assert ip["iters"].max() <= 25 and fo["iters"].max() > 200
    assert (ip["status"] == 0).sum() >= (fo["status"] == 0).sum()
    both = (fo["status"] == 0) & (ip["status"] == 0)
    assert both.mean() > 0.95
    assert synth.geodesic(ip["R"], fo["R"])[both].max() < 1e-7
    assert (np.linalg.norm(ip["t"] - fo["t"], axis=1) / np.linalg.norm(fo["t"], axis=1))[both].max() < 1e-7
    # a certified Z is the rank-one z z^T of the same pose in both paths
    np.testing.assert_allclose(ip["Z"][both], fo["Z"][both], atol=1e-6)
    # planar scenes (two-fold ambiguous, rank 2): both paths report the rank and the same pair of poses
    dp = synth.make_planar_pnp(60, 8, 0.5, seed=21, general=True)
    fo = hostsim.solve_batch(dp["pts_2d"], dp["pts_3d"], None, None, dp["K"])
    ip = hostsim.ipm_batch(dp["pts_2d"], dp["pts_3d"], None, None, dp["K"])
    assert (ip["status"] == fo["status"]).mean() > 0.95
    same = (ip["status"] == fo["status"]) & ~np.isnan(fo["R"]).any(axis=(1, 2)) & ~np.isnan(ip["R"]).any(axis=(1, 2))
    g = synth.geodesic(ip["R"], fo["R"])[same]  # (which of the two poses comes first is arbitrary: the mirror twin is half a turn away)
    g = np.minimum(g, np.abs(g - np.pi))
    # (a pair that is not exactly two-fold ambiguous -- noisy image points -- is rounded from an uncertified Z: close, not identical)
    assert same.sum() > 40 and (g < 1e-6).mean() > 0.9 and g.max() < 2e-2


@pytest.mark.parametrize("n_p,n_l,sigma,iters", [(10, 0, 2.0, 6), (10, 0, 0.0, 6), (5, 5, 1.0, 6), (4, 0, 1.0, 6), (0, 6, 1.0, 5), (7, 2, 3.0, 4), (10, 0, 2.0, 2)])
def test_lane_core_restatement_equals_the_general_core(n_p, n_l, sigma, iters):
    """cvxl::lane_phase (csrc/lane_core.h: the first phase of the lane-hybrid schedule written for a register budget --
    straight line, streamed projections, one certificate at the end) against the general scalar core it restates
    (cvx::solve_problem<TWIN = false>, first_check == hand-off point; run with its float64 eigen-solve as the yardstick -- measured
    on 512 four-point problems, parked iterates: restatement vs float64 core <= 3.4e-5, the general core's own single-precision
    mode vs its float64 mode 9.5e-2 on one ill-conditioned problem, 2e-5 at the 99th percentile), on the host: the same problems certify, with the same
    pose (both Newton-polish to the stationary point: 1e-10) and the same certified bound; the same problems are parked, with
    the same iterate to rounding (the restatement forms (W + sigma I) V in single precision and starts the polish from two
    polar steps + Gram-Schmidt instead of a converged polar iteration.  The eigen-solve stops at a column cosine of 6e-2, i.e. it
    is accurate to ~4e-3 by design, and whether one more sweep runs is a discrete decision that single-precision noise can flip:
    parked iterates of the two agree to 2e-3, measured <= 2e-4 -- the same bound the float64 / float32 sweep modes of the kernels
    meet, tests/test_precision_modes.py -- and certificate decisions may differ on the few problems at the acceptance threshold)."""
    import hostsim
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(512, n_p, n_l, sigma, seed=77 + n_p + 3 * n_l + iters)
    d["pts_3d"][:8, :, 2] = 0.0 if n_p else d["pts_3d"][:8, :, 2]  # a few planar scenes (canonical frame): nothing to certify, parked
    if n_p:
        d["pts_2d"][:8] = synth.project(d["pts_3d"][:8], d["K"], d["R_gt"][:8], d["t_gt"][:8])
    o = hostsim.default_opts(first_check=iters)
    args = (d["pts_2d"] if n_p else None, d["pts_3d"] if n_p else None, d["line_2d"] if n_l else None, d["line_3d"] if n_l else None, d["K"])
    a = hostsim.lane_phase(*args, iters, 0, o, dbl=True)  # the general core with the eigen-solve on float64 columns: the yardstick
    b = hostsim.lane_phase(*args, iters, 1, o)
    assert set(np.unique(a["status"])) <= {-1, 0} and set(np.unique(b["status"])) <= {-1, 0}
    same = a["status"] == b["status"]
    assert same.mean() >= 0.995, np.flatnonzero(~same)
    cert = same & (a["status"] == 0)
    park = same & (a["status"] == -1)
    if iters >= 5 and n_p + n_l >= 8:
        assert cert.mean() > 0.9
    assert (a["iters"][cert] == iters).all() and (b["iters"][cert] == iters).all()
    if cert.any():
        assert synth.geodesic(a["R"][cert], b["R"][cert]).max() < 1e-10  # (the polish takes its last Newton step from |g| < 1e-8: 1e-10 ... 1e-16)
        assert np.abs(a["t"][cert] - b["t"][cert]).max() < 1e-9
        assert np.abs(a["cost"][cert] - b["cost"][cert]).max() < 1e-12 * max(1.0, np.abs(a["cost"][cert]).max()) + 2e-10  # (dobj carries the dual: <= eps apart)
        gap = b["cost"][cert, 0] - b["cost"][cert, 1]
        assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][cert, 0])).all()
    assert park.sum() >= 8 or n_p == 0
    assert (a["handoff"][park, 55] == b["handoff"][park, 55]).all()
    dW = np.abs(a["handoff"][park, :55] - b["handoff"][park, :55]).max() if park.any() else 0.0
    print(f"lane core: n_p={n_p} n_l={n_l} iters={iters}: {cert.sum()} certified, {park.sum()} parked, status mismatches {(~same).sum()}, max |dW| parked {dW:.2e}")
    assert dW < 2e-3
    # a parked planar scene in a general frame carries the start iterate and iteration 0
    dg = synth.make_planar_pnp(16, 10, 0.0, seed=5, general=True)
    pa = hostsim.lane_phase(dg["pts_2d"], dg["pts_3d"], None, None, dg["K"], 6, 1, hostsim.default_opts(first_check=6))
    assert (pa["status"] == -1).all() and (pa["handoff"][:, 55] == 0).all() and (pa["handoff"][:, 54] == 1).all()
    # degenerate input: NaN pose, status 3
    bad = synth.make_pnp(4, 6, 0.0, seed=1)
    bad["pts_2d"][:] = bad["pts_2d"][:, :1]
    pb = hostsim.lane_phase(bad["pts_2d"], bad["pts_3d"], None, None, bad["K"], 6, 1, hostsim.default_opts(first_check=6))
    pc = hostsim.lane_phase(bad["pts_2d"], bad["pts_3d"], None, None, bad["K"], 6, 0, hostsim.default_opts(first_check=6))
    assert (pb["status"] == pc["status"]).all() and (pb["status"] == 3).all() and np.isnan(pb["R"]).all()


@pytest.mark.parametrize("n_p,n_l,sigma,iters", [(10, 0, 2.0, 6), (5, 5, 1.0, 6), (4, 0, 1.0, 6), (0, 6, 1.0, 5), (7, 2, 3.0, 4), (10, 0, 2.0, 2), (10, 0, 2.0, 3)])
def test_lane_core_float64_instantiation_equals_the_general_core(n_p, n_l, sigma, iters):
    """cvxl::lane_phase_f64 (csrc/lane_core.h: the lane phase with every sweep in float64 -- the reference's precision,
    cvxpnpl.py:475-513 -- where the positive part is never stored: the update, linear in it, is added into the iterate one eigen
    column at a time and its projection onto span A_i taken from the difference of the constraint sums, cvxl::pos_update_cols)
    against the general scalar core with its float64 eigen-solve (cvx::solve_problem<false, ., ., DBL = true>), on the host.  Both
    are float64 throughout and run the same sweeps, so unlike the single-precision restatement above the parked iterates agree to
    rounding (the two differ in the ORDER of the update's additions and in where the polish starts from: measured <= 2e-13), the same
    problems certify, and the sweep counts are identical.  With iters >= 4 the tail_from switch (iteration 3: sc != 1) falls into the
    column-wise update, with iters = 3 into the last iteration of the phase, with iters = 2 it does not happen."""
    import hostsim
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(512, n_p, n_l, sigma, seed=177 + n_p + 3 * n_l + iters)
    o = hostsim.default_opts(first_check=iters, f32_sweeps_until=0)
    args = (d["pts_2d"] if n_p else None, d["pts_3d"] if n_p else None, d["line_2d"] if n_l else None, d["line_3d"] if n_l else None, d["K"])
    a = hostsim.lane_phase(*args, iters, 0, o, dbl=True)
    b = hostsim.lane_phase(*args, iters, 2, o)
    assert set(np.unique(a["status"])) <= {-1, 0} and set(np.unique(b["status"])) <= {-1, 0}
    same = a["status"] == b["status"]
    assert same.mean() >= 0.998, np.flatnonzero(~same)  # (certificate decisions at the acceptance threshold may differ)
    cert = same & (a["status"] == 0)
    park = same & (a["status"] == -1)
    assert (a["sweeps"] == b["sweeps"]).mean() > 0.99
    assert (a["iters"][cert] == iters).all() and (b["iters"][cert] == iters).all()
    if cert.any():
        assert synth.geodesic(a["R"][cert], b["R"][cert]).max() < 1e-10
        assert np.abs(a["t"][cert] - b["t"][cert]).max() < 1e-9
        gap = b["cost"][cert, 0] - b["cost"][cert, 1]
        assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][cert, 0])).all()
    assert park.any()
    assert (a["handoff"][park, 55] == b["handoff"][park, 55]).all()
    eq = park & (a["sweeps"] == b["sweeps"])
    dW = np.abs(a["handoff"][eq, :55] - b["handoff"][eq, :55]).max()
    print(f"lane core f64: n_p={n_p} n_l={n_l} iters={iters}: {cert.sum()} certified, {park.sum()} parked, status mismatches {(~same).sum()}, max |dW| parked {dW:.2e}")
    assert dW < 1e-11  # (measured <= 1.6e-13)


def test_gram_sums_are_taken_about_a_robust_centre():
    """The Gram sums are shifted about the per-coordinate median of the first three 3D records (cvx::shift_centre; rounds 1-2: about
    the first record -- the round-2 advisor's finding: a far point in first position drags the centre away from the scene).  The
    shift is exact for ANY centre, so this is a property test of the rule, not a regression on digits (measured: with a point
    50 000 scene sizes away the assembled cost agrees to 1e-11 of its largest entry wherever that point stands in the list, under
    either rule -- such a point dominates the cost whatever the centre): the order of the correspondences does not matter, and
    points, lines, mixed records and fewer than three records all go through it."""
    import hostsim
    from cvxpnpl_amd import synth

    d = synth.make_pnp(1, 12, 0.5, seed=8)
    p2, p3 = d["pts_2d"][0].copy(), d["pts_3d"][0].copy()
    p3[0] += np.array([3.0e4, -2.0e4, 1.0e4])  # 50 000 scene sizes away
    outs = []
    for pos in (0, 1, 2, 11):
        idx = list(range(1, 12))
        idx.insert(pos, 0)
        _, B, Q = hostsim.assemble(p2[idx], p3[idx], None, None, d["K"])
        outs.append((B, Q))
    Q0 = outs[-1][1]  # outlier last: the centre is a scene point whatever the rule
    for B, Q in outs[:-1]:
        assert np.abs(Q - Q0).max() <= 1e-11 * np.abs(Q0).max()
        assert np.abs(B - outs[-1][0]).max() <= 1e-9 * np.abs(outs[-1][0]).max()
    # lines, mixed records and fewer than three records go through the same rule
    dl = synth.make_pnpl(1, 1, 3, 0.0, seed=3)
    rc, B, Q = hostsim.assemble(dl["pts_2d"][0], dl["pts_3d"][0], dl["line_2d"][0], dl["line_3d"][0], dl["K"])
    assert rc == 0 and np.isfinite(Q).all()
