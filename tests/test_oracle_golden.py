"""Pin the CPU oracle against vectors produced by the reference itself.

Every array named g*/ex_*/e2e_* was written by tests/golden/make_golden.py, which runs
/root/reference/cvxpnpl.py (stub `scs`) in the build container.  CPU only.
"""
import numpy as np
import pytest

from conftest import geodesic


def test_point_constraints_g1(golden, orc):
    for tag in ("n4", "n6_intK", "n10"):
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(golden[f"g1_{tag}_pts2d"], golden[f"g1_{tag}_pts3d"], golden[f"g1_{tag}_K"])
        np.testing.assert_allclose(np.stack((c1, c2, c3)), golden[f"g1_{tag}_C"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(np.stack((n1, n2, n3)), golden[f"g1_{tag}_N"], rtol=0, atol=1e-15)


def test_line_constraints_g2(golden, orc):
    for tag in ("n4", "n5"):
        C, N = orc.line_constraints(golden[f"g2_{tag}_line2d"], golden[f"g2_{tag}_line3d"], golden["K_kinect"])
        np.testing.assert_allclose(C, golden[f"g2_{tag}_C"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(N, golden[f"g2_{tag}_N"], rtol=0, atol=1e-15)


def test_elimination_and_cost_vector_g3(golden, orc):
    # pnp example
    (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(golden["ex_pnp_pts2d"], golden["ex_pnp_pts3d"], golden["ex_pnp_K"])
    B, A = orc.eliminate(np.vstack((c1, c2, c3)), np.vstack((n1, n2, n3)))
    np.testing.assert_allclose(B, golden["g3_pnp_B"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(A, golden["g3_pnp_A"], rtol=0, atol=1e-13)
    Q = np.zeros((10, 10))
    Q[:9, :9] = A.T @ A
    np.testing.assert_allclose(orc.vech10(Q, 2.0), golden["g3_pnp_c"], rtol=0, atol=1e-13)
    # pnl example
    C, N = orc.line_constraints(golden["ex_pnl_line2d"], golden["ex_pnl_line3d"], golden["ex_pnl_K"])
    B, A = orc.eliminate(C, N)
    np.testing.assert_allclose(B, golden["g3_pnl_B"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(A, golden["g3_pnl_A"], rtol=0, atol=1e-13)
    # pnpl example: rows stacked [Cp1; Cp2; Cp3; Cl] (cvxpnpl.py:619-620)
    (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(golden["ex_pnpl_pts2d"], golden["ex_pnpl_pts3d"], golden["ex_pnpl_K"])
    Cl, Nl = orc.line_constraints(golden["ex_pnpl_line2d"], golden["ex_pnpl_line3d"], golden["ex_pnpl_K"])
    B, A = orc.eliminate(np.vstack((c1, c2, c3, Cl)), np.vstack((n1, n2, n3, Nl)))
    np.testing.assert_allclose(B, golden["g3_pnpl_B"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(A, golden["g3_pnpl_A"], rtol=0, atol=1e-13)
    # the kwargs the reference hands to scs (cvxpnpl.py:478-484) with its defaults
    assert float(golden["g3_kw_eps_abs"]) == 1e-9 and int(golden["g3_kw_max_iters"]) == 2500


def test_static_sdp_data_g4(golden, orc):
    Ad, b = orc.sdp_constraints()
    assert Ad.shape == (77, 55)
    assert np.array_equal(Ad, golden["g4_A"])  # bit exact, incl. the -sqrt(2) cone block
    assert np.array_equal(b, golden["g4_b"])
    assert int(golden["g4_cone_zero"]) == 22 and list(golden["g4_cone_s"]) == [10]
    assert np.count_nonzero(Ad) == 125
    assert np.linalg.matrix_rank(Ad[:22]) == 21


def test_vech_g5(golden, orc):
    M = golden["g5_M"]
    assert np.array_equal(orc.vech10(M, 1.0), golden["g5_vech1"])
    assert np.array_equal(orc.vech10(M, 2.0), golden["g5_vech2"])
    np.testing.assert_allclose(orc.vech10(M, np.sqrt(2)), golden["g5_vechs2"], rtol=1e-16)
    assert np.array_equal(orc.vech10_inv(np.arange(55.0)), golden["g5_inv"])


def _match_poses(poses, Rg, tg, tol):
    """Order-free comparison of pose lists (np.roots' order is not part of the contract)."""
    assert len(poses) == len(Rg)
    used = set()
    for R, t in poses:
        d = [np.abs(R - Rg[k]).max() + np.abs(t - tg[k]).max() if k not in used else np.inf for k in range(len(Rg))]
        k = int(np.argmin(d))
        assert d[k] < tol, d
        used.add(k)


def test_recovery_rank1_g6(golden, orc):
    A, B = golden["g3_pnp_A"], golden["g3_pnp_B"]
    poses, st, rk = orc.recover(golden["g6_r1_x"], 0.0, A, B)
    assert rk == 1 and len(poses) == 1
    _match_poses(poses, golden["g6_r1_R"], golden["g6_r1_t"], 1e-12)
    assert geodesic(poses[0][0], golden["g6_r1_R_in"]) < 1e-14
    assert (st & 4) != 0 and int(golden["g6_r1_warned"]) == 1  # ||Ar||^2 != 0 here: the reference warned too
    # perturbed: exercises the SVD projection of a non-orthogonal 3x3
    poses, st, rk = orc.recover(golden["g6_r1p_x"], 0.0, A, B)
    _match_poses(poses, golden["g6_r1p_R"], golden["g6_r1p_t"], 1e-11)


def test_recovery_rank2_rank4_g6(golden, orc):
    A, B = golden["g3_pnp_A"], golden["g3_pnp_B"]
    poses, st, rk = orc.recover(golden["g6_r2_x"], 0.0, A, B)
    assert rk == 2 and (st & 2)
    _match_poses(poses, golden["g6_r2_R"], golden["g6_r2_t"], 1e-7)
    poses, st, rk = orc.recover(golden["g6_r4_x"], 0.0, A, B)
    assert rk == 4 and (st & 2)
    _match_poses(poses, golden["g6_r4_R"], golden["g6_r4_t"], 1e-6)
    # and the recovered rotations are the ones the mixtures were built from
    for R, _ in poses:
        assert min(geodesic(R, Rin) for Rin in golden["g6_r4_R_in"]) < 1e-6


def test_constraint_ortho_det_g7(golden, orc):
    for tag, rank in (("r2", 2), ("r4", 4)):
        # eigenvectors are defined up to sign; the reference's own vecs are fed in
        rc = orc.constraint_ortho_det(golden[f"g7_{tag}_vecs"], rank)
        ref = golden[f"g7_{tag}_rc"]
        assert rc.shape == ref.shape
        for row in rc:
            assert np.abs(ref - row).max(axis=1).min() < 1e-6


def test_re6q3_g7(golden, orc):
    a, b, c = orc.re6q3(golden["g7_re6q3_A"])
    ref = golden["g7_re6q3_abc"]  # (3, 4): a, b, c
    assert len(a) == 4
    for k in range(4):
        d = np.abs(ref[0] - a[k]) + np.abs(ref[1] - b[k]) + np.abs(ref[2] - c[k])
        assert d.min() < 1e-6, (k, d)


def test_nan_sentinel_and_certificate_g6(golden, orc):
    A, B = golden["g3_pnp_A"], golden["g3_pnp_B"]
    poses, st, rk = orc.recover(np.full(55, np.nan), 0.0, A, B)
    assert len(poses) == int(golden["g6_nan_n"]) == 1 and (st & 1)
    assert np.isnan(poses[0][0]).all() and np.isnan(poses[0][1]).all()
    assert np.isnan(golden["g6_nan_R"]).all()
    _, st, _ = orc.recover(golden["g6_r1_x"], 1.0, A, B)
    assert (st & 4) and int(golden["g6_cert_warned"]) == 1


@pytest.mark.parametrize("name", ["pnp", "pnl", "pnpl"])
def test_examples_known_answer_g8(golden, orc, name):
    """examples/*.py: deterministic inputs, literal ground truth (8 digits)."""
    if name == "pnp":
        poses, info = orc.pnp(golden["ex_pnp_pts2d"], golden["ex_pnp_pts3d"], golden["ex_pnp_K"], eps=1e-10, max_iters=200000)
    elif name == "pnl":
        poses, info = orc.pnl(golden["ex_pnl_line2d"], golden["ex_pnl_line3d"], golden["ex_pnl_K"], eps=1e-10, max_iters=200000)
    else:
        poses, info = orc.pnpl(golden["ex_pnpl_pts2d"], golden["ex_pnpl_line2d"], golden["ex_pnpl_pts3d"], golden["ex_pnpl_line3d"],
                               golden["ex_pnpl_K"], eps=1e-10, max_iters=200000)
    assert len(poses) == 1 and info.rank == 1 and info.scs_status == 0
    R, t = poses[0]
    assert geodesic(R, golden[f"ex_{name}_R"]) < 1e-6           # literals carry 8 digits
    assert np.linalg.norm(t - golden[f"ex_{name}_t"]) / np.linalg.norm(golden[f"ex_{name}_t"]) < 1e-6
    assert info.status == 0                                      # certified: | ||Ar||^2 - dobj | <= eps


def test_restated_scs_satisfies_kkt(golden, orc):
    """The solver is third party and absent; pin it to the SDP optimum through the KKT
    conditions of  min c^T x, A x + s = b, s in {0}^22 x PSD, written with the reference's
    own A and b (g4) and numpy only."""
    A, b = golden["g4_A"], golden["g4_b"]
    d = np.abs(np.diag(A[22:]))  # 1 / sqrt(2) svec weights

    def mat(v):  # svec (sqrt 2 scaled) -> symmetric matrix
        M = np.zeros((10, 10))
        k = 0
        for j in range(10):
            for i in range(j, 10):
                M[i, j] = M[j, i] = v[k] if i == j else v[k] / np.sqrt(2)
                k += 1
        return M

    for i in range(int(golden["e2e_count"])):
        c = golden[f"e2e_{i}_c"]
        tr = c[[0, 10, 19, 27, 34, 40, 45, 49, 52]].sum()
        r = orc.scs_solve(c, eps=1e-10, max_iters=400000, cscale=10.0 / tr)
        x, y = r["x"], r["y"]
        assert r["info"]["status"] == "solved"
        s = b - A @ x
        assert np.abs(s[:22]).max() < 1e-8                          # equalities
        assert np.linalg.eigvalsh(mat(s[22:])).min() > -1e-8        # Z PSD
        assert np.abs(A.T @ y + c).max() < 1e-8 * max(1.0, np.abs(c).max())  # dual feasibility
        assert np.linalg.eigvalsh(mat(y[22:])).min() > -1e-8 * max(1.0, tr)  # dual slack PSD
        assert abs(c @ x + b @ y) < 1e-8 * max(1.0, tr)             # zero gap
        np.testing.assert_allclose(s[22:], d * x, atol=1e-12)


def test_end_to_end_matches_reference_postprocessing(golden, orc):
    """orc_pnpl (restated driver) == the reference's own pnp/pnl/pnpl run on the same solve."""
    for i in range(int(golden["e2e_count"])):
        kind = str(golden[f"e2e_{i}_kind"])
        p2, p3 = golden[f"e2e_{i}_pts2d"], golden[f"e2e_{i}_pts3d"]
        l2, l3 = golden[f"e2e_{i}_line2d"], golden[f"e2e_{i}_line3d"]
        poses, info = orc.pnpl(p2 if len(p2) else None, l2 if len(l2) else None, p3 if len(p3) else None, l3 if len(l3) else None,
                               golden["K_kinect"], eps=1e-11, max_iters=400000)
        assert len(poses) == 1, kind
        R, t = poses[0]
        assert geodesic(R, golden[f"e2e_{i}_R"]) < 1e-8
        assert np.abs(t - golden[f"e2e_{i}_t"]).max() < 1e-8
        if float(golden[f"e2e_{i}_noise"]) == 0.0:
            assert geodesic(R, golden[f"e2e_{i}_Rgt"]) < 1e-7
