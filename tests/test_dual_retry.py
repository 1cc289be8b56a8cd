"""opts.dual_shift: the second tries of a failed dual (solver_core.h: dual_retry_entry6, dual_certificate; wave_kernel.h: coop_dual).

The duals that certify a pose z = [vec(R); 1] are the PSD members of  S1 + U,  U = { X in span A_i : X z = 0 }  (A_i: the equality
matrices of cvxpnpl.py:387-451).  The second tries move the recovered dual along D(R) = P_U(I - z z^T / 4), for which the code uses a
closed form.  Pinned here: (CPU) the closed form equals the projection computed numerically from the constraint matrices, for both
constraint sets; D z = 0; D lies in span A_i; the scalar core with and without the tries certifies the same poses in fewer iterations.
(GPU) a lane-hybrid launch with and without the tries: same statuses, same certified poses to 1e-9 rad, fewer iterations; option validated."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _span_A(rc):
    Ti = [[0, 3, 6], [0, 3, 6], [1, 4, 7], [0, 1, 2], [0, 1, 2], [3, 4, 5], [1, 2, 6], [2, 0, 7], [0, 1, 8], [4, 5, 0], [5, 3, 1], [3, 4, 2], [2, 1, 3], [0, 2, 4], [1, 0, 5]]
    Tj = [[1, 4, 7], [2, 5, 8], [2, 5, 8], [3, 4, 5], [6, 7, 8], [6, 7, 8], [5, 4, 9], [3, 5, 9], [4, 3, 9], [8, 7, 9], [6, 8, 9], [7, 6, 9], [7, 8, 9], [8, 6, 9], [6, 7, 9]]
    A = []
    for t in range(3 if rc else 0, 15):      # cvxpnpl.py:404-435 as sign triples (solver_core.h: tri_i / tri_j / tri_s); rc.py:16-35 drops the first three
        M = np.zeros((10, 10))
        for k in range(3):
            s = -1.0 if (t >= 6 and k >= 1) else 1.0
            M[Ti[t][k], Tj[t][k]] += 0.5 * s
            M[Tj[t][k], Ti[t][k]] += 0.5 * s
        A.append(M)
    if not rc:
        for i in range(3):                   # rows of R have unit norm
            M = np.zeros((10, 10))
            for j in range(3):
                M[3 * j + i, 3 * j + i] = 1
            A.append(M)
    for j in range(3):                       # columns of R have unit norm
        M = np.zeros((10, 10))
        for i in range(3):
            M[3 * j + i, 3 * j + i] = 1
        A.append(M)
    M = np.zeros((10, 10))
    M[9, 9] = 1
    A.append(M)
    return np.array([a.ravel() for a in A])


def _numeric_direction(z, Am):
    _, s, Vt = np.linalg.svd(Am, full_matrices=False)
    B = Vt[: int(np.sum(s > 1e-10))]                               # orthonormal basis of span A_i
    G = np.array([(b.reshape(10, 10) @ z) for b in B]).T            # X z = G c
    _, s2, vt = np.linalg.svd(G)
    Ub = vt[int(np.sum(s2 > 1e-10)):] @ B                          # orthonormal basis of U
    T = np.eye(10) - np.outer(z, z) / 4.0
    return ((Ub @ T.ravel()) @ Ub).reshape(10, 10), len(Ub)


@pytest.mark.parametrize("rc", [False, True])
def test_closed_form_of_the_retry_direction(rc):
    from scipy.spatial.transform import Rotation

    import hostsim

    Am = _span_A(rc)
    assert np.linalg.matrix_rank(Am) == (16 if rc else 21)  # (22 / 16 equalities, one redundant in the full set)
    for seed in range(5):
        R = Rotation.random(random_state=seed).as_matrix()
        z = np.concatenate([R.T.ravel(), [1.0]])                    # z[3 j + i] = R[i][j]
        D = hostsim.dual_retry_direction(R)
        Dn, dim = _numeric_direction(z, Am)
        assert rc or dim == 14
        if not rc:
            assert np.abs(D - Dn).max() < 1e-13
        else:  # the rc family is larger; the SAME direction is used and must lie in it: D in span A_rc, D z = 0
            c = np.linalg.lstsq(Am.T, D.ravel(), rcond=None)[0]
            assert np.abs(Am.T @ c - D.ravel()).max() < 1e-12
        assert np.abs(D @ z).max() < 1e-14 and np.abs(D - D.T).max() == 0.0
        # in span A_i: the projection onto the equalities' direction space (proj_affine, homogeneous) annihilates it
        vech = np.array([D[i, j] for i in range(10) for j in range(i, 10)])
        assert np.abs(hostsim.proj_affine_homog(vech, 1 if rc else 0)).max() < 1e-14
        lam = np.linalg.eigvalsh(D)
        assert abs(lam[0] + 2.0 / 3.0) < 1e-12 and np.abs(lam[-5:] - 1.0 / 3.0).max() < 1e-12 and np.abs(lam[1:5]).max() < 1e-12


def test_scalar_core_with_and_without_second_tries():
    import hostsim
    from cvxpnpl_amd import synth
    from test_gpu_parity import geodesic_np

    d = synth.make_pnpl(4000, 10, 0, 2.0, seed=42)
    a = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts(dual_shift=0.0))
    b = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts())
    assert hostsim.default_opts().dual_shift == 0.015
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    assert max(geodesic_np(a["R"][i], b["R"][i]) for i in range(4000)) < 1e-9          # both are the certified global optimum
    assert (b["iters"] <= a["iters"]).mean() > 0.999 and (b["iters"] >= 8).sum() < 0.75 * (a["iters"] >= 8).sum()  # (measured 32 against 52)
    gap = b["cost"][:, 0] - b["cost"][:, 1]
    assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][:, 0])).all()                # the certificate statement is unchanged


@pytest.mark.gpu
def test_second_tries_on_the_device():
    import torch
    from test_gpu_parity import _solve, geodesic_np

    from cvxpnpl_amd import synth

    from cvxpnpl_amd import _lib

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.lib()
    dev = torch.device("cuda:0")
    d = synth.make_pnpl(30000, 10, 0, 2.0, seed=7)   # lane-hybrid schedule: parked problems go to resume_wave_kernel, where the tries are made
    a = _solve(dev, d, 10, 0, dual_shift=0.0)
    b = _solve(dev, d, 10, 0)
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    worst = max(geodesic_np(a["R"][i], b["R"][i]) for i in np.flatnonzero(a["iters"] != b["iters"]))
    assert worst < 1e-9
    assert b["iters"].mean() < a["iters"].mean() and (b["iters"] >= 10).sum() <= (a["iters"] >= 10).sum()
    gap = b["cost"][:, 0] - b["cost"][:, 1]
    assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][:, 0])).all()
    with pytest.raises(RuntimeError, match="bad options"):
        _solve(dev, d, 10, 0, dual_shift=-0.1)


def test_scalar_core_with_and_without_the_eigen_gradient_step():
    """opts.dual_refine (cvx::dual_refine_step): same certified optimum, same certificate statement, fewer iterations for the stragglers."""
    import hostsim
    from cvxpnpl_amd import synth
    from test_gpu_parity import geodesic_np

    d = synth.make_pnpl(6000, 10, 0, 2.0, seed=42)
    a = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts(dual_refine=0))
    b = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts())
    assert hostsim.default_opts().dual_refine == 1
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    assert max(geodesic_np(a["R"][i], b["R"][i]) for i in np.flatnonzero(a["iters"] != b["iters"])) < 1e-8   # both are the certified global optimum
    # the step is made from the third attempt of a solve on (iteration 9): what needed 11 and more iterations mostly stops at 9
    assert (b["iters"] <= a["iters"]).all() and (b["iters"] >= 11).sum() <= 0.5 * (a["iters"] >= 11).sum() and (a["iters"] >= 11).sum() >= 10
    gap = b["cost"][:, 0] - b["cost"][:, 1]
    assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][:, 0])).all()                # the certificate statement is unchanged


def test_newton_tables_are_a_basis_of_the_dual_family_in_the_frame_of_R():
    """cvx::kNt* (solver_core.h, generated by tools/gen_newton_tables.py): 14 independent matrices of span A_i that annihilate z_I = [vec I3; 1]
    -- with the constraint matrices built from the oracle's own static data -- and T_I = I - z_I z_I^T / 4 as the 15th."""
    import hostsim
    import oracle

    M = hostsim.newton_tables()
    zI = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 1.0])
    assert np.abs(M[:14] @ zI).max() == 0.0 and np.allclose(M[14], np.eye(10) - np.outer(zI, zI) / 4)
    assert np.linalg.matrix_rank(M[:14].reshape(14, 100)) == 14
    Ad, _ = oracle.sdp_constraints()                     # the reference's _A (cvxpnpl.py:387-451): rows 0..21 are the equalities on vech(Z)
    A = np.stack([oracle.vech10_inv(Ad[i]) for i in range(22)])
    A = 0.5 * (A + np.stack([np.diag(np.diag(a)) for a in A]))   # <A_i, Z> = row . vech(Z): every off-diagonal pair counted once
    F = A.reshape(len(A), 100)
    coef, res, *_ = np.linalg.lstsq(F.T, M[:14].reshape(14, 100).T, rcond=None)
    assert np.abs(F.T @ coef - M[:14].reshape(14, 100).T).max() < 1e-12          # inside span A_i
    assert 14 == 21 - np.linalg.matrix_rank(np.stack([a @ zI for a in A], 1))    # ... and they span all of { X in span A_i : X z_I = 0 }


def test_scalar_core_with_the_newton_solve_of_the_dual():
    """opts.dual_refine = 2 (cvx::dual_newton; the scalar core only): the same certified optimum and certificate statement, and what the
    eigen-gradient step leaves at 11 and more iterations stops at the third attempt -- the dual family holds a certifying member as soon as
    the pose is final."""
    import hostsim
    from cvxpnpl_amd import synth
    from test_gpu_parity import geodesic_np

    d = synth.make_pnpl(20000, 10, 0, 2.0, seed=42)
    a = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts(first_check=6))
    b = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=hostsim.default_opts(first_check=6, dual_refine=2))
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    diff = np.flatnonzero(a["iters"] != b["iters"])
    assert len(diff) >= 1 and max(geodesic_np(a["R"][i], b["R"][i]) for i in diff) < 1e-8
    assert (b["iters"] <= a["iters"]).all() and (b["iters"] > 10).sum() < (a["iters"] > 10).sum()
    gap = b["cost"][:, 0] - b["cost"][:, 1]
    assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][:, 0])).all()


@pytest.mark.gpu
def test_the_eigen_gradient_step_on_the_device():
    """quad schedule (10 000 problems): the wave-per-problem phase behind the quad phase makes the step; statuses equal, stragglers shorter,
    poses the same certified optimum; fresh wave-per-problem launches and the lane schedule do not make it (same iteration counts)."""
    import torch
    from test_gpu_parity import _solve, geodesic_np

    from cvxpnpl_amd import synth

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    d = synth.make_pnpl(10000, 10, 0, 2.0, seed=3)
    a = _solve(dev, d, 10, 0, dual_refine=0)
    b = _solve(dev, d, 10, 0)
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    diff = np.flatnonzero(a["iters"] != b["iters"])
    assert len(diff) >= 5 and max(geodesic_np(a["R"][i], b["R"][i]) for i in diff) < 1e-8
    assert (b["iters"] <= a["iters"]).all() and (b["iters"] >= 11).sum() < (a["iters"] >= 11).sum()
    gap = b["cost"][:, 0] - b["cost"][:, 1]
    assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(b["cost"][:, 0])).all()
    for layout, n in ((2, 1500), (1, 30000)):
        dd = synth.make_pnpl(n, 10, 0, 2.0, seed=5)
        a = _solve(dev, dd, 10, 0, layout=layout, dual_refine=0)
        b = _solve(dev, dd, 10, 0, layout=layout)
        assert np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["status"], b["status"]), layout
    with pytest.raises(RuntimeError, match="bad options"):
        _solve(dev, d, 10, 0, dual_refine=2)
