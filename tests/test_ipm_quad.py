"""The interior-point solve with four problems per wavefront (cvxpnpl_amd/csrc/ipm_quad.h, C ABI cvxpnpl_ipm_batch) on its own:
against the optimality conditions of the SDP the reference hands to scs.solve (cvxpnpl.py:454-489: equality rows :387-451, 10 x 10 PSD
cone) and against the scalar host statement of the same method (csrc/ipm_core.h: ipm_solve, built into tests/hostsim).

CPU part: the host statement itself satisfies those conditions (so that it is a checker worth comparing with).
GPU part (-m gpu): every problem of ragged batches (1, 2, 3, 5, 1001 problems: every occupancy of a wavefront's four rows), both
constraint sets; tolerances are absolute, on trace-normalised costs: feasibility 5e-7 (measured 2e-9 in the host statement, 1.2e-7 on the device -- reciprocals from the hardware seed + two Newton steps, the
residual of the equalities taken into every right-hand side; the first-order iteration that takes the iterate over projects it back), duality gap 5e-6 (the solve stops at 1e-10 or, more often, where
rounding ends the progress of the gap: median 1e-9, worst 3e-8 in the host statement and 2e-6 over 40 000 solves on the device), primal objective within 5e-6
of the host statement's."""
import numpy as np
import pytest


def _costs(n, n_pts, sigma, seed):
    """trace-normalised costs of synthetic PnP problems: Q45 (packed 9 x 9) and Qs55 (vech of the 10 x 10, zero last column)"""
    from hostsim import assemble

    from cvxpnpl_amd import synth

    d = synth.make_pnp(n, n_pts, sigma=sigma, seed=seed)
    iu9, iu10 = np.triu_indices(9), np.triu_indices(10)
    sel = np.flatnonzero((iu10[0] < 9) & (iu10[1] < 9))
    Q45, Qs55 = np.zeros((n, 45)), np.zeros((n, 55))
    for b in range(n):
        rc, _, Q = assemble(d["pts_2d"][b], d["pts_3d"][b], None, None, d["K"])
        assert rc == 0
        Q45[b] = Q[iu9]
        Qs55[b, sel] = Q45[b] / np.trace(Q)
    return Q45, Qs55


def _check_optimality(Qs55, Z, S, gap, variant, tol_feas=5e-7, tol_gap=5e-6):
    from hostsim import ipm_rows

    A = ipm_rows(variant)                                   # [rows, 10, 10]
    n = len(Z)
    iu10 = np.triu_indices(10)
    Q = np.zeros((n, 10, 10))
    Q[:, iu10[0], iu10[1]] = Qs55
    Q = Q + np.transpose(Q, (0, 2, 1)) - np.einsum("bii->bi", Q)[:, :, None] * np.eye(10)
    assert np.abs(Z - np.transpose(Z, (0, 2, 1))).max() < 1e-12 and np.abs(S - np.transpose(S, (0, 2, 1))).max() < 1e-12
    # primal feasibility: <A_i, Z> = b_i (1 on the diagonal-sum rows and Z_99, 0 on the triples)
    b = np.array([np.trace(Ai @ np.diag(np.r_[np.full(9, 1 / 3), 1.0])) for Ai in A])   # (Z0 = blkdiag(I/3, 1) is feasible)
    feas = np.abs(np.einsum("rij,bij->br", A, Z) - b).max()
    assert feas < tol_feas, feas
    # dual feasibility: S - Q in the span of the rows
    Am = A.reshape(len(A), 100).T
    D = (S - Q).reshape(n, 100).T
    y = np.linalg.lstsq(Am, D, rcond=None)[0]
    dres = np.abs(Am @ y - D).max()
    assert dres < tol_feas, dres
    # cone membership and complementarity
    assert np.linalg.eigvalsh(Z).min() > -1e-12 and np.linalg.eigvalsh(S).min() > -1e-12
    g = np.einsum("bij,bij->b", Z, S)
    assert np.abs(g - gap).max() < 1e-12
    # (the 16-equality variant: a solve now and then ends early -- its Schur matrix stops being positive definite in rounding; the iterate
    #  it leaves is feasible and the first-order iteration goes on from it.  Host statement: 1 of 300 at 2e-3.)
    conv = gap < tol_gap
    assert conv.all() if variant == 0 else conv.mean() >= 0.98, (gap.max(), np.sum(~conv))
    return np.einsum("bij,bij->b", Q, Z), conv


@pytest.mark.parametrize("variant", [0, 1])
def test_host_statement_meets_the_optimality_conditions(variant):
    from hostsim import ipm_solve

    _, Qs = _costs(200, 4, 2.0, 11)
    Z, S, gap, it = ipm_solve(Qs, variant)
    _check_optimality(Qs, Z, S, gap, variant)
    assert it.min() >= 5 and it.max() <= 40


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n,n_pts,sigma", [(1, 4, 2.0), (2, 4, 2.0), (3, 6, 1.0), (5, 4, 0.0), (1001, 4, 2.0), (600, 10, 2.0), (400, 5, 3.0)])
def test_four_per_wavefront_solve(variant, n, n_pts, sigma):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from hostsim import ipm_solve

    import cvxpnpl_amd as ca

    Q45, Qs = _costs(n, n_pts, sigma, 100 + n)
    Zd, Sd, gapd, itd = ca.ipm_batch(torch.as_tensor(3.7 * Q45, device="cuda"), variant=variant)   # (any scale: normalised inside)
    Z, S, gap, it = Zd.cpu().numpy(), Sd.cpu().numpy(), gapd.cpu().numpy(), itd.cpu().numpy() & 255   # (reason of the stop: bits 8..)
    assert np.isfinite(Z).all() and np.isfinite(S).all()
    obj, conv = _check_optimality(Qs, Z, S, gap, variant)
    Zh, Sh, gaph, ith = ipm_solve(Qs, variant)
    objh, convh = _check_optimality(Qs, Zh, Sh, gaph, variant)
    both = conv & convh
    assert np.abs(obj - objh)[both].max() < 5e-6, np.abs(obj - objh)[both].max()
    assert it.min() >= 5 and it.max() <= 40 and abs(float(it.mean()) - float(ith.mean())) < 3.0, (it.mean(), ith.mean())
    # where the optimum is one point (rank-1 Z: the typical problem) the two solves agree on it
    lam = np.linalg.eigvalsh(Zh)
    tight = (lam[:, -2] < 1e-6) & both
    if tight.any():
        assert np.abs(Z[tight] - Zh[tight]).max() < 1e-3, np.abs(Z[tight] - Zh[tight]).max()


@pytest.mark.gpu
def test_degenerate_costs_end_the_solve_without_hanging():
    """a NaN cost, a zero cost, an empty batch: the kernel's loops are bounded by iteration counts, never by the data"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import _lib

    Q45, _ = _costs(8, 4, 2.0, 9)
    Q45[1] = np.nan                     # NaN: no factorisation succeeds -> the solve ends at once
    Q45[2, 1:] = 0.0                    # a rank-one cost e0 e0^T
    Z, S, gap, it = [x.cpu().numpy() for x in ca.ipm_batch(torch.as_tensor(Q45, device="cuda"))]
    assert (it[1] & 255) == 0 and (it[1] >> 8) in (2, 3, 4, 5)          # stopped by a failed factorisation / no step / no progress
    ok = np.array([0, 2, 3, 4, 5, 6, 7])
    assert np.isfinite(Z[ok]).all() and np.isfinite(S[ok]).all() and (gap[ok] < 1e-6).all()
    L = _lib.lib()
    assert L.cvxpnpl_ipm_batch(0, None, 0, None, None, None, None, None) == -1     # (null pointers are refused even for an empty batch)
    z = torch.empty((1, 100), dtype=torch.float64, device="cuda")
    assert L.cvxpnpl_ipm_batch(0, z.data_ptr(), 0, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None) == 0
    assert L.cvxpnpl_ipm_batch(4, z.data_ptr(), 7, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None) == -1   # unknown variant
