"""The reference's "rc" ablation (benchmarks/toolkit/methods/rc.py: 16 equalities, row orthonormality dropped) and
the solve at the _solve_relaxation(A, B) seam (cvxpnpl.py:454-460).

CPU: the oracle's restatement against vectors produced by the reference's own rc.py (tests/golden/
reference_vectors_rc.npz, make_golden_rc.py) -- _A_rc bit-exact, recovery on injected x, end-to-end poses -- and the
device algorithm (host build) against the oracle.  GPU (-m gpu): cvxpnpl_solve_cost_batch in every layout and the rc
variant against the oracle and the golden poses."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def grc():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors_rc.npz"), allow_pickle=False))


def _geo(Ra, Rb):
    from cvxpnpl_amd import synth

    return float(synth.geodesic(np.asarray(Ra)[None], np.asarray(Rb)[None])[0])


def test_rc_static_constraints_bit_exact(orc, grc):
    Ad, b = orc.sdp_constraints_rc()
    assert np.array_equal(Ad, grc["rc_A"]) and np.array_equal(b, grc["rc_b"])
    assert int(grc["cone_f"]) == 16 and list(np.atleast_1d(grc["cone_s"])) == [10]
    assert np.linalg.matrix_rank(Ad[:16]) == 16  # (the full set has rank 21 of 22)


def test_rc_recovery_on_injected_solutions(orc, grc):
    """rc.py:104-131 through the oracle's recovery: rank-1 (exact, perturbed), rank-2, NaN sentinel."""
    A, B = grc["inj_A"], grc["inj_B"]
    for tag in ("r1", "r1p", "r2"):
        poses, st, rk = orc.recover(grc[f"inj_{tag}_x"], 0.0, A, B)
        assert len(poses) == len(grc[f"inj_{tag}_R"])
        for R, t in poses:
            assert min(_geo(R, Rg) + np.abs(t - tg).max() for Rg, tg in zip(grc[f"inj_{tag}_R"], grc[f"inj_{tag}_t"])) < 1e-9
    assert np.isnan(grc["inj_nan_R"]).all() and np.isnan(grc["inj_nan_t"]).all()
    poses, st, rk = orc.recover(np.full(55, np.nan), 0.0, A, B)
    assert len(poses) == 1 and np.isnan(poses[0][0]).all()


def test_rc_end_to_end_oracle_and_device_algorithm(orc, grc):
    """_solve_relaxation_rc's poses (reference post-processing on the oracle's converged rc solve), reproduced by
    the oracle's own driver and by the device algorithm's rc instantiation (host build)."""
    import hostsim
    from cvxpnpl_amd.api import pack_cost

    for i in range(int(grc["e2e_count"])):
        A, B = grc[f"e2e_{i}_A"], grc[f"e2e_{i}_B"]
        poses, info = orc.solve_relaxation_rc(A, B, eps=1e-11, max_iters=400000)
        assert len(poses) == 1 and info.rank == 1
        assert _geo(poses[0][0], grc[f"e2e_{i}_R"]) < 1e-9 and np.abs(poses[0][1] - grc[f"e2e_{i}_t"]).max() < 1e-9
        # the reference hands scs c = vech(Q, 2) (rc.py:88): same numbers as the packed cost, off-diagonals doubled
        Q = A.T @ A
        assert np.allclose(grc[f"e2e_{i}_c"][:9], np.concatenate([[Q[0, 0]], 2 * Q[0, 1:9]]), rtol=1e-12, atol=1e-15)
        h = hostsim.solve_cost_batch(pack_cost(Q)[None], B.reshape(1, 27), variant=1, want_Z=True)
        assert h["status"][0] == 0
        assert _geo(h["R"][0], grc[f"e2e_{i}_R"]) < 1e-6 and np.abs(h["t"][0] - grc[f"e2e_{i}_t"]).max() < 1e-6
        # Z of the certified solve satisfies the 16 equalities and violates none of them
        Ad, b = orc.sdp_constraints_rc()
        x = h["Z"][0]
        assert np.abs(Ad[:16] @ x - b[:16]).max() < 1e-12


def test_rc_affine_projection_matches_dense_projector(orc):
    """closed-form projection of the rc instantiation == dense projection built from the reference's _A_rc rows"""
    import hostsim

    Ad, b = orc.sdp_constraints_rc()
    A16 = Ad[:16]
    P = A16.T @ np.linalg.solve(A16 @ A16.T, np.eye(16))
    rs = np.random.RandomState(3)
    D = np.where(np.array([i == j for i in range(10) for j in range(i, 10)]), 1.0, 2.0)  # <A, Z> weights of vech
    for homog in (0, 1):
        for _ in range(5):
            E = rs.normal(size=55)
            got = hostsim.proj_affine_rc(E, homog)
            # orthogonal projection in the Frobenius metric of the symmetric matrix: rows of _A weight off-diagonals by 2
            Aw = A16 / D
            rhs = (0 if homog else 1) * b[:16] - A16 @ E
            lam = np.linalg.solve(Aw @ (Aw * D).T, rhs)
            want = E + (Aw.T @ lam)
            assert np.abs(got - want).max() < 1e-12


def test_cost_seam_host_build_equals_correspondence_entry():
    """solve at the (Q, B) seam == solve from correspondences (same iterates: the assembly is the only difference)"""
    import hostsim
    from cvxpnpl_amd import synth
    from cvxpnpl_amd.api import pack_cost

    d = synth.make_pnpl(48, 5, 5, 1.0, seed=8)
    a = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], d["line_2d"], d["line_3d"], d["K"])
    Q, B = [], []
    for i in range(48):
        rc_, Bm, Qm = hostsim.assemble(d["pts_2d"][i], d["pts_3d"][i], d["line_2d"][i], d["line_3d"][i], d["K"])
        Q.append(pack_cost(Qm))
        B.append(Bm.reshape(27))
    b = hostsim.solve_cost_batch(np.array(Q), np.array(B))
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"])
    assert np.abs(a["R"] - b["R"]).max() == 0.0


# ------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu():
    import torch

    from cvxpnpl_amd import _lib

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [1, 2, 3])
def test_gpu_cost_entry_equals_correspondence_entry(gpu, layout):
    """cvxpnpl_solve_cost_batch(Q45, B27) fed with cvxpnpl_assemble_batch's outputs reproduces cvxpnpl_solve_batch
    (PnP and PnPL, every layout): same statuses, poses to rounding."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    for n_p, n_l, batch in ((10, 0, 4200), (5, 5, 900)):
        d = synth.make_pnpl(batch, n_p, n_l, 1.0, seed=31 + n_l)
        tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
        args = (tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
                tt(d["line_3d"]) if n_l else None, tt(d["K"]))
        ref = ca.pnpl_batch(*args, layout=layout, want_Z=True)
        Bt, Qt = ca.assemble_batch(*args)
        res = ca.solve_cost_batch(Qt, Bt, layout=layout, want_Z=True)
        torch.cuda.synchronize()
        st0, st1 = ref.status.cpu().numpy(), res.status.cpu().numpy()
        assert (st0 == st1).mean() > 0.999
        both = (st0 == 0) & (st1 == 0)
        assert both.mean() > 0.99
        R0, R1 = ref.R.cpu().numpy(), res.R.cpu().numpy()
        assert synth.geodesic(R0, R1)[both].max() < 1e-9
        assert np.abs(ref.t.cpu().numpy() - res.t.cpu().numpy())[both].max() < 1e-9
        assert np.abs(ref.Z.cpu().numpy() - res.Z.cpu().numpy())[both].max() < 1e-8


@pytest.mark.gpu
def test_gpu_rc_variant_vs_oracle_and_reference_vectors(gpu, orc, grc):
    """The 16-equality variant on the GPU: the reference's own rc poses (golden e2e), the oracle's rc solve on a
    seeded batch (<= 1e-6 rad / 1e-6 relative t), certificates valid, Z feasible for the 16 equalities."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    for i in range(int(grc["e2e_count"])):
        poses = ca.solve_relaxation_rc(grc[f"e2e_{i}_A"], grc[f"e2e_{i}_B"])
        assert len(poses) == 1
        assert _geo(poses[0][0], grc[f"e2e_{i}_R"]) < 1e-6 and np.abs(poses[0][1] - grc[f"e2e_{i}_t"]).max() < 1e-6
    # the full-set drop-in at the same seam agrees with pnp()
    d = synth.make_pnp(1, 8, 1.0, seed=12)
    (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][0], d["pts_3d"][0], d["K"])
    B, A = orc.eliminate(np.vstack((c1, c2, c3)), np.vstack((n1, n2, n3)))
    p_seam = ca.solve_relaxation(A, B)
    p_api = ca.pnp(d["pts_2d"][0], d["pts_3d"][0], d["K"])
    assert _geo(p_seam[0][0], p_api[0][0]) < 1e-9 and np.abs(p_seam[0][1] - p_api[0][1]).max() < 1e-9
    # batch, rc vs oracle
    d = synth.make_pnp(160, 10, 1.0, seed=77)
    Bt, Qt = ca.assemble_batch(torch.as_tensor(d["pts_2d"], device=gpu), None, torch.as_tensor(d["pts_3d"], device=gpu), None, d["K"])
    res = ca.solve_cost_batch(Qt, Bt, variant=ca.VARIANT_RC, want_Z=True)
    st = res.status.cpu().numpy()
    assert (st == 0).mean() > 0.97, np.bincount(st)
    Ad, b = orc.sdp_constraints_rc()
    Zc = res.Z.cpu().numpy()[st == 0]
    assert np.abs(Zc @ Ad[:16].T - b[:16]).max() < 1e-10
    c = res.cost.cpu().numpy()[st == 0]
    assert ((c[:, 0] - c[:, 1]) >= -1e-15).all() and ((c[:, 0] - c[:, 1]) <= 1e-9).all()
    R, t = res.R.cpu().numpy(), res.t.cpu().numpy()
    n_cmp = 0
    for i in range(0, 160, 4):
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        B, A = orc.eliminate(np.vstack((c1, c2, c3)), np.vstack((n1, n2, n3)))
        poses, info = orc.solve_relaxation_rc(A, B, eps=1e-11, max_iters=400000)
        if st[i] != 0 or len(poses) != 1:
            continue
        assert _geo(R[i], poses[0][0]) < 1e-6 and np.linalg.norm(t[i] - poses[0][1]) / np.linalg.norm(poses[0][1]) < 1e-6
        n_cmp += 1
    assert n_cmp >= 36


@pytest.mark.gpu
def test_gpu_rc_quad_schedule_equals_wave_and_oracle(gpu, orc):
    """Round 3: the 16-equality variant in the quad schedule (four problems per wavefront, solve_quad_kernel<..., VAR_RC>, the
    wavefront's leftovers and the planar queue through the rc wave kernel) -- same statuses as the wave-per-problem layout,
    certified poses equal to 1e-9 (both Newton-polish the same stationary point), Z feasible for the reference's 16 rows, a
    sample against the oracle's rc solve (<= 1e-6), and the AUTO policy picks it for a mid-size batch.
    Match: benchmarks/toolkit/methods/rc.py:67-131."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_pnp(3000, 10, 1.0, seed=78)
    Bt, Qt = ca.assemble_batch(torch.as_tensor(d["pts_2d"], device=gpu), None, torch.as_tensor(d["pts_3d"], device=gpu), None, d["K"])
    rw = ca.solve_cost_batch(Qt, Bt, variant=ca.VARIANT_RC, want_Z=True, layout=2)
    outs = {}
    for name, kw in (("quad", dict(layout=3)), ("auto", dict()), ("quad_short", dict(layout=3, lane_iters=9)), ("lane->quad", dict(layout=1))):
        r = ca.solve_cost_batch(Qt, Bt, variant=ca.VARIANT_RC, want_Z=True, **kw)
        outs[name] = {k: v.cpu().numpy() for k, v in r.items()}
    w = {k: v.cpu().numpy() for k, v in rw.items()}
    Ad, b = orc.sdp_constraints_rc()
    for name, r in outs.items():
        assert (r["status"] == w["status"]).mean() > 0.998, (name, np.flatnonzero(r["status"] != w["status"])[:10])
        both = (r["status"] == 0) & (w["status"] == 0)
        assert both.mean() > 0.97, name
        assert synth.geodesic(r["R"], w["R"])[both].max() < 1e-9 and np.abs(r["t"] - w["t"])[both].max() < 1e-9, name
        assert np.abs(r["Z"][r["status"] == 0] @ Ad[:16].T - b[:16]).max() < 1e-10
        c = r["cost"][r["status"] == 0]
        assert ((c[:, 0] - c[:, 1]) >= -1e-15).all() and ((c[:, 0] - c[:, 1]) <= 1.0001e-9).all()
    assert np.array_equal(outs["auto"]["status"], outs["quad"]["status"]) and np.array_equal(outs["auto"]["iters"], outs["quad"]["iters"])  # AUTO at 3 000 = quad
    q = outs["quad"]
    n_cmp = 0
    for i in range(0, 3000, 100):
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        B, A = orc.eliminate(np.vstack((c1, c2, c3)), np.vstack((n1, n2, n3)))
        poses, info = orc.solve_relaxation_rc(A, B, eps=1e-11, max_iters=400000)
        if q["status"][i] != 0 or len(poses) != 1:
            continue
        assert _geo(q["R"][i], poses[0][0]) < 1e-6 and np.linalg.norm(q["t"][i] - poses[0][1]) / np.linalg.norm(poses[0][1]) < 1e-6
        n_cmp += 1
    assert n_cmp >= 26


def test_rc_interior_point_statement_vs_oracle_and_first_order(orc):
    """The interior-point path on the 16-row constraint set (csrc/ipm_core.h, cvx::ipm_rows(VAR_RC): twelve triples, three column
    sums, Z99 -- the scalar statement of what cvxw::coop_ipm<VAR_RC> runs in rescue_wave_kernel_rc): the same SDP as the
    first-order rc solve and as the oracle's restated SCS on the reference's own _A_rc (rc.py:9-64), reached in <= 25
    iterations whatever the conditioning; and its optimal Z satisfies the reference's 16 rows."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hostsim
    from cvxpnpl_amd import synth

    d = synth.make_pnp(120, 10, 1.0, seed=77)
    o = hostsim.default_opts(variant=1)
    ip = hostsim.ipm_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], opts=o, want_Z=True)
    Bs, Qs = [], []
    for i in range(120):
        _, B, Q = hostsim.assemble(d["pts_2d"][i], d["pts_3d"][i], None, None, d["K"])
        Bs.append(B.reshape(-1))
        Qs.append(np.array([Q[a, b] for a in range(9) for b in range(a, 9)]))
    fo = hostsim.solve_cost_batch(np.array(Qs), np.array(Bs), variant=1, want_Z=True)
    assert ip["iters"].max() <= 25
    both = (ip["status"] == 0) & (fo["status"] == 0)
    assert both.sum() >= 110 and (ip["status"] == 0).sum() >= (fo["status"] == 0).sum()
    assert synth.geodesic(ip["R"], fo["R"])[both].max() < 1e-7 and np.abs(ip["t"] - fo["t"])[both].max() < 1e-7
    Ad, b = orc.sdp_constraints_rc()
    assert np.abs(ip["Z"][ip["status"] == 0] @ Ad[:16].T - b[:16]).max() < 1e-9
    n_cmp = 0
    for i in range(0, 120, 6):
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        B, A = orc.eliminate(np.vstack((c1, c2, c3)), np.vstack((n1, n2, n3)))
        poses, info = orc.solve_relaxation_rc(A, B, eps=1e-11, max_iters=400000)
        if ip["status"][i] != 0 or len(poses) != 1:
            continue
        assert _geo(ip["R"][i], poses[0][0]) < 1e-6 and np.linalg.norm(ip["t"][i] - poses[0][1]) / np.linalg.norm(poses[0][1]) < 1e-6
        n_cmp += 1
    assert n_cmp >= 16
    # minimal problems: the rc relaxation is rarely tight there; the interior-point iterate is the SDP optimum all the same
    dm = synth.make_pnp(64, 4, 1.0, seed=5)
    im = hostsim.ipm_batch(dm["pts_2d"], dm["pts_3d"], None, None, dm["K"], opts=o, want_Z=True)
    assert im["iters"].max() <= 30 and np.isfinite(im["R"]).all()
    fin = np.isfinite(im["Z"]).all(axis=1)
    assert np.abs(im["Z"][fin] @ Ad[:16].T - b[:16]).max() < 1e-7


@pytest.mark.gpu
def test_gpu_rc_interior_point_rescue(gpu, orc):
    """opts.rescue_from for the rc variant (round 3; default 48): a problem whose relaxation is not tight no longer runs to
    max_iters -- the launch's slowest problem stays below rescue_from + ~40 -- what certifies without the path certifies with it,
    to the same pose, and the rescued problems themselves match the oracle's rc solve."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_pnp(6000, 10, 2.0, seed=42)
    Bt, Qt = ca.assemble_batch(torch.as_tensor(d["pts_2d"], device=gpu), None, torch.as_tensor(d["pts_3d"], device=gpu), None, d["K"])
    base = {k: v.cpu().numpy() for k, v in ca.solve_cost_batch(Qt, Bt, variant=ca.VARIANT_RC, rescue_from=0, layout=2).items()}
    for layout in (2, 3, 0):
        r = {k: v.cpu().numpy() for k, v in ca.solve_cost_batch(Qt, Bt, variant=ca.VARIANT_RC, layout=layout, want_Z=True).items()}
        assert r["iters"].max() <= 48 + 16 + 64 + 4, r["iters"].max()  # hand-over at 48, <= ~16 second-order iterations, the 64-iteration grace
        both = (r["status"] == 0) & (base["status"] == 0)
        assert (r["status"] == 0).sum() >= (base["status"] == 0).sum() and both.mean() > 0.98
        assert synth.geodesic(r["R"], base["R"])[both].max() < 1e-7
        assert np.isfinite(r["R"]).all()
    resc = np.flatnonzero(r["iters"] > 48)
    assert len(resc) >= 3
    n_cmp = 0
    for i in resc[:24]:
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], d["K"])
        B, A = orc.eliminate(np.vstack((c1, c2, c3)), np.vstack((n1, n2, n3)))
        poses, info = orc.solve_relaxation_rc(A, B, eps=1e-11, max_iters=400000)
        if r["status"][i] != 0 or len(poses) != 1:
            continue
        assert _geo(r["R"][i], poses[0][0]) < 1e-6
        n_cmp += 1
    assert n_cmp >= 1 or (r["status"][resc] != 0).all()
