"""Row f2 of SURVEY.md section 8 pinned to the REFERENCE: the error metrics, the pose disambiguation and the synthetic
generator of the product (cvxpnpl_amd/metrics.py, synth.py on the host; cvxpnpl_pose_errors, cvxpnpl_disambiguate,
cvxpnpl_synth_batch on the device) against vectors produced by the reference's own functions
(tests/golden/make_golden_harness.py imports benchmarks/toolkit/suites/suite.py and synth.py; G9).

Tolerances: metrics 1e-9 deg / 1e-12 relative (acos near 0 and pi loses half the digits: 2e-6 deg there); the chosen
candidate is an index: exact; seeded single draws of the host generator: translation and 3D points bit-equal (the same
MT19937 stream), rotations to 5e-15 (the reference normalises angle * axis once more); the device generator is another stream (Philox): moments within 4 standard errors of the reference's."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g9():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors_harness.npz"), allow_pickle=False))


# ------------------------------------------------------------------------------------------------ host twins (CPU)
def test_rotation_angle_is_the_references_angle(g9):
    from cvxpnpl_amd import metrics

    mine = metrics.rotation_angle(g9["g9_angle_in"])
    ref = g9["g9_angle_out"]
    edge = (ref < 1e-3) | (ref > np.pi - 1e-3)
    assert np.abs(mine - ref)[~edge].max() < 1e-12 and np.abs(mine - ref)[edge].max() < 5e-8
    assert np.array_equal(g9["g9_angle_batched_out"], ref)  # (the reference's function on the stack == one by one)


def test_projection_is_the_references(g9):
    from cvxpnpl_amd import metrics, synth

    ref = g9["g9_proj_out"]
    a = metrics._project(g9["g9_proj_P"], g9["g9_proj_K"], g9["g9_proj_R"], g9["g9_proj_t"])
    b = synth.project(g9["g9_proj_P"][None], g9["g9_proj_K"], g9["g9_proj_R"][None], g9["g9_proj_t"][None])[0]
    assert np.abs(a - ref).max() < 1e-10 and np.abs(b - ref).max() < 1e-10  # pixels ~ 500: 1e-10 is 2 ulp


def test_pose_errors_are_the_references(g9):
    from cvxpnpl_amd import metrics

    ang, tr = metrics.pose_errors(g9["g9_err_Rgt"], g9["g9_err_tgt"], g9["g9_err_R"], g9["g9_err_t"])
    raised = g9["g9_err_raised"] == 1  # NaN estimates: the reference's SVD raises LinAlgError, the twin reports NaN
    assert raised.sum() == 20 and np.isnan(ang[raised]).all() and np.isnan(tr[raised]).all()
    ra, rt = g9["g9_err_ang_deg"][~raised], g9["g9_err_trans"][~raised]
    edge = (ra < 0.1) | (ra > 179.9)
    assert np.abs(ang[~raised] - ra)[~edge].max() < 1e-9 and np.abs(ang[~raised] - ra)[edge].max() < 2e-6
    assert np.abs(tr[~raised] - rt).max() < 1e-12 * max(1.0, rt.max())
    assert (ra > 90).sum() >= 30 and (ra < 1).sum() >= 40  # the vectors cover reflections / far and close estimates


def test_disambiguation_picks_what_the_references_loop_picks(g9):
    """suite.py:96-108 with the support points it drew (np.random seeded per case; RandomState(seed) reproduces them)"""
    from cvxpnpl_amd import metrics

    n = len(g9["g9_dis_seed"])
    assert np.bincount(g9["g9_dis_n"])[[1, 2, 3, 4]].min() >= 8
    for i in range(n):
        seed = int(g9["g9_dis_seed"][i])
        assert np.array_equal(np.random.RandomState(seed).random_sample((20, 3)) - 0.5, g9["g9_dis_support"][i])
        R, t, idx = metrics.disambiguate(g9["g9_dis_R_all"][i:i + 1], g9["g9_dis_t_all"][i:i + 1], g9["g9_dis_n"][i:i + 1], g9["g9_dis_K"],
                                         g9["g9_dis_Rgt"][i:i + 1], g9["g9_dis_tgt"][i:i + 1], n_support=20, seed=seed)
        assert idx[0] == g9["g9_dis_pick"][i], (i, idx[0], g9["g9_dis_pick"][i])
        assert np.array_equal(R[0], g9["g9_dis_R_all"][i, idx[0]], equal_nan=True)


def test_host_generator_reproduces_the_references_draws(g9):
    """one problem per RandomState: the reference's draw order (pose 3 + 1 + 2 + 1 uniforms, points, noise; synth.py:27-42,
    :277-285, run loop :238-246) -- the same MT19937 stream gives the same problem to rounding"""
    from cvxpnpl_amd import synth

    s0 = int(g9["g9_rp_seed0"])
    for k in range(len(g9["g9_rp_R"])):
        R, t = synth.random_poses(np.random.RandomState(s0 + k), 1)
        # (the reference re-normalises angle * axis in aa2rm: a few ulp)
        assert np.abs(R[0] - g9["g9_rp_R"][k]).max() < 5e-15 and np.abs(t[0] - g9["g9_rp_t"][k]).max() == 0.0
    for tag in ("pnp", "pnp0"):
        n, sigma, seed = int(g9[f"g9_gen_{tag}_n"]), float(g9[f"g9_gen_{tag}_sigma"]), int(g9[f"g9_gen_{tag}_seed"])
        d = synth.make_pnpl(1, n, 0, sigma, seed=seed)
        assert np.array_equal(d["K"], g9[f"g9_gen_{tag}_K"]) and synth.LENGTH == float(g9[f"g9_gen_{tag}_length"])
        assert np.abs(d["R_gt"][0] - g9[f"g9_gen_{tag}_R"]).max() < 5e-15
        assert np.abs(d["pts_3d"][0] - g9[f"g9_gen_{tag}_pts_3d"]).max() == 0.0
        assert np.abs(d["pts_2d"][0] - g9[f"g9_gen_{tag}_pts_2d"]).max() < 1e-10
    n, sigma, seed = int(g9["g9_gen_pnl_n"]), float(g9["g9_gen_pnl_sigma"]), int(g9["g9_gen_pnl_seed"])
    d = synth.make_pnpl(1, 0, n, sigma, seed=seed)
    assert np.abs(d["line_3d"][0] - g9["g9_gen_pnl_line_3d"]).max() == 0.0      # (line, end point, xyz): consecutive point pairs
    assert np.abs(d["line_2d"][0] - g9["g9_gen_pnl_line_2d"]).max() < 1e-10
    # aa2rm (synth.py:12-24) through the twin's Rodrigues formula
    for aa, Rr in zip(g9["g9_aa_in"], g9["g9_aa_out"]):
        ang = np.linalg.norm(aa)
        if ang < 2.220446049250313e-16:
            continue  # the reference returns I below machine epsilon; the twin never draws such an angle
        k = aa / ang
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        assert np.abs(np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx) - Rr).max() < 1e-15


def _moment_checks(g9, R, t, n):
    se = lambda s: 4.0 * s / np.sqrt(min(n, int(g9["g9_rp_moments_n"])) / 2.0)  # noqa: E731  (4 standard errors, both samples finite)
    assert np.abs(t.mean(0) - g9["g9_rp_t_mean"]).max() < se(0.47)
    assert np.abs(t.std(0) - g9["g9_rp_t_std"]).max() < se(0.47)
    assert (t.min(0) >= [-0.5, -0.5, 0.6]).all() and (t.max(0) < [0.5, 0.5, 2.2]).all()
    ang = np.arccos(np.clip(0.5 * (np.trace(R, axis1=1, axis2=2) - 1), -1, 1))
    assert abs(ang.mean() - float(g9["g9_rp_angle_mean"])) < se(0.91) and abs(ang.std() - float(g9["g9_rp_angle_std"])) < se(0.91)
    assert np.abs(R.mean(0) - g9["g9_rp_R_mean"]).max() < se(0.6)  # incl. the reference's non-uniform axis law (cube, normalised)


def test_generators_have_the_references_pose_moments(g9):
    from cvxpnpl_amd import synth

    R, t = synth.random_poses(np.random.RandomState(11), 20000)
    _moment_checks(g9, R, t, 20000)
    d = synth.philox_pnpl(20000, 1, 0, sigma=0.0, seed=11)  # the numpy restatement of the device generator
    _moment_checks(g9, d["R_gt"], d["t_gt"], 20000)


# ------------------------------------------------------------------------------------------------ device kernels (GPU)
@pytest.fixture(scope="module")
def gpu():
    import torch

    from cvxpnpl_amd import _lib

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_device_pose_errors_are_the_references(gpu, g9):
    import torch

    from cvxpnpl_amd import metrics

    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    ang, tr = metrics.pose_errors_device(tt(g9["g9_err_Rgt"]), tt(g9["g9_err_tgt"]), tt(g9["g9_err_R"]), tt(g9["g9_err_t"]))
    ang, tr = ang.cpu().numpy(), tr.cpu().numpy()
    raised = g9["g9_err_raised"] == 1
    assert np.isnan(ang[raised]).all()
    ra, rt = g9["g9_err_ang_deg"][~raised], g9["g9_err_trans"][~raised]
    edge = (ra < 0.1) | (ra > 179.9)
    assert np.abs(ang[~raised] - ra)[~edge].max() < 1e-8 and np.abs(ang[~raised] - ra)[edge].max() < 2e-6
    assert np.abs(tr[~raised] - rt).max() < 1e-12 * max(1.0, rt.max())
    # angle() alone: R_gt = I makes the kernel's R_gt^-1 R the input itself
    n = len(g9["g9_angle_in"])
    eye = np.broadcast_to(np.eye(3), (n, 3, 3)).copy()
    a2, _ = metrics.pose_errors_device(tt(eye), tt(np.ones((n, 3))), tt(g9["g9_angle_in"]), tt(np.ones((n, 3))))
    ref = np.degrees(g9["g9_angle_out"])
    e2 = (ref < 0.1) | (ref > 179.9)
    d = np.abs(a2.cpu().numpy() - ref)
    assert d[~e2].max() < 1e-8 and d[e2].max() < 5e-6


@pytest.mark.gpu
def test_device_disambiguation_picks_what_the_references_loop_picks(gpu, g9):
    import torch

    from cvxpnpl_amd import metrics

    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    for i in range(len(g9["g9_dis_seed"])):
        R, t, idx = metrics.disambiguate_device(tt(g9["g9_dis_R_all"][i:i + 1]), tt(g9["g9_dis_t_all"][i:i + 1]), g9["g9_dis_n"][i:i + 1],
                                                g9["g9_dis_K"], tt(g9["g9_dis_Rgt"][i:i + 1]), tt(g9["g9_dis_tgt"][i:i + 1]), n_support=20,
                                                seed=int(g9["g9_dis_seed"][i]))
        k = int(idx.cpu()[0])
        assert k == g9["g9_dis_pick"][i], (i, k)
        assert np.array_equal(R.cpu().numpy()[0], g9["g9_dis_R_all"][i, k], equal_nan=True)


@pytest.mark.gpu
def test_device_generator_has_the_references_pose_moments(gpu, g9):
    from cvxpnpl_amd import synth

    d = synth.device_pnpl(20000, 4, 0, sigma=0.0, seed=5, device=gpu)
    _moment_checks(g9, d["R_gt"].cpu().numpy(), d["t_gt"].cpu().numpy(), 20000)
    P = d["pts_3d"].cpu().numpy()
    assert P.min() >= -0.3 and P.max() < 0.3 and abs(P.std() - 0.6 / np.sqrt(12)) < 0.002  # synth.py:279
