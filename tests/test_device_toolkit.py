"""The reference's benchmark toolkit on the device: synthetic problem generator, error
metrics and pose disambiguation as HIP kernels, against their numpy counterparts (cvxpnpl_amd/synth.py, metrics.py --
themselves following benchmarks/toolkit/suites/synth.py:27-42, :276-346 and suite.py:8-33, :96-108).

CPU: the numpy restatement of the counter-based generator has the reference's distributions.  GPU: device == numpy."""
import numpy as np
import pytest


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): counter / key all zero, and all ones."""
    from cvxpnpl_amd.synth import _philox4x32

    out = _philox4x32(np.zeros((1, 4), dtype=np.uint64), 0, 0)
    assert [int(x[0]) for x in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = _philox4x32(np.full((1, 4), 0xFFFFFFFF, dtype=np.uint64), 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x[0]) for x in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


def test_philox_generator_has_the_reference_distributions():
    from cvxpnpl_amd import synth

    d = synth.philox_pnpl(20000, 3, 2, sigma=1.5, seed=7)
    R, t = d["R_gt"], d["t_gt"]
    assert np.abs(R @ np.swapaxes(R, 1, 2) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(R) - 1).max() < 1e-12
    ang = np.arccos(np.clip(0.5 * (np.trace(R, axis1=1, axis2=2) - 1), -1, 1))
    # rotation angle 2 pi U folded onto [0, pi]: uniform; t: U-.5, U-.5, 1.6 U + .6 (synth.py:36, :41)
    assert abs(ang.mean() - np.pi / 2) < 0.03 and abs(ang.std() - np.pi / np.sqrt(12)) < 0.03
    assert abs(t[:, 0].mean()) < 0.01 and abs(t[:, 1].std() - 1 / np.sqrt(12)) < 0.01
    assert t[:, 2].min() >= 0.6 and t[:, 2].max() < 2.2 and abs(t[:, 2].mean() - 1.4) < 0.01
    P = np.concatenate([d["pts_3d"].reshape(-1, 3), d["line_3d"].reshape(-1, 3)])
    assert P.min() >= -0.3 and P.max() < 0.3 and abs(P.std() - 0.6 / np.sqrt(12)) < 0.002  # 0.6 (U - .5), synth.py:279
    clean = synth.philox_pnpl(20000, 3, 2, sigma=0.0, seed=7)
    noise = np.concatenate([(d["pts_2d"] - clean["pts_2d"]).ravel(), (d["line_2d"] - clean["line_2d"]).ravel()])
    assert abs(noise.mean()) < 0.02 and abs(noise.std() - 1.5) < 0.02
    assert abs(np.mean(noise ** 4) / noise.std() ** 4 - 3.0) < 0.1  # Gaussian kurtosis
    # noise-free pixels are the projections (suite.py:17-19)
    assert np.abs(clean["pts_2d"] - synth.project(clean["pts_3d"], clean["K"], clean["R_gt"], clean["t_gt"])).max() < 1e-9
    # different seeds / problems are different streams
    e = synth.philox_pnpl(8, 3, 2, sigma=0.0, seed=8)
    assert np.abs(e["pts_3d"] - clean["pts_3d"][:8]).min() > 1e-9


@pytest.fixture(scope="module")
def gpu():
    import torch

    from cvxpnpl_amd import _lib

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("batch,n_p,n_l,sigma", [(1, 6, 0, 0.0), (1000, 10, 0, 2.0), (257, 5, 5, 1.0), (64, 0, 7, 0.5), (3, 0, 0, 0.0)])
def test_device_generator_matches_numpy_restatement(gpu, batch, n_p, n_l, sigma):
    """cvxpnpl_synth_batch == its numpy restatement: identical uniform draws (exact integer arithmetic), poses / points /
    pixels to libm rounding; and the solver recovers the generated ground truth on noise-free data."""
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.device_pnpl(batch, n_p, n_l, sigma=sigma, seed=1234 + batch, device=gpu)
    h = synth.philox_pnpl(batch, n_p, n_l, sigma=sigma, seed=1234 + batch)
    for k in ("pts_3d", "line_3d", "R_gt", "t_gt"):
        assert np.abs(d[k].cpu().numpy() - h[k]).max() < 1e-14 if h[k].size else True, k
    for k in ("pts_2d", "line_2d"):  # pixels ~ 1e2, noise through log / sin / cos
        assert np.abs(d[k].cpu().numpy() - h[k]).max() < 1e-9 if h[k].size else True, k
    if n_p + n_l >= 6 and sigma == 0.0:
        res = ca.pnpl_batch(d["pts_2d"] if n_p else None, d["line_2d"] if n_l else None, d["pts_3d"] if n_p else None,
                            d["line_3d"] if n_l else None, d["K"])
        assert (res.status.cpu().numpy() == 0).all()
        assert synth.geodesic(res.R.cpu().numpy(), d["R_gt"].cpu().numpy()).max() < 1e-6


@pytest.mark.gpu
def test_device_metrics_and_disambiguation_match_numpy(gpu):
    """cvxpnpl_pose_errors / cvxpnpl_disambiguate == cvxpnpl_amd.metrics (numpy): errors to 1e-9, identical choices;
    NaN poses, missing candidates and reflections handled like the reference's harness."""
    import torch

    from cvxpnpl_amd import metrics, synth

    rs = np.random.RandomState(5)
    B = 700
    Rg, tg = synth.random_poses(rs, B)
    dR, _ = synth.random_poses(rs, B)
    small = rs.uniform(0, 1, B) < 0.7
    w = rs.normal(size=(B, 3)) * 1e-3
    for i in np.where(small)[0]:  # mostly small errors, like a benchmark
        th = np.linalg.norm(w[i])
        k = w[i] / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        dR[i] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    R = Rg @ dR + rs.normal(size=(B, 3, 3)) * 1e-6  # not exactly orthogonal: the SVD projection matters
    t = tg * (1 + rs.normal(size=(B, 1)) * 0.01)
    R[3] = np.nan
    R[7] = -R[7]  # a reflection (cvxpnpl.py:510-511 can return one)
    a0, e0 = metrics.pose_errors(Rg, tg, R, t)
    a1, e1 = metrics.pose_errors_device(torch.as_tensor(Rg, device=gpu), tg, torch.as_tensor(R, device=gpu), t)
    a1, e1 = a1.cpu().numpy(), e1.cpu().numpy()
    assert np.isnan(a1[3]) and np.isnan(a0[3])
    ok = ~np.isnan(a0)
    # arccos near 0 amplifies rounding: 1e-9 rad differences in R appear as ~1e-5 deg for errors ~1e-4 deg; compare in cos
    assert np.abs(np.cos(np.radians(a0[ok])) - np.cos(np.radians(a1[ok]))).max() < 1e-12
    big = ok & (a0 > 1e-3)
    assert np.abs(a0[big] - a1[big]).max() < 1e-7 * np.maximum(1.0, a0[big]).max()
    assert np.abs(e0 - e1).max() < 1e-13
    # disambiguation: candidates = the true pose hidden among distractors
    K = synth.K_KINECT
    R_all = np.zeros((B, 4, 3, 3))
    t_all = np.zeros((B, 4, 3))
    n_poses = rs.choice([0, 1, 2, 4, -1], B).astype(np.int32)
    where = rs.randint(0, 4, B)
    for c in range(4):
        Rc, tc = synth.random_poses(rs, B)
        R_all[:, c], t_all[:, c] = Rc, tc
    for i in range(B):
        if n_poses[i] > 0:
            j = where[i] % n_poses[i]
            R_all[i, j], t_all[i, j] = Rg[i], tg[i]
    R_all[11, 0] = np.nan
    R0, t0, i0 = metrics.disambiguate(R_all, t_all, n_poses, K, Rg, tg)
    R1, t1, i1 = metrics.disambiguate_device(R_all, t_all, n_poses, K, torch.as_tensor(Rg, device=gpu), tg)
    assert np.array_equal(i0, i1.cpu().numpy())
    assert np.array_equal(R0, R1.cpu().numpy(), equal_nan=True) and np.array_equal(t0, t1.cpu().numpy(), equal_nan=True)
    picked = n_poses > 0
    picked[11] = picked[11] and (where[11] % max(n_poses[11], 1)) != 0
    assert (i0[picked] == (where % np.maximum(n_poses, 1))[picked]).all()


def test_minimal_set_sampler_restatement_is_a_uniform_draw_without_replacement():
    """cvxpnpl_amd.synth.philox_minimal_sets (the numpy twin of cvxpnpl_sample_minimal_sets): k distinct indices per hypothesis, every
    correspondence equally likely in every position, reproducible, different seeds differ"""
    from cvxpnpl_amd import synth

    idx = synth.philox_minimal_sets(40000, 25, 4, seed=11)
    assert idx.min() == 0 and idx.max() == 24
    s = np.sort(idx, axis=1)
    assert (s[:, 1:] != s[:, :-1]).all()
    for j in range(4):  # 1 600 expected per cell, sigma 39: 6 sigma
        cnt = np.bincount(idx[:, j], minlength=25)
        assert np.abs(cnt - 1600).max() < 240, cnt
    # pairs: every unordered pair of the 300 equally likely (6 per draw x 40 000 / 300 = 800 expected)
    pc = np.zeros((25, 25), int)
    for a in range(4):
        for b in range(a + 1, 4):
            np.add.at(pc, (np.minimum(idx[:, a], idx[:, b]), np.maximum(idx[:, a], idx[:, b])), 1)
    pu = pc[np.triu_indices(25, 1)]
    assert np.abs(pu - 800).max() < 170, (pu.min(), pu.max())
    assert np.array_equal(idx, synth.philox_minimal_sets(40000, 25, 4, seed=11)) and not np.array_equal(idx[:100], synth.philox_minimal_sets(100, 25, 4, seed=12))
    assert np.array_equal(np.sort(synth.philox_minimal_sets(50, 8, 8, seed=3), axis=1), np.tile(np.arange(8), (50, 1)))  # k = n_corr: a permutation


@pytest.mark.gpu
def test_device_minimal_set_sampler_matches_numpy_restatement(gpu):
    """cvxpnpl_sample_minimal_sets: the same indices as the numpy twin, the gathered correspondences are the scene's"""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    rs = np.random.RandomState(5)
    for n_corr, k, H, seed in ((100, 4, 50000, 7), (9, 6, 1000, 2**40 + 3), (8, 8, 257, 0), (5, 1, 64, 9)):
        x, X = rs.random_sample((n_corr, 2)), rs.random_sample((n_corr, 3))
        p2, p3, idx = ca.sample_minimal_sets(torch.as_tensor(x, device=gpu), torch.as_tensor(X, device=gpu), H, k, seed, want_idx=True)
        idx = idx.cpu().numpy()
        assert np.array_equal(idx, synth.philox_minimal_sets(H, n_corr, k, seed))
        assert np.array_equal(p2.cpu().numpy(), x[idx]) and np.array_equal(p3.cpu().numpy(), X[idx])
    with pytest.raises(ValueError):
        ca.sample_minimal_sets(torch.zeros((3, 2), device=gpu), torch.zeros((3, 3), device=gpu), 10, 4)


@pytest.mark.gpu
def test_subset_assembly_equals_the_assembly_of_the_gathered_subset():
    """cvxpnpl_assemble_subsets (the device-side refit of a RANSAC consensus set): B, Q of scene + mask against cvxpnpl_assemble_batch on the
    gathered correspondences -- same cost (1e-12 relative; the centres of the Gram sums differ) and the same solved pose; counts; a subset of
    two correspondences gives NaN."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_ransac(8, n_corr=60, outlier_frac=0.3, sigma=0.5, seed=11)
    x, X, K = (torch.as_tensor(d[k], device="cuda") for k in ("scene_2d", "scene_3d", "K"))
    rs = np.random.RandomState(5)
    mask = (rs.rand(6, 60) < 0.5).astype(np.uint8)
    mask[4] = 0; mask[4, [3, 17]] = 1            # two correspondences: singular
    mask[5] = 1                                  # the whole scene
    Bt, Qt, cnt = ca.assemble_subsets(x, X, K, torch.as_tensor(mask, device="cuda"))
    assert (cnt.cpu().numpy() == mask.sum(axis=1)).all()
    assert torch.isnan(Bt[4]).all() and torch.isnan(Qt[4]).all()
    for b in (0, 1, 2, 3, 5):
        sel = np.flatnonzero(mask[b])
        Bg, Qg = ca.assemble_batch(x[sel][None], None, X[sel][None], None, K)
        scale = float(Qg.abs().max())
        assert float((Qt[b] - Qg[0]).abs().max()) <= 1e-11 * scale
        assert float((Bt[b] - Bg[0]).abs().max()) <= 1e-9 * max(1.0, float(Bg.abs().max()))
    fit = ca.solve_cost_batch(Qt[[0, 1, 2, 3, 5]], Bt[[0, 1, 2, 3, 5]])
    ref = [ca.pnp_batch(x[np.flatnonzero(mask[b])][None], X[np.flatnonzero(mask[b])][None], K) for b in (0, 1, 2, 3, 5)]
    for i, r in enumerate(ref):
        assert int(fit.status[i]) == int(r.status[0])
        assert synth.geodesic(fit.R[i].cpu().numpy(), r.R[0].cpu().numpy()) < 1e-8
        assert float((fit.t[i] - r.t[0]).abs().max()) < 1e-8
    L = __import__("cvxpnpl_amd._lib", fromlist=["lib"]).lib()
    assert L.cvxpnpl_assemble_subsets(0, 60, None, None, None, None, None, None, None, None) == 0
    assert L.cvxpnpl_assemble_subsets(4, 0, None, None, None, None, None, None, None, None) == -1


@pytest.mark.gpu
def test_subset_assembly_ignores_what_the_mask_leaves_out():
    """advisor (round 5): the Gram sums' centre is taken from the SUBSET's own first correspondences -- a non-finite or far-away outlier among
    the scene's first three points no longer spoils the subsets that mask it out"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    d = synth.make_ransac(1, n_corr=40, outlier_frac=0.0, sigma=0.3, seed=4)
    x, X = d["scene_2d"].copy(), d["scene_3d"].copy()
    mask = np.ones((2, 40), np.uint8)
    mask[:, 1] = 0                                  # correspondence 1 is masked out of both subsets ...
    clean = ca.assemble_subsets(torch.as_tensor(x, device="cuda"), torch.as_tensor(X, device="cuda"), torch.as_tensor(d["K"], device="cuda"),
                                torch.as_tensor(mask, device="cuda"))
    X[1] = [np.nan, 1e9, -1e9]                      # ... and is rubbish
    x[1] = [np.inf, 0.0]
    dirty = ca.assemble_subsets(torch.as_tensor(x, device="cuda"), torch.as_tensor(X, device="cuda"), torch.as_tensor(d["K"], device="cuda"),
                                torch.as_tensor(mask, device="cuda"))
    assert torch.equal(clean[0], dirty[0]) and torch.equal(clean[1], dirty[1]) and torch.isfinite(dirty[1]).all()


@pytest.mark.gpu
def test_frame_selection_kernels_against_torch():
    """cvxpnpl_select_best / cvxpnpl_refit_update (round 6) against the torch ops they replace: arg-max with the lowest index on a tie, the
    winner's pose / status / mask / count, the number of certified hypotheses; the refit is taken -- pose, mask and count together -- exactly
    when it is usable and keeps the consensus."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    dev = torch.device("cuda:0")
    d = synth.make_ransac(6000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
    x, X, K = (torch.as_tensor(d[k], device=dev) for k in ("scene_2d", "scene_3d", "K"))
    res = ca.pnp_batch(torch.as_tensor(d["pts_2d"], device=dev), torch.as_tensor(d["pts_3d"], device=dev), K, eps=1e-6, max_iters=100)
    score = ca.score_hypotheses(res.R, res.t, K, x, X, 2.0, status=res.status, usable=(0, 2))
    for trial in range(3):
        sc = score.clone()
        if trial == 1:      # a tie between the best hypotheses: the lowest index wins
            top = int(sc.max())
            idx = torch.nonzero(sc >= top - 1).flatten()
            sc[idx] = top
        if trial == 2:      # the best hypothesis is the last one
            sc[-1] = int(sc.max()) + 1
        R, t, head, mask = ca.select_best(sc, res.R, res.t, res.status, K, x, X, 2.0)
        b = int(torch.argmax(sc))
        if trial == 1:
            b = int(torch.nonzero(sc == sc.max()).flatten()[0])
        h = head.cpu().tolist()
        cnt1, mask1 = ca.score_hypotheses(res.R[b:b + 1], res.t[b:b + 1], K, x, X, 2.0, want_mask=True)
        assert h[2] == b and h[0] == int(res.status[b]) and h[1] == int(cnt1[0]) and h[3] == int((res.status == 0).sum())
        assert torch.equal(R[0], res.R[b]) and torch.equal(t[0], res.t[b]) and torch.equal(mask, mask1)
    # refit: taken when usable and no smaller
    R, t, head, mask = ca.select_best(score, res.R, res.t, res.status, K, x, X, 2.0)
    Bt, Qt, cnt = ca.assemble_subsets(x, X, K, mask)
    fit = ca.solve_cost_batch(Qt, Bt)
    n_new, mask_new = ca.score_hypotheses(fit.R, fit.t, K, x, X, 2.0, want_mask=True)
    before = (R.clone(), t.clone(), head.clone(), mask.clone())
    ca.refit_update(fit, cnt, K, x, X, 2.0, R, t, head, mask)
    take = int(fit.status[0]) in (0, 2) and int(cnt[0]) >= 4 and int(n_new[0]) >= int(before[2][1])
    assert take, "the refit of the consensus set of this scene is expected to be taken"
    assert torch.equal(R, fit.R) and torch.equal(t, fit.t) and torch.equal(mask, mask_new) and head.cpu().tolist()[:2] == [int(fit.status[0]), int(n_new[0])]
    # ... and refused when it would lose inliers (a pose fitted to a corrupted set)
    bad = ca.solve_cost_batch(Qt.roll(1, 1), Bt)
    n_bad = ca.score_hypotheses(bad.R, bad.t, K, x, X, 2.0)
    keep = (R.clone(), t.clone(), head.clone(), mask.clone())
    ca.refit_update(bad, cnt, K, x, X, 2.0, R, t, head, mask)
    if not (int(bad.status[0]) in (0, 2) and int(n_bad[0]) >= int(keep[2][1])):
        assert torch.equal(R, keep[0]) and torch.equal(t, keep[1]) and torch.equal(head, keep[2]) and torch.equal(mask, keep[3])
