"""opts.f32_sweeps_until: the ONE place the float64 path uses single precision -- the Jacobi sweeps of the PSD projection
during the first iterations of a solve -- can be switched off (0 = every sweep, rotation angles included, in float64; the
reference is float64 throughout, cvxpnpl.py:475-513).  These tests make the claim "the single-precision sweeps change no
result" checkable from outside: both modes through the C ABI on the same inputs, all four layouts.

Bounds (stated, asserted):
  * statuses identical;
  * certified poses (status 0) equal to 2e-11 rad / 2e-11 relative translation -- a certified pose is the Newton-polished
    stationary point of r^T Q r on SO(3), found in float64 in both modes; the iterate is only its starting point (measured over
    the 32 case x layout combinations: <= 1e-12 on 31, 5.1e-12 on noise-free PnPL 5+5 -- the polish takes its last Newton step
    from |gradient| < 1e-8, which leaves the pose at 1e-10 ... 1e-16 depending on the Hessian; the same spread is seen
    between two layouts in the same mode);
  * uncertified exits (forced by max_iters = 2...12): the returned Z of the two modes within 5e-4 in Frobenius norm (|Z| = 4;
    measured worst 2.3e-4, four-point problems cut at 12 iterations: the eigen-solve stops at a column cosine of 6e-2, i.e. is
    accurate to ~4e-3 by design, and whether one more sweep runs is a discrete decision that single-precision noise can flip),
    same rank decision except where an eigenvalue sits within that distance of the 1e-3 threshold of cvxpnpl.py:502; what has
    certified by the cap has certified in both modes, except for at most two of 96 minimal problems whose attempt sits at the
    acceptance threshold.
  The certified-pose bound is 1e-9 for minimal (four-correspondence) problems: measured 4.6e-11.  The |dZ| bound is for problems
  with at least five correspondences; the young iterates of minimal, non-tight problems drift apart under any rounding difference
  (bulk statements instead: see the test)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import CASES, _solve, geodesic_np, gpu  # noqa: E402,F401
from test_gpu_parity import LAYOUTS as _ALL_LAYOUTS  # noqa: E402

# wave, quad, lane: the layouts that exist in both precisions.  (penta, the 12-lane experiment, has no float64 instantiation: asked for
# float64 sweeps it runs the 16-lane geometry, so an A/B there would compare two layouts, not two precisions.)
LAYOUTS = {k: v for k, v in _ALL_LAYOUTS.items() if k != "penta"}


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("n_p,n_l,sigma,batch", CASES)
def test_f32_and_f64_sweeps_agree(gpu, n_p, n_l, sigma, batch, layout):  # noqa: F811
    from cvxpnpl_amd import synth

    d = synth.make_pnpl(batch, n_p, n_l, sigma, seed=200 + n_p + 7 * n_l)
    a = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout], want_Z=True)                      # default: single-precision sweeps while young
    b = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout], want_Z=True, f32_sweeps_until=0)  # float64 throughout
    # minimal problems run long: an iteration count may differ by a certificate attempt between the modes, the outcome may not
    assert (a["status"] == b["status"]).all(), np.flatnonzero(a["status"] != b["status"])
    cert = a["status"] == 0
    assert cert.sum() >= batch - 2
    worst_r = max(geodesic_np(a["R"][i], b["R"][i]) for i in np.flatnonzero(cert))
    tn = np.maximum(1.0, np.linalg.norm(b["t"][cert], axis=1))
    worst_t = (np.linalg.norm(a["t"][cert] - b["t"][cert], axis=1) / tn).max()
    bound = 2e-11 if n_p + n_l >= 5 else 1e-9  # (minimal problems: a flat cost around the optimum, the polish leaves more)
    assert worst_r <= bound and worst_t <= bound, (worst_r, worst_t)
    # the certificate itself is a float64 statement in both modes
    for r in (a, b):
        gap = r["cost"][cert, 0] - r["cost"][cert, 1]
        assert (gap >= -1e-15).all() and (gap <= 1.0001e-9 + 1e-12 * np.abs(r["cost"][cert, 0])).all()
    if n_p + n_l >= 8:  # well-posed problems take the same number of iterations in both modes (an attempt at the acceptance threshold
        # may flip -- noise-free data sits there: cost 0 against the 8e-13 tr floor of the gap -- and then costs one more attempt, two iterations)
        assert (a["iters"] == b["iters"]).mean() >= 0.9, (a["iters"] != b["iters"]).sum()


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("max_iters", [2, 3, 5, 8, 12])
def test_uncertified_exits_of_both_modes(gpu, layout, max_iters):  # noqa: F811
    """solves cut short: the pose comes from the iterate, whose eigenvectors were single precision in one mode"""
    from cvxpnpl_amd import synth

    worst = 0.0
    for n_p, n_l, sigma in [(10, 0, 2.0), (5, 5, 1.0), (4, 0, 1.0)]:
        d = synth.make_pnpl(96, n_p, n_l, sigma, seed=31 + n_p + max_iters)
        a = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout], want_Z=True, max_iters=max_iters)
        b = _solve(gpu, d, n_p, n_l, layout=LAYOUTS[layout], want_Z=True, max_iters=max_iters, f32_sweeps_until=0)
        # certified: a pose (status 0) or an exactly two-fold ambiguous pair (status 1 with its certified lower bound in cost[:, 1])
        ca_ = (a["status"] == 0) | ((a["status"] == 1) & np.isfinite(a["cost"][:, 1]))
        cb_ = (b["status"] == 0) | ((b["status"] == 1) & np.isfinite(b["cost"][:, 1]))
        open_ = ~ca_ & ~cb_
        # what certifies by the cap, certifies in both modes -- up to attempts that sit at the acceptance threshold (minimal problems)
        assert (ca_ != cb_).sum() <= (2 if n_p + n_l < 6 else 0), (n_p, n_l, np.flatnonzero(ca_ != cb_))
        if not open_.any():
            continue
        dz = np.linalg.norm((a["Z"][open_] - b["Z"][open_]) * _VECH_W, axis=1)
        if n_p + n_l < 5:
            # Minimal problems are often not tight (rank 3-4 at the optimum) and their young iterates do not contract yet: two runs that
            # differ by rounding drift apart (measured: |dZ| 0.21 on one of 96 after 12 iterations, eigenvalues (.88, 1.10, 1.27) against
            # (.88, 1.05, 1.19)).  So for them: the bulk of the iterates agree, and the poses read off them (rounded from the iterate, not
            # polished to a certificate: they follow it -- measured 1e-16 ... 4e-4 rad) stay within 1e-3 rad on nine of ten.
            assert np.percentile(dz, 75) <= 5e-3, np.percentile(dz, [50, 75, 90, 100])
            fin = open_ & np.isfinite(a["R"]).all(axis=(1, 2)) & np.isfinite(b["R"]).all(axis=(1, 2))
            g = np.array([geodesic_np(a["R"][i], b["R"][i]) for i in np.flatnonzero(fin)])
            assert (g < 1e-3).mean() >= 0.9, (g > 1e-3).sum()
            continue
        worst = max(worst, float(dz.max()))
        same = a["status"][open_] == b["status"][open_]
        if not same.all():  # only where an eigenvalue of Z sits at the rank threshold (cvxpnpl.py:502)
            for i in np.flatnonzero(open_)[~same]:
                lam = np.linalg.eigvalsh(_unvech(b["Z"][i]))
                assert np.abs(lam - 1e-3).min() < 2e-4, (i, lam)
    assert worst <= 5e-4, worst


def test_f64_mode_against_the_oracle(gpu, orc):  # noqa: F811
    """the float64 mode is a full citizen: same oracle parity as the default (<= 1e-6 rad, north-star tolerance)"""
    from cvxpnpl_amd import synth

    d = synth.make_pnp(64, 10, sigma=2.0, seed=77)
    o = orc.pnpl_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"], eps=1e-11, max_iters=200000)
    for layout in sorted(LAYOUTS):
        r = _solve(gpu, d, 10, 0, layout=LAYOUTS[layout], f32_sweeps_until=0)
        ok = (r["status"] == 0) & (o["n_poses"] == 1)
        assert ok.sum() >= 63
        g = synth.geodesic(r["R"][ok], o["R"][ok, 0])
        assert g.max() < 1e-6 and np.abs(r["t"][ok] - o["t"][ok, 0]).max() < 1e-6


def test_penta_request_with_float64_sweeps_runs_the_quad_geometry(gpu):  # noqa: F811
    """Round-3 advisor finding: CVXPNPL_LAYOUT_PENTA with f32_sweeps_until = 0 (the twelve-lane geometry has no float64 instantiation)
    used to fall through to the LANE branch with a workspace fetched for another stride -- the rescue queue's pointers could dangle and
    problems past rescue_from never finish.  The request now runs the sixteen-lane quad kernel: every problem comes back, and with
    exactly what a QUAD request returns.  Minimal problems, so that the resume and the interior-point queues are really used; two
    launch sizes on one stream, the second larger, so that the workspace has to grow between them."""
    import torch

    from cvxpnpl_amd import synth

    torch.cuda.synchronize()
    for batch in (300, 3000):
        d = synth.make_pnpl(batch, 4, 0, 1.0, seed=5 + batch)
        a = _solve(gpu, d, 4, 0, layout=_ALL_LAYOUTS["penta"], f32_sweeps_until=0, want_Z=True)
        b = _solve(gpu, d, 4, 0, layout=_ALL_LAYOUTS["quad"], f32_sweeps_until=0, want_Z=True)
        assert ((a["status"] >= 0) & (a["status"] <= 4)).all() and (a["iters"] >= 1).all()
        assert (a["iters"] > 32).sum() >= 10  # (some of them did go through the queues)
        assert (a["status"] == b["status"]).all() and (a["iters"] == b["iters"]).all()
        assert np.array_equal(np.nan_to_num(a["R"]), np.nan_to_num(b["R"])) and np.array_equal(np.nan_to_num(a["t"]), np.nan_to_num(b["t"]))


def test_a_lane_request_the_lane_kernel_cannot_serve_runs_the_next_best_schedule(gpu):  # noqa: F811
    """Round-5 advisor finding: a LANE request with options the register-budgeted lane kernel does not cover (lane_iters != first_check,
    warm_start = 0) used to run the wave layout -- one problem per wavefront also at 30 000 problems -- with the attempt schedule of the
    lane layout, and nothing said so.  The layout is now settled before anything is derived from it: quad from 2 560 problems, wave below,
    and cvxpnpl_last_layout() reports what ran."""
    from cvxpnpl_amd import _lib, synth

    L = _lib.lib()
    d = synth.make_pnpl(30000, 10, 0, 2.0, seed=9)
    r = _solve(gpu, d, 10, 0)
    assert L.cvxpnpl_last_layout() == 1 and r["iters"].min() == 6                    # AUTO at this size: the lane-hybrid schedule
    q = _solve(gpu, d, 10, 0, layout=3)
    assert L.cvxpnpl_last_layout() == 3 and q["iters"].min() == 5
    for kw in (dict(lane_iters=4), dict(warm_start=0)):
        a = _solve(gpu, d, 10, 0, layout=1, **kw)
        assert L.cvxpnpl_last_layout() == 3, kw                                      # ... served by the quad schedule, with ITS first attempt
        assert a["iters"].min() == 5 and (a["status"] == 0).all()
    small = {k: (v[:1000] if isinstance(v, np.ndarray) and v.ndim > 2 else v) for k, v in d.items()}
    _solve(gpu, small, 10, 0, layout=1, lane_iters=4)
    assert L.cvxpnpl_last_layout() == 2
    _solve(gpu, small, 10, 0, layout=4, f32_sweeps_until=0)
    assert L.cvxpnpl_last_layout() == 3                                              # (a PENTA request with float64 sweeps: the sixteen-lane kernel)


def test_f32_sweeps_until_is_validated(gpu):  # noqa: F811
    from cvxpnpl_amd import synth

    d = synth.make_pnp(8, 10, sigma=1.0, seed=1)
    with pytest.raises(RuntimeError, match="bad options"):
        _solve(gpu, d, 10, 0, f32_sweeps_until=-2)
    with pytest.raises(RuntimeError, match="bad options"):
        _solve(gpu, d, 10, 0, f32_sweeps_until=65)  # (the documented maximum: 64, the window the accuracy experiments cover)
    assert (_solve(gpu, d, 10, 0, f32_sweeps_until=64)["status"] == 0).all()
    r = _solve(gpu, d, 10, 0, f32_sweeps_until=3)  # in between: phases shorter than the bound stay single, the others float64
    assert (r["status"] == 0).all()


def _unvech(z):
    M = np.zeros((10, 10))
    k = 0
    for i in range(10):
        for j in range(i, 10):
            M[i, j] = M[j, i] = z[k]
            k += 1
    return M


_VECH_W = np.array([1.0 if i == j else np.sqrt(2.0) for i in range(10) for j in range(i, 10)])
