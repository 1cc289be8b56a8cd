"""Problems with many correspondences (the reference's scalability regime, benchmarks/scalability/pnp.py:37-40:
n = 200 .. 10 000 points per problem): the blocked assembly kernel (cvxpnpl_assemble_large_batch) and the solve
behind it, against the CPU oracle -- which builds the explicit C, N, A the way the reference does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    from cvxpnpl_amd import _lib

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


def _oracle_QB(orc, d, i, n_p, n_l):
    Cs, Ns = [], []
    K = d["K"] if d["K"].ndim == 2 else d["K"][i]
    if n_p:
        (c1, c2, c3), (n1, n2, n3) = orc.point_constraints(d["pts_2d"][i], d["pts_3d"][i], K)
        Cs += [c1, c2, c3]
        Ns += [n1, n2, n3]
    if n_l:
        cl, nl = orc.line_constraints(d["line_2d"][i], d["line_3d"][i], K)
        Cs.append(cl)
        Ns.append(nl)
    B, A = orc.eliminate(np.vstack(Cs), np.vstack(Ns))
    return A.T @ A, B


@pytest.mark.parametrize("n_p,n_l,batch", [(200, 0, 33), (2000, 0, 9), (10000, 0, 5), (0, 150, 7), (301, 77, 6), (1, 0, 1)])
def test_blocked_assembly_matches_reference_matrices(gpu, orc, n_p, n_l, batch):
    """B and Q = A^T A of cvxpnpl.py:545-549 / :475 from the blocked kernel vs the oracle's explicit matrices, and vs
    the one-lane-per-problem assembly kernel; bit-identical between two runs (fixed summation order)."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth
    from cvxpnpl_amd.api import pack_cost

    d = synth.make_pnpl(batch, n_p, n_l, 1.0, seed=n_p + 3 * n_l)
    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    args = (tt(d["pts_2d"]) if n_p else None, tt(d["line_2d"]) if n_l else None, tt(d["pts_3d"]) if n_p else None,
            tt(d["line_3d"]) if n_l else None, tt(d["K"]))
    Bt, Qt = ca.assemble_batch(*args, blocked=True)
    B2, Q2 = ca.assemble_batch(*args, blocked=True)
    Bn, Qn = Bt.cpu().numpy(), Qt.cpu().numpy()
    assert np.array_equal(Bn, B2.cpu().numpy(), equal_nan=True) and np.array_equal(Qn, Q2.cpu().numpy(), equal_nan=True)
    if n_p + n_l == 1:  # one correspondence: N^T N singular, NaN like the small-problem path (reference: LinAlgError)
        assert np.isnan(Bn).all()
        return
    Bs, Qs = ca.assemble_batch(*args, blocked=False)
    scaleQ = np.abs(Qn).max(axis=1, keepdims=True)
    assert np.abs(Qn - Qs.cpu().numpy()).max() < 1e-11 * scaleQ.max() and np.abs(Bn - Bs.cpu().numpy()).max() < 1e-10
    for i in range(min(batch, 4)):
        Q, B = _oracle_QB(orc, d, i, n_p, n_l)
        assert np.abs(Qn[i] - pack_cost(Q)).max() < 1e-11 * np.abs(Q).max(), (i, np.abs(Qn[i] - pack_cost(Q)).max())
        assert np.abs(Bn[i].reshape(3, 9) - B).max() < 1e-10 * max(1.0, np.abs(B).max())


@pytest.mark.parametrize("n_p", [63, 64, 65, 128, 1000, 4099, 20000])
def test_ring_and_direct_loads_agree(gpu, n_p):
    """assemble_large_kernel streams full tiles of 64 records through its LDS ring when a problem's arrays start on a 16-byte boundary
    and loads directly otherwise (and for the partial last tile): same lane, same order -- bit-identical sums.  The same batch is
    assembled from aligned tensors and from views that start 8 bytes into their storage."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth

    batch = 6
    d = synth.make_pnpl(batch, n_p, 0, 1.0, seed=5 + n_p)
    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731

    def off8(x):  # the same values, storage shifted by one double
        t = tt(x)
        buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=gpu)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 8
        return v

    K = tt(d["K"])
    Ba, Qa = ca.assemble_batch(tt(d["pts_2d"]), None, tt(d["pts_3d"]), None, K, blocked=True)
    Bb, Qb = ca.assemble_batch(off8(d["pts_2d"]), None, off8(d["pts_3d"]), None, K, blocked=True)
    assert torch.equal(Ba, Bb) and torch.equal(Qa, Qb)
    Bs, Qs = ca.assemble_batch(tt(d["pts_2d"]), None, tt(d["pts_3d"]), None, K, blocked=False)
    assert (Qa - Qs).abs().max().item() < 1e-11 * Qa.abs().max().item() and (Ba - Bs).abs().max().item() < 1e-10


@pytest.mark.parametrize("n", [200, 2000, 10000])
def test_large_n_poses_vs_oracle(gpu, orc, n):
    """pnp_batch at n = 200, 2 000, 10 000 points per problem (from 768 points on routed through the blocked assembly): poses within 1e-6
    rad / 1e-6 relative translation of the oracle's converged solve of the reference's explicit system."""
    from cvxpnpl_amd import synth
    import cvxpnpl_amd as ca

    batch = 6
    d = synth.make_pnp(batch, n, 2.0, seed=n)
    res = ca.pnp_batch(d["pts_2d"], d["pts_3d"], d["K"])
    st = res.status.cpu().numpy()
    assert (st == 0).all(), st
    o = orc.pnpl_batch(d["pts_2d"], None, d["pts_3d"], None, d["K"], eps=1e-11, max_iters=200000)
    assert (o["n_poses"] == 1).all()
    geo = synth.geodesic(res.R.cpu().numpy(), o["R"][:, 0])
    terr = np.linalg.norm(res.t.cpu().numpy() - o["t"][:, 0], axis=1) / np.linalg.norm(o["t"][:, 0], axis=1)
    assert geo.max() < 1e-6 and terr.max() < 1e-6, (geo.max(), terr.max())
    # statistical sanity: 2 px noise averaged over n points
    assert synth.geodesic(res.R.cpu().numpy(), d["R_gt"]).max() < 0.05 / np.sqrt(n / 200)


def test_blocked_assembly_far_origin_conditioning(gpu, orc):
    """World origin 1e3 scene sizes away (advisor finding): the sums about the problem's first point keep the digits the
    plain Gram difference C^T C - (N^T C)^T B loses; Q is compared with the reference's A^T A (A = C - N B, formed
    explicitly by the oracle) relative to |Q|."""
    import torch

    import cvxpnpl_amd as ca
    from cvxpnpl_amd import synth
    from cvxpnpl_amd.api import pack_cost

    d = synth.make_pnp(8, 400, 0.5, seed=5)
    c = np.random.RandomState(1).normal(size=(8, 1, 3)) * 600.0
    d["pts_3d"] = d["pts_3d"] + c
    d["t_gt"] = d["t_gt"] - np.einsum("bij,bj->bi", d["R_gt"], c[:, 0])
    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    Bb, Qb = ca.assemble_batch(tt(d["pts_2d"]), None, tt(d["pts_3d"]), None, tt(d["K"]), blocked=True)
    Bs, Qs = ca.assemble_batch(tt(d["pts_2d"]), None, tt(d["pts_3d"]), None, tt(d["K"]), blocked=False)
    eb = es = 0.0
    for i in range(8):
        Q, B = _oracle_QB(orc, d, i, 400, 0)
        eb = max(eb, np.abs(Qb[i].cpu().numpy() - pack_cost(Q)).max() / np.abs(Q).max())
        es = max(es, np.abs(Qs[i].cpu().numpy() - pack_cost(Q)).max() / np.abs(Q).max())
        assert np.abs(Bb[i].cpu().numpy().reshape(3, 9) - B).max() < 1e-7 * np.abs(B).max()
    assert eb < 1e-9, eb  # shifted sums: conditioning of the scene, not of the origin
    assert eb < es or es < 1e-12, (eb, es)
    res = ca.pnp_batch(d["pts_2d"], d["pts_3d"], d["K"])
    assert (res.status.cpu().numpy() == 0).all()
    assert synth.geodesic(res.R.cpu().numpy(), d["R_gt"]).max() < 5e-3
