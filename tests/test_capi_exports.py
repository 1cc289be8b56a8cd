"""The C-ABI library loads, exports every symbol include/cvxpnpl_amd.h declares, and its
host-side entry points work -- no GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from cvxpnpl_amd import _lib, build

    build.build()
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from cvxpnpl_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "cvxpnpl_amd.h")).read()
    declared = set(re.findall(r"\b(cvxpnpl_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None


def test_default_opts_match_reference_defaults(L):
    from cvxpnpl_amd import _lib

    o = _lib.default_opts()
    assert o.eps == 1e-9 and o.max_iters == 2500  # cvxpnpl.py:527-528
    with pytest.raises(TypeError):
        _lib.default_opts(nonsense=1)


def test_opts_struct_is_size_guarded(L):
    """cvxpnpl_opts_t carries its own size: a caller built against another revision of the struct (e.g. round 1's,
    shorter) is refused with a launch-level error instead of having fields read from beyond its buffer."""
    from cvxpnpl_amd import _lib

    assert L.cvxpnpl_opts_size() == C.sizeof(_lib.Opts)
    o = _lib.default_opts()
    assert o.struct_size == C.sizeof(_lib.Opts) and o.f32_sweeps_until == -1
    dummy = (C.c_double * 64)()  # never dereferenced: the size check comes before any launch
    p = C.cast(dummy, C.c_void_p)

    def call(opts_ptr):
        return L.cvxpnpl_solve_batch(1, 4, p, p, 0, None, None, p, 0, opts_ptr, p, p, p, None, None, None, None, None)

    class OldOpts(C.Structure):  # the struct as round 2 shipped it: no struct_size, no f32_sweeps_until -- a short buffer
        _fields_ = [f for f in _lib.Opts._fields_ if f[0] not in ("struct_size", "f32_sweeps_until")]

    old = OldOpts()
    C.memmove(C.byref(old), C.byref(o, 8), C.sizeof(old))
    assert call(C.cast(C.pointer(old), C.POINTER(_lib.Opts))) == -1
    assert b"options block" in L.cvxpnpl_last_error() and str(C.sizeof(_lib.Opts)).encode() in L.cvxpnpl_last_error()
    o.struct_size = C.sizeof(_lib.Opts) - 8
    assert call(C.byref(o)) == -1 and b"options block" in L.cvxpnpl_last_error()
    o.struct_size = 0
    assert call(C.byref(o)) == -1
    rc = L.cvxpnpl_solve_cost_batch(1, p, p, C.byref(o), p, p, p, None, None, None, None, None)
    assert rc == -1 and b"options block" in L.cvxpnpl_last_error()


def test_unknown_layouts_are_refused(L):
    """opts.layout outside the public enum (the experiment layouts 9-13 of rounds 2-4 live in experiment builds only) is a launch-level
    error, not a silent fall-through to some kernel (round-4 advisor; argument check only: no launch, no GPU needed)"""
    from cvxpnpl_amd import _lib

    dummy = (C.c_double * 64)()
    p = C.cast(dummy, C.c_void_p)
    for bad in (-1, 5, 9, 10, 11, 12, 13, 99):
        o = _lib.default_opts(layout=bad)
        rc = L.cvxpnpl_solve_batch(1, 4, p, p, 0, None, None, p, 0, C.byref(o), p, p, p, None, None, None, None, None)
        assert rc == -1 and b"layout" in L.cvxpnpl_last_error(), (bad, rc, L.cvxpnpl_last_error())
        rc = L.cvxpnpl_solve_cost_batch(1, p, p, C.byref(o), p, p, p, None, None, None, None, None)
        assert rc == -1 and b"layout" in L.cvxpnpl_last_error()


def test_empty_sampling_call_is_a_no_op(L):
    """n_hyp == 0 returns 0 whatever the (empty, possibly NULL) output pointers are (round-4 advisor: the null check used to come first)"""
    dummy = (C.c_double * 64)()
    p = C.cast(dummy, C.c_void_p)
    assert L.cvxpnpl_sample_minimal_sets(0, 100, p, p, 4, 1, None, None, None, None) == 0
    assert L.cvxpnpl_sample_minimal_sets(-1, 100, p, p, 4, 1, None, p, p, None) == -1


def test_integration_doc_mirrors_the_opts_struct():
    """INTEGRATION.md's ctypes block is generated from _lib.Opts (tools/gen_integration_opts.py): field names, order, types"""
    from cvxpnpl_amd import _lib

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blk = doc[doc.index("<!-- opts:begin -->"):doc.index("<!-- opts:end -->")]
    fields = re.findall(r'\("([a-z_0-9]+)", C\.(c_[a-z0-9]+)\)', blk)
    assert fields == [(n, t.__name__) for n, t in _lib.Opts._fields_], "run python tools/gen_integration_opts.py"
    hdr = open(os.path.join(ROOT, "include", "cvxpnpl_amd.h")).read()
    body = hdr[hdr.index("typedef struct {"):hdr.index("} cvxpnpl_opts_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for typ, decl in re.findall(r"\b(uint32_t|int32_t|double)\s+([a-z_0-9, ]+);", body):
        names += [(d.strip(), typ) for d in decl.split(",")]
    ct = {"uint32_t": "c_uint", "int32_t": "c_int", "double": "c_double"}  # (ctypes aliases c_uint32 -> c_uint, c_int32 -> c_int)
    assert [(n, ct[t]) for n, t in names] == [(n, t.__name__) for n, t in _lib.Opts._fields_]


def test_bad_arguments_are_rejected_without_gpu(L):
    from cvxpnpl_amd import _lib

    o = _lib.default_opts()
    rc = L.cvxpnpl_solve_batch(4, 0, None, None, 0, None, None, None, 0, C.byref(o), None, None, None, None, None, None, None, None)
    assert rc == -1 and b"bad arguments" in L.cvxpnpl_last_error()
    assert b"gfx950" in L.cvxpnpl_version()
    rc = L.cvxpnpl_pack_results(8, None, None, None, None, None)
    assert rc == -1 and b"cvxpnpl_pack_results: bad arguments" in L.cvxpnpl_last_error()
    rc = L.cvxpnpl_score_hypotheses(8, None, None, None, 5, None, 10, None, None, 2.0, None, None, None)
    assert rc == -1 and b"cvxpnpl_score_hypotheses: bad arguments" in L.cvxpnpl_last_error()
    for n_corr, k in ((100, 9), (3, 4), (100, 0)):  # more than 8 per set, fewer correspondences than a set, empty sets
        rc = L.cvxpnpl_sample_minimal_sets(16, n_corr, C.c_void_p(8), C.c_void_p(8), k, 1, None, C.c_void_p(8), C.c_void_p(8), None)
        assert rc == -1 and b"cvxpnpl_sample_minimal_sets: bad arguments" in L.cvxpnpl_last_error()


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cvxpnpl_amd

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cvxpnpl_amd.pnp(np.zeros((6, 2)), np.zeros((6, 3)), np.eye(3))


def test_product_does_not_import_oracle():
    """The shipped package must not reach into oracle/ or the test-only host build."""
    pkg = os.path.join(ROOT, "cvxpnpl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "import hostsim" not in src, f
                assert "oracle.h" not in src and "liboracle" not in src and "libhostsim" not in src, f


def test_recover_multi_host_path_matches_reference(L, golden):
    """cvxpnpl_recover_multi (host C++) against the reference's rank-2 / rank-4 outputs."""
    from cvxpnpl_amd.api import recover_multi

    B = golden["g3_pnp_B"]
    for tag, n, tol in (("r1", 1, 1e-10), ("r2", 2, 1e-7), ("r4", 4, 1e-6)):
        poses = recover_multi(golden[f"g6_{tag}_x"], B)
        Rg, tg = golden[f"g6_{tag}_R"], golden[f"g6_{tag}_t"]
        assert len(poses) == n == len(Rg)
        used = set()
        for R, t in poses:
            d = [np.abs(R - Rg[k]).max() + np.abs(t - tg[k]).max() if k not in used else np.inf for k in range(n)]
            k = int(np.argmin(d))
            assert d[k] < tol, (tag, d)
            used.add(k)
    with pytest.raises(NotImplementedError):
        recover_multi(np.full(55, np.nan), B)


def test_planar_scene_two_fold_ambiguity_host_recovery(L):
    """Planar scenes are exactly two-fold ambiguous for the algebraic cost: Z has eigenvalues (2, 2),
    the device algorithm flags rank > 1 and the host recovery returns both poses.  With the robust basis
    choice and the optional polish the true pose is recovered to 1e-12 (the reference's formula divides by
    the last entry of the top eigenvector, which is arbitrary here, and returns NaN for ~1/3 of these)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hostsim
    from cvxpnpl_amd import synth
    from cvxpnpl_amd.api import recover_multi

    d = synth.make_pnp(48, 10, 0.0, seed=1)
    d["pts_3d"][:, :, 2] = 0.0
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"])
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    assert (hs["status"] == 1).all() and (hs["rank"] == 2).all()
    for i in range(48):
        _, B, Q = hostsim.assemble(d["pts_2d"][i], d["pts_3d"][i], None, None, d["K"])
        q45 = np.array([Q[a, b] for a in range(9) for b in range(a, 9)])
        poses = recover_multi(hs["Z"][i], B, q45)
        assert len(poses) == 2
        err = min(synth.geodesic(R, d["R_gt"][i]) + np.linalg.norm(t - d["t_gt"][i]) for R, t in poses)
        assert err < 1e-10, (i, err)
        rough = recover_multi(hs["Z"][i], B)  # unpolished: limited by the first-order solve
        assert min(synth.geodesic(R, d["R_gt"][i]) for R, t in rough) < 1e-2


def test_recover_multi_batch_equals_per_problem_recovery(L):
    """cvxpnpl_recover_multi_batch (host threads over a batch) returns exactly what
    cvxpnpl_recover_multi returns problem by problem, and skips problems whose status is not RANK_GT1."""
    import sys
    import types

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hostsim
    from cvxpnpl_amd import synth
    from cvxpnpl_amd.api import recover_multi, recover_multi_batch

    d = synth.make_pnp(96, 10, 0.0, seed=2)
    d["pts_3d"][::2, :, 2] = 0.0  # every other scene planar (rank 2), the rest rank 1
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"])
    hs = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"], want_Z=True)
    assert (hs["status"][::2] == 1).all() and (hs["status"][1::2] == 0).all()
    Bs, Qs = [], []
    for i in range(96):
        _, B, Q = hostsim.assemble(d["pts_2d"][i], d["pts_3d"][i], None, None, d["K"])
        Bs.append(B.reshape(-1))
        Qs.append(np.array([Q[a, b] for a in range(9) for b in range(a, 9)]))
    res = types.SimpleNamespace(Z=hs["Z"], status=hs["status"])
    for nt in (1, 3, 0):
        R, t, cnt = recover_multi_batch(res, np.array(Bs), np.array(Qs), n_threads=nt)
        assert np.isin(cnt[::2], (2, 4)).all() and (cnt[::2] == 2).mean() > 0.9 and (cnt[1::2] == 0).all()
        for i in range(0, 96, 2):
            poses = recover_multi(hs["Z"][i], Bs[i], Qs[i])
            assert len(poses) == cnt[i]
            for k, (Rk, tk) in enumerate(poses):
                assert np.array_equal(R[i, k], Rk, equal_nan=True) and np.array_equal(t[i, k], tk, equal_nan=True)
            if cnt[i] == 2:
                assert min(synth.geodesic(R[i, k], d["R_gt"][i]) for k in range(2)) < 1e-10
    # status = None: every problem is recovered (rank-1 ones return their single pose)
    import ctypes as C

    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    Z = np.ascontiguousarray(hs["Z"])
    Bn, Qn = np.ascontiguousarray(np.array(Bs)), np.ascontiguousarray(np.array(Qs))
    R, t, cnt = np.zeros((96, 4, 3, 3)), np.zeros((96, 4, 3)), np.zeros(96, dtype=np.int32)
    rc = L.cvxpnpl_recover_multi_batch(96, None, Z.ctypes.data_as(dp), Bn.ctypes.data_as(dp), Qn.ctypes.data_as(dp),
                                       R.ctypes.data_as(dp), t.ctypes.data_as(dp), cnt.ctypes.data_as(ip), 0)
    assert rc == 0 and (cnt[1::2] == 1).all() and np.isin(cnt[::2], (2, 4)).all()
    assert max(synth.geodesic(R[i, 0], d["R_gt"][i]) for i in range(1, 96, 2)) < 1e-9
