#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors_harness.npz (G9) by running the REFERENCE's benchmark-toolkit functions
themselves: benchmarks/toolkit/suites/suite.py (angle :8-14, project_points :17-19, compute_pose_error :22-33, the pose
disambiguation inside Suite.estimate_pose :74-110) and suites/synth.py (aa2rm :12-24, random_pose :27-42,
PnPSynth / PnLSynth.generate_correspondences :277-310).  They pin cvxpnpl_amd/metrics.py, cvxpnpl_amd/synth.py and the
device kernels cvxpnpl_pose_errors / cvxpnpl_disambiguate / cvxpnpl_synth_batch.

Build container only:    python -B tests/golden/make_golden_harness.py [out.npz]

suite.py needs numpy only; synth.py also imports cycler / matplotlib (both in this image) and `.suite`, so the two files
are loaded as the modules `suites.suite` / `suites.synth` of an empty package object (the package's own __init__ pulls in
the real-data suite, which is out of scope).  One alias is restored for the run: `np.float` (removed in numpy 1.24; the
reference's disambiguation loop writes np.float("+inf"), suite.py:99) -- the builtin float, which is what it was.
Only numeric arrays are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
SUITES = "/root/reference/benchmarks/toolkit/suites"

if not hasattr(np, "float"):
    np.float = float  # the alias numpy < 1.24 had (suite.py:99)

os.environ.setdefault("MPLBACKEND", "Agg")
pkg = types.ModuleType("suites")
pkg.__path__ = [SUITES]
sys.modules["suites"] = pkg


def _load(name):
    spec = importlib.util.spec_from_file_location(f"suites.{name}", os.path.join(SUITES, f"{name}.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[f"suites.{name}"] = m
    spec.loader.exec_module(m)
    return m


suite = _load("suite")
synth = _load("synth")

out = {}
rs = np.random.RandomState(20240929)


def rot(rs_):
    q = rs_.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


# ---- angle (suite.py:8-14): rotations, reflections, non-orthogonal matrices, near-identity, near-pi
mats = [rot(rs) for _ in range(40)]
mats += [rot(rs) @ np.diag([1.0, 1.0, -1.0]) for _ in range(20)]               # reflections (what U Vh without det fix returns)
mats += [rot(rs) + 0.05 * rs.normal(size=(3, 3)) for _ in range(20)]            # not orthogonal
mats += [np.eye(3), np.diag([1.0, -1.0, -1.0]), rot(rs) * 3.0]
small = synth.aa2rm(np.array([1e-9, -2e-9, 3e-9]))
mats += [small, synth.aa2rm(np.array([np.pi - 1e-7, 0.0, 0.0]))]
mats = np.array(mats)
out["g9_angle_in"] = mats
out["g9_angle_out"] = np.array([suite.angle(m) for m in mats])
out["g9_angle_batched_out"] = suite.angle(mats)  # the function takes stacks as well

# ---- project_points (suite.py:17-19)
K = np.array([[572.41140, 0, 325.26110], [0, 573.57043, 242.04899], [0, 0, 1]])
P = rs.random_sample((20, 3)) - 0.5
Rp, tp = rot(rs), np.array([0.1, -0.2, 1.3])
out["g9_proj_K"], out["g9_proj_P"], out["g9_proj_R"], out["g9_proj_t"] = K, P, Rp, tp
out["g9_proj_out"] = suite.project_points(P, K, Rp, tp)

# ---- compute_pose_error (suite.py:22-33): 200 pairs -- close estimates, far ones, reflections, non-orthogonal, NaN
Rg, tg, Re, te, ang, tr, raised = [], [], [], [], [], [], []
for i in range(200):
    R_gt = rot(rs)
    t_gt = np.array([rs.random_sample() - 0.5, rs.random_sample() - 0.5, 1.6 * rs.random_sample() + 0.6])
    kind = i % 5
    if kind == 0:
        R = R_gt @ synth.aa2rm(1e-3 * rs.normal(size=3)); t = t_gt + 1e-3 * rs.normal(size=3)
    elif kind == 1:
        R = rot(rs); t = t_gt + 0.3 * rs.normal(size=3)
    elif kind == 2:
        R = R_gt @ np.diag([1.0, -1.0, 1.0]) @ synth.aa2rm(0.1 * rs.normal(size=3)); t = -t_gt
    elif kind == 3:
        R = R_gt + 0.02 * rs.normal(size=(3, 3)); t = t_gt * 1.01
    else:
        R = np.full((3, 3), np.nan) if i % 10 == 4 else R_gt.copy(); t = np.full(3, np.nan) if i % 10 == 4 else t_gt.copy()
    try:
        a, e = suite.compute_pose_error((R_gt, t_gt), (R, t))
        a, e, r = float(a), float(e), 0
    except np.linalg.LinAlgError:  # the reference's SVD on a NaN estimate
        a, e, r = np.nan, np.nan, 1
    Rg.append(R_gt); tg.append(t_gt); Re.append(R); te.append(t); ang.append(a); tr.append(e); raised.append(r)
out["g9_err_Rgt"], out["g9_err_tgt"], out["g9_err_R"], out["g9_err_t"] = map(np.array, (Rg, tg, Re, te))
out["g9_err_ang_deg"], out["g9_err_trans"], out["g9_err_raised"] = np.array(ang), np.array(tr), np.array(raised, dtype=np.int32)

# ---- pose disambiguation (Suite.estimate_pose, suite.py:74-110): a stub method that returns prepared candidate lists.
# The support points come from np.random (suite.py:95): the global stream is seeded right before every call, and the
# same seed reproduces them -- stored as g9_dis_support for the callers that inject them.


class _Stub:
    name = "stub"
    poses = None

    @staticmethod
    def estimate_pose(K, **kw):
        return _Stub.poses


st = suite.Suite(methods=[_Stub], timed=False)
cR, ct, cn, gR, gt_, pick, seeds, supp = [], [], [], [], [], [], [], []
for i in range(64):
    R_gt = rot(rs)
    t_gt = np.array([rs.random_sample() - 0.5, rs.random_sample() - 0.5, 1.6 * rs.random_sample() + 0.6])
    n = (1, 2, 4, 2, 4, 3)[i % 6]
    cands = []
    for k in range(n):
        if k == (i // 6) % n:  # the good one somewhere in the list
            cands.append((R_gt @ synth.aa2rm(0.02 * rs.normal(size=3)), t_gt + 0.01 * rs.normal(size=3)))
        elif k % 2:
            cands.append((R_gt @ np.diag([-1.0, -1.0, 1.0]), t_gt.copy()))          # the planar twin
        else:
            cands.append((rot(rs), t_gt + 0.2 * rs.normal(size=3)))
    if i % 16 == 7 and n > 1:  # a non-finite candidate in front: its error is NaN, `err < min_error` is False, it is never chosen unless first
        cands[1] = (np.full((3, 3), np.nan), np.full(3, np.nan))
    _Stub.poses = cands
    seed = 1000 + i
    np.random.seed(seed)
    (Rc, tc), _ = st.estimate_pose(_Stub, (R_gt, t_gt), K)
    idx = [k for k, (a, b) in enumerate(cands) if a is Rc or (np.array_equal(a, Rc, equal_nan=True) and np.array_equal(b, tc, equal_nan=True))][0]
    Ra = np.full((4, 3, 3), np.nan); ta = np.full((4, 3), np.nan)
    for k, (a, b) in enumerate(cands):
        Ra[k], ta[k] = a, b
    np.random.seed(seed)
    supp.append(np.random.random((20, 3)) - 0.5)  # what suite.py:95 drew
    cR.append(Ra); ct.append(ta); cn.append(n); gR.append(R_gt); gt_.append(t_gt); pick.append(idx); seeds.append(seed)
out["g9_dis_K"] = K
out["g9_dis_R_all"], out["g9_dis_t_all"], out["g9_dis_n"] = np.array(cR), np.array(ct), np.array(cn, dtype=np.int32)
out["g9_dis_Rgt"], out["g9_dis_tgt"] = np.array(gR), np.array(gt_)
out["g9_dis_pick"], out["g9_dis_seed"], out["g9_dis_support"] = np.array(pick, dtype=np.int32), np.array(seeds, dtype=np.int64), np.array(supp)

# ---- aa2rm / random_pose (synth.py:12-42): seeded single draws (the draw order 3 + 1 + 2 + 1 is part of the spec) and moments
aas = np.array([[0.3, -0.2, 0.9], [1e-17, 0.0, 0.0], [0.0, 0.0, np.pi], [2.0, 2.0, -1.0], [1e-8, 1e-8, 0.0]])
out["g9_aa_in"], out["g9_aa_out"] = aas, np.array([synth.aa2rm(a) for a in aas])
rpR, rpt = [], []
for s in range(32):
    np.random.seed(500 + s)
    R, t = synth.random_pose()
    rpR.append(R); rpt.append(t)
out["g9_rp_seed0"], out["g9_rp_R"], out["g9_rp_t"] = np.int64(500), np.array(rpR), np.array(rpt)
np.random.seed(7)
N = 20000
Rs, ts = zip(*[synth.random_pose() for _ in range(N)])
Rs, ts = np.array(Rs), np.array(ts)
ang_all = suite.angle(Rs)
out["g9_rp_moments_n"] = np.int64(N)
out["g9_rp_t_mean"], out["g9_rp_t_std"] = ts.mean(0), ts.std(0)
out["g9_rp_t_min"], out["g9_rp_t_max"] = ts.min(0), ts.max(0)
# rotation angle in [0, pi] of a rotation by 2 pi U about a random axis: angle = min(a, 2 pi - a), mean pi / 2
out["g9_rp_angle_mean"], out["g9_rp_angle_std"] = ang_all.mean(), ang_all.std()
out["g9_rp_trace_mean"] = np.trace(Rs, axis1=1, axis2=2).mean()
out["g9_rp_R_mean"] = Rs.mean(0)  # E[R] = E[cos] I + E[1 - cos] E[k k^T] with the reference's (non-uniform) axis law

# ---- generate_correspondences (synth.py:277-310): seed, random_pose(), then the suite's generator -- the order of
# SynthSuite.run (synth.py:238-246).  PnP and PnL (PnPL draws its split with randint first, :323: not reproduced by a twin).
for tag, cls, n, sigma in (("pnp", synth.PnPSynth, 10, 2.0), ("pnp0", synth.PnPSynth, 6, 0.0), ("pnl", synth.PnLSynth, 5, 1.0)):
    sess = cls(methods=[], n_runs=1, timed=False)
    np.random.seed(900)
    R, t = synth.random_pose()
    d = sess.generate_correspondences(n, R, t, sigma)
    out[f"g9_gen_{tag}_n"], out[f"g9_gen_{tag}_sigma"], out[f"g9_gen_{tag}_seed"] = np.int64(n), np.float64(sigma), np.int64(900)
    out[f"g9_gen_{tag}_R"], out[f"g9_gen_{tag}_t"] = R, t
    out[f"g9_gen_{tag}_K"], out[f"g9_gen_{tag}_length"] = sess.K, np.float64(sess.LENGTH)
    for k, v in d.items():
        out[f"g9_gen_{tag}_{k}"] = v

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "reference_vectors_harness.npz")
np.savez_compressed(path, **out)
print(f"{path}: {len(out)} arrays; compute_pose_error raised on {int(np.sum(raised))} NaN estimates; "
      f"disambiguation picks {np.bincount(pick, minlength=4).tolist()}")
