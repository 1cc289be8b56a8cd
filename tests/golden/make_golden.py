#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz by running the REFERENCE ITSELF.

Run in the build container only (the GPU box has no /root/reference):

    python -B tests/golden/make_golden.py

The reference module (/root/reference/cvxpnpl.py) imports `scs`, which is absent from
this image.  A stub module named `scs` is placed in sys.modules first;
every function of the reference EXCEPT scs.solve then runs as the reference's own code.
The stub's solve() either records its arguments, returns an injected x, or -- for the
end-to-end vectors only -- calls the oracle's restated SCS (marked `e2e_*`, these pin
the reference's post-processing on a converged solve, not the solver).

Only numeric arrays are stored: inputs and the reference's outputs.  No reference
source text is copied.  Always run with -B so no bytecode is written into /root/reference.
"""
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

_state = {"mode": "record", "x": None, "dobj": 0.0, "last_c": None, "last_kw": None}


def _solve(data, cones, **kw):
    _state["last_c"] = np.array(data["c"], dtype=np.float64)
    _state["last_kw"] = dict(kw)
    _state["last_cones"] = dict(cones)
    if _state["mode"] == "inject":
        return {"x": np.array(_state["x"], dtype=np.float64), "info": {"dobj": float(_state["dobj"])}}
    if _state["mode"] == "oracle":
        import oracle

        c = _state["last_c"]
        tr = c[[0, 10, 19, 27, 34, 40, 45, 49, 52]].sum()
        r = oracle.scs_solve(c, eps=1e-11, max_iters=400000, cscale=10.0 / tr)
        _state["last_x"] = r["x"]
        _state["last_dobj"] = r["info"]["dobj"]
        return r
    # record mode: a harmless feasible point (Z = e9 e9^T) so post-processing does not crash
    x = np.zeros(55)
    x[54] = 1.0
    return {"x": x, "info": {"dobj": 0.0}}


stub = types.ModuleType("scs")
stub.__version__ = "3.2.4"
stub.solve = _solve
sys.modules["scs"] = stub
sys.path.insert(0, "/root/reference")
import cvxpnpl as ref  # noqa: E402

G = {}
rs = np.random.RandomState(20240928)
K_kinect = np.array([[572.41140, 0, 325.26110], [0, 573.57043, 242.04899], [0, 0, 1]])
K_int = np.array([[160, 0, 320], [0, 120, 240], [0, 0, 1]])


def aa2rm(aa):
    ang = np.linalg.norm(aa)
    k = aa / ang
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def rand_pose():
    axis = rs.random_sample(3) - 0.5
    axis /= np.linalg.norm(axis)
    R = aa2rm(2 * np.pi * rs.random_sample() * axis)
    t = np.concatenate([rs.random_sample(2) - 0.5, 1.6 * rs.random_sample(1) + 0.6])
    return R, t


def project(P, K, R, t):
    x = (P @ R.T + t) @ K.T
    return (x / x[:, -1, None])[:, :-1]


# ---- G1: _point_constraints (cvxpnpl.py:20-104) -----------------------------------
for tag, n, K in (("n4", 4, K_kinect), ("n6_intK", 6, K_int), ("n10", 10, K_kinect)):
    R, t = rand_pose()
    P = 0.6 * (rs.random_sample((n, 3)) - 0.5)
    x = project(P, K, R, t) + rs.normal(scale=1.0, size=(n, 2))
    (C1, C2, C3), (N1, N2, N3) = ref._point_constraints(x, P, K)
    G[f"g1_{tag}_pts2d"], G[f"g1_{tag}_pts3d"], G[f"g1_{tag}_K"] = x, P, np.asarray(K, float)
    G[f"g1_{tag}_C"] = np.stack((C1, C2, C3))
    G[f"g1_{tag}_N"] = np.stack((N1, N2, N3))

# ---- G2: _line_constraints (cvxpnpl.py:107-153) -----------------------------------
for tag, n in (("n4", 4), ("n5", 5)):
    R, t = rand_pose()
    P = 0.6 * (rs.random_sample((2 * n, 3)) - 0.5)
    x = project(P, K_kinect, R, t) + rs.normal(scale=0.5, size=(2 * n, 2))
    l3, l2 = P.reshape(n, 2, 3), x.reshape(n, 2, 2)
    Cm, N = ref._line_constraints(l2, l3, K_kinect)
    G[f"g2_{tag}_line2d"], G[f"g2_{tag}_line3d"] = l2, l3
    G[f"g2_{tag}_C"], G[f"g2_{tag}_N"] = Cm, N
G["K_kinect"] = K_kinect

# ---- G4: static SDP data (cvxpnpl.py:387-451) -------------------------------------
G["g4_A"] = ref._A.toarray()
G["g4_b"] = ref._b
G["g4_cone_zero"] = np.array(ref._CONES.get("z", ref._CONES.get("f")))
G["g4_cone_s"] = np.array(ref._CONES["s"])

# ---- G5: _vech10 / _vech10_inv (cvxpnpl.py:346-384) --------------------------------
M = np.arange(100, dtype=float).reshape(10, 10)
M = M + M.T
G["g5_M"] = M
G["g5_vech1"] = ref._vech10(M)
G["g5_vech2"] = ref._vech10(M, 2)
G["g5_vechs2"] = ref._vech10(M, np.sqrt(2))
G["g5_inv"] = ref._vech10_inv(np.arange(55, dtype=float))

# ---- G8 + G3: the three examples (examples/pnp.py, pnl.py, pnpl.py) -----------------
_cap = {}
_orig_sr = ref._solve_relaxation


def _capture_sr(A, B, eps=1e-9, max_iters=2500, verbose=False):
    _cap["A"], _cap["B"] = np.array(A), np.array(B)
    _cap["eps"], _cap["max_iters"] = eps, max_iters
    return _orig_sr(A, B, eps=eps, max_iters=max_iters, verbose=verbose)


ref._solve_relaxation = _capture_sr

# examples/pnp.py:5-26
np.random.seed(0)
np.random.seed(42)
pts = 0.6 * (np.random.random((6, 3)) - 0.5)
R_gt = np.array([[-0.48048015, 0.1391384, -0.86589799], [-0.0333282, -0.98951829, -0.14050899],
                 [-0.8763721, -0.03865296, 0.48008113]])
t_gt = np.array([-0.10266772, 0.25450789, 1.70391109])
pts_2d = project(pts, K_int, R_gt, t_gt)
G["ex_pnp_pts3d"], G["ex_pnp_pts2d"], G["ex_pnp_K"] = pts, pts_2d, K_int
G["ex_pnp_R"], G["ex_pnp_t"] = R_gt, t_gt

# examples/pnl.py:5-31
np.random.seed(0)
np.random.seed(42)
line_3d = 0.6 * (np.random.random((6, 2, 3)) - 0.5)
R_gt2 = np.array([[0.89802142, -0.41500101, 0.14605372], [0.24509948, 0.7476071, 0.61725997],
                  [-0.36535431, -0.51851499, 0.77308372]])
t_gt2 = np.array([-0.0767557, 0.13917375, 1.9708239])
line_2d = project(line_3d.reshape(-1, 3), K_int, R_gt2, t_gt2).reshape(-1, 2, 2)
G["ex_pnl_line3d"], G["ex_pnl_line2d"], G["ex_pnl_K"] = line_3d, line_2d, K_int
G["ex_pnl_R"], G["ex_pnl_t"] = R_gt2, t_gt2

# examples/pnpl.py:5-37
np.random.seed(0)
np.random.seed(42)
pts_b = 0.6 * (np.random.random((4, 3)) - 0.5)
line_3d_b = 0.6 * (np.random.random((4, 2, 3)) - 0.5)
all2d = project(np.vstack((pts_b, line_3d_b.reshape(-1, 3))), K_int, R_gt2, t_gt2)
G["ex_pnpl_pts3d"], G["ex_pnpl_pts2d"] = pts_b, all2d[:4]
G["ex_pnpl_line3d"], G["ex_pnpl_line2d"] = line_3d_b, all2d[4:].reshape(-1, 2, 2)
G["ex_pnpl_K"], G["ex_pnpl_R"], G["ex_pnpl_t"] = K_int, R_gt2, t_gt2

_state["mode"] = "record"
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref.pnp(pts_2d, pts, K_int)
    G["g3_pnp_c"], G["g3_pnp_A"], G["g3_pnp_B"] = _state["last_c"], _cap["A"], _cap["B"]
    G["g3_kw_eps_abs"] = np.array(_state["last_kw"]["eps_abs"])
    G["g3_kw_max_iters"] = np.array(_state["last_kw"]["max_iters"])
    ref.pnl(line_2d, line_3d, K_int)
    G["g3_pnl_c"], G["g3_pnl_A"], G["g3_pnl_B"] = _state["last_c"], _cap["A"], _cap["B"]
    ref.pnpl(all2d[:4], all2d[4:].reshape(-1, 2, 2), pts_b, line_3d_b, K_int)
    G["g3_pnpl_c"], G["g3_pnpl_A"], G["g3_pnpl_B"] = _state["last_c"], _cap["A"], _cap["B"]

# ---- G6/G7: post-solve recovery on injected x (cvxpnpl.py:492-520, 221-343, 156-218) --


def z_of(R):
    return np.append(R.T.reshape(9), 1.0)  # column-major vec(R), homogenised


def inject(Zm, A, B, dobj):
    _state["mode"] = "inject"
    _state["x"] = ref._vech10(Zm)
    _state["dobj"] = dobj
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        poses = _orig_sr(A, B)
    return poses, len(w)


A_ex, B_ex = G["g3_pnp_A"], G["g3_pnp_B"]
# rank 1, exact
Ra, _ = rand_pose()
poses, nw = inject(np.outer(z_of(Ra), z_of(Ra)), A_ex, B_ex, 0.0)
G["g6_r1_R_in"] = Ra
G["g6_r1_x"] = _state["x"]
G["g6_r1_R"], G["g6_r1_t"] = np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])
G["g6_r1_warned"] = np.array(nw)
# rank 1, perturbed (Z not exactly rank one / not exactly a rotation -> exercises the SVD projection)
E = rs.normal(scale=1e-3, size=(10, 10))
Zp = np.outer(z_of(Ra), z_of(Ra)) + 0.5 * (E + E.T)
poses, nw = inject(Zp, A_ex, B_ex, 0.0)
G["g6_r1p_x"] = _state["x"]
G["g6_r1p_R"], G["g6_r1p_t"] = np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])
# rank 2
Rb, _ = rand_pose()
Z2 = 0.6 * np.outer(z_of(Ra), z_of(Ra)) + 0.4 * np.outer(z_of(Rb), z_of(Rb))
poses, nw = inject(Z2, A_ex, B_ex, 0.0)
G["g6_r2_R_in"] = np.stack((Ra, Rb))
G["g6_r2_x"] = _state["x"]
G["g6_r2_R"], G["g6_r2_t"] = np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])
# rank 4
Rc, _ = rand_pose()
Rd, _ = rand_pose()
Z4 = sum(w_ * np.outer(z_of(Rk), z_of(Rk)) for w_, Rk in zip((0.3, 0.3, 0.2, 0.2), (Ra, Rb, Rc, Rd)))
poses, nw = inject(Z4, A_ex, B_ex, 0.0)
G["g6_r4_R_in"] = np.stack((Ra, Rb, Rc, Rd))
G["g6_r4_x"] = _state["x"]
G["g6_r4_R"], G["g6_r4_t"] = np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])
# the intermediate stages for rank 2 and rank 4: eigenvectors in, r_c out; E6Q3 in/out
for tag, Zm, rank in (("r2", Z2, 2), ("r4", Z4, 4)):
    vals, vecs = np.linalg.eigh(Zm)
    G[f"g7_{tag}_vecs"] = vecs
    G[f"g7_{tag}_rc"] = ref._constraint_ortho_det(vecs, rank)
_re6_in = {}
_orig_re6 = ref._re6q3


def _cap_re6(Am):
    _re6_in["A"] = np.array(Am)
    out = _orig_re6(Am)
    _re6_in["out"] = np.stack(out)
    return out


ref._re6q3 = _cap_re6
ref._constraint_ortho_det(G["g7_r4_vecs"], 4)
ref._re6q3 = _orig_re6
G["g7_re6q3_A"], G["g7_re6q3_abc"] = _re6_in["A"], _re6_in["out"]
# NaN sentinel (cvxpnpl.py:493-498)
_state["mode"] = "inject"
_state["x"] = np.full(55, np.nan)
_state["dobj"] = 0.0
poses = _orig_sr(A_ex, B_ex)
G["g6_nan_n"] = np.array(len(poses))
G["g6_nan_R"], G["g6_nan_t"] = poses[0][0], poses[0][1]
# certificate warning fires when | ||Ar||^2 - dobj | > eps (cvxpnpl.py:516-519)
poses, nw = inject(np.outer(z_of(Ra), z_of(Ra)), A_ex, B_ex, 1.0)
G["g6_cert_warned"] = np.array(nw)

# ---- e2e: the reference's pnp/pnl/pnpl driven by the oracle's restated SCS ----------
_state["mode"] = "oracle"
e2e = []
cases = [("pnp", 10, 0, 0.0), ("pnp", 10, 0, 2.0), ("pnp", 6, 0, 0.0), ("pnp", 6, 0, 1.0),
         ("pnpl", 5, 5, 0.0), ("pnpl", 5, 5, 1.0), ("pnl", 0, 6, 0.0), ("pnl", 0, 8, 1.0)]
for i, (kind, n_p, n_l, noise) in enumerate(cases):
    R, t = rand_pose()
    P = 0.6 * (rs.random_sample((n_p + 2 * n_l, 3)) - 0.5)
    x = project(P, K_kinect, R, t)
    if noise > 0:
        x = x + rs.normal(scale=noise, size=x.shape)
    p2, p3 = x[:n_p], P[:n_p]
    l2, l3 = x[n_p:].reshape(n_l, 2, 2), P[n_p:].reshape(n_l, 2, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "pnp":
            poses = ref.pnp(p2, p3, K_kinect)
        elif kind == "pnl":
            poses = ref.pnl(l2, l3, K_kinect)
        else:
            poses = ref.pnpl(p2, l2, p3, l3, K_kinect)
    assert len(poses) == 1, (kind, n_p, n_l, noise, len(poses))
    G[f"e2e_{i}_kind"] = np.array(kind)
    G[f"e2e_{i}_pts2d"], G[f"e2e_{i}_pts3d"] = p2, p3
    G[f"e2e_{i}_line2d"], G[f"e2e_{i}_line3d"] = l2, l3
    G[f"e2e_{i}_Rgt"], G[f"e2e_{i}_tgt"], G[f"e2e_{i}_noise"] = R, t, np.array(noise)
    G[f"e2e_{i}_R"], G[f"e2e_{i}_t"] = poses[0][0], poses[0][1]
    G[f"e2e_{i}_x"], G[f"e2e_{i}_dobj"] = _state["last_x"], np.array(_state["last_dobj"])
    G[f"e2e_{i}_c"] = _state["last_c"]
G["e2e_count"] = np.array(len(cases))

out = os.path.join(HERE, "reference_vectors.npz")
np.savez_compressed(out, **G)
print("wrote", out, "with", len(G), "arrays,", os.path.getsize(out), "bytes")
assert not os.path.exists("/root/reference/__pycache__"), "bytecode leaked into the reference tree"
