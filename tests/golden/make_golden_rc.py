#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors_rc.npz by running the REFERENCE's rc variant itself
(benchmarks/toolkit/methods/rc.py: the relaxation without the six row-orthonormality equalities).

Build container only:    python -B tests/golden/make_golden_rc.py

Same recipe as make_golden.py: a stub module named `scs` stands in for the absent solver; every line of the
reference except scs.solve runs as the reference's own code.  rc.py is loaded as a stand-alone module (its package
__init__ imports MATLAB / OpenCV wrappers that are out of scope).  The stub either returns an injected x (pins the
reference's recovery, rc.py:104-131) or calls the oracle's restated SCS on the 16-equality set (`e2e_*`: pins the
post-processing on a converged solve, and records the c / cones / x the reference handed over).
Only numeric arrays are stored.  A separate file so that reference_vectors.npz stays bit-reproducible.
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

_state = {"mode": "inject", "x": None, "last_c": None, "last_cones": None, "last_kw": None}


def _solve(data, cones, **kw):
    _state["last_c"] = np.array(data["c"], dtype=np.float64)
    _state["last_cones"] = dict(cones)
    _state["last_kw"] = dict(kw)
    _state["last_A_shape"] = data["A"].shape
    if _state["mode"] == "inject":
        return {"x": np.array(_state["x"], dtype=np.float64), "info": {"dobj": 0.0}}
    import oracle

    c = _state["last_c"]
    tr = c[[0, 10, 19, 27, 34, 40, 45, 49, 52]].sum()
    r = oracle.scs_solve_rc(c, eps=1e-11, max_iters=400000, cscale=10.0 / tr)
    _state["last_x"] = r["x"]
    return r


stub = types.ModuleType("scs")
stub.__version__ = "2.1.4"  # rc.py calls scs.solve with the scs 2 keywords (eps=, cone key "f")
stub.solve = _solve
sys.modules["scs"] = stub
sys.path.insert(0, "/root/reference")
import cvxpnpl as ref  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_rc", "/root/reference/benchmarks/toolkit/methods/rc.py")
rc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rc)

G = {}
rs = np.random.RandomState(20260928)
K = np.array([[572.41140, 0, 325.26110], [0, 573.57043, 242.04899], [0, 0, 1]])


def aa2rm(aa):
    ang = np.linalg.norm(aa)
    k = aa / ang
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def rand_pose():
    axis = rs.random_sample(3) - 0.5
    axis /= np.linalg.norm(axis)
    return aa2rm(2 * np.pi * rs.random_sample() * axis), np.concatenate([rs.random_sample(2) - 0.5, 1.6 * rs.random_sample(1) + 0.6])


def project(P, R, t):
    x = (P @ R.T + t) @ K.T
    return (x / x[:, -1, None])[:, :-1]


def A_B(x, P):
    (C1, C2, C3), (N1, N2, N3) = ref._point_constraints(x, P, K)
    C, N = np.vstack((C1, C2, C3)), np.vstack((N1, N2, N3))
    B = np.linalg.solve(N.T @ N, N.T) @ C
    return C - N @ B, B


# ---- static data (rc.py:9-64) ------------------------------------------------------------
G["rc_A"] = rc._A_rc.toarray()
G["rc_b"] = rc._b_rc
G["K"] = K

# ---- recovery on injected x (rc.py:104-131) -------------------------------------------------
R0, t0 = rand_pose()
P0 = 0.6 * (rs.random_sample((8, 3)) - 0.5)
A0, B0 = A_B(project(P0, R0, t0), P0)
G["inj_A"], G["inj_B"] = A0, B0


def z_of(R):
    return np.append(R.T.reshape(9), 1.0)


Ra, _ = rand_pose()
Rb, _ = rand_pose()
E = rs.normal(scale=1e-3, size=(10, 10))
for tag, Zm in (("r1", np.outer(z_of(Ra), z_of(Ra))), ("r1p", np.outer(z_of(Ra), z_of(Ra)) + 0.5 * (E + E.T)),
                ("r2", 0.6 * np.outer(z_of(Ra), z_of(Ra)) + 0.4 * np.outer(z_of(Rb), z_of(Rb)))):
    _state["mode"], _state["x"] = "inject", ref._vech10(Zm)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        poses = rc._solve_relaxation_rc(A0, B0)
    G[f"inj_{tag}_x"] = _state["x"]
    G[f"inj_{tag}_R"], G[f"inj_{tag}_t"] = np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])
G["cone_f"] = np.array(_state["last_cones"]["f"])
G["cone_s"] = np.array(_state["last_cones"]["s"])
G["kw_eps"] = np.array(_state["last_kw"]["eps"])
G["kw_max_iters"] = np.array(_state["last_kw"]["max_iters"])
_state["x"] = np.full(55, np.nan)
poses = rc._solve_relaxation_rc(A0, B0)
G["inj_nan_R"], G["inj_nan_t"] = poses[0][0], poses[0][1]

# ---- e2e: _solve_relaxation_rc driven by the oracle's restated SCS (16-equality set) ---------
_state["mode"] = "oracle"
cases = [(10, 0.0), (10, 2.0), (8, 1.0), (6, 0.5), (12, 1.0), (20, 2.0)]
for i, (n, noise) in enumerate(cases):
    R, t = rand_pose()
    P = 0.6 * (rs.random_sample((n, 3)) - 0.5)
    x = project(P, R, t)
    if noise > 0:
        x = x + rs.normal(scale=noise, size=x.shape)
    A, B = A_B(x, P)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        poses = rc._solve_relaxation_rc(A, B)
    assert len(poses) == 1, (n, noise, len(poses))
    G[f"e2e_{i}_pts2d"], G[f"e2e_{i}_pts3d"] = x, P
    G[f"e2e_{i}_A"], G[f"e2e_{i}_B"] = A, B
    G[f"e2e_{i}_Rgt"], G[f"e2e_{i}_tgt"] = R, t
    G[f"e2e_{i}_R"], G[f"e2e_{i}_t"] = poses[0][0], poses[0][1]
    G[f"e2e_{i}_c"], G[f"e2e_{i}_x"] = _state["last_c"], _state["last_x"]
G["e2e_count"] = np.array(len(cases))

out = os.path.join(HERE, "reference_vectors_rc.npz")
np.savez_compressed(out, **G)
print("wrote", out, "with", len(G), "arrays,", os.path.getsize(out), "bytes")
assert not os.path.exists("/root/reference/__pycache__"), "bytecode leaked into the reference tree"
assert not os.path.exists("/root/reference/benchmarks/toolkit/methods/__pycache__"), "bytecode leaked into the reference tree"
