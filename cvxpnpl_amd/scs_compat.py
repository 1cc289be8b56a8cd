"""The reference's ONE native call, `scs.solve(data, cones, **kw)`, answered by the HIP solver.

cvxpnpl hands its Shor relaxation to SCS as  scs.solve({"A": _A, "b": _b, "c": vech(Q, 2)}, _CONES, eps_abs|eps=..., max_iters=...,
verbose=...)  and reads exactly two fields of the result, results["x"] (cvxpnpl.py:485-492) and results["info"]["dobj"] (:517); the
`rc` ablation does the same with its 16-equality data (benchmarks/toolkit/methods/rc.py:90-96).  `_A`, `_b` and the cones never change
between calls (cvxpnpl.py:451): only the cost vector c carries a problem.  This module is that call for exactly that family:

    import sys, cvxpnpl_amd.scs_compat
    sys.modules["scs"] = cvxpnpl_amd.scs_compat      # before `import cvxpnpl`
    import cvxpnpl                                    # the reference, unmodified: its solves now run on the GPU

`solve` checks that (A, b, cones) ARE one of the two static sets (22 or 16 equalities + the 10 x 10 PSD cone, in the reference's row order
and scaling) and raises ValueError for anything else -- it is not a general conic solver and does not pretend to be one.  The cost is
unpacked to the packed 9 x 9 block the C ABI takes (cvxpnpl_solve_cost_batch, include/cvxpnpl_amd.h) and the answer is the solver's
returned Z in the reference's vech order plus its certified dual bound.  `solve_batch` is the same for [B, 55] costs in one launch.

No CPU fallback: without the HIP library or a GPU these functions raise (cvxpnpl_amd.api).
"""
from typing import Dict, Optional

import numpy as np

from . import _lib

__version__ = "3.2.4"  # the reference switches its cone key ("z") and tolerance keyword ("eps_abs") on the major version (cvxpnpl.py:11-17)

__all__ = ["solve", "solve_batch", "static_data", "cost_from_c"]

_SQRT2 = np.sqrt(2.0)


def _vidx(i: int, j: int) -> int:
    """index of entry (i, j) of a 10 x 10 symmetric matrix in the column-major lower-triangle vech of cvxpnpl.py:346-370"""
    if i < j:
        i, j = j, i
    return 10 * j - j * (j - 1) // 2 + (i - j)


def static_data(variant: int = _lib.VARIANT_FULL):
    """(A [22|16 + 55, 55] dense, b) of the family this module answers for: the equalities <A_i, Z> = b_i as rows on x = vech(Z) --
    Z_99 = 1, R R^T = I (absent from the rc variant), R^T R = I, the three cross products c_i x c_j = c_k -- followed by the diagonal
    block that places  -D x  (D = 1 on diagonal entries, sqrt 2 off) in the PSD cone.  Row order and signs are the reference's
    (cvxpnpl.py:387-451; rc.py:9-64), which tests/test_scs_compat.py checks entry by entry against the golden dump of `_A`."""
    rows = []

    def row(terms):
        r = np.zeros(55)
        for (i, j, v) in terms:
            r[_vidx(i, j)] += v
        rows.append(r)

    row([(9, 9, 1.0)])
    if variant == _lib.VARIANT_FULL:
        for i in range(3):          # rows of R: sum_k r[3k + i] r[3k + j] = delta_ij
            for j in range(i, 3):
                row([(3 * k + i, 3 * k + j, 1.0) for k in range(3)] + ([(9, 9, -1.0)] if i == j else []))
    for i in range(3):              # columns of R: sum_k r[3i + k] r[3j + k] = delta_ij
        for j in range(i, 3):
            row([(3 * i + k, 3 * j + k, 1.0) for k in range(3)] + ([(9, 9, -1.0)] if i == j else []))
    for (a, b_, c) in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):   # c_a x c_b = c_c, component by component, linear through the homogenising column
        for k in range(3):
            k1, k2 = (k + 1) % 3, (k + 2) % 3
            row([(3 * a + k1, 3 * b_ + k2, 1.0), (3 * a + k2, 3 * b_ + k1, -1.0), (3 * c + k, 9, -1.0)])
    n_eq = len(rows)
    A = np.zeros((n_eq + 55, 55))
    A[:n_eq] = np.array(rows)
    for j in range(10):
        for i in range(j, 10):
            e = _vidx(i, j)
            A[n_eq + e, e] = -1.0 if i == j else -_SQRT2
    b = np.zeros(n_eq + 55)
    b[0] = 1.0
    return A, b


_STATIC: Dict[int, tuple] = {}


def _static(variant):
    if variant not in _STATIC:
        _STATIC[variant] = static_data(variant)
    return _STATIC[variant]


def _dense(A):
    return np.asarray(A.todense() if hasattr(A, "todense") else A, dtype=np.float64)


def _identify(data, cones) -> int:
    """VARIANT_FULL / VARIANT_RC if (data["A"], data["b"], cones) is the reference's static problem data, else ValueError"""
    A = _dense(data["A"])
    b = np.asarray(data["b"], dtype=np.float64).reshape(-1)
    n_zero = int(cones.get("z", cones.get("f", -1)))
    if list(np.atleast_1d(cones.get("s", []))) != [10] or int(cones.get("l", 0)) != 0 or len(cones.get("q", [])) != 0 or \
            int(cones.get("ep", 0)) != 0 or int(cones.get("ed", 0)) != 0:
        raise ValueError("cvxpnpl_amd.scs_compat solves cvxpnpl's relaxation only: cones must be {z|f: 22 or 16, s: [10]}")
    variant = {22: _lib.VARIANT_FULL, 16: _lib.VARIANT_RC}.get(n_zero)
    if variant is None or A.shape != (n_zero + 55, 55) or b.shape != (n_zero + 55,):
        raise ValueError(f"cvxpnpl_amd.scs_compat: {n_zero} equalities / A {A.shape} is neither cvxpnpl's 22-row nor its rc 16-row problem")
    As, bs = _static(variant)
    if not (np.allclose(A, As, rtol=0.0, atol=1e-12) and np.array_equal(b, bs)):
        raise ValueError("cvxpnpl_amd.scs_compat: data['A'] / data['b'] are not cvxpnpl's static constraint data (cvxpnpl.py:387-451)")
    return variant


_TRIU9 = np.triu_indices(9)
_C_OF_Q45 = np.array([_vidx(i, j) for i, j in zip(*_TRIU9)])
_SCALE_OF_Q45 = np.array([1.0 if i == j else 0.5 for i, j in zip(*_TRIU9)])
_LAST = np.array([_vidx(9, j) for j in range(10)])


def cost_from_c(c: np.ndarray) -> np.ndarray:
    """c = vech(Q, 2) [..., 55] (cvxpnpl.py:486; off-diagonals doubled) -> the packed upper triangle of Q[:9, :9] [..., 45], row by
    row (api.pack_cost).  The last row / column of Q is zero in every problem the reference poses (cvxpnpl.py:475); anything else
    is outside the solver's family and refused."""
    c = np.asarray(c, dtype=np.float64)
    if c.shape[-1] != 55:
        raise ValueError(f"c must have 55 entries (vech of a 10 x 10 cost), got {c.shape}")
    if np.any(c[..., _LAST] != 0.0):
        raise ValueError("cvxpnpl_amd.scs_compat: the cost's homogenising row / column must be zero (Q = blkdiag(A^T A, 0), cvxpnpl.py:475)")
    return np.ascontiguousarray(c[..., _C_OF_Q45] * _SCALE_OF_Q45)


_SCS_SETTINGS = {"eps", "eps_abs", "eps_rel", "eps_infeas", "max_iters", "verbose", "normalize", "scale", "adaptive_scale", "rho_x", "alpha",
                 "acceleration_lookback", "acceleration_interval", "time_limit_secs", "write_data_filename", "log_csv_filename",
                 "use_indirect", "gpu", "linear_solver", "cg_rate", "warm_start"}


def solve_batch(c, variant: int = _lib.VARIANT_FULL, eps: float = 1e-9, max_iters: int = 2500, device=None, **solver_opts) -> dict:
    """[B, 55] cost vectors of one of the two static problems -> {"x" [B, 55], "dobj" [B], "pobj" [B], "status" [B] int32,
    "iter" [B] int32} as numpy arrays, one HIP launch (cvxpnpl_solve_cost_batch with the returned Z)."""
    import torch

    from . import api

    Q45 = cost_from_c(np.atleast_2d(np.asarray(c, dtype=np.float64)))
    B27 = np.zeros((len(Q45), 27))  # the translation map is not part of this seam (t = -B r is the caller's, cvxpnpl.py:513)
    res = api.solve_cost_batch(Q45, B27, eps=eps, max_iters=max_iters, want_Z=True, variant=variant, device=device, res_tol=0.0, **solver_opts)
    torch.cuda.synchronize(res.Z.device)
    x = res.Z.cpu().numpy()
    cost = res.cost.cpu().numpy()
    status = res.status.cpu().numpy()
    dobj = cost[:, 1].copy()
    # an uncertified exit carries no dual bound (NaN): -inf is the honest one, and it makes the reference's check
    # |cost - dobj| > eps (cvxpnpl.py:516-519) fire, where a NaN would silently pass it
    dobj[~np.isfinite(dobj) & (status != 3)] = -np.inf  # (status 3: non-finite input, x is NaN and the reference returns its NaN pose)
    pobj = np.einsum("bi,bi->b", np.atleast_2d(np.asarray(c, dtype=np.float64)), np.where(np.isfinite(x), x, 0.0))
    return {"x": x, "dobj": dobj, "pobj": pobj, "status": status, "iter": res.iters.cpu().numpy()}


def solve(data, cones, **kw) -> dict:
    """scs.solve(data, cones, **settings) for cvxpnpl's two problems (see the module docstring).  Returns the fields SCS's result has and
    the reference reads -- {"x": ndarray (55,), "info": {"dobj", ...}} -- plus y / s as None (nothing in the reference reads them).

    Settings: eps_abs (SCS 3) or eps (SCS 2) is the certificate tolerance  0 <= cost - dobj <= eps  of the HIP solver, max_iters its
    iteration cap; the other SCS settings are accepted and ignored (they tune an algorithm this is not); unknown names raise, as SCS does."""
    unknown = set(kw) - _SCS_SETTINGS
    if unknown:
        raise TypeError(f"unknown SCS setting(s): {sorted(unknown)}")
    variant = _identify(data, cones)
    eps = float(kw.get("eps_abs", kw.get("eps", 1e-4 if "eps_abs" in kw or "eps" not in kw else 1e-5)))
    max_iters = int(kw.get("max_iters", 100000))
    c = np.asarray(data["c"], dtype=np.float64).reshape(55)
    r = solve_batch(c[None], variant=variant, eps=eps, max_iters=max_iters)
    status = int(r["status"][0])
    if kw.get("verbose"):
        print(f"cvxpnpl_amd.scs_compat: status={_lib.STATUS_NAMES[status]} iter={int(r['iter'][0])} pobj={r['pobj'][0]:.6e} dobj={r['dobj'][0]:.6e}")
    solved = status == 0 or (status == 1 and np.isfinite(r["dobj"][0]))
    info = {"dobj": float(r["dobj"][0]), "pobj": float(r["pobj"][0]), "iter": int(r["iter"][0]),
            "status": "solved" if solved else "solved (inaccurate - reached max_iters)", "status_val": 1 if solved else 2,
            "cvxpnpl_status": status, "gap": float(r["pobj"][0] - r["dobj"][0])}
    return {"x": r["x"][0], "y": None, "s": None, "info": info}
