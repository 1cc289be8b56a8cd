"""TEST-ONLY host build of the device algorithm (see hostsim.cpp).  Not part of the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhostsim.so")
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "cvxpnpl_amd", "csrc")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Opts(C.Structure):
    _fields_ = [("eps", C.c_double), ("max_iters", C.c_int), ("rho", C.c_double), ("alpha", C.c_double),
                ("first_check", C.c_int), ("check_every", C.c_int), ("res_tol", C.c_double), ("jacobi_sweeps", C.c_int),
                ("jacobi_tol", C.c_double), ("warm_start", C.c_int), ("rho_tail", C.c_double), ("tail_from", C.c_int),
                ("adapt_every", C.c_int), ("adapt_from", C.c_int), ("adapt_mu", C.c_double), ("adapt_tau", C.c_double),
                ("stall_from", C.c_int), ("stall_lam", C.c_double), ("stall_res", C.c_double), ("stall_drop", C.c_double), ("variant", C.c_int), ("rescue_from", C.c_int), ("f32_sweeps_until", C.c_int), ("sweep_schedule", C.c_int), ("dual_shift", C.c_double), ("dual_refine", C.c_int)]


def build(force=False):
    srcs = [os.path.join(_HERE, "hostsim.cpp"), os.path.join(_CSRC, "solver_core.h"), os.path.join(_CSRC, "problem_io.h"), os.path.join(_CSRC, "ipm_core.h"),
            os.path.join(_CSRC, "lane_core.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
                               "-o", _SO, srcs[0]])
    return _SO


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("CVXPNPL_HOSTSIM_LIB")  # tools/sanitize.sh: the same source built with ASan + UBSan
        if override:
            _lib = C.CDLL(override)
        else:
            build()
            _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def set_threads(n):
    """OpenMP threads of the batch entry points; returns the previous maximum."""
    return int(lib().hs_set_threads(int(n)))


def default_opts(**kw):
    o = Opts()
    lib().hs_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def solve_batch(pts_2d, pts_3d, line_2d, line_3d, K, opts=None, want_Z=False):
    a = [np.ascontiguousarray(v, dtype=np.float64) if v is not None else None for v in (pts_2d, pts_3d, line_2d, line_3d)]
    K = np.ascontiguousarray(K, dtype=np.float64)
    Bn = len(a[1]) if a[1] is not None else len(a[3])
    n_p = a[1].shape[1] if a[1] is not None else 0
    n_l = a[3].shape[1] if a[3] is not None else 0
    o = opts or default_opts()
    R, t = np.zeros((Bn, 3, 3)), np.zeros((Bn, 3))
    st, it, rk, sw = (np.zeros(Bn, np.int32) for _ in range(4))
    cost = np.zeros((Bn, 2))
    Z = np.zeros((Bn, 55)) if want_Z else None
    lib().hs_solve_batch(Bn, n_p, _p(a[0]), _p(a[1]), n_l, _p(a[2]), _p(a[3]), _p(K), int(K.ndim == 3), C.byref(o), _p(R), _p(t),
                         st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(cost), rk.ctypes.data_as(_ip), sw.ctypes.data_as(_ip), _p(Z))
    return {"R": R, "t": t, "status": st, "iters": it, "cost": cost, "rank": rk, "sweeps": sw, "Z": Z}


def ipm_batch(pts_2d, pts_3d, line_2d, line_3d, K, opts=None, want_Z=False):
    """The interior-point path (csrc/ipm_core.h) on the host: same outputs as solve_batch."""
    a = [np.ascontiguousarray(v, dtype=np.float64) if v is not None else None for v in (pts_2d, pts_3d, line_2d, line_3d)]
    K = np.ascontiguousarray(K, dtype=np.float64)
    Bn = len(a[1]) if a[1] is not None else len(a[3])
    n_p = a[1].shape[1] if a[1] is not None else 0
    n_l = a[3].shape[1] if a[3] is not None else 0
    o = opts or default_opts()
    R, t = np.zeros((Bn, 3, 3)), np.zeros((Bn, 3))
    st, it, rk = (np.zeros(Bn, np.int32) for _ in range(3))
    cost = np.zeros((Bn, 2))
    Z = np.zeros((Bn, 55)) if want_Z else None
    lib().hs_ipm_batch(Bn, n_p, _p(a[0]), _p(a[1]), n_l, _p(a[2]), _p(a[3]), _p(K), int(K.ndim == 3), C.byref(o), _p(R), _p(t),
                       st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(cost), rk.ctypes.data_as(_ip), _p(Z))
    return {"R": R, "t": t, "status": st, "iters": it, "cost": cost, "rank": rk, "Z": Z}


def assemble(pts_2d, pts_3d, line_2d, line_3d, K):
    a = [np.ascontiguousarray(v, dtype=np.float64) if v is not None else None for v in (pts_2d, pts_3d, line_2d, line_3d)]
    K = np.ascontiguousarray(K, dtype=np.float64)
    B, Q9 = np.zeros((3, 9)), np.zeros(45)
    rc = lib().hs_assemble(len(a[1]) if a[1] is not None else 0, _p(a[0]), _p(a[1]), len(a[3]) if a[3] is not None else 0,
                           _p(a[2]), _p(a[3]), _p(K), _p(B), _p(Q9))
    Q = np.zeros((9, 9))
    k = 0
    for i in range(9):
        for j in range(i, 9):
            Q[i, j] = Q[j, i] = Q9[k]
            k += 1
    return rc, B, Q


def proj_affine(E55, homog):
    E = np.ascontiguousarray(E55, dtype=np.float64).copy()
    lib().hs_proj_affine(_p(E), int(homog))
    return E


def pospart(W55):
    W = np.ascontiguousarray(W55, dtype=np.float64)
    Wp, lam = np.zeros(55), np.zeros(10)
    s = lib().hs_pospart(_p(W), _p(Wp), _p(lam))
    return Wp, lam, s


def dual_lambda(R, rhs, symm=False):
    R = np.ascontiguousarray(R, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    out = np.zeros(10)
    lib().hs_dual_lambda(_p(R), _p(rhs), int(symm), _p(out))
    return out


def rounds_to(v, Rp, tol=0.1):
    v = np.ascontiguousarray(v, dtype=np.float64)
    Rp = np.ascontiguousarray(Rp, dtype=np.float64)
    d0 = C.c_double()
    lib().hs_rounds_to.restype = C.c_int
    lib().hs_rounds_to.argtypes = [_dp, _dp, C.c_double, C.POINTER(C.c_double)]
    ok = lib().hs_rounds_to(_p(v), _p(Rp), float(tol), C.byref(d0))
    return bool(ok), d0.value


def solve_cost_batch(Q45, B27, opts=None, variant=0, want_Z=False):
    """host build of the device algorithm at the cost seam (cvxpnpl_solve_cost_batch); variant 0 full / 1 rc"""
    Q45 = np.ascontiguousarray(Q45, dtype=np.float64).reshape(-1, 45)
    B27 = np.ascontiguousarray(B27, dtype=np.float64).reshape(-1, 27)
    Bn = len(Q45)
    o = opts or default_opts()
    o.variant = int(variant)
    R, t = np.zeros((Bn, 3, 3)), np.zeros((Bn, 3))
    st, it, rk = (np.zeros(Bn, np.int32) for _ in range(3))
    cost = np.zeros((Bn, 2))
    Z = np.zeros((Bn, 55)) if want_Z else None
    lib().hs_solve_cost_batch(Bn, _p(Q45), _p(B27), C.byref(o), _p(R), _p(t), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(cost),
                              rk.ctypes.data_as(_ip), _p(Z))
    return {"R": R, "t": t, "status": st, "iters": it, "cost": cost, "rank": rk, "Z": Z}


def proj_affine_rc(E55, homog):
    E = np.ascontiguousarray(E55, dtype=np.float64).copy()
    lib().hs_proj_affine_rc(_p(E), int(homog))
    return E


def lane_phase(pts_2d, pts_3d, line_2d, line_3d, K, iters, impl, opts=None, dbl=False):
    """First phase of the lane-hybrid schedule on the host (hs_lane_phase): impl 0 = the general scalar core
    (cvx::solve_problem<TWIN = false>), impl 1 = the register-budgeted restatement (cvxl::lane_phase).  status -1 = parked,
    handoff [B,56] = W (55) + iteration count."""
    a = [np.ascontiguousarray(v, dtype=np.float64) if v is not None else None for v in (pts_2d, pts_3d, line_2d, line_3d)]
    K = np.ascontiguousarray(K, dtype=np.float64)
    Bn = len(a[1]) if a[1] is not None else len(a[3])
    n_p = a[1].shape[1] if a[1] is not None else 0
    n_l = a[3].shape[1] if a[3] is not None else 0
    o = opts or default_opts()
    R, t = np.zeros((Bn, 3, 3)), np.zeros((Bn, 3))
    st, it, sw = (np.zeros(Bn, np.int32) for _ in range(3))
    cost, ho, Z = np.zeros((Bn, 2)), np.zeros((Bn, 56)), np.full((Bn, 55), np.nan)
    lib().hs_lane_phase(Bn, n_p, _p(a[0]), _p(a[1]), n_l, _p(a[2]), _p(a[3]), _p(K), int(K.ndim == 3), C.byref(o), int(iters), int(impl), int(dbl),
                        _p(R), _p(t), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(cost), sw.ctypes.data_as(_ip), _p(ho), _p(Z))
    return {"R": R, "t": t, "status": st, "iters": it, "cost": cost, "sweeps": sw, "handoff": ho, "Z": Z}


def dual_retry_direction(R):
    R = np.ascontiguousarray(R, dtype=np.float64)
    D = np.zeros((10, 10))
    lib().hs_dual_retry_direction.argtypes = [_dp, _dp]
    lib().hs_dual_retry_direction(_p(R), _p(D))
    return D


def newton_tables():
    """constant tables of cvx::dual_newton as dense matrices [15, 10, 10]"""
    M = np.zeros((15, 10, 10))
    lib().hs_newton_tables.argtypes = [_dp]
    lib().hs_newton_tables(_p(M))
    return M


def dual_newton(S55, R, delta, lam=float("nan")):
    """cvx::dual_newton on one failed dual (S55: packed, delta on the diagonal) -> (min pivot, |S z|_inf, z^T S z, Newton steps)"""
    S = np.ascontiguousarray(S55, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    out = np.zeros(3)
    lib().hs_dual_newton.argtypes = [_dp, _dp, C.c_double, C.c_double, _dp]
    lib().hs_dual_newton.restype = C.c_int
    steps = lib().hs_dual_newton(_p(S), _p(R), float(delta), float(lam), _p(out))
    return out[0], out[1], out[2], steps


def proj_affine_homog(E55, variant=0):
    E = np.ascontiguousarray(E55, dtype=np.float64).copy()
    lib().hs_proj_affine_homog.argtypes = [_dp, C.c_int]
    lib().hs_proj_affine_homog(_p(E), int(variant))
    return E


def ipm_solve(Qs55, variant=0):
    """cvx::ipm_solve on trace-normalised costs [B, 55] -> Z [B,10,10], S [B,10,10], gap [B], iters [B]"""
    Qs55 = np.ascontiguousarray(Qs55, dtype=np.float64)
    Bn = len(Qs55)
    Z, S, gap, it = np.zeros((Bn, 10, 10)), np.zeros((Bn, 10, 10)), np.zeros(Bn), np.zeros(Bn, dtype=np.int32)
    lib().hs_ipm_solve(Bn, _p(Qs55), int(variant), _p(Z), _p(S), _p(gap), it.ctypes.data_as(_ip))
    return Z, S, gap, it


def ipm_rows(variant=0):
    """the constraint rows A_i of the interior-point solve as dense [rows, 10, 10]"""
    A = np.zeros((21, 10, 10))
    nr = lib().hs_ipm_rows(int(variant), _p(A))
    return A[:nr]
