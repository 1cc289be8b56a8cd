"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``cvxpnpl_amd``) never does.  See ``oracle.c`` for
the reference file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OrcInfo(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int),
        ("status", C.c_int),
        ("rank", C.c_int),
        ("iters", C.c_int),
        ("scs_status", C.c_int),
        ("dobj", C.c_double),
        ("pobj", C.c_double),
        ("x", C.c_double * 55),
        ("eigs", C.c_double * 10),
        ("res", C.c_double * 3),
        ("B", C.c_double * 27),
    ]


def build(force=False):
    """Compile liboracle.so with gcc (Makefile next to this file)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_scs_solve.argtypes = [_dp, C.c_double, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, _dp]
        _lib.orc_recover.argtypes = [_dp, C.c_double, _dp, C.c_int, _dp, C.c_double, _dp, _dp, _ip, _ip, _dp]
        _lib.orc_solve_relaxation.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_int, _dp, _dp, C.POINTER(OrcInfo)]
        _lib.orc_pnpl.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, _dp, _dp, C.POINTER(OrcInfo)]
        _lib.orc_pnpl_batch.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_double,
                                        C.c_int, _dp, _dp, _ip, _ip, _ip, _dp]
        _lib.orc_vech10.argtypes = [_dp, C.c_double, _dp]
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def point_constraints(pts_2d, pts_3d, K):
    pts_2d, pts_3d, K = _c(pts_2d), _c(pts_3d), _c(K)
    n = len(pts_3d)
    Cm, N = np.zeros((3 * n, 9)), np.zeros((3 * n, 3))
    rc = lib().orc_point_constraints(n, _p(pts_2d), _p(pts_3d), _p(K), _p(Cm), _p(N))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return (Cm[:n], Cm[n:2 * n], Cm[2 * n:]), (N[:n], N[n:2 * n], N[2 * n:])


def line_constraints(line_2d, line_3d, K):
    line_2d, line_3d, K = _c(line_2d), _c(line_3d), _c(K)
    n = len(line_2d)
    Cm, N = np.zeros((2 * n, 9)), np.zeros((2 * n, 3))
    rc = lib().orc_line_constraints(n, _p(line_2d), _p(line_3d), _p(K), _p(Cm), _p(N))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return Cm, N


def eliminate(Cm, N):
    Cm, N = _c(Cm), _c(N)
    m = len(Cm)
    B, A = np.zeros((3, 9)), np.zeros((m, 9))
    rc = lib().orc_eliminate(m, _p(Cm), _p(N), _p(B), _p(A))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return B, A


def vech10(A, scale=1.0):
    A = _c(A)
    v = np.zeros(55)
    lib().orc_vech10(_p(A), float(scale), _p(v))
    return v


def vech10_inv(v):
    v = _c(v)
    A = np.zeros((10, 10))
    lib().orc_vech10_inv(_p(v), _p(A))
    return A


def sdp_constraints():
    Ad, b = np.zeros((77, 55)), np.zeros(77)
    lib().orc_sdp_constraints(_p(Ad), _p(b))
    return Ad, b


def sdp_constraints_rc():
    """dense _A_rc (71 x 55), _b_rc of benchmarks/toolkit/methods/rc.py:9-64"""
    Ad, b = np.zeros((71, 55)), np.zeros(71)
    lib().orc_sdp_constraints_rc(_p(Ad), _p(b))
    return Ad, b


def eigh(A):
    A = _c(A).copy()
    n = len(A)
    w, V = np.zeros(n), np.zeros((n, n))
    lib().orc_eigh(n, _p(A), _p(w), _p(V))
    return w, V


def svd3_uvh(M):
    M = _c(M)
    R = np.zeros((3, 3))
    lib().orc_svd3_uvh(_p(M), _p(R))
    return R


def scs_solve(c, eps=1e-9, max_iters=2500, cscale=1.0):
    """Restated SCS: returns dict with the fields of scs.solve's result the reference reads."""
    c = _c(c)
    x, y, res = np.zeros(55), np.zeros(77), np.zeros(3)
    dobj, pobj, iters = C.c_double(), C.c_double(), C.c_int()
    st = lib().orc_scs_solve(_p(c), eps, max_iters, cscale, _p(x), _p(y), C.byref(dobj), C.byref(pobj), C.byref(iters), _p(res))
    return {"x": x, "y": y, "info": {"dobj": dobj.value, "pobj": pobj.value, "iter": iters.value,
                                      "status": "solved" if st == 0 else "max_iters", "res": res}}


def re6q3(A):
    A = _c(A)
    a, b, c = np.zeros(4), np.zeros(4), np.zeros(4)
    n = lib().orc_re6q3(len(A), _p(A), _p(a), _p(b), _p(c))
    return a[:n], b[:n], c[:n]


def constraint_ortho_det(vecs, rank):
    vecs = _c(vecs)
    out = np.zeros((4, 9))
    n = lib().orc_constraint_ortho_det(_p(vecs), int(rank), _p(out))
    if n < 0:
        raise NotImplementedError
    return out[:n]


def recover(x, dobj, A, B, eps=1e-9):
    x, A, B = _c(x), _c(A), _c(B)
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    st, rk = C.c_int(), C.c_int()
    eig = np.zeros(10)
    n = lib().orc_recover(_p(x), dobj, _p(A), len(A), _p(B), eps, _p(R), _p(t), C.byref(st), C.byref(rk), _p(eig))
    return [(R[i].copy(), t[i].copy()) for i in range(n)], st.value, rk.value


def solve_relaxation(A, B, eps=1e-9, max_iters=2500):
    A, B = _c(A), _c(B)
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    info = OrcInfo()
    n = lib().orc_solve_relaxation(len(A), _p(A), _p(B), eps, max_iters, _p(R), _p(t), C.byref(info))
    return [(R[i].copy(), t[i].copy()) for i in range(n)], info


def solve_relaxation_rc(A, B, eps=1e-9, max_iters=2500):
    """_solve_relaxation_rc (benchmarks/toolkit/methods/rc.py:67-131) restated"""
    A, B = _c(A), _c(B)
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    info = OrcInfo()
    lib().orc_solve_relaxation_rc.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_int, _dp, _dp, C.POINTER(OrcInfo)]
    n = lib().orc_solve_relaxation_rc(len(A), _p(A), _p(B), eps, max_iters, _p(R), _p(t), C.byref(info))
    return [(R[i].copy(), t[i].copy()) for i in range(n)], info


def scs_solve_rc(c, eps=1e-9, max_iters=2500, cscale=1.0):
    """the restated SCS on the rc constraint set (16 zero-cone rows)"""
    c = _c(c)
    x, y, res = np.zeros(55), np.zeros(71), np.zeros(3)
    dobj, pobj, iters = C.c_double(), C.c_double(), C.c_int()
    lib().orc_scs_solve_var.argtypes = [C.c_int, _dp, C.c_double, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, _dp]
    st = lib().orc_scs_solve_var(1, _p(c), eps, max_iters, cscale, _p(x), _p(y), C.byref(dobj), C.byref(pobj), C.byref(iters), _p(res))
    return {"x": x, "y": y, "info": {"dobj": dobj.value, "pobj": pobj.value, "iter": iters.value,
                                      "status": "solved" if st == 0 else "max_iters", "res": res}}


def pnpl(pts_2d, line_2d, pts_3d, line_3d, K, eps=1e-9, max_iters=2500):
    """cvxpnpl.pnpl restated end to end.  Returns (poses, info)."""
    pts_2d = _c(np.reshape(pts_2d, (-1, 2))) if pts_2d is not None else np.zeros((0, 2))
    pts_3d = _c(np.reshape(pts_3d, (-1, 3))) if pts_3d is not None else np.zeros((0, 3))
    line_2d = _c(np.reshape(line_2d, (-1, 2, 2))) if line_2d is not None else np.zeros((0, 2, 2))
    line_3d = _c(np.reshape(line_3d, (-1, 2, 3))) if line_3d is not None else np.zeros((0, 2, 3))
    K = _c(K)
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    info = OrcInfo()
    n = lib().orc_pnpl(len(pts_3d), _p(pts_2d), _p(pts_3d), len(line_3d), _p(line_2d), _p(line_3d), _p(K), eps, max_iters,
                       _p(R), _p(t), C.byref(info))
    return [(R[i].copy(), t[i].copy()) for i in range(n)], info


def pnp(pts_2d, pts_3d, K, eps=1e-9, max_iters=2500):
    return pnpl(pts_2d, None, pts_3d, None, K, eps, max_iters)


def pnl(line_2d, line_3d, K, eps=1e-9, max_iters=2500):
    return pnpl(None, line_2d, None, line_3d, K, eps, max_iters)


def pnpl_batch(pts_2d, line_2d, pts_3d, line_3d, K, eps=1e-9, max_iters=2500):
    """Batch of same-shape problems.  pts_2d [B,np,2], line_2d [B,nl,2,2] (either may be None)."""
    Bn = len(pts_3d) if pts_3d is not None else len(line_3d)
    n_p = pts_3d.shape[1] if pts_3d is not None else 0
    n_l = line_3d.shape[1] if line_3d is not None else 0
    a = [(_c(v) if v is not None else None) for v in (pts_2d, pts_3d, line_2d, line_3d)]
    K = _c(K)
    R, t = np.zeros((Bn, 4, 3, 3)), np.zeros((Bn, 4, 3))
    npz, st, it = np.zeros(Bn, np.int32), np.zeros(Bn, np.int32), np.zeros(Bn, np.int32)
    cost = np.zeros((Bn, 2))
    lib().orc_pnpl_batch(Bn, n_p, _p(a[0]), _p(a[1]), n_l, _p(a[2]), _p(a[3]), _p(K), int(K.ndim == 3), eps, max_iters,
                         _p(R), _p(t), npz.ctypes.data_as(_ip), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(cost))
    return {"R": R, "t": t, "n_poses": npz, "status": st, "iters": it, "cost": cost}


def num_threads():
    return lib().orc_num_threads()
