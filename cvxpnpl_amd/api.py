"""Host side of the drop-in boundary: cvxpnpl's pnp / pnl / pnpl and batched variants.

Signatures, defaults, return type and warning / NaN behaviour of the three public functions
follow the reference (cvxpnpl.py:523-530, :555-562, :586-595, :493-498, :516-519).  The
arithmetic runs in the HIP library (include/cvxpnpl_amd.h) on the current torch device;
PyTorch is used for device memory and streams only.
"""
import ctypes as C
import threading
import warnings
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib

__all__ = ["pnp", "pnl", "pnpl", "pnp_batch", "pnl_batch", "pnpl_batch", "BatchResult", "score_hypotheses",
           "solve_cost_batch", "solve_relaxation", "solve_relaxation_rc", "pack_cost"]


class BatchResult(dict):
    """R [B,3,3], t [B,3], status [B] int32, iters [B] int32, cost [B,2] (||Ar||^2, dobj),
    work [B,2] (rank, sweeps) and optionally Z [B,55]; torch tensors on the input device."""

    def __getattr__(self, name):
        # AttributeError (not KeyError) for missing names: hasattr(), copy.deepcopy() and pickle rely on it
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("cvxpnpl_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")


def _as_dev(x, device, shape_tail):
    if x is None:
        return None
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64))
    x = x.to(device=device, dtype=torch.float64).contiguous()
    if tuple(x.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError(f"expected trailing shape {shape_tail}, got {tuple(x.shape)}")
    return x


def _pair_2d(pts_2d, line_2d, device, batch, n_p, n_l):
    """The 2D halves of the correspondences, checked against the 3D halves: both members of a pair present,
    same [batch, n] leading shape."""
    p2 = l2 = None
    if n_p:
        if pts_2d is None:
            raise ValueError("pts_3d given without pts_2d")
        p2 = _as_dev(pts_2d, device, (2,))
        if p2.numel() != batch * n_p * 2:
            raise ValueError(f"pts_2d {tuple(p2.shape)} does not match pts_3d [{batch},{n_p},3]")
        p2 = p2.reshape(batch, n_p, 2)
    if n_l:
        if line_2d is None:
            raise ValueError("line_3d given without line_2d")
        l2 = _as_dev(line_2d, device, (2, 2))
        if l2.numel() != batch * n_l * 4:
            raise ValueError(f"line_2d {tuple(l2.shape)} does not match line_3d [{batch},{n_l},2,3]")
        l2 = l2.reshape(batch, n_l, 2, 2)
    return p2, l2


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def pnpl_batch(pts_2d, line_2d, pts_3d, line_3d, K, eps: float = 1e-9, max_iters: int = 2500, want_Z: bool = False,
               device=None, **solver_opts) -> BatchResult:
    """Solve B independent PnPL problems of identical shape on the GPU.

    pts_2d [B,n_p,2], line_2d [B,n_l,2,2], pts_3d [B,n_p,3], line_3d [B,n_l,2,3]
    (argument order of cvxpnpl.pnpl, cvxpnpl.py:586-591); K [3,3] or [B,3,3].  Either the
    point or the line pair may be None / empty.  torch tensors (any device) or numpy arrays.
    """
    _require_gpu()
    L = _lib.lib()
    if device is None:
        for a in (pts_3d, line_3d, pts_2d, line_2d):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                device = a.device
                break
        else:
            device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    p3 = _as_dev(pts_3d, device, (3,)) if pts_3d is not None else None
    l3 = _as_dev(line_3d, device, (2, 3)) if line_3d is not None else None
    n_p = p3.shape[-2] if p3 is not None and p3.dim() >= 3 else 0
    n_l = l3.shape[-3] if l3 is not None and l3.dim() >= 4 else 0
    if n_p == 0 and n_l == 0:
        raise ValueError("need at least one point or line correspondence ([B,n,3] points / [B,n,2,3] lines)")
    batch = (p3 if n_p else l3).shape[0]
    if n_p and n_l and l3.shape[0] != batch:
        raise ValueError(f"points and lines describe different batches ({batch} vs {l3.shape[0]})")
    Kd = _as_dev(K, device, (3, 3))
    per = int(Kd.dim() == 3)
    if Kd.dim() not in (2, 3) or (per and Kd.shape[0] != batch):
        raise ValueError("K must be [3,3] or [batch,3,3]")
    if batch > 0 and use_blocked_assembly(n_p + 2 * n_l, batch):
        # the scalability regime (benchmarks/scalability/pnp.py:37-40: up to 10^4 points per problem): bandwidth-shaped
        # blocked assembly, then the solve at the cost seam -- instead of one wavefront streaming the problem
        Bt, Qt = assemble_batch(pts_2d, line_2d, p3 if n_p else None, l3 if n_l else None, Kd, device=device, blocked=True)
        return solve_cost_batch(Qt, Bt, eps=eps, max_iters=max_iters, want_Z=want_Z, device=device, **solver_opts)
    p2, l2 = _pair_2d(pts_2d, line_2d, device, batch, n_p, n_l)
    p3 = p3.reshape(batch, n_p, 3) if n_p else None
    l3 = l3.reshape(batch, n_l, 2, 3) if n_l else None
    opts = _lib.default_opts(eps=float(eps), max_iters=int(max_iters), **solver_opts)
    with torch.cuda.device(device):
        R, t, status, iters, cost, work, Z = _alloc_outputs(batch, device, want_Z)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = L.cvxpnpl_solve_batch(batch, n_p, _ptr(p2), _ptr(p3), n_l, _ptr(l2), _ptr(l3), _ptr(Kd), per, C.byref(opts),
                                   _ptr(R), _ptr(t), _ptr(status), _ptr(iters), _ptr(cost), _ptr(Z), _ptr(work),
                                   C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_solve_batch failed ({rc}): {_lib.last_error()}")
    out = BatchResult(R=R, t=t, status=status, iters=iters, cost=cost, work=work)
    if want_Z:
        out["Z"] = Z
    return out


def pnp_batch(pts_2d, pts_3d, K, eps: float = 1e-9, max_iters: int = 2500, **kw) -> BatchResult:
    """B independent PnP problems: pts_2d [B,n,2], pts_3d [B,n,3] (cvxpnpl.pnp, cvxpnpl.py:523)."""
    return pnpl_batch(pts_2d, None, pts_3d, None, K, eps=eps, max_iters=max_iters, **kw)


def pnl_batch(line_2d, line_3d, K, eps: float = 1e-9, max_iters: int = 2500, **kw) -> BatchResult:
    """B independent PnL problems: line_2d [B,n,2,2], line_3d [B,n,2,3] (cvxpnpl.pnl, cvxpnpl.py:555)."""
    return pnpl_batch(None, line_2d, None, line_3d, K, eps=eps, max_iters=max_iters, **kw)


def recover_multi(Z55: np.ndarray, B27: np.ndarray, Q45: Optional[np.ndarray] = None) -> List[Tuple[np.ndarray, np.ndarray]]:
    """All poses of a rank > 1 SDP solution (cvxpnpl.py:507 -> :221-343), host side.  With Q45 (packed
    A^T A) every pose is Newton-polished on SO(3)."""
    L = _lib.lib()
    Z55 = np.ascontiguousarray(Z55, dtype=np.float64)
    B27 = np.ascontiguousarray(B27, dtype=np.float64)
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    dp = C.POINTER(C.c_double)
    q = np.ascontiguousarray(Q45, dtype=np.float64).ctypes.data_as(dp) if Q45 is not None else None
    n = L.cvxpnpl_recover_multi(Z55.ctypes.data_as(dp), B27.ctypes.data_as(dp), q, R.ctypes.data_as(dp), t.ctypes.data_as(dp))
    if n < 0:
        raise NotImplementedError  # cvxpnpl.py:340-341
    return [(R[i].copy(), t[i].copy()) for i in range(n)]


def recover_multi_batch(res: "BatchResult", B, Q=None, n_threads: int = 0):
    """All poses of every rank > 1 problem of a batch result (solved with want_Z=True): the host-side cold path
    (cvxpnpl.py:507 -> :221-343) on all host cores.  B, Q: the outputs of assemble_batch for the same inputs.
    Returns (R [batch,4,3,3], t [batch,4,3], n_poses [batch]) as numpy arrays; n_poses is 0 for problems whose
    status is not CVXPNPL_RANK_GT1."""
    L = _lib.lib()
    if getattr(res, "Z", None) is None:
        raise ValueError("recover_multi_batch needs the SDP solutions: solve with want_Z=True")
    Z = np.ascontiguousarray(res.Z.detach().cpu().numpy() if isinstance(res.Z, torch.Tensor) else res.Z, dtype=np.float64)
    st = np.ascontiguousarray(res.status.detach().cpu().numpy() if isinstance(res.status, torch.Tensor) else res.status, dtype=np.int32)
    tonp = lambda x: np.ascontiguousarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, dtype=np.float64)  # noqa: E731
    Bn = tonp(B).reshape(-1, 27)
    Qn = tonp(Q).reshape(-1, 45) if Q is not None else None
    n = len(st)
    if not (len(Z) == n and len(Bn) == n and (Qn is None or len(Qn) == n)):
        raise ValueError("recover_multi_batch: Z, B, Q and status must describe the same batch")
    R, t, cnt = np.zeros((n, 4, 3, 3)), np.zeros((n, 4, 3)), np.zeros(n, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    rc = L.cvxpnpl_recover_multi_batch(n, st.ctypes.data_as(ip), Z.ctypes.data_as(dp), Bn.ctypes.data_as(dp),
                                       Qn.ctypes.data_as(dp) if Qn is not None else None, R.ctypes.data_as(dp), t.ctypes.data_as(dp),
                                       cnt.ctypes.data_as(ip), int(n_threads))
    if rc != 0:
        raise RuntimeError("cvxpnpl_recover_multi_batch: bad arguments")
    return R, t, cnt


def recover_multi_device(res: "BatchResult", B, Q=None):
    """recover_multi_batch on the DEVICE (cvxpnpl_recover_multi_device): every rank > 1 problem of a batch result solved
    with want_Z=True, one HIP launch, nothing leaves the GPU.  B, Q: outputs of assemble_batch (device tensors).  Returns
    device tensors (R [batch,4,3,3], t [batch,4,3], n_poses [batch] int32)."""
    _require_gpu()
    L = _lib.lib()
    if getattr(res, "Z", None) is None:
        raise ValueError("recover_multi_device needs the SDP solutions: solve with want_Z=True")
    dev = res.Z.device
    Z = res.Z.contiguous()
    st = res.status.to(torch.int32).contiguous()
    Bd = torch.as_tensor(B).to(device=dev, dtype=torch.float64).contiguous().reshape(-1, 27)
    Qd = torch.as_tensor(Q).to(device=dev, dtype=torch.float64).contiguous().reshape(-1, 45) if Q is not None else None
    n = Z.shape[0]
    if Bd.shape[0] != n or (Qd is not None and Qd.shape[0] != n) or st.shape[0] != n:
        raise ValueError("recover_multi_device: Z, B, Q and status must describe the same batch")
    with torch.cuda.device(dev):
        R = torch.empty((n, 4, 3, 3), dtype=torch.float64, device=dev)
        t = torch.empty((n, 4, 3), dtype=torch.float64, device=dev)
        cnt = torch.empty((n,), dtype=torch.int32, device=dev)
        rc = L.cvxpnpl_recover_multi_device(n, _ptr(st), _ptr(Z), _ptr(Bd), _ptr(Qd), _ptr(R), _ptr(t), _ptr(cnt),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_recover_multi_device failed ({rc}): {_lib.last_error()}")
    return R, t, cnt


def _alloc_outputs(batch, device, want_Z):
    R = torch.empty((batch, 3, 3), dtype=torch.float64, device=device)
    t = torch.empty((batch, 3), dtype=torch.float64, device=device)
    status = torch.empty((batch,), dtype=torch.int32, device=device)
    iters = torch.empty((batch,), dtype=torch.int32, device=device)
    cost = torch.empty((batch, 2), dtype=torch.float64, device=device)
    work = torch.empty((batch, 2), dtype=torch.int32, device=device)
    Z = torch.empty((batch, 55), dtype=torch.float64, device=device) if want_Z else None
    return R, t, status, iters, cost, work, Z


_IU9 = np.triu_indices(9)


def pack_cost(Q):
    """[..., 9, 9] symmetric (A^T A, cvxpnpl.py:475) -> [..., 45]: the upper triangle row by row, the layout of
    cvxpnpl_solve_cost_batch's d_Q45 (and of assemble_batch's second output).  numpy or torch."""
    return Q[..., _IU9[0], _IU9[1]]


def solve_cost_batch(Q45, B27, eps: float = 1e-9, max_iters: int = 2500, want_Z: bool = False, variant: int = _lib.VARIANT_FULL,
                     device=None, **solver_opts) -> BatchResult:
    """The batched solve at the seam of the reference's _solve_relaxation(A, B, ...) (cvxpnpl.py:454-460): the caller
    brings Q45 [B,45] (packed A^T A, see pack_cost / assemble_batch) and B27 [B,27] or [B,3,9] (t = -B r) instead of
    correspondences.  variant=VARIANT_RC solves the reference's 16-equality ablation (benchmarks/toolkit/methods/rc.py)."""
    _require_gpu()
    L = _lib.lib()
    if device is None:
        device = Q45.device if isinstance(Q45, torch.Tensor) and Q45.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    Qd = _as_dev(Q45, device, (45,))
    batch = Qd.shape[0] if Qd.dim() == 2 else 1
    Qd = Qd.reshape(batch, 45)
    Bd = torch.as_tensor(np.ascontiguousarray(B27, dtype=np.float64)) if not isinstance(B27, torch.Tensor) else B27
    Bd = Bd.to(device=device, dtype=torch.float64).contiguous()
    if Bd.numel() != batch * 27:
        raise ValueError(f"B {tuple(Bd.shape)} does not match {batch} problems x (3 x 9)")
    Bd = Bd.reshape(batch, 27)
    opts = _lib.default_opts(eps=float(eps), max_iters=int(max_iters), variant=int(variant), **solver_opts)
    with torch.cuda.device(device):
        R, t, status, iters, cost, work, Z = _alloc_outputs(batch, device, want_Z)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = L.cvxpnpl_solve_cost_batch(batch, _ptr(Qd), _ptr(Bd), C.byref(opts), _ptr(R), _ptr(t), _ptr(status), _ptr(iters),
                                        _ptr(cost), _ptr(Z), _ptr(work), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_solve_cost_batch failed ({rc}): {_lib.last_error()}")
    out = BatchResult(R=R, t=t, status=status, iters=iters, cost=cost, work=work)
    if want_Z:
        out["Z"] = Z
    return out


def ipm_batch(Q45, variant: int = _lib.VARIANT_FULL, device=None):
    """The interior-point solve of the relaxation alone (cvxpnpl_ipm_batch; four problems per wavefront, csrc/ipm_quad.h): Q45 [B,45]
    packed A^T A (pack_cost / assemble_batch; any positive scale -- it is normalised to trace 1 here) -> Z [B,10,10], S [B,10,10]
    (primal / dual iterates of  min <Q/tr Q, Z> s.t. the reference's equality rows, cvxpnpl.py:387-451), gap [B] = <Z, S>, iters [B].
    No rounding, polish or certificate: what opts.rescue_from runs for a slow problem before the first-order iteration takes over again."""
    _require_gpu()
    L = _lib.lib()
    if device is None:
        device = Q45.device if isinstance(Q45, torch.Tensor) and Q45.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    Qd = _as_dev(Q45, device, (45,)).reshape(-1, 45)
    batch = Qd.shape[0]
    di = torch.as_tensor(np.flatnonzero(_IU9[0] == _IU9[1]), device=device)
    Qn = Qd / Qd[:, di].sum(dim=1, keepdim=True)
    iu10 = np.triu_indices(10)
    sel = torch.as_tensor(np.flatnonzero((iu10[0] < 9) & (iu10[1] < 9)), device=device)
    Qs = torch.zeros((batch, 55), dtype=torch.float64, device=device)
    Qs[:, sel] = Qn  # (vech order of the 10 x 10 = the 9 x 9 rows with one more column each)
    with torch.cuda.device(device):
        Z = torch.empty((batch, 10, 10), dtype=torch.float64, device=device)
        S = torch.empty((batch, 10, 10), dtype=torch.float64, device=device)
        gap = torch.empty((batch,), dtype=torch.float64, device=device)
        iters = torch.empty((batch,), dtype=torch.int32, device=device)
        rc = L.cvxpnpl_ipm_batch(batch, _ptr(Qs), int(variant), _ptr(Z), _ptr(S), _ptr(gap), _ptr(iters), C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_ipm_batch failed ({rc}): {_lib.last_error()}")
    return Z, S, gap, iters  # (iters: iterations | reason << 8, see cvxpnpl_ipm_batch)


def _poses_of_single(res, Bt, Qt, verbose, certify_warning=True) -> List[Tuple[np.ndarray, np.ndarray]]:
    """List[(R, t)] of a one-problem BatchResult, with the reference's NaN sentinel (cvxpnpl.py:493-498), rank > 1
    branch (:507) and "not certifiably optimal" warning (:516-519)."""
    status = int(res.status[0])
    if status == 3:  # cvxpnpl.py:493-498
        if verbose:
            warnings.warn("The SDP solver did not return a valid solution. Increasing max_iters might solve the issue.")
        return [(np.full((3, 3), np.nan), np.full(3, np.nan))]
    if status == 1:  # rank > 1: cvxpnpl.py:507
        poses = recover_multi(res.Z[0].cpu().numpy(), Bt, Qt)
        # cvxpnpl.py:516-519: warn unless |cost - dobj| <= eps.  A certified twin pair (exact two-fold ambiguity,
        # e.g. a planar scene) carries its certified lower bound in cost[1]; an uncertified exit has NaN there.
        # (the kernel writes a finite dobj only with 0 <= cost - dobj <= max(eps, 8e-13 tr Q) established)
        if certify_warning and not np.isfinite(float(res.cost[0, 1])):
            warnings.warn("The solution is not certifiably optimal.")
        return poses
    if status != 0 and certify_warning:  # cvxpnpl.py:517-519
        warnings.warn("The solution is not certifiably optimal.")
    if verbose:
        print(f"cvxpnpl_amd: status={_lib.STATUS_NAMES[status]} iters={int(res.iters[0])} "
              f"cost={float(res.cost[0, 0]):.3e} dobj={float(res.cost[0, 1]):.3e}")
    return [(res.R[0].cpu().numpy(), res.t[0].cpu().numpy())]


class _SingleCtx:
    """Staging of one single-problem call (pnp / pnl / pnpl: the reference's unit of use and of timing, suite.py:75-85): ONE pinned host buffer
    and ONE device buffer for the inputs [pts_2d | pts_3d | line_2d | line_3d | K], one of each for every output, so that a call is one
    H2D copy, the solve, one D2H copy and one synchronisation -- no allocation, no pageable copy, no per-field read-back."""

    N_OUT = 9 + 3 + 2 + 55  # R, t, cost | dobj, Z (doubles); then status, iters, work[2] as int32 in two more doubles

    def __init__(self, device, n_p, n_l):
        self.n_p, self.n_l = n_p, n_l
        # (n_p = -1: the cost seam -- [Q45 | B27] instead of correspondences and K)
        self.sizes = (2 * n_p, 3 * n_p, 4 * n_l, 6 * n_l, 9) if n_p >= 0 else (45, 27, 0, 0, 0)
        self.offs = np.cumsum((0,) + self.sizes)
        n_in = int(self.offs[-1])
        self.h_in = torch.empty(n_in, dtype=torch.float64).pin_memory()
        self.h_np = self.h_in.numpy()
        self.d_in = torch.empty(n_in, dtype=torch.float64, device=device)
        self.d_out = torch.empty(self.N_OUT + 2, dtype=torch.float64, device=device)
        self.h_out = torch.empty(self.N_OUT + 2, dtype=torch.float64).pin_memory()
        self.o_np = self.h_out.numpy()
        self.i_np = self.o_np[self.N_OUT:].view(np.int32)
        base_in, base_out = self.d_in.data_ptr(), self.d_out.data_ptr()
        pin = [C.c_void_p(base_in + 8 * int(o)) if sz else C.c_void_p(0) for o, sz in zip(self.offs[:-1], self.sizes)]
        self.p2, self.p3, self.l2, self.l3, self.K = pin
        o = lambda k: C.c_void_p(base_out + 8 * k)  # noqa: E731
        self.R, self.t, self.cost, self.Z = o(0), o(9), o(12), o(14)
        self.status, self.iters, self.work = o(self.N_OUT), C.c_void_p(base_out + 8 * self.N_OUT + 4), C.c_void_p(base_out + 8 * self.N_OUT + 8)


_single_tls = threading.local()


def _single_fast(p2, l2, p3, l3, Kn, eps, max_iters):
    """One problem through the staging of _SingleCtx; the outputs as a BatchResult of host tensors (batch 1)."""
    _require_gpu()
    L = _lib.lib()
    device = torch.device("cuda", torch.cuda.current_device())
    n_p = 0 if p3 is None else p3.shape[1]
    n_l = 0 if l3 is None else l3.shape[1]
    cache = getattr(_single_tls, "ctx", None)
    if cache is None:
        cache = _single_tls.ctx = {}
    key = (device.index, n_p, n_l)
    ctx = cache.get(key)
    if ctx is None:
        if len(cache) > 64:
            cache.clear()
        ctx = cache[key] = _SingleCtx(device, n_p, n_l)
    h, of = ctx.h_np, ctx.offs
    if n_p:
        h[of[0]:of[1]] = p2.reshape(-1)
        h[of[1]:of[2]] = p3.reshape(-1)
    if n_l:
        h[of[2]:of[3]] = l2.reshape(-1)
        h[of[3]:of[4]] = l3.reshape(-1)
    h[of[4]:of[5]] = Kn.reshape(-1)
    okey = (float(eps), int(max_iters))
    opts = getattr(_single_tls, "opts", {}).get(okey)
    if opts is None:
        if not hasattr(_single_tls, "opts"):
            _single_tls.opts = {}
        opts = _single_tls.opts[okey] = _lib.default_opts(eps=float(eps), max_iters=int(max_iters), res_tol=0.0)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device)
        ctx.d_in.copy_(ctx.h_in, non_blocking=True)
        rc = L.cvxpnpl_solve_batch(1, n_p, ctx.p2, ctx.p3, n_l, ctx.l2, ctx.l3, ctx.K, 0, C.byref(opts), ctx.R, ctx.t, ctx.status, ctx.iters, ctx.cost, ctx.Z,
                                   ctx.work, C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"cvxpnpl_solve_batch failed ({rc}): {_lib.last_error()}")
        ctx.h_out.copy_(ctx.d_out, non_blocking=True)
        stream.synchronize()
    o, i = ctx.o_np, ctx.i_np
    return BatchResult(R=torch.from_numpy(o[0:9].reshape(1, 3, 3).copy()), t=torch.from_numpy(o[9:12].reshape(1, 3).copy()),
                       status=torch.from_numpy(i[0:1].copy()), iters=torch.from_numpy(i[1:2].copy()), cost=torch.from_numpy(o[12:14].reshape(1, 2).copy()),
                       work=torch.from_numpy(i[2:4].reshape(1, 2).copy()), Z=torch.from_numpy(o[14:69].reshape(1, 55).copy()))


def _single_cost_fast(Q45, B27, eps, max_iters, variant):
    """The same staging for the cost seam (solve_relaxation / solve_relaxation_rc: one problem given as packed A^T A and B)."""
    _require_gpu()
    L = _lib.lib()
    device = torch.device("cuda", torch.cuda.current_device())
    cache = getattr(_single_tls, "ctx", None)
    if cache is None:
        cache = _single_tls.ctx = {}
    key = (device.index, -1, 0)
    ctx = cache.get(key)
    if ctx is None:
        ctx = cache[key] = _SingleCtx(device, -1, 0)
    ctx.h_np[0:45] = Q45
    ctx.h_np[45:72] = B27
    if not hasattr(_single_tls, "opts"):
        _single_tls.opts = {}
    okey = (float(eps), int(max_iters), int(variant))
    opts = _single_tls.opts.get(okey)
    if opts is None:
        opts = _single_tls.opts[okey] = _lib.default_opts(eps=float(eps), max_iters=int(max_iters), res_tol=0.0, variant=int(variant))
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device)
        ctx.d_in.copy_(ctx.h_in, non_blocking=True)
        rc = L.cvxpnpl_solve_cost_batch(1, ctx.p2, ctx.p3, C.byref(opts), ctx.R, ctx.t, ctx.status, ctx.iters, ctx.cost, ctx.Z, ctx.work, C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"cvxpnpl_solve_cost_batch failed ({rc}): {_lib.last_error()}")
        ctx.h_out.copy_(ctx.d_out, non_blocking=True)
        stream.synchronize()
    o, i = ctx.o_np, ctx.i_np
    return BatchResult(R=torch.from_numpy(o[0:9].reshape(1, 3, 3).copy()), t=torch.from_numpy(o[9:12].reshape(1, 3).copy()),
                       status=torch.from_numpy(i[0:1].copy()), iters=torch.from_numpy(i[1:2].copy()), cost=torch.from_numpy(o[12:14].reshape(1, 2).copy()),
                       work=torch.from_numpy(i[2:4].reshape(1, 2).copy()), Z=torch.from_numpy(o[14:69].reshape(1, 55).copy()))


def _single(pts_2d, line_2d, pts_3d, line_3d, K, eps, max_iters, verbose) -> List[Tuple[np.ndarray, np.ndarray]]:
    def b(x, tail):
        if x is None:
            return None
        x = np.asarray(x, dtype=np.float64).reshape((-1,) + tail)
        return x[None] if len(x) else None

    p2, p3 = b(pts_2d, (2,)), b(pts_3d, (3,))
    l2, l3 = b(line_2d, (2, 2)), b(line_3d, (2, 3))
    Kn = np.asarray(K, dtype=np.float64)
    # res_tol = 0: the fixed-point-residual exit of the batch entry points is off.  This is NOT "to max_iters like the reference"
    # for every problem (the advisor's finding): a problem still open after opts.rescue_from first-order iterations (32 / 64 /
    # 128 by size) gets its iterate from the interior-point solve of the same SDP -- the SDP optimum to a gap of 1e-10, where the
    # reference + SCS would hand back whatever 2 500 first-order iterations reached -- and a Z that has settled at rank > 1
    # (opts.stall_from = 300) stops as rank > 1.  Certified poses are unaffected; for non-tight problems the poses recovered
    # from Z (recover_multi) are those of the converged Z, not of SCS's iterate.  INTEGRATION.md section 3 says the same.
    n_corr = (0 if p3 is None else p3.shape[1]) + 2 * (0 if l3 is None else l3.shape[1])
    paired = (p3 is None) == (p2 is None) and (l3 is None) == (l2 is None) and (p3 is None or p2.shape[1] == p3.shape[1]) and \
        (l3 is None or l2.shape[1] == l3.shape[1])  # (anything else: the batch entry point raises the error the caller should see)
    if paired and Kn.shape == (3, 3) and (p3 is not None or l3 is not None) and not use_blocked_assembly(n_corr, 1):
        res = _single_fast(p2, l2, p3, l3, Kn, eps, max_iters)  # one H2D, the solve, one D2H (same entry point, same options)
    else:
        res = pnpl_batch(p2, l2, p3, l3, Kn, eps=eps, max_iters=max_iters, want_Z=True, res_tol=0.0)
    Bt = Qt = None
    if int(res.status[0]) == 1:
        Bt, Qt = _translation_map(p2, l2, p3, l3, Kn)
    return _poses_of_single(res, Bt, Qt, verbose)


def solve_relaxation(A: np.ndarray, B: np.ndarray, eps: float = 1e-9, max_iters: int = 2500, verbose: bool = False,
                     variant: int = _lib.VARIANT_FULL) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Drop-in for the reference's private _solve_relaxation(A, B, eps, max_iters, verbose) (cvxpnpl.py:454-520),
    the seam its benchmark harness calls directly (benchmarks/toolkit/methods/pnp.py:4): A (m x 9) with A r = 0,
    B (3 x 9) with t = -B r.  The cost A^T A is formed here (cvxpnpl.py:475) and solved on the GPU."""
    A = np.asarray(A, dtype=np.float64)
    Bn = np.ascontiguousarray(B, dtype=np.float64).reshape(27)
    Q45 = np.ascontiguousarray(pack_cost(A.T @ A))
    res = _single_cost_fast(Q45, Bn, eps, max_iters, variant)  # (= solve_cost_batch(Q45[None], Bn[None], ..., want_Z=True, res_tol=0.0) through one staging buffer each way)
    # the reference's rc variant returns its poses without the certificate check (rc.py:118-131)
    return _poses_of_single(res, Bn, Q45, verbose, certify_warning=(variant == _lib.VARIANT_FULL))


def solve_relaxation_rc(A: np.ndarray, B: np.ndarray, eps: float = 1e-9, max_iters: int = 2500, verbose: bool = False):
    """Drop-in for _solve_relaxation_rc (benchmarks/toolkit/methods/rc.py:67-131): the relaxation without the six
    row-orthonormality equalities."""
    return solve_relaxation(A, B, eps=eps, max_iters=max_iters, verbose=verbose, variant=_lib.VARIANT_RC)


# Correspondence records (points + 2 x lines) from which pnpl_batch assembles with the blocked kernel.  Measured, wall clock of pnpl_batch,
# blocked / in-kernel (profiles/r03/large_n_crossover.jsonl, with the LDS-ring assembly): 384 records 1.27-1.34 (1-256 problems), 1.10 (1 k),
# 0.72 (4 k), 0.79 (16 k), 0.93 (50 k); 768: 1.08-1.11 (1-256) / 0.91 (1 k) / 0.54 (4 k) / 0.58 (16 k) / 0.72 (50 k); 1536: 0.80-0.82 / 0.73 /
# 0.38-0.44; 192 records: 1.0 at best.  In-kernel assembly costs one memory round trip and is parallel over problems only; the blocked
# kernel is parallel over correspondences and pays one more launch (two when a problem is split over several workgroups).
# (Round 2 had 192, from launch-time measurements of 1 000-problem batches with the general lane core.)
LARGE_N = 768
LARGE_N_MANY = 384      # ... in batches of at least
LARGE_N_MANY_BATCH = 2048


def use_blocked_assembly(n_records, batch):
    """the routing rule of pnpl_batch / assemble_batch(blocked=None)"""
    return n_records >= LARGE_N or (n_records >= LARGE_N_MANY and batch >= LARGE_N_MANY_BATCH)


def assemble_batch(pts_2d, line_2d, pts_3d, line_3d, K, device=None, blocked=None):
    """Device-side constraint assembly only (cvxpnpl.py:432-452): returns (B [batch,27], Q [batch,45]),
    the translation map t = -B r and the packed 9x9 cost r^T Q r, as float64 device tensors.  What
    the rank > 1 recovery (recover_multi) needs next to Z."""
    _require_gpu()
    L = _lib.lib()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    p3 = _as_dev(pts_3d, dev, (3,)) if pts_3d is not None else None
    l3 = _as_dev(line_3d, dev, (2, 3)) if line_3d is not None else None
    n_p = p3.shape[-2] if p3 is not None and p3.dim() >= 3 else 0
    n_l = l3.shape[-3] if l3 is not None and l3.dim() >= 4 else 0
    if n_p == 0 and n_l == 0:
        raise ValueError("need at least one point or line correspondence ([B,n,3] points / [B,n,2,3] lines)")
    batch = (p3 if n_p else l3).shape[0]
    p2, l2 = _pair_2d(pts_2d, line_2d, dev, batch, n_p, n_l)
    Kd = _as_dev(K, dev, (3, 3))
    per = int(Kd.dim() == 3)
    if Kd.dim() not in (2, 3) or (per and Kd.shape[0] != batch):  # (a short [m,3,3] would be read out of bounds by the kernels)
        raise ValueError("K must be [3,3] or [batch,3,3]")
    if n_p and n_l and l3.shape[0] != batch:
        raise ValueError(f"points and lines describe different batches ({batch} vs {l3.shape[0]})")
    if blocked is None:
        blocked = use_blocked_assembly(n_p + 2 * n_l, batch)
    p3 = p3.reshape(batch, n_p, 3) if n_p else None
    l3 = l3.reshape(batch, n_l, 2, 3) if n_l else None
    with torch.cuda.device(dev):
        Bt = torch.empty((batch, 27), dtype=torch.float64, device=dev)
        Qt = torch.empty((batch, 45), dtype=torch.float64, device=dev)
        sh = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if blocked:
            # many correspondences per problem: several workgroups per problem at HBM rate (cvxpnpl_assemble_large_batch)
            for lo in range(0, batch, 65535):
                hi = min(batch, lo + 65535)
                nb = L.cvxpnpl_assemble_large_scratch_bytes(hi - lo, n_p, n_l)
                scratch = torch.empty((nb,), dtype=torch.uint8, device=dev)
                sl = lambda x: None if x is None else x[lo:hi]  # noqa: E731
                rc = L.cvxpnpl_assemble_large_batch(hi - lo, n_p, _ptr(sl(p2)), _ptr(sl(p3)), n_l, _ptr(sl(l2)), _ptr(sl(l3)),
                                                    _ptr(Kd[lo:hi] if per else Kd), per, _ptr(Bt[lo:hi]), _ptr(Qt[lo:hi]),
                                                    _ptr(scratch), nb, sh)
                if rc != 0:
                    raise RuntimeError(f"cvxpnpl_assemble_large_batch failed ({rc}): {_lib.last_error()}")
            return Bt, Qt
        rc = L.cvxpnpl_assemble_batch(batch, n_p, _ptr(p2), _ptr(p3 if n_p else None), n_l, _ptr(l2), _ptr(l3 if n_l else None),
                                      _ptr(Kd), per, _ptr(Bt), _ptr(Qt), sh)
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_assemble_batch failed ({rc}): {_lib.last_error()}")
    return Bt, Qt


def score_hypotheses(R, t, K, pts_2d, pts_3d, thresh: float = 2.0, status=None, usable=(0, 2), want_mask: bool = False):
    """Inlier counts of pose hypotheses against one scene (cvxpnpl_score_hypotheses, the HIP scoring kernel).

    R [H,3,3], t [H,3] device tensors (e.g. of a BatchResult); scene pts_2d [M,2], pts_3d [M,3]; K [3,3].
    status [H] int32 (optional): only hypotheses whose status is in `usable` are scored, the others count 0.
    Returns count [H] int32, or (count, mask [H,M] uint8) with want_mask."""
    _require_gpu()
    L = _lib.lib()
    dev = R.device if isinstance(R, torch.Tensor) and R.is_cuda else torch.device("cuda", torch.cuda.current_device())
    Rd = _as_dev(R, dev, (3, 3))
    td = _as_dev(t, dev, (3,))
    Kd = _as_dev(K, dev, (3, 3))
    x = _as_dev(pts_2d, dev, (2,))
    X = _as_dev(pts_3d, dev, (3,))
    if Rd.dim() != 3 or td.shape[0] != Rd.shape[0] or Kd.dim() != 2 or x.dim() != 2 or X.dim() != 2 or x.shape[0] != X.shape[0]:
        raise ValueError("expected R [H,3,3], t [H,3], K [3,3], pts_2d [M,2], pts_3d [M,3]")
    H, M = Rd.shape[0], X.shape[0]
    if H == 0:
        count = torch.empty((0,), dtype=torch.int32, device=dev)
        return (count, torch.empty((0, M), dtype=torch.uint8, device=dev)) if want_mask else count
    st = None
    if status is not None:
        st = torch.as_tensor(status).to(device=dev, dtype=torch.int32).contiguous()
        if st.shape != (H,):
            raise ValueError("status must be [H]")
    um = 0
    for s_ in usable:
        um |= 1 << int(s_)
    with torch.cuda.device(dev):
        count = torch.empty((H,), dtype=torch.int32, device=dev)
        mask = torch.empty((H, M), dtype=torch.uint8, device=dev) if want_mask else None
        rc = L.cvxpnpl_score_hypotheses(H, _ptr(Rd), _ptr(td), _ptr(st), um, _ptr(Kd), M, _ptr(x), _ptr(X), float(thresh),
                                        _ptr(count), _ptr(mask), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_score_hypotheses failed ({rc}): {_lib.last_error()}")
    return (count, mask) if want_mask else count


def select_best(count, R, t, status, K, pts_2d, pts_3d, thresh: float = 2.0):
    """The selection of a RANSAC frame in one launch (cvxpnpl_select_best): arg-max of the inlier counts `count` [H] (score_hypotheses) with the
    lowest index winning a tie, the winner's pose, and its inlier mask over the scene.  Device tensors in, device tensors out, no
    synchronisation: returns R [1,3,3], t [1,3], head [4] int32 = (status, inliers, index, certified hypotheses), mask [1,M] uint8."""
    _require_gpu()
    L = _lib.lib()
    dev = R.device
    H, M = R.shape[0], pts_3d.shape[0]
    if H < 1:
        raise ValueError("select_best needs at least one hypothesis")
    with torch.cuda.device(dev):
        oR = torch.empty((1, 3, 3), dtype=torch.float64, device=dev)
        ot = torch.empty((1, 3), dtype=torch.float64, device=dev)
        head = torch.empty((4,), dtype=torch.int32, device=dev)
        mask = torch.empty((1, M), dtype=torch.uint8, device=dev)
        rc = L.cvxpnpl_select_best(H, _ptr(count), _ptr(R), _ptr(t), _ptr(status), _ptr(K), M, _ptr(pts_2d), _ptr(pts_3d), float(thresh),
                                   _ptr(oR), _ptr(ot), _ptr(head), _ptr(mask), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_select_best failed ({rc}): {_lib.last_error()}")
    return oR, ot, head, mask


def refit_update(fit: "BatchResult", fit_count, K, pts_2d, pts_3d, thresh, R, t, head, mask):
    """cvxpnpl_refit_update: take the refitted pose `fit` (a one-problem BatchResult) -- pose, status, mask and inlier count together, in
    place -- when it is usable and keeps at least head[1] inliers.  One launch, no synchronisation."""
    L = _lib.lib()
    dev = R.device
    with torch.cuda.device(dev):
        rc = L.cvxpnpl_refit_update(_ptr(fit.R), _ptr(fit.t), _ptr(fit.status), _ptr(fit_count), _ptr(K), pts_3d.shape[0], _ptr(pts_2d), _ptr(pts_3d),
                                    float(thresh), _ptr(R), _ptr(t), _ptr(head), _ptr(mask), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_refit_update failed ({rc}): {_lib.last_error()}")


def assemble_subsets(pts_2d, pts_3d, K, mask):
    """Constraint assembly for subsets of ONE scene (cvxpnpl_assemble_subsets): pts_2d [M,2], pts_3d [M,3], mask [B,M] (non-zero = taken, e.g.
    the mask of score_hypotheses) -> (B27 [B,27], Q45 [B,45], count [B] int32) on the device; feed solve_cost_batch.  The refit of a RANSAC
    consensus set without a host round trip for its size."""
    _require_gpu()
    L = _lib.lib()
    dev = mask.device if isinstance(mask, torch.Tensor) and mask.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = _as_dev(pts_2d, dev, (2,)).reshape(-1, 2)
    X = _as_dev(pts_3d, dev, (3,)).reshape(-1, 3)
    Kd = _as_dev(K, dev, (3, 3)).reshape(3, 3)
    M = X.shape[0]
    mk = torch.as_tensor(mask).to(device=dev).ne(0).to(torch.uint8).reshape(-1, M).contiguous()
    Bn = mk.shape[0]
    if x.shape[0] != M:
        raise ValueError(f"{x.shape[0]} 2D points for {M} 3D points")
    with torch.cuda.device(dev):
        Bt = torch.empty((Bn, 27), dtype=torch.float64, device=dev)
        Qt = torch.empty((Bn, 45), dtype=torch.float64, device=dev)
        cnt = torch.empty((Bn,), dtype=torch.int32, device=dev)
        rc = L.cvxpnpl_assemble_subsets(Bn, M, _ptr(x), _ptr(X), _ptr(mk), _ptr(Kd), _ptr(Bt), _ptr(Qt), _ptr(cnt), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_assemble_subsets failed ({rc}): {_lib.last_error()}")
    return Bt, Qt, cnt


def sample_minimal_sets(pts_2d, pts_3d, n_hyp: int, k: int = 4, seed: int = 0, want_idx: bool = False):
    """k distinct correspondences of the scene per hypothesis, drawn uniformly and gathered into the inputs of a minimal solve
    (cvxpnpl_sample_minimal_sets: one HIP launch; counter-based Philox stream, reproducible per (seed, hypothesis)).

    pts_2d [M,2], pts_3d [M,3] device tensors (or anything torch.as_tensor takes).  Returns (p2 [H,k,2], p3 [H,k,3]) or, with want_idx,
    (p2, p3, idx [H,k] int32)."""
    _require_gpu()
    L = _lib.lib()
    dev = pts_3d.device if isinstance(pts_3d, torch.Tensor) and pts_3d.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = _as_dev(pts_2d, dev, (2,))
    X = _as_dev(pts_3d, dev, (3,))
    if x.dim() != 2 or X.dim() != 2 or x.shape[0] != X.shape[0]:
        raise ValueError("expected pts_2d [M,2], pts_3d [M,3]")
    M, H = int(X.shape[0]), int(n_hyp)
    if not (1 <= k <= 8) or M < k or H < 0:
        raise ValueError(f"k must be 1..8 and at most the number of correspondences (k={k}, M={M})")
    with torch.cuda.device(dev):
        p2 = torch.empty((H, k, 2), dtype=torch.float64, device=dev)
        p3 = torch.empty((H, k, 3), dtype=torch.float64, device=dev)
        idx = torch.empty((H, k), dtype=torch.int32, device=dev) if want_idx else None
        rc = L.cvxpnpl_sample_minimal_sets(H, M, _ptr(x), _ptr(X), int(k), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(idx), _ptr(p2), _ptr(p3),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cvxpnpl_sample_minimal_sets failed ({rc}): {_lib.last_error()}")
    return (p2, p3, idx) if want_idx else (p2, p3)


def _translation_map(p2, l2, p3, l3, Kn):
    Bt, Qt = assemble_batch(p2, l2, p3, l3, Kn)
    return Bt[0].cpu().numpy(), Qt[0].cpu().numpy()


def pnp(pts_2d: np.ndarray, pts_3d: np.ndarray, K: np.ndarray, eps: float = 1e-9, max_iters: int = 2500,
        verbose: bool = False) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Compute object poses from point 2D-3D correspondences (drop-in for cvxpnpl.pnp, cvxpnpl.py:523-552).

    pts_2d -- n x 2 pixels; pts_3d -- n x 3 points; K -- 3 x 3 intrinsics; eps -- numerical
    precision of the solver; max_iters -- iteration cap; verbose -- print solver information.
    Returns a list of (R 3x3, t 3) with x_cam = R X + t.
    """
    return _single(pts_2d, None, pts_3d, None, K, eps, max_iters, verbose)


def pnl(line_2d: np.ndarray, line_3d: np.ndarray, K: np.ndarray, eps: float = 1e-9, max_iters: int = 2500,
        verbose: bool = False) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Compute object poses from line 2D-3D correspondences (drop-in for cvxpnpl.pnl, cvxpnpl.py:555-583).

    line_2d -- n x 2 x 2 (line, sampled point, xy); line_3d -- n x 2 x 3 (line, end point, xyz).
    """
    return _single(None, line_2d, None, line_3d, K, eps, max_iters, verbose)


def pnpl(pts_2d: np.ndarray, line_2d: np.ndarray, pts_3d: np.ndarray, line_3d: np.ndarray, K: np.ndarray,
         eps: float = 1e-9, max_iters: int = 2500, verbose: bool = False) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Compute object poses from point and line correspondences (drop-in for cvxpnpl.pnpl, cvxpnpl.py:586-627)."""
    return _single(pts_2d, line_2d, pts_3d, line_3d, K, eps, max_iters, verbose)
