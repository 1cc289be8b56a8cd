"""Multi-GPU: shard the batch across the ranks of one node, gather the poses over RCCL.

The path partitions trivially -- problems are independent, the only shared data are
read-only constants -- so every rank solves a contiguous slice with no data-path
collective; the one exchange step is the gather of the results (north-star config 4:
"1M PnP problems sharded across 8 MI355X, RCCL gather over xGMI"): 13 doubles per pose
(R, t, status).  One process per GPU, `torch.distributed` backend "nccl" (= RCCL on ROCm);
the same code runs on "gloo" for the CPU tests.
"""
import datetime
import os
import sys
import threading
import time
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

PACK = 13  # R (9) + t (3) + status (1)


class _Watchdog:
    """A stage that does not finish in `timeout_s` ends the PROCESS with a one-line diagnosis on stderr (exit code 3).  A hung rendezvous or
    a hung first collective sits inside C++ (RCCL / gloo) where no Python exception can reach it; the first contact with an 8-GPU node
    must produce a sentence, not a hang (round-5 verdict, item 6)."""

    def __init__(self, timeout_s, describe):
        self.timeout_s, self.describe, self.stage, self.t0 = float(timeout_s), describe, "start", time.time()
        self._timer = threading.Timer(self.timeout_s, self._fire)
        self._timer.daemon = True

    def __enter__(self):
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False

    def _fire(self):
        sys.stderr.write(f"cvxpnpl_amd.dist preflight: NO ANSWER after {self.timeout_s:.0f} s in stage '{self.stage}' -- {self.describe()} -- "
                         "a rank died or never started, the rendezvous address is wrong, or the ranks cannot reach each other's GPUs "
                         "(RCCL: HSA_ENABLE_IPC_MODE_LEGACY=0 must be set; NCCL_DEBUG=INFO shows the transport).\n")
        sys.stderr.flush()
        os._exit(3)


def init_with_preflight(backend: str, rank: int, world: int, device=None, timeout_s: float = 60.0) -> dict:
    """init_process_group + a tiny all_reduce and all_gather_into_tensor (all_gather on gloo), each under a watchdog: returns
    {"backend", "ranks_seen", "init_ms", "all_reduce_ms", "all_gather_ms", "timeout_s"} or ends the process with a one-line diagnosis
    (exit code 3) after `timeout_s` -- also when a peer has died, which the collectives' own error paths turn into an exception
    first where they can (re-raised as RuntimeError with the same one-line description)."""

    def describe():
        return (f"rank {rank} of {world}, backend {backend}, rendezvous {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}, "
                f"device {device}, pid {os.getpid()}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")

    info = {"backend": backend, "timeout_s": float(timeout_s)}
    with _Watchdog(timeout_s, describe) as wd:
        try:
            wd.stage = "init_process_group (rendezvous)"
            t0 = time.perf_counter()
            kw = {"device_id": device} if backend == "nccl" and device is not None else {}
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
            info["init_ms"] = 1e3 * (time.perf_counter() - t0)
            dev = device if backend == "nccl" else torch.device("cpu")
            wd.stage = "all_reduce of one double per rank"
            t0 = time.perf_counter()
            ones = torch.ones(1, dtype=torch.float64, device=dev)
            dist.all_reduce(ones)
            seen = int(ones.item())  # (.item() synchronises: the collective has really run)
            info["all_reduce_ms"] = 1e3 * (time.perf_counter() - t0)
            info["ranks_seen"] = seen
            wd.stage = f"all_gather of {PACK} doubles per rank"
            t0 = time.perf_counter()
            mine = torch.full((1, PACK), float(rank), dtype=torch.float64, device=dev)
            full = torch.empty((world, PACK), dtype=torch.float64, device=dev)
            if backend == "nccl":
                dist.all_gather_into_tensor(full, mine)
            else:
                dist.all_gather(list(full.split(1)), mine)
            got = full[:, 0].cpu().tolist()
            info["all_gather_ms"] = 1e3 * (time.perf_counter() - t0)
        except Exception as e:  # a dead peer, a refused connection, a timeout the backend noticed itself
            raise RuntimeError(f"cvxpnpl_amd.dist preflight failed in stage '{wd.stage}' ({type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}) -- {describe()}") from e
    if seen != world or got != [float(r) for r in range(world)]:
        raise RuntimeError(f"cvxpnpl_amd.dist preflight: the collectives ran but returned ranks_seen = {seen}, slices {got} -- {describe()}")
    return info


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `batch` problems owned by `rank`."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(R: torch.Tensor, t: torch.Tensor, status: torch.Tensor) -> torch.Tensor:
    """[n, 13] float64 records (R row-major, t, status) of one shard.  Device tensors: one HIP launch
    (cvxpnpl_pack_results) on the current stream; host tensors (the gloo tests of the sharding logic): torch ops."""
    n = R.shape[0]
    out = torch.empty((n, PACK), dtype=torch.float64, device=R.device)
    if R.is_cuda and n > 0:
        import ctypes as C

        from . import _lib

        Rc, tc, sc = R.contiguous(), t.contiguous(), status.to(torch.int32).contiguous()
        with torch.cuda.device(R.device):
            rc = _lib.lib().cvxpnpl_pack_results(n, C.c_void_p(Rc.data_ptr()), C.c_void_p(tc.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(R.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"cvxpnpl_pack_results failed ({rc}): {_lib.last_error()}")
        return out
    if n == 0:
        return out
    out[:, :9] = R.reshape(n, 9)
    out[:, 9:12] = t
    out[:, 12] = status.to(torch.float64)
    return out


def unpack_results(packed: torch.Tensor):
    n = packed.shape[0]
    return packed[:, :9].reshape(n, 3, 3), packed[:, 9:12], packed[:, 12].to(torch.int32)


def gather_results(packed_local: torch.Tensor, batch: int, group=None, out: Optional[torch.Tensor] = None,
                   async_op: bool = False):
    """All-gather the per-rank [n_r, 13] slices into the full [batch, 13] on every rank.

    Equal slices use one all_gather_into_tensor (a single RCCL collective; 13 MB per rank
    for config 4 -- a direct exchange over the 7 xGMI links, far below a millisecond);
    ragged slices are padded to the largest one.  With async_op=True (equal slices only) the
    collective is only enqueued and (out, work) is returned: the caller overlaps it with the next
    batch's solve and calls work.wait() before reading `out` (keep `packed_local` alive until then).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
    assert packed_local.shape[0] == sizes[rank], (packed_local.shape, sizes, rank)
    nmax = max(sizes)
    dev = packed_local.device
    if out is None:
        out = torch.empty((batch, PACK), dtype=torch.float64, device=dev)
    if all(s == nmax for s in sizes):
        if dist.get_backend(group) == "nccl":
            work = dist.all_gather_into_tensor(out, packed_local.contiguous(), group=group, async_op=async_op)
        else:
            chunks = list(out.split(nmax))
            work = dist.all_gather(chunks, packed_local.contiguous(), group=group, async_op=async_op)
        return (out, work) if async_op else out
    assert not async_op, "async gather needs equal shards"
    buf = torch.zeros((nmax, PACK), dtype=torch.float64, device=dev)
    buf[: sizes[rank]] = packed_local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    lo = 0
    for r in range(world):
        out[lo:lo + sizes[r]] = parts[r][: sizes[r]]
        lo += sizes[r]
    return out


def gather_to_root(packed_local: torch.Tensor, batch: int, group=None, out: Optional[torch.Tensor] = None, async_op: bool = False,
                   root: int = 0):
    """The exchange when ONE rank consumes the poses: `dist.gather` of equal [n, 13] slices to `root` (every other rank only sends:
    1/N of the all-gather's bytes on its links, nothing received).  Returns (out, work): out is the [batch, 13] tensor on root, None
    elsewhere; with async_op the caller waits on `work` before reading it.  Ragged shards: pad to the largest one first (as bench.py does)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = packed_local.shape[0]
    assert n * world == batch, "gather_to_root needs equal (padded) shards"
    chunks = None
    if rank == root:
        if out is None:
            out = torch.empty((batch, PACK), dtype=torch.float64, device=packed_local.device)
        chunks = list(out.split(n))
    work = dist.gather(packed_local.contiguous(), chunks, dst=dist.get_global_rank(group, root) if group is not None else root, group=group,
                       async_op=async_op)
    return (out if rank == root else None), work


def solve_sharded(pts_2d, line_2d, pts_3d, line_3d, K, group=None, solver: Optional[Callable] = None, inputs_are_shards: bool = False,
                  to_root: bool = False, **kw):
    """Solve a batch over the ranks of `group` and exchange the poses; returns (R [B,3,3], t [B,3], status [B]) of the WHOLE batch on
    every rank (on rank 0 only, None elsewhere, with to_root).

    inputs_are_shards=False: every rank holds (or can index) the full inputs and solves its own contiguous slice (shard_range).
    inputs_are_shards=True: every rank passes ITS OWN slice only -- what a data loader per GPU produces; nothing but results crosses a
    link.  Shard sizes are agreed with one small all_gather of the local counts and may be ragged; the result order is rank order.
    `solver` defaults to cvxpnpl_amd.pnpl_batch (HIP); tests inject a CPU stand-in to exercise the sharding and the collective on gloo."""
    if solver is None:
        from .api import pnpl_batch as solver
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ref = pts_3d if pts_3d is not None else line_3d
    if inputs_are_shards:
        res = solver(pts_2d, line_2d, pts_3d, line_3d, K, **kw)
        n_local = int(ref.shape[0])
    else:
        batch = ref.shape[0]
        lo, hi = shard_range(batch, rank, world)

        def sl(x):
            return None if x is None else x[lo:hi]

        Kl = K[lo:hi] if getattr(K, "ndim", 2) == 3 else K
        res = solver(sl(pts_2d), sl(line_2d), sl(pts_3d), sl(line_3d), Kl, **kw)
        n_local = hi - lo
    packed = pack_results(torch.as_tensor(res["R"]), torch.as_tensor(res["t"]), torch.as_tensor(res["status"]))
    if not inputs_are_shards and not to_root:
        return unpack_results(gather_results(packed, batch, group=group))
    # shard sizes as the ranks report them (they need not follow shard_range), padded to the largest: one collective of equal slices
    counts = torch.zeros(world, dtype=torch.int64, device=packed.device)
    counts[rank] = n_local
    dist.all_reduce(counts, group=group)
    sizes = [int(c) for c in counts.tolist()]
    nmax = max(sizes)
    buf = packed
    if n_local != nmax:
        buf = torch.zeros((nmax, PACK), dtype=torch.float64, device=packed.device)
        buf[:n_local] = packed
    if to_root:
        full_p, _ = gather_to_root(buf, nmax * world, group=group)
        if full_p is None:
            return None
    else:
        full_p = torch.empty((nmax * world, PACK), dtype=torch.float64, device=packed.device)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(full_p, buf.contiguous(), group=group)
        else:
            dist.all_gather(list(full_p.split(nmax)), buf.contiguous(), group=group)
    if all(sz == nmax for sz in sizes):
        return unpack_results(full_p)
    return unpack_results(torch.cat([full_p[r * nmax:r * nmax + sizes[r]] for r in range(world)]))
