"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, solve their slices with
a CPU stand-in for the HIP solver (the test-only host build of the device algorithm), gather
with cvxpnpl_amd.dist, and must reproduce the single-process result exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _cpu_solver(p2, l2, p3, l3, K, **kw):
    import hostsim

    def n(x):
        return None if x is None else np.asarray(x)

    return hostsim.solve_batch(n(p2), n(p3), n(l2), n(l3), np.asarray(K))


def _worker(rank, world, port, batch, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvxpnpl_amd import dist as cd
    from cvxpnpl_amd import synth

    d = synth.make_pnp(batch, 8, 1.0, seed=5)
    R, t, st = cd.solve_sharded(torch.as_tensor(d["pts_2d"]), None, torch.as_tensor(d["pts_3d"]), None, d["K"], solver=_cpu_solver)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), R=R.numpy(), t=t.numpy(), st=st.numpy())
    if batch % world == 0:  # the overlapped (async) gather bench.py uses, equal shards
        lo, hi = cd.shard_range(batch, rank, world)
        packed = cd.pack_results(R[lo:hi], t[lo:hi], st[lo:hi])
        out, work = cd.gather_results(packed, batch, async_op=True)
        work.wait()
        R2, t2, st2 = cd.unpack_results(out)
        assert torch.equal(R2, R) and torch.equal(t2, t) and torch.equal(st2, st)
    dist.barrier()
    dist.destroy_process_group()


def _cheap_solver(p2, l2, p3, l3, K, **kw):
    """a closed-form stand-in (a function of each problem's own inputs, so that shards can be compared with the whole): the large
    ragged test is about the sharding and the exchange, not about the solve"""
    p2, p3 = np.asarray(p2), np.asarray(p3)
    n = p3.shape[0]
    R = np.zeros((n, 3, 3))
    R[:, 0, :] = p3[:, 0, :]
    R[:, 1, :2] = p2[:, 0, :]
    R[:, 2, 2] = p3[:, 0, :].sum(axis=1)
    t = p3[:, 0, :] * 2.0 - 1.0
    status = (np.floor(np.abs(p3[:, 0, 0]) * 1e4).astype(np.int64) % 5).astype(np.int32)
    return {"R": R, "t": t, "status": status}


def _big_inputs(batch):
    rs = np.random.RandomState(11)
    return rs.random_sample((batch, 1, 2)), rs.random_sample((batch, 1, 3)), np.eye(3)


def _worker_big(rank, world, port, batch, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvxpnpl_amd import dist as cd

    p2, p3, K = _big_inputs(batch)
    R, t, st = cd.solve_sharded(torch.as_tensor(p2), None, torch.as_tensor(p3), None, K, solver=_cheap_solver)
    lo, hi = cd.shard_range(batch, rank, world)
    # every rank holds the WHOLE result; each writes a digest of it and its own span
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), lo=lo, hi=hi, n=R.shape[0], sR=R.numpy().sum(axis=(1, 2))[::997], st=st.numpy()[::997],
             t_last=t.numpy()[-1], R_first=R.numpy()[0], cs=np.array([R.numpy().sum(), t.numpy().sum(), st.numpy().astype(np.int64).sum()]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gloo_sharding_of_a_ragged_million(tmp_path):
    """config 4's shape on CPU: 1 000 003 problems over EIGHT ranks (ragged by one: three shards of 125 001, five of 125 000), the padded all-gather of cvxpnpl_amd.dist.gather_results, every rank ends with the whole result, identical to the
    single-process one.  (A closed-form stand-in for the solve: this is the sharding and the exchange at size.)"""
    batch, world = 1_000_003, 8
    mp.spawn(_worker_big, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    p2, p3, K = _big_inputs(batch)
    ref = _cheap_solver(p2, None, p3, None, K)
    cs = np.array([ref["R"].sum(), ref["t"].sum(), ref["status"].astype(np.int64).sum()])
    spans = []
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert int(got["n"]) == batch
        assert np.array_equal(got["sR"], ref["R"].sum(axis=(1, 2))[::997]) and np.array_equal(got["st"], ref["status"][::997])
        assert np.array_equal(got["t_last"], ref["t"][-1]) and np.array_equal(got["R_first"], ref["R"][0])
        assert np.allclose(got["cs"], cs, rtol=1e-12, atol=0) and got["cs"][2] == cs[2]
        spans.append((int(got["lo"]), int(got["hi"])))
    assert spans[0][0] == 0 and spans[-1][1] == batch and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert sorted(b - a for a, b in spans) == [125_000] * 5 + [125_001] * 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("batch", [64, 37])  # even and ragged shards
def test_two_rank_gloo_sharding_equals_single_process(tmp_path, batch):
    import hostsim
    from cvxpnpl_amd import synth

    hostsim.build()
    mp.spawn(_worker, args=(2, _free_port(), batch, str(tmp_path)), nprocs=2, join=True)
    d = synth.make_pnp(batch, 8, 1.0, seed=5)
    ref = hostsim.solve_batch(d["pts_2d"], d["pts_3d"], None, None, d["K"])
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(got["R"], ref["R"]) and np.array_equal(got["t"], ref["t"])
        assert np.array_equal(got["st"], ref["status"])


def test_shard_range_partitions_the_batch():
    from cvxpnpl_amd.dist import shard_range

    for batch in (0, 1, 7, 8, 1_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_shards(rank, world, port, sizes, out_dir):
    """every rank brings ITS OWN slice only (what a per-GPU loader produces): uneven shard sizes that do not follow shard_range"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvxpnpl_amd import dist as cd

    batch = sum(sizes)
    p2, p3, K = _big_inputs(batch)
    lo = sum(sizes[:rank])
    hi = lo + sizes[rank]
    mine = (torch.as_tensor(p2[lo:hi]), None, torch.as_tensor(p3[lo:hi]), None, K)
    R, t, st = cd.solve_sharded(*mine, solver=_cheap_solver, inputs_are_shards=True)
    root = cd.solve_sharded(*mine, solver=_cheap_solver, inputs_are_shards=True, to_root=True)
    assert (root is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(root[0], R) and torch.equal(root[1], t) and torch.equal(root[2], st)
    # gather_to_root with equal (padded) slices, overlapped form
    n = max(sizes)
    pk = torch.zeros((n, cd.PACK), dtype=torch.float64)
    pk[: sizes[rank]] = cd.pack_results(R[lo:hi], t[lo:hi], st[lo:hi])
    out, work = cd.gather_to_root(pk, n * world, async_op=True)
    work.wait()
    if rank == 0:
        for r in range(world):
            a = sum(sizes[:r])
            assert torch.equal(out[r * n:r * n + sizes[r]], cd.pack_results(R[a:a + sizes[r]], t[a:a + sizes[r]], st[a:a + sizes[r]]))
    else:
        assert out is None
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), R=R.numpy(), t=t.numpy(), st=st.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_per_rank_input_shards_and_gather_to_root(tmp_path):
    """solve_sharded(inputs_are_shards=True): ranks hold only their own problems (sizes 5, 0 is not allowed by the solver stand-in, so 5, 9, 2), every
    rank ends with the whole result in rank order; to_root=True leaves it on rank 0 only; gather_to_root delivers every slice"""
    sizes = [5, 9, 2]
    mp.spawn(_worker_shards, args=(3, _free_port(), sizes, str(tmp_path)), nprocs=3, join=True)
    p2, p3, K = _big_inputs(sum(sizes))
    ref = _cheap_solver(p2, None, p3, None, K)
    for r in range(3):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(got["R"], ref["R"]) and np.array_equal(got["t"], ref["t"]) and np.array_equal(got["st"], ref["status"])


_PREFLIGHT_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, {root!r})
rank, world, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[4]
if mode == "never_starts" and rank == 1:
    sys.exit(0)                      # a rank that died before the rendezvous
from cvxpnpl_amd import dist as cd
if mode == "dies_after_rendezvous" and rank == 1:
    import datetime
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    os._exit(0)                      # ... or right after it
info = cd.init_with_preflight("gloo", rank, world, timeout_s=float(sys.argv[5]))
print("PREFLIGHT_OK", info["ranks_seen"], sorted(info))
"""


def _run_preflight(mode, timeout_s, world=2):
    import subprocess
    import time

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = _PREFLIGHT_SCRIPT.format(root=os.path.dirname(HERE))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world), mode, str(port), str(timeout_s)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=120) for p in procs]
    return [p.returncode for p in procs], outs, time.time() - t0


def test_preflight_passes_and_reports_its_stages():
    rcs, outs, _ = _run_preflight("ok", 60)
    assert rcs == [0, 0], outs
    for so, _ in outs:
        assert "PREFLIGHT_OK 2" in so and "all_gather_ms" in so and "all_reduce_ms" in so and "init_ms" in so


@pytest.mark.parametrize("mode", ["never_starts", "dies_after_rendezvous"])
def test_a_dead_rank_gives_a_diagnosis_not_a_hang(mode):
    """first contact with a multi-GPU node must not be a debugging session: a rank whose peer is gone says so in one line within the
    preflight's time limit (here 6 s; bench.py: 60 s) and exits non-zero"""
    rcs, outs, took = _run_preflight(mode, 6)
    assert rcs[0] != 0 and took < 45, (rcs, took)
    err = outs[0][1]
    assert "preflight" in err and "rank 0 of 2" in err and "backend gloo" in err and "127.0.0.1" in err, err[-600:]
    assert "PREFLIGHT_OK" not in outs[0][0]
