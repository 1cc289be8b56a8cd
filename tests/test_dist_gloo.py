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
