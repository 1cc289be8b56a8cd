"""(a) The interior-point path (opts.rescue_from; cvxpnpl_amd/csrc/ipm_wave.h) against the ORACLE, explicitly: the problems a
launch actually sends through it are selected (iters > rescue_from) and compared with the oracle's converged solve of the
reference's SDP -- not with another GPU run.  (b) The multi-rank path of bench.py (process group, cvxpnpl_pack_results, the
asynchronous all-gather on a side stream) in the driver-run suite: a one-rank RCCL group and two gloo ranks sharing the
device, so that the first 8-GPU run is not also the first run of that code.

Tolerance: rotation geodesic <= 1e-6 rad and |t - t_oracle| <= 1e-6 (north-star), wherever the oracle converged to one pose."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import _solve, gpu  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_compare(orc, d, r, idx, n_p):
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.ndim > 2 and len(v) == len(d["pts_3d"]) else v) for k, v in d.items()}
    o = orc.pnpl_batch(sub["pts_2d"], None, sub["pts_3d"], None, d["K"], eps=1e-11, max_iters=400000)
    from cvxpnpl_amd import synth

    st = r["status"][idx]
    single = (o["n_poses"] == 1) & np.isfinite(o["R"][:, 0]).all(axis=(1, 2))
    both = (st == 0) & single
    geo = synth.geodesic(r["R"][idx], o["R"][:, 0])
    dt = np.abs(r["t"][idx] - o["t"][:, 0]).max(axis=1)
    return both, geo, dt, o


def test_rescued_minimal_problems_against_the_oracle(gpu, orc):  # noqa: F811
    """3 000 four-point problems at 2 px: about a fifth is still open after the 32 first-order iterations N = 4 gets and is
    finished through the interior-point solve; >= 128 of exactly those are compared with the oracle."""
    from cvxpnpl_amd import synth

    d = synth.make_pnp(3000, 4, sigma=2.0, seed=4242)
    r = _solve(gpu, d, 4, 0, max_iters=2500)  # defaults: rescue_from = -1 -> 32 for four correspondences
    rescued = np.flatnonzero(r["iters"] > 32)
    assert len(rescued) >= 300, len(rescued)
    # nothing the path touches may come back worse than uncertified: certified or flagged, never NaN
    assert np.isin(r["status"][rescued], (0, 1, 2, 4)).all() and np.isfinite(r["R"][rescued]).all()
    assert (r["status"][rescued] == 0).mean() > 0.95
    idx = rescued[:176]
    both, geo, dt, o = _oracle_compare(orc, d, r, idx, 4)
    assert both.sum() >= 128, both.sum()
    assert geo[both].max() <= 1e-6 and dt[both].max() <= 1e-6, (geo[both].max(), dt[both].max())
    # "certified-then-wrong" cannot happen: a certified pose is the SDP optimum, so wherever the oracle reports one pose it is that pose
    assert not ((r["status"][idx] == 0) & (o["n_poses"] == 1) & (geo > 1e-6)).any()
    # and its certificate is the float64 statement 0 <= cost - dobj <= eps
    c = r["cost"][rescued][r["status"][rescued] == 0]
    assert ((c[:, 0] - c[:, 1]) >= -1e-15).all() and ((c[:, 0] - c[:, 1]) <= 1.0001e-9 + 1e-12 * np.abs(c[:, 0])).all()


def test_rescued_near_planar_problems_against_the_oracle(gpu, orc):  # noqa: F811
    """ten points close to a plane (thickness 2 % of the scene): near two-fold ambiguity, the slow N = 10 workload;
    rescue_from = 40 sends the stragglers through the interior-point solve"""
    from cvxpnpl_amd import synth

    d = synth.make_pnp(2500, 10, sigma=0.0, seed=99)
    d["pts_3d"][:, :, 2] *= 0.02
    d["pts_2d"] = synth.project(d["pts_3d"], d["K"], d["R_gt"], d["t_gt"]) + np.random.RandomState(5).normal(scale=1.0, size=d["pts_2d"].shape)
    r = _solve(gpu, d, 10, 0, max_iters=2500, rescue_from=40)
    rescued = np.flatnonzero(r["iters"] > 40)
    assert len(rescued) >= 40, len(rescued)
    assert np.isfinite(r["R"][rescued]).all() and (r["status"][rescued] == 0).mean() > 0.9
    idx = rescued[:160]
    both, geo, dt, o = _oracle_compare(orc, d, r, idx, 10)
    assert both.sum() >= min(128, int(0.8 * len(idx))), (both.sum(), len(idx))
    assert geo[both].max() <= 1e-6 and dt[both].max() <= 1e-6, (geo[both].max(), dt[both].max())
    # the same launch without the path: identical poses wherever both certify (the path only supplies a better iterate)
    r0 = _solve(gpu, d, 10, 0, max_iters=2500, rescue_from=0)
    c2 = (r["status"] == 0) & (r0["status"] == 0)
    assert synth.geodesic(r["R"][c2], r0["R"][c2]).max() < 1e-7  # (near-planar: the cost is almost flat along the twin direction; measured 8.5e-9)
    assert (r["status"] == 0).sum() >= (r0["status"] == 0).sum()


def test_config5_hypotheses_against_the_oracle(gpu, orc):  # noqa: F811
    """BASELINE config 5 as SURVEY.md 8(d) defines it -- ONE scene of 100 correspondences, 30 % of the 2D points replaced by
    uniform clutter, random 4-subsets -- against the oracle on a sample of the hypotheses, inlier-only and outlier-contaminated subsets
    separately (>= 128 compared in all).  An all-inlier subset is a noisy but consistent minimal problem; a contaminated one has a
    large residual and is often not tight (rank > 1) -- there the statuses are compared, and the poses wherever both sides report
    exactly one."""
    from cvxpnpl_amd import synth

    d = synth.make_ransac(20_000, n_corr=100, outlier_frac=0.3, sigma=0.5, seed=46)
    r = _solve(gpu, d, 4, 0, max_iters=2500)
    clean = d["inlier"][d["idx"]].all(axis=1)
    assert 0.15 < clean.mean() < 0.35  # (0.7^4 = 0.24)
    assert np.isin(r["status"], (0, 1, 2, 3, 4)).all()
    n_cmp = 0
    for sel, n_take, min_cert in ((np.flatnonzero(clean), 112, 0.9), (np.flatnonzero(~clean), 112, 0.5)):
        idx = sel[:n_take]
        both, geo, dt, o = _oracle_compare(orc, d, r, idx, 4)
        st = r["status"][idx]
        assert (st == 0).mean() >= min_cert, (st == 0).mean()
        # a certified pose is the global optimum of the SDP: wherever the oracle converged to one pose it is that pose
        assert both.sum() >= 48, both.sum()
        assert geo[both].max() <= 1e-6 and dt[both].max() <= 1e-6, (geo[both].max(), dt[both].max())
        assert not ((st == 0) & (o["n_poses"] == 1) & np.isfinite(geo) & (geo > 1e-6)).any()
        # the certificate: 0 <= cost - dobj <= eps, in float64
        c = r["cost"][idx][st == 0]
        assert ((c[:, 0] - c[:, 1]) >= -1e-15).all() and ((c[:, 0] - c[:, 1]) <= 1.0001e-9 + 1e-12 * np.abs(c[:, 0])).all()
        n_cmp += int(both.sum())
    assert n_cmp >= 128, n_cmp
    # consensus: an all-inlier hypothesis explains the 70 inliers of the scene (2 px), no contaminated one does
    import torch

    import cvxpnpl_amd as ca

    tt = lambda x: torch.as_tensor(x, device=gpu)  # noqa: E731
    cnt = ca.score_hypotheses(tt(r["R"]), tt(r["t"]), tt(d["K"]), tt(d["scene_2d"]), tt(d["scene_3d"]), 2.0, status=tt(r["status"]), usable=(0, 2)).cpu().numpy()
    n_inl = int(d["inlier"].sum())
    assert cnt.max() >= n_inl - 2 and clean[int(cnt.argmax())]
    assert cnt[~clean].max() < n_inl - 5


def _run_bench(extra, timeout=420):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pmc", "off",
                        "--no-overlap"] + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines  # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_one_rank_rccl_group(gpu):  # noqa: F811
    """python bench.py --force-dist: process group on RCCL (backend nccl), cvxpnpl_pack_results, all_gather_into_tensor on the
    side stream, the gathered records equal the local ones"""
    out = _run_bench(["--force-dist"])
    col = out["config"]["collective"]
    assert col["ranks"] == 1 and "RCCL" in col["backend"], col
    assert col["gather_check"]["all_ranks_ok"] and col["gather_check"]["records"] == 10000, col
    assert out["n_gpus"] == 1 and out["value"] > 1e6 and out["solver"]["certified_frac"] > 0.999


def test_bench_two_ranks_sharing_the_device(gpu):  # noqa: F811
    """python bench.py --gpus 2 --backend gloo: bench.py starts its own two ranks (torch.distributed.run on 127.0.0.1), each
    solves its shard, the packed records are exchanged and checked on both ranks.  (RCCL refuses two ranks on one device:
    this exercises launch, sharding, packing and the asynchronous gather, not xGMI.)"""
    out = _run_bench(["--gpus", "2", "--backend", "gloo"], timeout=600)
    col = out["config"]["collective"]
    assert col["ranks"] == 2 and out["n_gpus"] == 2, col
    assert col["gather_check"]["all_ranks_ok"] and col["gather_check"]["records"] == 20000, col
    assert out["scaling"] == "weak" and out["value"] > 1e4  # (two ranks on one device, records through gloo on the host, 3 steps: a functional check)
    # the line describes what it ran on (round-3 verdict item 5): ranks counted by a collective, every rank's device, the exchange priced
    assert col["ranks_seen"] == 2 and len(col["per_rank"]) == 2 and [x["rank"] for x in col["per_rank"]] == [0, 1], col
    assert all(x["problems_per_step"] == 10000 and x["own_ms_per_step"] > 0 and "device" in x for x in col["per_rank"]), col
    assert col["distinct_devices"] == 1 and "gloo" in col["backend"]
    g = col["gather_ms_per_step"]
    assert g["alone"] > 0 and g["exposed"] >= 0 and g["hidden"] >= 0 and g["bytes_received_per_rank_per_step"] == 2 * 10000 * 13 * 8, g
    assert g["in_flight"] == 2 and col["gather_check"]["held_by"] == "every rank"
    assert "slowest rank" in col["value_is"] and col["handover"]


def test_bench_strong_scaling_with_ragged_shards(gpu):  # noqa: F811
    """--scaling strong --total T: the T problems of a step are split over the ranks (contiguous, balanced: shard_range), the short shard is
    padded in the exchange, `value` counts T problems per step"""
    out = _run_bench(["--gpus", "2", "--backend", "gloo", "--scaling", "strong", "--total", "10001"], timeout=600)
    col = out["config"]["collective"]
    assert out["scaling"] == "strong" and out["n_gpus"] == 2
    assert sorted(x["problems_per_step"] for x in col["per_rank"]) == [5000, 5001], col["per_rank"]
    assert col["gather_check"]["all_ranks_ok"] and col["gather_check"]["records"] == 2 * 5001, col
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 10001) < 1e-6 * 10001


def test_bench_config4_eight_ranks_on_one_device(gpu):  # noqa: F811
    """north-star config 4's shape end to end on the 1-GPU box: `--gpus 8 --scaling strong --total 1000000` = eight ranks (sharing the device,
    records through gloo: RCCL needs a device per rank), 125 000 problems each, one exchange of 8 x 125 000 records per step with two in
    flight; the line must prove all eight ranks took part and price the exchange"""
    out = _run_bench(["--gpus", "8", "--backend", "gloo", "--scaling", "strong", "--total", "1000000", "--workload", "pnp_n10_125k", "--no-transfer"], timeout=1500)
    col = out["config"]["collective"]
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and col["ranks"] == 8 and col["ranks_seen"] == 8, col
    # the rendezvous and a tiny all_reduce / all_gather ran under the watchdog before anything else (cvxpnpl_amd.dist.init_with_preflight)
    pf = col["preflight"]
    assert pf["ranks_seen"] == 8 and pf["timeout_s"] == 60.0 and pf["all_reduce_ms"] > 0 and pf["all_gather_ms"] > 0, pf
    assert all(x["own_ms_per_step"] > 0 for x in col["per_rank"]) and "exposed" in col["gather_ms_per_step"], col
    assert [x["rank"] for x in col["per_rank"]] == list(range(8)) and all(x["problems_per_step"] == 125000 for x in col["per_rank"])
    assert col["gather_check"]["all_ranks_ok"] and col["gather_check"]["records"] == 1000000 and col["gather_check"]["held_by"] == "every rank"
    g = col["gather_ms_per_step"]
    assert g["bytes_received_per_rank_per_step"] == 1000000 * 13 * 8 and g["in_flight"] == 2 and g["alone"] > 0, g
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 1000000) < 1.0
    assert out["dtype"] == "f64" and out["solver"]["certified_frac"] > 0.999


def test_bench_gather_to_root(gpu):  # noqa: F811
    """--collective gather: one consumer -- only rank 0 receives the records"""
    out = _run_bench(["--gpus", "2", "--backend", "gloo", "--collective", "gather"], timeout=600)
    col = out["config"]["collective"]
    assert col["ranks_seen"] == 2 and col["gather_check"]["all_ranks_ok"] and col["gather_check"]["held_by"] == "rank 0", col
    assert "gather to rank 0" in col["exchange"]
    assert col["gather_ms_per_step"]["bytes_received_per_rank_per_step"] == {"rank 0": 2 * 10000 * 13 * 8, "other ranks": 0}


def test_bench_headline_is_reference_precision(gpu):  # noqa: F811
    """the default line is timed with every sweep in float64 (the reference's precision, cvxpnpl.py:475-513); the library's default mode is
    reported beside it, with identical statuses and certified poses to 1e-9 rad"""
    out = _run_bench(["--no-transfer"])
    assert out["dtype"] == "f64" and out["value_all_f64"] == out["value"] and "mixed" in out and out["value_mixed"] == out["mixed"]["value"]
    assert out["mixed"]["status_equal_to_timed_region_frac"] == 1.0 and out["mixed"]["max_rot_diff_vs_timed_region_rad"] < 1e-9
    out = _run_bench(["--no-transfer", "--precision", "mixed"])
    assert out["dtype"].startswith("f64 (f32") and "all_f64" in out and out["value_all_f64"] == out["all_f64"]["value"]


def test_bench_config5_workload(gpu):  # noqa: F811
    """python bench.py --workload ransac_n4_50k (BASELINE config 5 as a driver-runnable line): the solve is `value`, the frame (sample ->
    solve -> score -> arg-max -> refit) and the scoring kernel are beside it"""
    out = _run_bench(["--workload", "ransac_n4_50k", "--no-f64-ab"], timeout=600)
    assert out["config"]["workload"] == "ransac_n4_50k" and out["config"]["problems_per_gpu_per_step"] == 50000
    s_ = out["solver"]
    assert s_["certified_frac"] > 0.5 and 0 <= s_["rank_gt1_frac"] < 0.5 and s_["status_hist"][3] == 0, s_
    f = out["ransac_frame"]
    assert f["frames_per_s"] > 10 and f["score_kernel"]["ms"] > 0 and f["score_kernel"]["scene_correspondences"] == 100, f
    assert f["last_frame"]["n_inliers"] >= f["last_frame"]["true_inliers"] - 2 and f["last_frame"]["rot_err_vs_gt_rad"] < 0.02, f
    assert f["fixed_subsets"]["best_hypothesis_inliers"] >= f["fixed_subsets"]["true_inliers"] - 2, f
    assert out["median_ms_per_step"] > 0 and out["transfer_inclusive"]["records_equal_device_run"], out.get("transfer_inclusive")


def test_stream_wait_fails_closed(gpu):  # noqa: F811
    """cvxpnpl_stream_wait_value (round 6): a wait whose value never arrives gives up after ~0.25 s and then HOLDS its stream until the host
    has acknowledged the give-up -- nothing behind it runs before; cvxpnpl_stream_wait_gave_up reports it (1), clears the word and thereby
    releases the stream.  A wait whose value arrives passes and reports 0.  The explicitly fail-open form lets its stream go on."""
    import ctypes as C
    import time

    import torch

    from cvxpnpl_amd import _lib

    L = _lib.lib()
    prod, cons = torch.cuda.Stream(gpu), torch.cuda.Stream(gpu)
    flag = torch.zeros(2, dtype=torch.int64, device=gpu)
    behind = torch.zeros(1, dtype=torch.int64, device=gpu)
    torch.cuda.synchronize(gpu)
    fp, cs, ps = C.c_void_p(flag.data_ptr()), C.c_void_p(cons.cuda_stream), C.c_void_p(prod.cuda_stream)
    # 1. the value arrives: no give-up, the consumer goes on
    assert L.cvxpnpl_stream_wait_value(fp, 1, cs) == 0
    with torch.cuda.stream(cons):
        behind.add_(1)
    assert L.cvxpnpl_stream_write_value(fp, 1, ps) == 0
    assert L.cvxpnpl_stream_wait_gave_up(fp, 1, cs) == 0
    assert int(behind.item()) == 1 and int(flag[0].item()) == 1
    # 2. the value never arrives: the stream is held beyond the give-up, until acknowledged
    assert L.cvxpnpl_stream_wait_value(fp, 7, cs) == 0
    with torch.cuda.stream(cons):
        behind.add_(1)
    time.sleep(1.0)                                   # (the wait gives up after ~0.25 s)
    assert not cons.query()                           # still held: the kernel behind the wait has not run
    t0 = time.time()
    assert L.cvxpnpl_stream_wait_gave_up(fp, 1, cs) == 1   # told, cleared, released
    cons.synchronize()
    assert time.time() - t0 < 5.0 and int(behind.item()) == 2 and int(flag[1].item()) == 0
    # 3. the explicitly fail-open form: gives up, marks the word, lets its stream go on
    assert L.cvxpnpl_stream_wait_value_bounded(fp, 9, 1 << 12, cs) == 0
    with torch.cuda.stream(cons):
        behind.add_(1)
    cons.synchronize()
    assert int(behind.item()) == 3 and int(flag[1].item()) == 1
    assert L.cvxpnpl_stream_wait_gave_up(fp, 1, cs) == 1 and L.cvxpnpl_stream_wait_gave_up(fp, 1, cs) == 0
