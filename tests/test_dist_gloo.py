"""N>1 path: batch sharding + final all-gather, exercised with world_size=2 on gloo (CPU).
The per-shard compute is injected (the numpy oracle stands in for the HIP call) so that the
sharding, padding and gather logic of c3_amd/dist.py is what is tested."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from c3_amd import dist as c3dist, workloads
from oracle import c3_oracle as o


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_compute(h0, hks, signals, dt, fr_phase=None, **kw):
    U = o.propagate_batch(np.asarray(h0), np.asarray(hks), np.asarray(signals), dt, fr_phase=None if fr_phase is None else np.asarray(fr_phase))
    return {"U": torch.as_tensor(U)}


def _worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = workloads.make_workload(1, B=B, N=24)
        r = c3dist.propagate_batch_sharded(wl.h0, wl.hks, wl.signals, wl.dt, compute=_cpu_compute, fr_phase=wl.fr_phase)
        lo, hi = r["bounds"]
        np.save(os.path.join(out_dir, f"U_rank{rank}.npy"), r["U"].numpy())
        np.save(os.path.join(out_dir, f"b_rank{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_sharded_batch_matches_single_process(tmp_path, B):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    wl = workloads.make_workload(1, B=B, N=24)
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, fr_phase=wl.fr_phase)
    spans = []
    for r in range(world):
        U = np.load(tmp_path / f"U_rank{r}.npy")
        assert U.shape == ref.shape
        assert np.abs(U - ref).max() < 1e-12  # every rank holds the full, ordered result
        spans.append(tuple(np.load(tmp_path / f"b_rank{r}.npy")))
    assert spans == [c3dist.shard_bounds(B, world, r) for r in range(world)]


def _robust_worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(9)
        goals = rng.uniform(size=B)
        grads = rng.normal(size=(B, 2, 3))
        r = c3dist.robust_goal_sharded(lambda lo, hi: (goals[lo:hi], grads[lo:hi]), B)
        np.save(os.path.join(out_dir, f"g_rank{rank}.npy"), np.concatenate([[float(r["goal"]), float(r["goal_std"])], r["grad"].numpy().ravel()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [1, 4, 7])
def test_robust_goal_all_reduce(tmp_path, B):
    """mean goal / gradient / std over instances sharded on two ranks (uneven and empty shards included)"""
    world = 2
    mp.spawn(_robust_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(9)
    goals = rng.uniform(size=B)
    grads = rng.normal(size=(B, 2, 3))
    want = np.concatenate([[goals.mean(), goals.std()], grads.mean(axis=0).ravel()])
    for r in range(world):
        got = np.load(tmp_path / f"g_rank{r}.npy")
        assert np.abs(got - want).max() < 1e-12


# --------------------------------------------------------------------------
# bench.py's multi-GPU schedule (plan_batch + SlabRing: step / flush / drain) with an injected compute
# --------------------------------------------------------------------------


def _ring_worker(rank, world, port, B_glob, G, steps, out_dir, overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        cfg = dict(B=B_glob, gpus=world)
        Bg, lo, hi, b_pad = bench.plan_batch(cfg, "strong", None, world, rank)
        assert (lo, hi) == c3dist.shard_bounds(B_glob, world, rank) and Bg == B_glob and b_pad == c3dist.max_shard(B_glob, world)
        B = hi - lo
        ring = c3dist.SlabRing(B, b_pad, (2, 2), G, use_dist=True, overlap=overlap)
        ring.warm({min(G, steps), steps % G})
        calls = [0]

        def compute(out):
            # sample b of step k on this rank: a recognisable value in every element
            k = calls[0]
            calls[0] += 1
            for b in range(B):
                out[b] = complex(1000 * (lo + b) + k, -k)

        for _ in range(steps):
            ring.step(compute)
        ring.drain()
        assert ring.counter == 0 and ring.pending == 0
        # collectives: the warm-up sizes + ceil(steps / G)
        g_last = steps % G or G
        assert ring.last_flushed == g_last
        got = np.stack([np.stack([ring.gathered_slab(r, i).numpy() for i in range(g_last)]) for r in range(world)])
        np.save(os.path.join(out_dir, f"ring_rank{rank}.npy"), got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B_glob,G,steps,overlap", [(5, 3, 7, False), (4, 2, 4, False), (5, 1, 6, True), (5, 3, 7, True), (4, 2, 4, True)])
def test_bench_slab_ring_two_ranks(tmp_path, B_glob, G, steps, overlap):
    """(overlap: the collective of a full bank of slabs is asynchronous and the next steps compute into the second bank; every
    slab still arrives, the last collective's slabs are readable after drain())"""
    world = 2
    port = _free_port()
    mp.spawn(_ring_worker, args=(world, port, B_glob, G, steps, str(tmp_path), overlap), nprocs=world, join=True)
    g_last = steps % G or G
    first = steps - g_last  # step index of slab 0 of the last collective
    b_pad = c3dist.max_shard(B_glob, world)
    outs = [np.load(tmp_path / f"ring_rank{r}.npy") for r in range(world)]
    assert np.array_equal(outs[0], outs[1])  # every rank holds every rank's slabs
    for r in range(world):
        lo, hi = c3dist.shard_bounds(B_glob, world, r)
        for i in range(g_last):
            for b in range(hi - lo):
                assert np.all(outs[0][r, i, b] == complex(1000 * (lo + b) + first + i, -(first + i)))
            assert outs[0].shape[2] == b_pad


def test_bench_plan_batch_defaults():
    import bench

    C = workloads.CONFIGS
    # weak: BASELINE's batch divided by the GPUs it is quoted on; strong: the whole batch, sharded
    assert bench.plan_batch(C[2], "weak", None, 1, 0) == (256, 0, 256, 256)
    assert bench.plan_batch(C[4], "weak", None, 1, 0) == (512, 0, 512, 512)
    assert bench.plan_batch(C[3], "weak", None, 8, 3) == (4096, 3 * 512, 4 * 512, 512)
    assert bench.plan_batch(C[5], "weak", None, 2, 1) == (2048, 1024, 2048, 1024)
    assert bench.plan_batch(C[3], "strong", None, 8, 7) == (4096, 7 * 512, 4096, 512)
    assert bench.plan_batch(C[5], "strong", 10, 4, 3) == (10, 8, 10, 3)
    assert bench.plan_batch(C[2], "weak", 64, 4, 2) == (256, 128, 192, 64)


# --------------------------------------------------------------------------
# `python bench.py --gpus N` launches its own ranks.  C3P_BENCH_STANDIN=1 swaps the HIP propagator for a
# labelled stand-in so that the launcher, plan_batch, every exchange schedule and the gathered check run here on gloo.
# --------------------------------------------------------------------------


def _run_bench(args, extra_env=None, launcher=None):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, C3P_BENCH_STANDIN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable] + (launcher or []) + [os.path.join(root, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    return out, (json.loads(out.stdout) if out.returncode == 0 and out.stdout.strip() else None)


@pytest.mark.parametrize("scaling,batch", [("strong", 5), ("weak", 3)])
def test_bench_self_launch_two_ranks(scaling, batch):
    out, d = _run_bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", str(batch), "--scaling", scaling, "--ramp-ms", "0"])
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(out.stdout.splitlines()) == 1  # ONE JSON line, from rank 0
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == scaling
    assert d["config"]["global_batch"] == (5 if scaling == "strong" else 6)
    la = d["launch"]
    assert la["gloo_world_size"] == 2 and "self-launch" in la["launcher"] and la["oversubscribed"] is False
    assert la["ms_per_step_fastest_rank"] <= la["ms_per_step_slowest_rank"] == d["ms_per_step"]
    assert abs(d["value"] - d["config"]["global_batch"] * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    # --check is the default at N > 1: own shard, own slab as received, one sample of the other rank's slab per rank
    assert d["max_fro_err_vs_oracle"] == 0.0 and d["gathered_samples_of_other_ranks_checked"] == 2
    # both forms of the one-gather-per-step schedule were calibrated; the one not chosen is timed beside the headline
    cal = la["gather_overlap"]
    assert cal["mode"] == "auto" and set(cal["calibration_ms_per_step"]) == {"in_stream_order", "overlapped"}
    other = "all_gather_every_step" if cal["chosen"] == "overlapped" else "all_gather_every_step_overlapped"
    assert set(d["other_exchange_schedules"]) >= {"all_gather_every_32_steps", other}
    assert "STAND-IN" in d["metric"]  # a stand-in line can never pass for a measurement


def test_bench_under_torchrun_two_ranks():
    """the driver's launch form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2"""
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    out, d = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--ramp-ms", "0", "--no-alt-schedules"], launcher=launcher)
    assert out.returncode == 0, out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["launch"]["gloo_world_size"] == 2 and "external" in d["launch"]["launcher"]
    assert d["max_fro_err_vs_oracle"] == 0.0


def test_bench_self_launch_propagates_a_failing_rank():
    out, d = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--config", "99"])
    assert out.returncode != 0 and d is None
