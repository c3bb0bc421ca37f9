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
