"""Multi-GPU sharding of the sample axis.

Each parameter sample is an independent chain, so B samples are partitioned
contiguously over the ranks of a `torch.distributed` group (one process per GPU,
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  The shared
operators (h0, hks, col_ops; a few KB) are simply present on every rank.  The ONLY
data-path collective is the final all-gather of the per-rank U slabs
(B/G x Dm x Dm complex128: 0.3 MB per GPU at cfg2, 6 MB at cfg3) -- no reduction,
no exchange inside a chain.  The reference has no distributed code at all
(SURVEY.md 2); this is the build's batch-axis extension.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of B samples for `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def max_shard(B: int, world: int) -> int:
    return (B + world - 1) // world


def gather_slabs(U_local, B: int, group=None):
    """All-gather per-rank result slabs [b_local, ...] into [B, ...] on every rank.

    Shards may be uneven: slabs are padded to the largest shard for the collective
    (all_gather_into_tensor needs equal sizes) and trimmed afterwards.  complex128 is
    moved as float64 pairs, which both RCCL and gloo support.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(B, world, rank)
    if int(U_local.shape[0]) != hi - lo:
        raise ValueError(f"rank {rank} holds {int(U_local.shape[0])} samples, expected {hi - lo}")
    m = max_shard(B, world)
    tail = tuple(U_local.shape[1:])
    is_c = U_local.is_complex()
    loc = torch.view_as_real(U_local.contiguous()) if is_c else U_local.contiguous()
    if hi - lo < m:
        pad = torch.zeros((m - (hi - lo),) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
        loc = torch.cat([loc, pad], dim=0)
    out = torch.empty((world * m,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
    dist.all_gather_into_tensor(out, loc, group=group)
    pieces = []
    for r in range(world):
        l, h = shard_bounds(B, world, r)
        pieces.append(out[r * m : r * m + (h - l)])
    full = torch.cat(pieces, dim=0)
    if is_c:
        full = torch.view_as_complex(full)
    return full.reshape((B,) + tail)


def propagate_batch_sharded(
    h0,
    hks,
    signals,
    dt: float,
    *,
    compute: Optional[Callable] = None,
    group=None,
    gather: bool = True,
    **kwargs,
):
    """Propagate the rank's shard of `signals` [B,K,N] and (optionally) gather all U.

    `signals` is the GLOBAL batch (every rank holds or can build it; only the local
    shard is touched).  `compute(h0, hks, signals_local, dt, **kwargs) -> {"U": ...}`
    defaults to the HIP path `c3_amd.propagation.propagate_batch`; tests inject a CPU
    stand-in to exercise the sharding/gather logic under gloo.
    Per-sample extras (`fr_phase`) given for the global batch are sliced to the shard.
    """
    import torch
    import torch.distributed as dist

    if compute is None:
        from .propagation import propagate_batch as compute
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = int(signals.shape[0])
    lo, hi = shard_bounds(B, world, rank)
    local_kwargs = dict(kwargs)
    if local_kwargs.get("fr_phase") is not None:
        local_kwargs["fr_phase"] = local_kwargs["fr_phase"][lo:hi]
    if hasattr(h0, "ndim") and h0.ndim == 3 and hks is not None and int(h0.shape[0]) == B:
        h0 = h0[lo:hi]
    if hks is not None and hasattr(hks, "ndim") and hks.ndim == 4:
        hks = hks[lo:hi]
    res = compute(h0, hks, signals[lo:hi], dt, **local_kwargs)
    U_local = res["U"]
    if not torch.is_tensor(U_local):
        U_local = torch.as_tensor(np.asarray(U_local))
    if not gather or world == 1:
        return {"U": U_local, "bounds": (lo, hi)}
    return {"U": gather_slabs(U_local, B, group), "bounds": (lo, hi)}


def robust_goal_sharded(goal_and_grad: Callable, B: int, *, group=None):
    """Mean goal and mean gradient over B noise instances sharded across the ranks -- the multi-GPU form of
    `OptimalControlRobust.goal_run_with_grad` (optimizers/optimalcontrol_robust.py:49-70), whose serial loop
    averages goals and gradients.  `goal_and_grad(lo, hi)` evaluates the rank's instances [lo, hi) and returns
    `(goals [hi-lo], grads [hi-lo, ...])` (on the GPU box: `optimal_control.goal_run_with_grad` on the shard).
    Here the path HAS an exchange step: one all-reduce(sum) of the B-weighted partial sums (a few hundred
    bytes over RCCL / xGMI), instead of gathering propagators.
    Returns {"goal": mean, "grad": mean gradient, "goal_std": std over all B instances, "bounds": (lo, hi)}.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(B, world, rank)
    goals, grads = goal_and_grad(lo, hi)
    goals = torch.as_tensor(goals, dtype=torch.float64)
    grads = torch.as_tensor(grads, dtype=torch.float64)
    if hi > lo:
        part = torch.cat([goals.sum().reshape(1), (goals * goals).sum().reshape(1), grads.sum(dim=0).reshape(-1)])
    else:
        part = torch.zeros(2 + int(np.prod(grads.shape[1:])), dtype=torch.float64, device=grads.device)
    part = part.contiguous()
    if world > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    mean = part[0] / B
    var = torch.clamp(part[1] / B - mean * mean, min=0.0)
    return {"goal": mean, "goal_std": torch.sqrt(var), "grad": (part[2:] / B).reshape(tuple(grads.shape[1:])), "bounds": (lo, hi)}


class SlabRing:
    """The exchange schedule of `bench.py --gpus N`: every rank computes its slab of a step into one of G
    buffers; after G steps (or at `drain()`) ONE all-gather moves the G slabs of every rank (fewer, larger
    collectives: an xGMI all-gather of a single 0.3 MB slab is latency-bound against a 0.15 ms batch).

    `compute(out)` fills `out` [b_pad, ...] (rows past the rank's own `b_local` samples are padding so that uneven
    shards -- strong scaling of a batch that does not divide by the world size -- use one equal-size collective).
    Works on any backend / device (RCCL on the GPU box, gloo on CPU in tests/test_dist_gloo.py).  `stage_host`: device
    slabs exchanged by a CPU-only backend (gloo between ranks that SHARE a device, which RCCL refuses): the bank is copied
    to host memory, gathered there and copied back -- synchronous, for executing the N > 1 path on one device, not for speed.
    """

    def __init__(self, b_local: int, b_pad: int, tail: tuple, G: int, *, dtype=None, device=None, group=None, use_dist=None, overlap=False, stage_host=False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.use_dist = (dist.is_available() and dist.is_initialized()) if use_dist is None else use_dist
        self.world = dist.get_world_size(group) if self.use_dist else 1
        self.rank = dist.get_rank(group) if self.use_dist else 0
        self.b_local, self.b_pad, self.G = int(b_local), int(b_pad), max(1, int(G))
        if self.b_local > self.b_pad:
            raise ValueError("b_local exceeds the padded slab")
        dtype = torch.complex128 if dtype is None else dtype
        # overlap: TWO banks of G slabs.  The collective of a full bank is issued asynchronously (RCCL's own stream, ordered
        # after the kernels that filled the bank) and the next G steps compute into the other bank; a bank is waited for only
        # when it is about to be overwritten (and at drain()).  Still one collective per G steps, every slab gathered.
        self.stage_host = bool(stage_host) and self.use_dist
        self.overlap = bool(overlap) and self.use_dist and not self.stage_host
        nb = 2 if self.overlap else 1
        self._banks = [torch.zeros((self.G, self.b_pad) + tuple(tail), dtype=dtype, device=device) for _ in range(nb)]
        self.buf = self._banks[0]
        self.slab = int(np.prod((self.b_pad,) + tuple(tail)))
        self._gathered = [torch.empty((self.world * self.G * self.slab,), dtype=dtype, device=device) for _ in range(nb)] if self.use_dist else [None]
        self.gathered = self._gathered[0]
        self._work = [None] * nb
        self._cur = 0
        self.counter = 0
        self.pending = 0
        self.last_flushed = 0  # slabs moved by the most recent collective
        self.collectives = 0

    def _real(self, t):
        return self.torch.view_as_real(t) if t.is_complex() else t

    def _wait(self, bank):
        w = self._work[bank]
        if w is not None:
            w.wait()  # RCCL: the CURRENT STREAM waits for the collective (no host block); gloo: the host does
            self._work[bank] = None

    def flush(self):
        g = self.pending
        if self.use_dist and g > 0:
            n = g * self.slab
            bank = self._cur
            dst = self._real(self._gathered[bank][: self.world * n])
            src = self._real(self._banks[bank][:g].reshape(-1))
            if self.stage_host:
                src_h = src.cpu()  # waits for the kernels that filled the bank
                dst_h = self.torch.empty(dst.shape, dtype=dst.dtype)
                self.dist.all_gather_into_tensor(dst_h, src_h, group=self.group)
                dst.copy_(dst_h)
            elif self.overlap:
                self._work[bank] = self.dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
                self._cur = bank ^ 1
                self._wait(self._cur)  # the bank the next steps compute into: its previous collective must have read it
            else:
                self.dist.all_gather_into_tensor(dst, src, group=self.group)
            self.gathered = self._gathered[bank]
            self._gathered_bank = bank
            self.collectives += 1
            self.last_flushed = g
        self.pending = 0

    def step(self, compute):
        out = self._banks[self._cur][self.pending if self.overlap else self.counter % self.G]
        self.counter += 1
        res = compute(out)
        self.pending += 1
        if self.pending == self.G:
            self.flush()
        return out if res is None else res

    def drain(self):
        self.flush()
        for bank in range(len(self._work)):
            self._wait(bank)
        self.counter = 0

    def warm(self, sizes):
        """Run every message size in `sizes` once (communicator set-up before anything is timed)."""
        if not self.use_dist:
            return
        for g in sorted({int(s) for s in sizes if 0 < int(s) <= self.G}):
            self.pending = g
            self.flush()

    def sent_slab(self, i: int):
        """This rank's own slab i of the most recent collective, as computed (the bank it was gathered from)."""
        return self._banks[getattr(self, "_gathered_bank", 0)][i]

    def gathered_slab(self, r: int, i: int):
        """Slab i (of the most recent collective) as sent by rank r: [b_pad, ...].  With `overlap` the collective may still be
        in flight: it is waited for here (RCCL: the current stream is ordered after it; gloo: the host waits)."""
        if self.use_dist:
            self._wait(getattr(self, "_gathered_bank", 0))
        g = self.last_flushed
        flat = self.gathered[r * g * self.slab : (r + 1) * g * self.slab]
        return flat.reshape((g,) + tuple(self.buf.shape[1:]))[i]
