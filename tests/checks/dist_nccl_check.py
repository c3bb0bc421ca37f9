"""Run under `python -m torch.distributed.run --nproc-per-node N` on the GPU box: the RCCL ("nccl") paths of c3_amd.dist
-- communicator set-up, all_gather_into_tensor of the U slabs, the all-reduce of the robust goal -- against the local
result and the oracle.  Prints one line "DIST_NCCL_OK world=N" on rank 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

from c3_amd import dist as c3dist, propagation as prop, workloads
from oracle import c3_oracle as o

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
assert dist.get_backend() == "nccl"
B = 4 * world + 1  # uneven shards
wl = workloads.make_workload(2, B=B, N=60)
h0, hks, sig, ph = (torch.as_tensor(x, device=dev) for x in (wl.h0, wl.hks, wl.signals, wl.fr_phase))
r = c3dist.propagate_batch_sharded(h0, hks, sig, wl.dt, fr_phase=ph)
U = r["U"].cpu().numpy()
ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, fr_phase=wl.fr_phase)
err = max(np.linalg.norm(U[b] - ref[b]) for b in range(B))
assert U.shape == (B, wl.D, wl.D) and err < 1e-10, err
lo, hi = r["bounds"]
assert (lo, hi) == c3dist.shard_bounds(B, world, rank)

# the exchange schedule of bench.py: G steps per collective
ring = c3dist.SlabRing(hi - lo, c3dist.max_shard(B, world), (wl.D, wl.D), 3, device=dev)
loc = prop.propagate_batch(h0, hks, sig[lo:hi], wl.dt, fr_phase=ph[lo:hi])["U"]
for step in range(5):
    ring.step(lambda out: out[: hi - lo].copy_(loc))
ring.drain()
torch.cuda.synchronize()
assert torch.equal(ring.gathered_slab(rank, 0)[: hi - lo], loc)

# robust goal: one all-reduce of the partial sums (optimalcontrol_robust.py:49-70)
def goal_and_grad(a, b):
    g = torch.arange(a, b, dtype=torch.float64, device=dev)
    return g, torch.stack([g, 2 * g], dim=1)

res = c3dist.robust_goal_sharded(goal_and_grad, B)
exp = np.arange(B, dtype=float)
assert abs(float(res["goal"]) - exp.mean()) < 1e-12 and abs(float(res["grad"][1]) - 2 * exp.mean()) < 1e-12
dist.barrier()
if rank == 0:
    print(f"DIST_NCCL_OK world={world} err={err:.2e}", flush=True)
dist.destroy_process_group()
