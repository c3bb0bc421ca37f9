import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import propagation, workloads
wl = workloads.make_workload(2, B=4, N=8)
dev = torch.device("cuda", 0)
bp = propagation.BatchPropagator(torch.as_tensor(wl.h0, device=dev), torch.as_tensor(wl.hks, device=dev), torch.as_tensor(wl.signals, device=dev), wl.dt, fr_phase=torch.as_tensor(wl.fr_phase, device=dev))
out = torch.empty((4, 9, 9), dtype=torch.complex128, device=dev)
for _ in range(50): bp.run(out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): bp.run(out=out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("cpu per call us", (t1 - t0) / 2000 * 1e6, "incl sync", (t2 - t0) / 2000 * 1e6)
