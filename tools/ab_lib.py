"""A/B timing of one bench configuration with an alternative build of the library:
    python tools/ab_lib.py <lib.so> <config> [batch] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation, workloads
cfg = int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else None; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
per = {1: 256, 2: 256, 3: 512, 4: 512, 5: 1024}[cfg]
wl = workloads.make_workload(cfg, B=B or per)
dev = torch.device("cuda:0")
fr = wl.fr_phase
if wl.lindblad:
    fr = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
bp = propagation.BatchPropagator(*(torch.as_tensor(x, device=dev) for x in (wl.h0, wl.hks, wl.signals)), wl.dt,
                                 col_ops=torch.as_tensor(wl.col_ops, device=dev) if wl.lindblad else None, fr_phase=torch.as_tensor(fr, device=dev))
t_r = time.perf_counter()
while time.perf_counter() - t_r < 0.3:
    bp.run(); torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); bp.run(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{os.path.basename(sys.argv[1])} cfg{cfg} B={wl.B}: {best*1e3:.3f} ms  {wl.B/best:.4g} props/s  kernel={_lib.last_kernel()}")
