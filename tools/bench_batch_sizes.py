"""Throughput of the cfg2 workload at batch sizes that do not fill the machine evenly (segment-count policy)."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import workloads as w, propagation as p

out = {}
dev = torch.device("cuda:0")
for B in (64, 200, 256, 300, 400, 512, 1000):
    wl = w.make_workload(2, B=B)
    h0 = torch.as_tensor(wl.h0, device=dev); hks = torch.as_tensor(wl.hks, device=dev); sig = torch.as_tensor(wl.signals, device=dev)
    ph = torch.as_tensor(wl.fr_phase, device=dev)
    for _ in range(300):
        p.propagate_batch(h0, hks, sig, wl.dt, fr_phase=ph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        p.propagate_batch(h0, hks, sig, wl.dt, fr_phase=ph)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    out[B] = {"ms_per_batch": ms, "propagators_per_s": B / ms * 1e3}
print(json.dumps(out))
