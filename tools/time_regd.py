import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from c3_amd import propagation, workloads, _lib
if os.environ.get('C3P_LIB'):
    _lib.LIB_PATH = os.environ['C3P_LIB']
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
wl = workloads.make_workload(4, B=B, N=N)
a = [torch.as_tensor(x, device=dev) for x in (wl.h0, wl.hks, wl.signals)]
col = torch.as_tensor(wl.col_ops, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = propagation.propagate_batch(a[0], a[1], a[2], wl.dt, col_ops=col, lindbladian=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"B={B} N={N}: {1e3*(t1-t0):.1f} ms -> {B/(t1-t0):.0f} props/s", flush=True)
