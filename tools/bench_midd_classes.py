import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib, propagation as prop
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(30): x @ x
torch.cuda.synchronize()
rng = np.random.default_rng(0)
for D in (13, 16, 17, 20, 21, 24, 25, 28, 29, 32, 33, 36, 40):
    sym = lambda s: (lambda m: (s * (m + m.T) / 2).astype(complex))(rng.normal(size=(D, D)))
    h0 = np.diag(rng.uniform(0, 1, D)).astype(complex) + sym(0.05); hks = np.stack([sym(0.3) for _ in range(2)])
    B, N = 512, 400
    sig = rng.uniform(-1, 1, size=(B, 2, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    dt = 1.5 / (one(h0) + sum(one(h) for h in hks))
    a = (t(h0), t(hks), t(sig), dt)
    for _ in range(3): prop.propagate_batch(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): prop.propagate_batch(*a)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    flops = B * N * (11 * 2 * D**3)  # 7 + 4 real products of the slice
    print(f"D={D:3d}  {ms:8.3f} ms  {B*1e3/ms:10.0f} propagators/s  useful real-product flops {flops/ms/1e9:7.2f} TFLOP/s  {_lib.last_kernel_detail()[:90]}", flush=True)
