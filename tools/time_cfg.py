"""Forward (and gradient) timing of one BASELINE configuration with one library:
python tools/time_cfg.py <library> <config> <B> [grad]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
from oracle import c3_oracle
cfg, B = int(sys.argv[2]), int(sys.argv[3])
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(20): x @ x
torch.cuda.synchronize()
w = make_workload(cfg, B=B)
kw = dict(fr_phase=t(w.fr_phase)) if getattr(w, "fr_phase", None) is not None else {}
if getattr(w, "col_ops", None) is not None: kw.update(col_ops=t(w.col_ops), lindbladian=True)
h0, hks, sig = t(w.h0), t(w.hks), t(w.signals)
f = lambda: prop.propagate_batch(h0, hks, sig, w.dt, **kw)
for _ in range(3): r = f()
torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
U = r["U"][:1].cpu().numpy()
kwo = {k: (v[:1].cpu().numpy() if k == "fr_phase" else (v.cpu().numpy() if hasattr(v, "cpu") else v)) for k, v in kw.items()}
ref = c3_oracle.propagate_batch(w.h0, w.hks, w.signals[:1], w.dt, **kwo)
print(os.path.basename(_lib.LIB_PATH), f"cfg{cfg} B={B} kernel {_lib.last_kernel()} ms {1e3 * min(ts):.3f} propagators/s {B / min(ts):.4e} err {np.linalg.norm(U[0] - ref[0]):.2e}", flush=True)
if len(sys.argv) > 4:
    Bg = min(B, 256)
    Ubar = torch.randn(Bg, U.shape[1], U.shape[2], dtype=torch.complex128, device="cuda:0")
    kwg = dict(kw)
    if "fr_phase" in kwg: kwg["fr_phase"] = kwg["fr_phase"][:Bg]
    g = lambda: prop.propagate_batch_vjp(h0, hks, sig[:Bg], w.dt, Ubar, **kwg)
    g(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); g(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(os.path.basename(_lib.LIB_PATH), f"cfg{cfg} gradient B={Bg} ms {1e3 * min(ts):.3f}", flush=True)
