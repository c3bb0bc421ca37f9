"""cfg2-sized batch with a complex Hermitian control operator: times the COMPLEX path of the small-D kernel."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from c3_amd import propagation as prop, _lib
from c3_amd.workloads import make_workload
w = make_workload(2)
hks = w.hks.copy()
hks[1] = hks[1] + 1j * (np.triu(hks[1].real, 1) - np.triu(hks[1].real, 1).T) * 0.3  # Hermitian, complex
dev = "cuda:0"
h0, hk, sig, ph = (torch.as_tensor(x, device=dev) for x in (w.h0, hks, w.signals, w.fr_phase))
lib = _lib.load()
for _ in range(300): prop.propagate_batch(h0, hk, sig, w.dt, fr_phase=ph)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): prop.propagate_batch(h0, hk, sig, w.dt, fr_phase=ph)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 100 * 1e3
from oracle import c3_oracle as o
U = prop.propagate_batch(h0, hk, sig, w.dt, fr_phase=ph)["U"][:2].cpu().numpy()
ref = o.propagate_batch(w.h0, hks, w.signals[:2], w.dt, fr_phase=w.fr_phase[:2])
print(json.dumps({"complex_path_ms_per_batch": ms, "propagators_per_s": w.B / ms * 1e3, "err": float(max(np.linalg.norm(U[b] - ref[b]) for b in range(2)))}))
