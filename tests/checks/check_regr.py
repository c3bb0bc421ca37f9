"""Lindblad chains in the Hermitian basis (c3p_regr.hip) against the complex kernel (c3p_regd.hip) and the oracle.
   python tests/checks/check_regr.py [--time] [--waves 8|2] [--rolled]   (the last two: tools/ab_build.sh variants c3p_regr.hip -DC3P_REGR_VARIANTS, C3P_LIB=...)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from c3_amd import propagation as prop, _lib, workloads
from oracle import c3_oracle as o

if os.environ.get("C3P_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["C3P_LIB"])
dev = torch.device("cuda:0")
if "--waves" in sys.argv:
    _lib.set_option("regr_waves", sys.argv[sys.argv.index("--waves") + 1])
if "--rolled" in sys.argv:
    _lib.set_option("regr_rolled", 1)
t = lambda a: torch.as_tensor(a, device=dev)
rng = np.random.default_rng(7)
worst = 0.0
for D, B, N, cplx_h, nonherm in ((9, 3, 12, False, False), (9, 2, 40, True, False), (7, 3, 9, True, False), (8, 2, 10, False, False), (9, 3, 10, True, True)):
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + (1j * rng.normal(size=(D, D)) if cplx_h else 0))
    K, C = 2, 2
    h0, hks = herm(0.25), np.stack([herm(0.1) for _ in range(K)])
    h0 = h0.astype(complex); hks = hks.astype(complex)
    if nonherm:
        hks[1] = hks[1] + 0.02 * rng.normal(size=(D, D))  # not Hermitian: the complex kernel must take these samples
    col = np.stack([0.2 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    fr = rng.uniform(0, 6.28, size=(B, D * D))
    r = prop.propagate_batch(t(h0), t(hks), t(sig), 1.0, col_ops=t(col), lindbladian=True, fr_phase=t(fr), want_dUs=True)
    U, dUs = r["U"].cpu().numpy(), r["dUs"].cpu().numpy()
    with _lib.options(no_hermitian_basis=1):
        r2 = prop.propagate_batch(t(h0), t(hks), t(sig), 1.0, col_ops=t(col), lindbladian=True, fr_phase=t(fr), want_dUs=True)
    U2, dUs2 = r2["U"].cpu().numpy(), r2["dUs"].cpu().numpy()
    ref = o.propagate_batch(h0, hks, sig, 1.0, col_ops=col, lindbladian=True, fr_phase=None)
    ref = np.stack([np.exp(1j * fr[b])[:, None] * ref[b] for b in range(B)])
    e1 = max(np.linalg.norm(U[b] - ref[b]) for b in range(B))
    e2 = max(np.linalg.norm(U2[b] - ref[b]) for b in range(B))
    ed = float(np.abs(dUs - dUs2).max())
    worst = max(worst, e1, ed)
    print(f"D={D} B={B} N={N} complexH={cplx_h} nonherm={nonherm}: |U_hb - oracle|_F {e1:.2e}   |U_complex - oracle|_F {e2:.2e}   |dUs_hb - dUs_complex|_max {ed:.2e}", flush=True)
assert worst < 1e-10, worst
# segments (S > 1: B < 256) on cfg4 operators
wl = workloads.make_workload(4, B=6, N=64)
frl = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
U = prop.propagate_batch(t(wl.h0), t(wl.hks), t(wl.signals), wl.dt, col_ops=t(wl.col_ops), lindbladian=True, fr_phase=t(frl))["U"].cpu().numpy()
ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, fr_phase=wl.fr_phase)
e = max(np.linalg.norm(U[b] - ref[b]) for b in range(wl.B))
print(f"cfg4 operators B=6 N=64 (segments): {e:.2e}", flush=True)
assert e < 1e-10
if "--time" in sys.argv:
    for B in (512, 256):
        wl = workloads.make_workload(4, B=B, N=1000)
        args = (t(wl.h0), t(wl.hks), t(wl.signals), wl.dt)
        kw = dict(col_ops=t(wl.col_ops), lindbladian=True)
        for name, opt in (("hermitian basis (real)", {}), ("complex kernel", {"no_hermitian_basis": 1})):
            with _lib.options(**opt):
                prop.propagate_batch(*args, **kw); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    Ux = prop.propagate_batch(*args, **kw)["U"]
                torch.cuda.synchronize()
                dt_ = (time.perf_counter() - t0) / 3
            print(f"cfg4 B={B}: {name}: {dt_*1e3:.1f} ms  {B/dt_:.0f} propagators/s", flush=True)
            if name.startswith("herm"): Ua = Ux
        print("   |U_hb - U_complex|_max", float((Ua - Ux).abs().max()), flush=True)
print("OK")
