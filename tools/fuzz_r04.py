"""Randomised parity sweep of the round-4 entry points (insurance beyond the fixed cases of tests/test_gpu_round4.py):
  (a) fused goal + gradient (c3p_pwc_unitary_goal_vjp) against the three-call form, random D, batch, slices, control lines,
      goal subspaces, real / complex Hamiltonians, with / without frame phases;
  (b) Hermitian-basis Lindblad sweep (D = 7, 8, 9) against the tiled sweep, random segments, K, N, per-sample operators, drive
      strength (Taylor degree / squarings), and the taped pair against the untaped one;
  (c) core + border form (D = 5, 9) against the padded tiles, random N, B, amplitude, MW on / off;
  (d) ODE trajectories in time segments against the direct integration; every other time: five to eight control lines against
      the workgroup kernel, and the taped Lindblad pair at D = 2, 3 against the untaped one.
    python tools/fuzz_r04.py --seconds 120 --seed 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import _lib, fidelities as fid, propagation as prop

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(x, device=dev)


def herm(D, s, real=False):
    m = rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))
    return (s * (m + m.conj().T) / 2).astype(complex)


def factor_dims(D):
    for d in (2, 3, 4, 5, 6):
        if D % d == 0 and D // d >= 2:
            return [d, D // d]
    return [D]


counts = {"goal": 0, "lind": 0, "split": 0, "ode": 0}
worst = {"goal": 0.0, "lind": 0.0, "split": 0.0, "ode": 0.0}
t_end = time.time() + a.seconds
it = 0
while time.time() < t_end:
    kind = ("goal", "lind", "split", "ode")[it % 4]
    it += 1
    if kind == "goal":
        D = int(rng.choice([2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 16, 20, 24, 27, 32, 36, 40, 44, 50]))
        B, K, N = int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(8, 60))
        real = bool(rng.integers(0, 2))
        dims = factor_dims(D)
        index = sorted(rng.choice(len(dims), size=int(rng.integers(1, len(dims) + 1)), replace=False).tolist())
        L = 2 ** len(index)
        G, _ = np.linalg.qr(rng.normal(size=(L, L)) + 1j * rng.normal(size=(L, L)))
        h0, hks = herm(D, 3e11 / D, real), np.stack([herm(D, 1.0, real) for _ in range(K)])
        sig = rng.normal(size=(B, K, N)) * 2e9 * rng.choice([0.3, 1.0, 4.0])
        ph = rng.uniform(0, 6, size=(B, D)) if rng.integers(0, 2) else None
        kindf = str(rng.choice(["unitary", "average"]))
        r = prop.propagate_batch_goal_vjp(h0, hks, sig, 1e-11, G, index, dims, kind=kindf, fr_phase=ph)
        U = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph)["U"])
        cot = fid.unitary_infid_cotangent if kindf == "unitary" else fid.average_infid_cotangent
        Ubar, goal = cot(G, U, index, dims)
        g3 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
        e = max(np.abs(np.asarray(r["goal"]) - np.asarray(goal)).max(), np.abs(np.asarray(r["grad_signals"]) - g3).max() / max(np.abs(g3).max(), 1e-300),
                np.abs(np.asarray(r["U"]) - U).max())
        assert e < 1e-10, ("goal", D, B, K, N, real, index, kindf, e)
    elif kind == "lind":
        D = int(rng.choice([7, 8, 9]))
        B, K, N = int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(4, 40))
        C = int(rng.integers(1, 3))
        per_sample = bool(rng.integers(0, 2))
        nb = B if per_sample else 1
        h0 = np.stack([herm(D, 0.8) for _ in range(nb)])
        hks = np.stack([np.stack([herm(D, 0.5) for _ in range(K)]) for _ in range(nb)])
        if not per_sample:
            h0, hks = h0[0], hks[0]
        col = np.stack([0.25 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)])
        sig = rng.uniform(-1, 1, size=(B, K, N))
        Dm = D * D
        Ubar = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
        ph = rng.uniform(0, 6, size=(B, Dm)) if rng.integers(0, 2) else None
        dt = float(rng.choice([0.05, 0.15, 0.4]))
        segs = int(rng.integers(1, 6))
        with _lib.options(segments=segs):
            g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
            r = prop.propagate_batch_lindblad_taped(t(h0), t(hks), t(sig), dt, t(col), fr_phase=None if ph is None else t(ph))
            gt2 = r["tape"].vjp(t(Ubar)).cpu().numpy()
        with _lib.options(tiled_grad=1):
            gt = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
        U0 = np.asarray(prop.propagate_batch(h0, hks, sig, dt, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
        e = max(np.abs(g - gt).max(), np.abs(gt2 - gt).max()) / np.abs(gt).max()
        e = max(e, np.abs(r["U"].cpu().numpy() - U0).max())
        assert e < 1e-9, ("lind", D, B, K, N, C, per_sample, dt, segs, e)
    elif kind == "split":
        D = int(rng.choice([5, 9]))
        B, N = int(rng.integers(1, 40)), int(rng.integers(3, 300))
        amp = float(rng.choice([0.5, 1.0, 3.0, 12.0]))
        h0, hks = herm(D, 6e10, True), np.stack([herm(D, 1.0, True), herm(D, 1.0, True)])
        sig = rng.normal(size=(B, 2, N)) * 2e9 * amp
        ph = rng.uniform(0, 6, size=(B, D))
        with _lib.options(no_mw=int(rng.integers(0, 2)) or None):
            x = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph)["U"])
            with _lib.options(no_split81=1):
                y = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph)["U"])
        e = np.abs(x - y).max()
        assert e < 1e-10, ("split", D, B, N, amp, e)
    elif kind == "ode" and it % 8 == 0:
        # more than four control lines (assembled Hamiltonians, lane rows / the eight-line instance at D >= 17) and the small taped
        # Lindblad pair, against the workgroup kernel / the untaped pair
        D = int(rng.choice([3, 5, 9, 14, 16, 18, 24, 27]))
        B, K, N = int(rng.integers(1, 6)), int(rng.integers(5, 9)), int(rng.integers(8, 60))
        h0, hks = herm(D, 0.3), np.stack([herm(D, 0.2, bool(rng.integers(0, 2))) for _ in range(K)])
        sig = rng.uniform(-1, 1, size=(B, K, N))
        psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
        solver = str(rng.choice(["rk4", "rk38", "rk5", "tsit5"]))
        fin = bool(rng.integers(0, 2))
        x = np.asarray(prop.ode_solve_batch(h0, hks, sig, 0.05, psi, solver, "schrodinger", final_only=fin))
        with _lib.options(ode_wg=1):
            y = np.asarray(prop.ode_solve_batch(h0, hks, sig, 0.05, psi, solver, "schrodinger", final_only=fin))
        e = np.abs(x - y).max() / max(1.0, np.abs(y).max())
        assert e < 1e-11, ("ode K>4", D, B, K, N, solver, fin, e)
        Dl = int(rng.choice([2, 3, 4]))
        Bl, Kl, Nl = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.integers(4, 90))
        ps = bool(rng.integers(0, 2))
        nb = Bl if ps else 1
        # (lossy Hamiltonians: the complex kernels; at D = 4 the taped pair exists in the Hermitian basis only)
        h0l = np.stack([herm(Dl, 0.8) - (0.05j * np.diag(np.arange(Dl)) if (Dl < 4 and rng.integers(0, 2)) else 0) for _ in range(nb)])
        hkl = np.stack([np.stack([herm(Dl, 0.5) for _ in range(Kl)]) for _ in range(nb)])
        if not ps:
            h0l, hkl = h0l[0], hkl[0]
        col = np.stack([0.3 * (rng.normal(size=(Dl, Dl)) + 1j * rng.normal(size=(Dl, Dl)))])
        sgl = rng.uniform(-1, 1, size=(Bl, Kl, Nl))
        Ub = rng.normal(size=(Bl, Dl**2, Dl**2)) + 1j * rng.normal(size=(Bl, Dl**2, Dl**2))
        phl = rng.uniform(0, 6, size=(Bl, Dl**2)) if rng.integers(0, 2) else None
        g0 = np.asarray(prop.propagate_batch_lindblad_vjp(h0l, hkl, sgl, 0.2, col, Ub, fr_phase=phl))
        U0 = np.asarray(prop.propagate_batch(h0l, hkl, sgl, 0.2, col_ops=col, lindbladian=True, fr_phase=phl)["U"])
        e2 = 0.0
        if prop.lindblad_tape_supported(Bl, Kl, Nl, Dl):  # (D = 4: only with the real kernels enabled)
            r = prop.propagate_batch_lindblad_taped(t(h0l), t(hkl), t(sgl), 0.2, t(col), fr_phase=None if phl is None else t(phl))
            g1 = r["tape"].vjp(t(Ub)).cpu().numpy()
            e2 = max(np.abs(g1 - g0).max() / max(np.abs(g0).max(), 1e-300), np.abs(r["U"].cpu().numpy() - U0).max())
        assert e2 < 1e-10, ("taped small", Dl, Bl, Kl, Nl, ps, e2)
        e = max(e, e2)
        # the real Hermitian-basis forward kernels (Hermitian Hamiltonians: the flag is set) against the complex ones
        h0h = np.stack([herm(Dl, float(rng.choice([0.3, 0.8, 6.0]))) for _ in range(nb)])
        if not ps:
            h0h = h0h[0]
        Nr = int(rng.integers(1, 400))
        sgr = rng.uniform(-1, 1, size=(Bl, Kl, Nr))
        xr = np.asarray(prop.propagate_batch(h0h, hkl, sgr, 0.2, col_ops=col, lindbladian=True, fr_phase=phl)["U"])
        with _lib.options(no_smallr=1):
            yr = np.asarray(prop.propagate_batch(h0h, hkl, sgr, 0.2, col_ops=col, lindbladian=True, fr_phase=phl)["U"])
        e3 = np.abs(xr - yr).max() / max(1.0, np.abs(yr).max())
        assert e3 < 1e-11, ("small real", Dl, Bl, Kl, Nr, ps, e3)
        e = max(e, e3)
    else:
        D = int(rng.choice([3, 6, 9, 12, 18, 27, 33, 40]))
        B, K, N = int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(70, 260))
        real = bool(rng.integers(0, 2))
        h0, hks = herm(D, 0.4, real), np.stack([herm(D, 0.3, real) for _ in range(K)])
        sig = rng.uniform(-1, 1, size=(B, K, N))
        psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
        solver = str(rng.choice(["rk4", "rk38", "rk5", "tsit5"]))
        x = np.asarray(prop.ode_solve_batch(h0, hks, sig, 0.05, psi, solver, "schrodinger"))
        with _lib.options(ode_no_seg=1):
            y = np.asarray(prop.ode_solve_batch(h0, hks, sig, 0.05, psi, solver, "schrodinger"))
        e = np.abs(x - y).max() / max(1.0, np.abs(y).max())
        assert e < 1e-10, ("ode", D, B, K, N, real, solver, e)
    counts[kind] += 1
    worst[kind] = max(worst[kind], float(e))
print("fuzz ok:", counts, "worst deviations:", {k: f"{v:.2e}" for k, v in worst.items()})
