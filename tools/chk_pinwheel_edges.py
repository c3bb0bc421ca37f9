"""Edge sizes of the pinwheel class (D = 25..28 real): N = 1..7, B = 1 / 2, K = 1 / 3, segment overrides, zero operators."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from c3_amd import _lib, propagation as prop
from oracle import c3_oracle as o
rng = np.random.default_rng(2)
sym = lambda D, s: (lambda m: (s * (m + m.T) / 2).astype(complex))(rng.normal(size=(D, D)))
for D in (25, 27, 28):
    for N in (1, 2, 3, 4, 7):
        for K in (1, 3):
            for B in (1, 2):
                h0 = sym(D, 8e9); hks = np.stack([sym(D, 1.0) for _ in range(K)])
                sig = rng.normal(size=(B, K, N)) * 3e9
                for S in (0, 1, 3):
                    with _lib.options(segments=S):
                        r = prop.propagate_batch(h0, hks, sig, 1e-11, want_dUs=True)
                    U = np.asarray(r["U"]); ref = o.propagate_batch(h0, hks, sig, 1e-11)
                    e = max(np.linalg.norm(U[b] - ref[b]) for b in range(B))
                    assert e < 1e-12, (D, N, K, B, S, e)
                Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
                g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar))
                with _lib.options(no_real_grad=1):
                    g2 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar))
                assert np.abs(g - g2).max() <= 1e-11 * np.abs(g2).max(), (D, N, K, B)
# zero control amplitudes, zero drift
D = 27
h0 = np.zeros((D, D), complex); hks = np.stack([sym(D, 1.0)])
sig = np.zeros((2, 1, 5))
U = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11)["U"])
assert np.abs(U - np.eye(D)).max() < 1e-15, np.abs(U - np.eye(D)).max()
print("edge cases OK", _lib.last_kernel())
