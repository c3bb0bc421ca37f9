#!/usr/bin/env python3
"""The dispatch table of libc3prop as DATA: which kernels a call shape lands in.

  measure (GPU):   python tools/dispatch_table.py --measure profiles/r06/dispatch_table.json
                   every row is one REAL call through the host layer on cuda:0; the kernels are read back from the library's
                   launch log (c3p_last_kernel_detail: host function pointers resolved to kernel names -- what ran, not what
                   the dispatcher meant to run)
  render:          python tools/dispatch_table.py --write   profiles/r06/dispatch_table.json      (rewrites INTEGRATION.md between its markers)
                   python tools/dispatch_table.py --check   profiles/r06/dispatch_table.json      (exit 1 if INTEGRATION.md differs)
  one shape (GPU): python tools/dispatch_table.py --probe pwc_unitary D=9 B=256 K=2 N=1000

Rows of neighbouring dimensions that run the same kernel family (template arguments aside) are merged; the example column shows
the launch log of the first dimension of the range.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = "<!-- BEGIN generated: dispatch table -->", "<!-- END generated: dispatch table -->"


# ---------------------------------------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------------------------------------
def _ops(D, K, rng, real=True, scale=0.05):
    import numpy as np

    def herm(s):
        m = rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))
        return (s * (m + m.conj().T) / 2).astype(np.complex128)

    h0 = np.diag(rng.uniform(0.0, 1.0, D)).astype(np.complex128) + herm(0.01)
    hks = np.stack([herm(scale) for _ in range(K)]) if K else np.zeros((0, D, D), dtype=np.complex128)
    return h0, hks


def probe(entry, D, B=4, K=2, N=64, real=True, flags="", seed=0):
    """one call -> (family id, launch log)"""
    import numpy as np
    import torch

    from c3_amd import _lib
    from c3_amd import propagation as p

    rng = np.random.default_rng(seed + 7 * D + K)
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, device=dev)
    h0, hks = _ops(D, K, rng, real)
    if "nonherm" in flags:
        h0 = h0 + 0.02j * rng.normal(size=(D, D))
    sig = rng.uniform(-1, 1, size=(B, K, N))
    dt = 0.3 / D**0.5
    col = np.stack([0.05 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
    if entry == "pwc_unitary":
        if "per_slice" in flags:
            hs = h0[None, None] + np.einsum("bkn,kij->bnij", sig, hks)
            p.propagate_batch(t(hs), None, None, dt)
        else:
            p.propagate_batch(t(h0), t(hks), t(sig), dt, want_dUs="dUs" in flags)
    elif entry == "pwc_lindblad":
        p.propagate_batch(t(h0), t(hks), t(sig), dt, col_ops=t(col), lindbladian=True, want_dUs="dUs" in flags)
    elif entry == "pwc_unitary_vjp":
        Ub = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
        p.propagate_batch_vjp(t(h0), t(hks), t(sig), dt, t(Ub))
    elif entry == "pwc_unitary_goal_vjp":
        dims = [D]
        ideal = np.eye(2, dtype=np.complex128)
        p.propagate_batch_goal_vjp(t(h0), t(hks), t(sig), dt, t(ideal), [0], dims, want_U=False)
    elif entry == "pwc_lindblad_vjp":
        Ub = rng.normal(size=(B, D * D, D * D)) + 1j * rng.normal(size=(B, D * D, D * D))
        p.propagate_batch_lindblad_vjp(t(h0), t(hks), t(sig), dt, t(col), t(Ub))
    elif entry == "pwc_lindblad_taped":
        r = p.propagate_batch_lindblad_taped(t(h0), t(hks), t(sig), dt, t(col))
        del r
    elif entry.startswith("ode_"):
        step = entry[4:]
        psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
        init = psi if step == "schrodinger" else np.einsum("bik,bjk->bij", psi, psi.conj())
        p.ode_solve_batch(t(h0), t(hks), t(sig), dt, t(init), "rk4", step, col_ops=t(col) if step == "lindblad" else None, final_only="final" in flags)
    elif entry == "expm":
        p.expm(t(np.stack([-1j * dt * h0] * B)))
    elif entry == "matmul_chain":
        M = np.stack([np.linalg.qr(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))[0] for _ in range(N)])
        p.tf_matmul_left(t(np.stack([M] * B)))
    else:
        raise SystemExit(f"unknown entry {entry}")
    torch.cuda.synchronize()
    return _lib.last_kernel(), _lib.last_kernel_detail()


GRID = [
    # (entry, label of the regime, dict of fixed arguments, list of D)
    ("pwc_unitary", "real symmetric operators, B = 256, N = 1000 (cfg2's shape)", dict(B=256, K=2, N=1000), [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]),
    ("pwc_unitary", "real symmetric operators, B = 8, N = 200", dict(B=8, K=2, N=200), [3, 9, 12, 13, 14, 16, 17, 20, 24, 25, 27, 28, 29, 32, 33, 36, 37, 40, 41, 48, 64, 81, 92, 93, 128]),
    ("pwc_unitary", "complex Hermitian operators, B = 8, N = 200", dict(B=8, K=2, N=200, real=False), [3, 9, 12, 13, 16, 24, 27, 32, 36, 40, 41, 64, 92, 93]),
    ("pwc_unitary", "slice propagators requested (want_dUs), B = 8, N = 200", dict(B=8, K=2, N=200, flags="dUs"), [3, 9, 13, 27, 36, 41, 93]),
    ("pwc_unitary", "one Hamiltonian per slice (branch B), B = 4, N = 100", dict(B=4, K=2, N=100, flags="per_slice"), [3, 9, 13, 27, 40, 41, 64]),
    ("pwc_lindblad", "Hermitian H, one collapse operator, B = 8, N = 200 (D = system dimension, kernels see D^2)", dict(B=8, K=2, N=200), [2, 3, 4, 5, 6, 7, 8, 9, 10, 12]),
    ("pwc_lindblad", "non-Hermitian H (complex generator tables), B = 8, N = 200", dict(B=8, K=2, N=200, real=False, flags="nonherm"), [2, 3, 7, 9]),
    ("pwc_unitary_vjp", "real symmetric operators, B = 64, N = 200", dict(B=64, K=2, N=200), [3, 9, 12, 13, 16, 24, 27, 32, 36, 40, 41, 48, 64, 65, 81]),
    ("pwc_unitary_vjp", "complex Hermitian operators, B = 64, N = 200", dict(B=64, K=2, N=200, real=False), [3, 9, 13, 27, 36, 40, 41, 64]),
    ("pwc_unitary_vjp", "real symmetric operators, B = 512, N = 100 (large batch: tiled sweep above D = 40)", dict(B=512, K=2, N=100), [41, 48, 64]),
    ("pwc_unitary_goal_vjp", "fused goal + gradient, B = 64, N = 200", dict(B=64, K=2, N=200), [3, 9, 12, 13, 27, 36, 40, 48]),
    ("pwc_lindblad_vjp", "Hermitian H, B = 8, N = 100", dict(B=8, K=2, N=100), [2, 3, 4, 5, 6, 7, 8, 9, 10]),
    ("pwc_lindblad_taped", "forward pass + tape, B = 8, N = 100", dict(B=8, K=2, N=100), [2, 3, 4, 7, 8, 9]),
    ("ode_schrodinger", "rk4, vector states, K = 2, B = 256, N = 100, whole trajectory", dict(B=256, K=2, N=100), [3, 9, 12, 16, 17, 27, 36, 48, 49, 64]),
    ("ode_schrodinger", "rk4, vector states, K = 2, B = 4, N = 400, final state only", dict(B=4, K=2, N=400, flags="final"), [3, 9, 16, 27, 48, 64]),
    ("ode_von_neumann", "rk4, rho states, K = 2, B = 64, N = 100", dict(B=64, K=2, N=100), [3, 9, 16, 17, 27, 36, 48, 49, 64]),
    ("ode_von_neumann", "rk4, rho states, K = 6 control lines, B = 64, N = 100", dict(B=64, K=6, N=100), [3, 9, 16, 27]),
    ("ode_lindblad", "rk4, rho states, one collapse operator, K = 2, B = 64, N = 100", dict(B=64, K=2, N=100), [3, 9, 16, 17, 27, 36, 48, 49]),
    ("expm", "B = 64 matrices", dict(B=64), [3, 9, 12, 13, 27, 40, 41, 81, 93]),
    ("matmul_chain", "ordered product of N = 64 matrices, B = 8", dict(B=8, N=64), [3, 9, 12, 13, 27, 40, 41, 81, 93]),
]


def measure(path):
    rows = []
    for entry, label, fixed, dims in GRID:
        for D in dims:
            kw = dict(fixed)
            flags = kw.pop("flags", "")
            real = kw.pop("real", True)
            try:
                fam, det = probe(entry, D, real=real, flags=flags, **kw)
            except Exception as e:  # noqa: BLE001  (a shape the library refuses is a row too)
                fam, det = "refused", str(e)[:160]
            rows.append({"entry": entry, "regime": label, "D": D, "family": fam, "kernels": det})
            print(entry, label[:40], D, fam, det[:120], flush=True)
    json.dump({"_comment": "tools/dispatch_table.py --measure on one MI355X: launch logs (c3p_last_kernel_detail) of real calls", "rows": rows}, open(path, "w"), indent=1)


# ---------------------------------------------------------------------------------------------------------------------------
# rendering
# ---------------------------------------------------------------------------------------------------------------------------
def _norm(kernels):
    """kernel families of a launch log, grouped by source file: template arguments and launch counts dropped"""
    files = {}
    for part in kernels.split("; "):
        part = re.sub(r" x\d+$", "", part)
        part = re.sub(r"<.*>", "", part)
        f, _, k = part.partition(": ")
        files.setdefault(f, [])
        if k not in files[f]:
            files[f].append(k)
    return "; ".join(f"{f}: {', '.join(ks)}" for f, ks in files.items())


def render(path):
    rows = json.load(open(path))["rows"]
    out = [BEGIN,
           f"Generated by `python tools/dispatch_table.py --write {os.path.relpath(path, ROOT)}` from launch logs measured on one MI355X (`c3p_last_kernel_detail`); "
           "`tests/test_abi_and_host.py` checks this block against the committed JSON, `tests/test_gpu_round6.py` re-measures sample rows on the GPU box.",
           "",
           "| entry (C ABI `c3p_*`) | regime | D | kernels launched (file: kernel), in order | example (first D of the range) |",
           "|---|---|---|---|---|"]
    i = 0
    while i < len(rows):
        r = rows[i]
        j = i
        while j + 1 < len(rows) and rows[j + 1]["entry"] == r["entry"] and rows[j + 1]["regime"] == r["regime"] and _norm(rows[j + 1]["kernels"]) == _norm(r["kernels"]):
            j += 1
        dims = [x["D"] for x in rows[i : j + 1]]
        dtxt = str(dims[0]) if len(dims) == 1 else f"{dims[0]} .. {dims[-1]} (measured: {', '.join(map(str, dims))})"
        ex = r["kernels"].replace("|", "\\|")
        if len(ex) > 260:
            ex = ex[:257] + "..."
        out.append(f"| `{r['entry']}` | {r['regime']} | {dtxt} | {_norm(r['kernels']) if r['family'] != 'refused' else 'REFUSED'} | `{ex}` |")
        i = j + 1
    out.append(END)
    return "\n".join(out)


def main():
    a = sys.argv[1:]
    if not a:
        raise SystemExit(__doc__)
    if a[0] == "--measure":
        measure(a[1])
        return
    if a[0] == "--probe":
        kw = {}
        for x in a[2:]:
            k, v = x.split("=")
            kw[k] = v if k == "flags" else (v == "True" if k == "real" else int(v))
        print(probe(a[1], **kw))
        return
    text = render(a[1])
    doc = os.path.join(ROOT, "INTEGRATION.md")
    s = open(doc).read()
    if a[0] == "--write":
        if BEGIN in s:
            s = s[: s.index(BEGIN)] + text + s[s.index(END) + len(END) :]
        else:
            s = s.rstrip() + "\n\n## 8. Dispatch table: which kernel a call lands in\n\n" + text + "\n"
        open(doc, "w").write(s)
    elif a[0] == "--check":
        if BEGIN not in s or s[s.index(BEGIN) : s.index(END) + len(END)] != text:
            print("INTEGRATION.md's dispatch table is not the rendering of", a[1])
            raise SystemExit(1)
    else:
        print(text)


if __name__ == "__main__":
    main()
