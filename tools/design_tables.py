#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 7 / 5.7 / 6 GENERATED from the committed measurement files of a round
(profiles/<round>/*.json), so that the document cannot drift from the numbers.

    python tools/design_tables.py r04            # prints the tables
    python tools/design_tables.py r04 --write    # rewrites them in DESIGN.md (between the BEGIN / END generated markers)
    python tools/design_tables.py r04 --check    # exit 1 if DESIGN.md does not hold exactly the generated tables
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(rnd, name):
    p = os.path.join(ROOT, "profiles", rnd, name)
    try:
        return json.load(open(p))
    except Exception:
        return None


def f3(x, nd=3):
    if x is None:
        return ""
    return f"{x:.{nd}g}"


def busy(r):
    m = r.get("issue_busy_model")
    if not m:
        return ""
    return f"{m['total']:.2f} = {m['mfma']:.2f} + {m['valu_f64']:.2f} + {m['valu_other']:.2f}"


def bench_table(rnd):
    out = ["| config | batch / GPU | kernel | propagators/s | ms / step | frac (useful issued) | issued / peak | tile utilisation | issue busy (model: MFMA + fp64 VALU + other) | frac_algorithmic | HBM-side bytes / launch | max_b err vs oracle |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for tag, fn in (("cfg1", "bench_cfg1.json"), ("**cfg2 (headline)**", "bench_cfg2.json"), ("cfg2, complex control operator", "bench_cfg2_complex.json"),
                    ("cfg3", "bench_cfg3.json"), ("cfg4", "bench_cfg4.json"), ("cfg5", "bench_cfg5.json")):
        d = load(rnd, fn)
        if not d:
            continue
        r = d["roofline"]
        tr = r.get("traffic")
        trs = "" if tr is None else (f"{tr / 1e6:.1f} MB" if tr < 1e9 else f"{tr / 1e9:.2f} GB")
        out.append(f"| {tag}: {d['config']['workload']} | {d['config']['batch_per_gpu']} | {d['config'].get('kernel', '')} | {f3(d['value'], 4)} | {f3(d['ms_per_step'], 4)} | "
                   f"{f3(r.get('frac'))} | {f3(r.get('issued_frac'))} | {f3(r.get('mfma_tile_utilisation'), 4)} | {busy(r)} | {f3(r.get('frac_algorithmic'))} | {trs} | "
                   f"{f3(d.get('max_fro_err_vs_oracle'), 2)} |")
    return "\n".join(out)


def exchange_table(rnd):
    d = load(rnd, "bench_cfg2_rccl_default.json")
    g = load(rnd, "bench_cfg2_rccl_goal.json")
    if not d:
        return ""
    out = ["| schedule (one rank under torch.distributed.run, RCCL) | ms / step | propagators/s |", "|---|---|---|"]
    out.append(f"| default: {d['config']['parallelism']} | {f3(d['ms_per_step'], 4)} | {f3(d['value'], 4)} |")
    for k, v in (d.get("other_exchange_schedules") or {}).items():
        if isinstance(v, dict) and "value" in v:
            out.append(f"| {k} (extra key of the same line) | {f3(v['ms_per_step'], 4)} | {f3(v['value'], 4)} |")
    if g:
        out.append(f"| `--exchange goal`: {g['config']['parallelism']} | {f3(g['ms_per_step'], 4)} | {f3(g['value'], 4)} |")
    return "\n".join(out)


def goal_table(rnd):
    d = load(rnd, "goal_run.json")
    if not d:
        return ""
    out = ["| case | B | N | forward | forward, graph replay | three calls | fused | fused, one hipGraph | best / forward | evaluations/s | gradients/s |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    ms = lambda x: "" if x is None else f"{x:.3g} ms"
    for r in d["rows"]:
        out.append(f"| {r['case']} | {r['B']} | {r['N']} | {ms(r['forward_ms'])} | {ms(r.get('forward_graph_ms'))} | {ms(r['three_call_ms'])} | {ms(r.get('fused_ms'))} | "
                   f"{ms(r.get('fused_graph_ms'))} | {r['best_over_forward']:.2f} | {r['iterations_per_s']:.3g} | {r['gradients_per_s']:.3g} |")
    return "\n".join(out)


def grad_table(rnd):
    out = ["| config | B | forward | gradient | x forward | gradients/s |", "|---|---|---|---|---|---|"]
    for c in (2, 3, 5):
        d = load(rnd, f"grad_cfg{c}.json")
        if not d:
            continue
        f, g = d.get("forward_ms"), d.get("vjp_ms") or d.get("grad_ms")
        if f is None or g is None:
            continue
        out.append(f"| cfg{c} | {d.get('B', d.get('batch', ''))} | {f:.3g} ms | {g:.3g} ms | {g / f:.2f} | {1e3 * d.get('B', d.get('batch', 0)) / g:.3g} |")
    return "\n".join(out) if len(out) > 2 else ""


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r04"
    parts = {"bench": bench_table(rnd), "exchange": exchange_table(rnd), "goal": goal_table(rnd), "grad": grad_table(rnd)}
    path = os.path.join(ROOT, "DESIGN.md")
    doc = open(path).read()
    if "--write" in sys.argv or "--check" in sys.argv:
        # DESIGN.md holds each table between <!-- BEGIN generated:NAME rNN --> and <!-- END generated:NAME -->
        new = doc
        for name, tab in parts.items():
            b, e = f"<!-- BEGIN generated:{name} {rnd} -->", f"<!-- END generated:{name} -->"
            if b in new and e in new and tab:
                i, j = new.index(b) + len(b), new.index(e)
                new = new[:i] + "\n" + tab + "\n" + new[j:]
        if "--check" in sys.argv:
            if new != doc:
                print("DESIGN.md is out of date: run python tools/design_tables.py", rnd, "--write")
                sys.exit(1)
            print("DESIGN.md holds the generated tables")
            return
        open(path, "w").write(new)
        print("DESIGN.md updated")
        return
    for name, tab in parts.items():
        print(f"<!-- {name} -->\n{tab}\n")


if __name__ == "__main__":
    main()
