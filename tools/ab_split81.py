"""A/B of the core + border form of the real small-D path (8 + 1 split at D = 9; c3p_smalld.hip SMat / GMat) against the padded
12 x 12 tiles (no_split81 = 1) at cfg2's operators: ms per batch (HIP events over `reps` launches after a clock ramp), the same
process, alternating.      python tools/ab_split81.py --out gpurun_out/final/ab_split81.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import propagation as prop
from c3_amd import _lib
from c3_amd.workloads import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--batches", default="256,512,1024")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
dev = "cuda:0"
t = lambda x: torch.as_tensor(x, device=dev)
rows = []
for B in (int(x) for x in a.batches.split(",")):
    w = make_workload(a.config, B=B)
    bp = prop.BatchPropagator(t(w.h0), t(w.hks), t(w.signals), w.dt, fr_phase=t(w.fr_phase))
    out = {}
    res = {}
    for rnd in range(3):
        for name, opt in (("core_plus_border", None), ("padded_tiles", 1)):
            _lib.set_option("no_split81", opt)
            for _ in range(300):  # clock ramp
                bp.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                U = bp.run()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(name, []).append(e0.elapsed_time(e1) / a.reps)
            res[name] = U.clone()
    _lib.set_option("no_split81", None)
    ms = {k: float(np.median(v)) for k, v in out.items()}
    row = {"config": w.name, "B": B, "N": w.N, "ms_per_batch": ms, "propagators_per_s": {k: 1e3 * B / v for k, v in ms.items()},
           "gain": ms["padded_tiles"] / ms["core_plus_border"], "max_abs_diff": float((res["core_plus_border"] - res["padded_tiles"]).abs().max())}
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"rows": rows}, open(a.out, "w"), indent=1)
