import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import propagation as prop, _lib
from c3_amd.workloads import make_workload
t = lambda x: torch.as_tensor(x, device="cuda:0")
w = make_workload(2, B=256)
bp = prop.BatchPropagator(t(w.h0), t(w.hks), t(w.signals), w.dt, fr_phase=t(w.fr_phase))
def run(reps=300):
    for _ in range(300): bp.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): bp.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for skew in (640, 660, 680, 700, 720, 740):
    _lib.set_option("mw_skew", skew)
    print("skew", skew, "ms", round(min(run(), run()), 5), flush=True)
_lib.set_option("mw_skew", None)
for seg in (16, 32, 64):
    _lib.set_option("smalld_segments", seg)
    print("segments", seg, "ms", round(min(run(), run()), 5), flush=True)
_lib.set_option("smalld_segments", None)
