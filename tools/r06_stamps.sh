R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
C3P_LIB=c3_amd/libc3prop_timing.so python - > $O/timing_stamps.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from c3_amd import propagation as prop, _lib
from c3_amd.workloads import make_workload
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(30): x @ x
torch.cuda.synchronize()
w = make_workload(2, B=256)
h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
for skew in (None,):
    _lib.set_option("mw_skew", skew)
    print("skew", skew, flush=True)
    for _ in range(4):
        prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)
    torch.cuda.synchronize()
PY
grep -A12 "^skew" $O/timing_stamps.txt | tail -60
