# round 6, first GPU visit: baseline cfg2 timing, in-kernel time stamps (timing build), changed tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python tools/time_cfg2.py c3_amd/libc3prop.so > $O/base_time.txt 2>&1
python - > $O/timing_stamps.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath("c3_amd/libc3prop_timing.so")
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(30): x @ x
torch.cuda.synchronize()
w = make_workload(2, B=256)
h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
for _ in range(6):
    prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)
torch.cuda.synchronize()
PY
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_tf_bridge.py tests/test_bench_contract.py -x -q -m gpu > $O/tests_partial.txt 2>&1
tail -3 $O/base_time.txt; tail -4 $O/timing_stamps.txt; tail -3 $O/tests_partial.txt
