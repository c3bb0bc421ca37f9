"""PCIe-inclusive rates at cfg2 (never the reported `value`): host numpy in / out through C3P_HOST_PTRS, and the
parameter-row route (envelope rows in, signals synthesised on the device, U out)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import propagation as prop, signals as sg
from c3_amd.workloads import make_workload
w = make_workload(2)
def timed(fn, reps=20):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
host = timed(lambda: prop.propagate_batch(w.h0, w.hks, w.signals, w.dt, fr_phase=w.fr_phase))
dev = "cuda:0"
h0, hks, sig, ph = (torch.as_tensor(x, device=dev) for x in (w.h0, w.hks, w.signals, w.fr_phase))
def resident():
    prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph); torch.cuda.synchronize()
res = timed(resident)
T = w.N * w.dt
TWO_PI = 2 * np.pi
rng = np.random.default_rng(0)
chans = [[dict(shape="gaussian_nonorm", amp=rng.uniform(0.1, 0.6, w.B), xy_angle=0.0, freq_offset=-50e6 * TWO_PI, t_final=T, sigma=T / 4, use_t_before=True)],
         [dict(shape="gaussian_nonorm", amp=rng.uniform(0.1, 0.6, w.B), xy_angle=0.3, freq_offset=-50e6 * TWO_PI, t_final=T, sigma=T / 4, use_t_before=True)]]
env, shapes = sg.pack_components(chans, B=w.B)
car = np.tile(np.array([[5.05e9 * TWO_PI, 1e9 * TWO_PI], [5.65e9 * TWO_PI, 1e9 * TWO_PI]]), (w.B, 1, 1))
def rows():
    s = sg.synthesize_signals(torch.as_tensor(env, device=dev), shapes, torch.as_tensor(car, device=dev), 0.0, T, 2e9, 100e9)
    U = prop.propagate_batch(h0, hks, s, w.dt, fr_phase=ph)["U"].cpu()
rw = timed(rows)
print(json.dumps({"config": w.name, "B": w.B, "host_numpy_in_out_ms": host, "host_numpy_props_per_s": w.B / host * 1e3,
                  "device_resident_ms": res, "device_resident_props_per_s": w.B / res * 1e3,
                  "parameter_rows_in_U_out_ms": rw, "parameter_rows_props_per_s": w.B / rw * 1e3,
                  "bytes_in_signals": int(w.signals.nbytes), "bytes_in_rows": int(env.nbytes + car.nbytes), "bytes_out": int(w.B * 81 * 16)}))
