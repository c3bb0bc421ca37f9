"""The CPU column of SURVEY 8d at full host width: the numpy oracle (Higham Pade expm per slice + pairwise
tree product, i.e. tf_propagation_vectorized + tf_matmul_n semantics) process-parallel over samples."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import multiprocessing as mp
import numpy as np

def work(args):
    cfg, b0, nb = args
    from c3_amd.workloads import make_workload
    from oracle import c3_oracle as o
    w = make_workload(cfg, B=nb, b_offset=b0)
    t0 = time.perf_counter()
    o.propagate_batch(w.h0, w.hks, w.signals, w.dt, col_ops=w.col_ops, lindbladian=w.lindblad, fr_phase=w.fr_phase)
    return time.perf_counter() - t0

if __name__ == "__main__":
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cores = os.cpu_count()
    with mp.Pool(cores) as pool:
        pool.map(work, [(cfg, 0, 1)] * cores)  # warm the workers (imports)
        t0 = time.perf_counter()
        pool.map(work, [(cfg, i * per, per) for i in range(cores)])
        wall = time.perf_counter() - t0
    print(json.dumps({"config": cfg, "cores": cores, "samples": cores * per, "wall_s": wall, "propagators_per_s": cores * per / wall}))
