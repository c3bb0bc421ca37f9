"""The driver's bench.py contract: one JSON line with the agreed keys (GPU box only)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line(lib):
    from c3_amd import _lib

    _lib.require_gpu()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--check"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]  # stdout is EXACTLY the one JSON line (library banners go to stderr)
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "full-gate propagators/s" and d["unit"] == "propagators/s"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "cfg2" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert r["bound"] in ("hbm", "mfma")
    # frac = useful issued flops over the peak, quoted only for the exact build / batch / slices the committed PMC profile
    # was taken on; otherwise null (never scaled from another run, never clamped) and only frac_algorithmic is given
    if r["pmc_exact_match"]:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] <= 1.0 and r["traffic"] is not None
    else:
        assert r["frac"] is None and r["achieved"] is None and r["traffic"] is None
    assert r["frac_algorithmic"] > 0.0 and "device_ms_per_step" in r
    c = d["cpu_baseline"]
    assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and c["kind"] in ("reference", "port") and c["cores"] >= 1
    # the CPU legs are bounded: they must not hold the GPU lease for minutes
    assert c["wall_s"] < 15.0 and d["cpu_baseline_allcores"]["wall_s"] < 40.0
    assert abs(d["value"] - 256 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert d["max_fro_err_vs_oracle"] < 1e-10


@pytest.mark.gpu
def test_bench_two_ranks_on_the_visible_devices(lib):
    """`python bench.py --gpus 2` launches its own two ranks.  On a one-GPU box they share the device (RCCL refuses that, so the
    slabs travel over gloo through host memory -- flagged `oversubscribed`); on a node with >= 2 GPUs the same command is the
    RCCL run.  Either way: sharding, every exchange schedule and the gathered check (own slab + the other rank's) execute."""
    import torch
    from c3_amd import _lib

    _lib.require_gpu()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "128"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 256 and d["config"]["batch_per_gpu"] == 128
    la = d["launch"]
    assert la["oversubscribed"] == (torch.cuda.device_count() < 2)
    assert la.get("rccl_world_size", la.get("gloo_world_size")) == 2
    assert d["max_fro_err_vs_oracle"] < 1e-10 and d["gathered_samples_of_other_ranks_checked"] == 2
    assert set(d["other_exchange_schedules"]) >= {"all_gather_every_32_steps", "goal_all_reduce_every_step"}
    assert {"all_gather_every_step", "all_gather_every_step_overlapped"} & set(d["other_exchange_schedules"])
    assert d["metric"] == "full-gate propagators/s"


def test_design_tables_are_generated_from_the_committed_profiles():
    """DESIGN.md's measurement tables are the output of tools/design_tables.py over profiles/r06/*.json"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "design_tables.py"), "r06", "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
