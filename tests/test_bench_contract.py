"""bench.py's output contract (one JSON line with the driver's keys, `roofline` and `cpu_baseline`), checked on a
small batch so that it runs in seconds on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "256", "--steps", "8", "--warmup", "24", "--pipelined-leg"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 24 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 0 and d["solver"]["solved_frac"] == 1.0
    # value = instances x steps / elapsed
    assert abs(d["value"] - 256 * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-6
    # the overlapped-steps figure rides beside value (same K steps, solve_pipeline = 1), never in its place
    wp = d["with_step_pipelining"]
    assert wp["steps"] == 8 and wp["solved_frac"] == 1.0 and wp["value"] > 0
    assert abs(wp["value"] - 256 * 8 / (wp["ms_per_step"] * 8e-3)) / wp["value"] < 1e-6


@pytest.mark.gpu
def test_bench_at_the_8_gpu_per_rank_batch_on_one_gpu():
    """BASELINE config 4 gives every rank 8192 instances; `bench.py --gpus 8` has never had its 8 GPUs.  What one GPU can keep warm:
    the very command line a rank runs, at the per-rank batch, with a single-rank world — workload string from the actual batch, no
    collective, no committed PMC figure quoted for a batch it was not collected at, every instance solved on the three-workgroup
    kernel.  The first real --gpus 8 lease is then a matter of RCCL only."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8192", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["window_fill_steps_before_warmup"] == 62
    assert d["config"]["workload"] == "Go1, batch=8192 synthetic IMU+encoder+vision streams, 20-step MHE, 1xMI355X"
    assert d["config"]["batch_per_gpu"] == 8192 and d["config"]["global_batch"] == 8192
    assert d["config"]["allgather"] == "none"
    assert d["roofline"]["traffic"] is None and "batch 8192" in d["roofline"]["traffic_source"]
    assert d["roofline"]["kernel"] == "k_mhe_solve_r3_4_n20" and d["roofline"]["units_per_launch"] == 8192
    assert d["roofline"]["solve_workgroups"] == 768
    assert d["solver"]["solved_frac"] == 1.0
    assert abs(d["value"] - 8192 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6
