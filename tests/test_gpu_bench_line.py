"""bench.py's contract with the driver, on the GPU: stdout is ONE JSON line -- nothing the library prints on the way may
share it -- carrying the metric, the roofline object and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_exactly_one_json_line(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--cpu-frames", "2", "--no-pcie"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "frames/s" and d["dtype"] == "u8"
    assert d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    rl = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12 and rl["unit"] == "GB/s"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


def test_bench_two_ranks_launch_themselves_and_print_one_line(gpu):
    """`--gpus 2` with no launcher around it: bench.py starts the two ranks itself (here both on the one GPU, which
    --oversubscribe allows for exactly this rehearsal); rank 0 prints the line, with the job's aggregate rate."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--cpu-frames", "0", "--no-pcie"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["output_frames_per_step_all_gpus"] == 16 and d["config"]["frames_per_step"] == 8
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only
