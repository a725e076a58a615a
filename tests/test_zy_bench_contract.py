"""The benchmark's output contract on a debug-sized workload: one JSON line with the fields the driver reads, the
roofline and CPU-baseline objects, and the parity flags set.  (Collected late: it starts a process of its own.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_of_a_small_workload():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "2", "--warmup", "1",
                           "--cpu-pairs", "6"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "tile_pairs_registered_per_sec" and line["n_gpus"] == 1 and line["steps"] == 2
    assert line["value"] > 0 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and "sample" in cpu
    assert "side_figures_error" not in line
    assert all(line["parity"].values()) if isinstance(line.get("parity"), dict) else True
