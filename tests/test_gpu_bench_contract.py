"""The driver's contract for bench.py and __graft_entry__ (one JSON line with roofline and cpu_baseline; smoke()
checks the HIP path against the oracle), exercised end to end on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "6", *extra],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # ONE JSON line
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _run_bench()
    assert d["metric"].startswith("frames/sec STFT->chromagram") and d["unit"] == "frames/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 60, 6)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "model" not in d["config"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["frames_per_gpu"] == 8192
    assert d["value"] == pytest.approx(8192 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert 1e7 < d["value"] < 2e9                  # between a broken launch and the HBM roofline (1.93e9 frames/s)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(4144 * 8192 / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)   # algorithmic bytes / kernel time
    assert 0.0 < r["kernel_ms"] <= r["step_ms_hip_events"] * 1.05     # the kernel is inside the (one-at-a-time) step
    assert d["config"]["batches_in_flight"] == 2 and r["in_flight"]["batches"] == 2
    assert d["ms_per_step"] <= r["step_ms_hip_events"] * 1.05         # two batches in flight are not slower than one
    assert r["in_flight"]["frac"] == pytest.approx(4144 * 8192 / (d["ms_per_step"] * 1e-3) / 8.0e12, rel=1e-6)
    assert r["traffic"] is None or r["traffic"] >= 0.9 * 4144 * 8192
    assert r["secondary"]["bound"] == "valu_f64" and 0.0 < r["secondary"]["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "frames/s" and c["value"] > 0 and "sample" in c
    assert d["value"] > 100 * c["value"]


def test_bench_without_cpu_leg_and_smoke():
    d = _run_bench("--no-cpu-baseline", "--streams", "1")
    assert d["config"]["batches_in_flight"] == 1
    assert "roofline" in d and d.get("cpu_baseline") in (None, {}) or "cpu_baseline" not in d
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stderr[-2000:]
