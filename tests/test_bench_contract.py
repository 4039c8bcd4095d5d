"""bench.py's output contract: one JSON line with the fields the driver reads, the roofline and
the CPU baseline objects.  CPU: the workload table against BASELINE.json and the CPU-baseline leg
on the smallest workload.  GPU: a short run of the real command."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workloads_are_the_baseline_configurations():
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    text = json.dumps(base)
    H, W, N, kind, fp = bench.WORKLOADS["kodak30k"]
    assert (H, W, N, kind, fp) == (512, 768, 30000, "cholesky", 1.0)
    assert "768" in text and "512" in text and "30" in text
    assert bench.WORKLOADS["uhd200k"][:3] == (2160, 3840, 200000)
    assert bench.WORKLOADS["kodak30k_rs"][3] == "rs"
    # algorithmic bytes of one rasteriser launch: 40 B per intersection, 12 B per pixel, 8 B per tile
    assert bench.algorithmic_bytes_raster(73429, 393216, 1536) == 40 * 73429 + 12 * 393216 + 8 * 1536


def test_cpu_baseline_leg_reports_what_it_sampled():
    import bench
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    r = bench.time_cpu_oracle("kodak5k", budget_s=0.5)
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["value"] > 0
    assert r["cores"] >= 1 and "kodak5k" in r["sample"]


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup",
                          "3", "--no-extras", "--cpu-budget-s", "1"], capture_output=True, text=True,
                         timeout=240, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 3 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    # one step renders one batch of frames: value [frames/s] x ms_per_step = frames per step
    B = d["config"]["frames_per_step"]
    assert B >= 1 and abs(d["value"] * d["ms_per_step"] * 1e-3 - B) < 1e-6 * B
    assert d["timed_regions"] >= 3 and d["frames_timed"] == d["timed_regions"] * 24 * B
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert r["bytes_with_lists_held"] <= r["algorithmic_bytes_per_launch"]
