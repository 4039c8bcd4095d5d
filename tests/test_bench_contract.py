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


def test_gpus_without_a_launcher_starts_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` typed without torch.distributed.run: the script runs its ranks
    as a child launcher process with the same arguments and hands back their exit status."""
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    # enough GPUs: RCCL (the default backend) is left alone; fewer GPUs than ranks: a gloo rehearsal
    assert "GI2D_BENCH_BACKEND" not in bench.launch_env(4, 8, {})
    assert bench.launch_env(4, 1, {})["GI2D_BENCH_BACKEND"] == "gloo"
    assert bench.launch_env(4, 1, {"GI2D_BENCH_BACKEND": "nccl"})["GI2D_BENCH_BACKEND"] == "nccl"
    assert bench.launch_env(2, 2, {})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    assert seen["cmd"][-4:] == ["--gpus", "2", "--steps", "3"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and \
        seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "2"


@pytest.mark.gpu
def test_gpus_2_typed_without_a_launcher_reaches_the_rank_code():
    """On a box with one GPU the two ranks share it and exchange through gloo: the line is a
    rehearsal of the multi-rank path and says so."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["GI2D_SHARD_TIMEOUT_S"] = "200"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
                          "10", "--warmup", "2"], capture_output=True, text=True, timeout=420,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "tile_row_sharded" in d and "error" not in d["tile_row_sharded"], d["tile_row_sharded"]
    import torch
    if torch.cuda.device_count() < 2:
        assert d["config"]["collective_backend"] == "gloo" and d["config"]["ranks_share_gpus"]


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
