"""bench.py's output contract, checked on the GPU box: exactly one JSON line with
the driver's keys, the roofline and cpu_baseline objects, and consistent arithmetic."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "20",
                          "--no-also", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 20
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "4 096" in d["config"]["workload"]
    # value = envs * steps / time
    assert abs(d["value"] - d["config"]["total_envs"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["bytes_per_launch"] == 84 * d["config"]["envs_per_gpu"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["launch_us_events"] / 1e3) / r["achieved"] < 1e-6
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 1e7          # BASELINE.json's target for one MI355X
    assert d["sanity"]["finite"]


def test_bench_collective_path_with_one_rccl_rank():
    """The multi-rank code path (process group over RCCL, in-line statistics all-reduce
    every STATS_EVERY steps, barrier, max-reduce) rehearsed with world size 1: still one
    JSON line on stdout (RCCL's banner must not reach it), the all-reduced statistics
    count every env step up to the last reduce, and the collective costs no throughput."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29633", ROBOY_BENCH_DIST_AT_1="1", ROBOY_BENCH_STATS_EVERY="400")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1600",
                          "--warmup", "100", "--no-also", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 1600
    stats = d["sanity"]["allreduced_stats"]
    # n_env_steps at the last reduce: 16 decorrelation + 100 warm-up + 400 rehearsal + 1600 timed steps
    assert stats[6] == 4096.0 * (16 + 100 + 400 + 1600)
    assert d["ms_per_step"] < 0.0035      # 2.2-2.3 us per step without a process group; 3.7 with a concurrent collective
