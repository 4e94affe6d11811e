"""bench.py's output contract, checked on the GPU box: exactly one JSON line with
the driver's keys, the roofline and cpu_baseline objects, and consistent arithmetic."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def _check_roofline(r, envs):
    assert r["bound"] in ("hbm", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    h, v = r["hbm"], r["valu"]
    assert h["peak"] == 8000.0 and h["bytes_per_launch"] == 84 * envs
    assert abs(h["achieved"] - h["bytes_per_launch"] / r["launch_us_events"] / 1e3) / h["achieved"] < 1e-6
    assert v["peak"] == 157.3 and abs(v["achieved"] - v["flops_per_launch"] / r["launch_us_events"] / 1e6) / v["achieved"] < 1e-6
    assert r["frac"] == max(h["frac"], v["frac"])
    assert r["traffic"] is None or r["traffic"] >= 0.9 * h["bytes_per_launch"]


def test_bench_prints_one_json_line_with_the_contract_keys():
    """The driver's invocation shape (a small --steps): the headline is the 262 144-env workload and the
    per-step time does not degrade into launch latency."""
    d = _run(["--steps", "20", "--warmup", "5", "--no-also", "--cpu-seconds", "3"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "repeats"):
        assert key in d, key
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "262 144" in d["config"]["workload"]
    assert d["config"]["envs_per_gpu"] == 262144 and d["config"]["integrator"] == "rk4"
    assert d["repeats"] >= 3 and d["repeats"] * 20 * d["ms_per_step"] >= 40.0      # >= ~50 ms timed in all
    # value = envs * steps / time
    assert abs(d["value"] - d["config"]["total_envs"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    _check_roofline(d["roofline"], 262144)
    assert d["roofline"]["bound"] == "valu"       # four acceleration evaluations per env step
    assert d["roofline"]["launches_per_step"] == 2 and d["roofline"]["one_launch_us"] > d["roofline"]["launch_us_events"]
    assert d["roofline"]["configs"] == {}         # --no-also
    # wall time per step (barrier + sync around 20 launches) within 12 % of the device-event time per launch
    assert d["ms_per_step"] * 1e3 < 1.12 * d["roofline"]["launch_us_events"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["python_env_processes"]["processes"] >= 1 and c["python_env_processes"]["value"] > 0
    assert d["value"] > 1e7          # BASELINE.json's target for one MI355X
    assert d["sanity"]["finite"] and d["collective"] is None


def test_bench_default_run_carries_every_config_with_sanity_fields():
    d = _run(["--no-cpu-baseline"])
    assert d["steps"] == 400 and "262 144" in d["config"]["workload"]
    names = [a["workload"] for a in d["also"]]
    for w in ("msj-4096-euler", "msj-262144-euler", "msj-2097152-euler", "upper-body-8192-euler",
              "upper-body-8192-rk4", "fused-env-2097152", "ppo-65536-fused", "ppo-65536-torch"):
        assert w in names, names
    ppo = {a["workload"]: a for a in d["also"] if a["workload"].startswith("ppo-")}
    assert ppo["ppo-65536-fused"]["value"] > 3 * ppo["ppo-65536-torch"]["value"]      # the consumer on the matrix cores
    cfg = d["roofline"]["configs"]                # the driver's record keeps `roofline`: the secondary workloads in compact form
    for w in ("msj-4096-euler", "msj-262144-euler", "msj-2097152-euler", "upper-body-8192-euler", "upper-body-8192-rk4", "fused-env-2097152"):
        assert cfg[w]["us_events"] > 0 and cfg[w]["hbm_frac"] > 0 and cfg[w]["finite"] is True, w
    assert cfg["msj-262144-euler"]["hbm_frac"] > 0.4 and cfg["ppo-65536-fused"]["rollout_us_per_step"] > 0
    for a in d["also"]:
        assert a["finite"] is True and 0.0 <= a["feasible_frac"] <= 1.0, a["workload"]
        if "roofline" in a:
            assert a["roofline"]["hbm"]["frac"] > 0


def test_bench_collective_path_with_one_rccl_rank():
    """The multi-rank code path (process group over RCCL, in-line statistics all-reduce every STATS_EVERY
    steps counted across the timed regions, barrier, max-reduce) rehearsed with world size 1
    at the driver's --steps 20: one JSON line on stdout (RCCL's banner must not reach it), the
    collective object audits the all-reduce, and the collective costs little throughput."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29633", ROBOY_BENCH_DIST_AT_1="1")
    d = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-also", "--no-cpu-baseline"], env=env)
    assert d["n_gpus"] == 1 and d["steps"] == 20
    c = d["collective"]
    assert c["ok"] and c["world_size"] == 1 and c["backend"].startswith("rccl")
    total = 16 + 5 + 20 * (d["repeats"] + 2)                   # decorrelation + warm-up + the regions (rehearsal and sizing region included)
    assert c["allreduce_calls"] == total // 100 + 1            # every 100 steps across the regions, + the closing one for the audit
    assert c["n_env_steps_allreduced"] == c["expected"] == 262144.0 * total
    assert d["roofline"]["one_launch"]["us_events"] > d["roofline"]["launch_us_events"]      # the one-launch form beside the chains
    assert d["sanity"]["allreduced_stats"][6] == c["expected"]
    assert d["ms_per_step"] * 1e3 < 1.25 * d["roofline"]["launch_us_events"]


def test_bench_collective_with_full_chunks():
    """--steps above STATS_EVERY: an all-reduce every STATS_EVERY steps, inside the regions too, and the closing one."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29634", ROBOY_BENCH_DIST_AT_1="1", ROBOY_BENCH_STATS_EVERY="100")
    d = _run(["--gpus", "1", "--steps", "250", "--warmup", "10", "--no-also", "--no-cpu-baseline",
              "--workload", "msj-4096-euler", "--repeats", "3"], env=env)
    c = d["collective"]
    assert c["ok"] and c["allreduce_calls"] == (16 + 10 + 250 * 4) // 100 + 1 and c["every_steps"] == 100     # --repeats given: no sizing region
    assert c["expected"] == 4096.0 * (16 + 10 + 250 * 4)


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """The N > 1 launch shape of the driver (torch.distributed.run, one rank per process) rehearsed with two ranks
    on the one GPU of the box and the tiny collectives over gloo: both ranks run the headline shard, rank 0 prints
    the one line, the all-reduced env-step count covers both ranks.  Throughput means nothing here (the ranks share a GPU)."""
    env = dict(os.environ, ROBOY_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29671", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "20", "--warmup", "5", "--repeats", "3"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 2 * 262144 and d["scaling"] == "weak"
    c = d["collective"]
    assert c["ok"] and c["world_size"] == 2 and c["backend"] == "gloo"
    assert c["allreduce_calls"] == (16 + 5 + 20 * 4) // 100 + 1       # three timed regions + the rehearsal (--repeats given: no sizing region)
    assert c["n_env_steps_allreduced"] == 2 * 262144.0 * (16 + 5 + 20 * 4)
    assert d["cpu_baseline"] is None and d["also"] == []
