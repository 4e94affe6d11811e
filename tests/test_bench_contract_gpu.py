"""bench.py's output contract, checked on the GPU box: exactly one JSON line on stdout, at most 4 096 bytes, strict
JSON, with the driver's keys, the roofline and cpu_baseline objects, and consistent arithmetic; the long form of the
secondary workloads goes to bench_also.json and stderr, never stdout."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "repeats", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "collective", "sanity")
LINE_CAP = 4096


def _strict(text):
    def refuse(token):
        raise ValueError("non-finite JSON constant %r" % token)
    return json.loads(text, parse_constant=refuse)


def _check_line(stdout):
    """Exactly one non-empty stdout line, under the cap, strict JSON, every contract key."""
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, [l[:200] for l in lines]
    assert len(lines[0].encode()) <= LINE_CAP, len(lines[0])
    d = _strict(lines[0])
    for key in CONTRACT_KEYS:
        assert key in d, key
    assert "also" not in d
    return d


def _run(args, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return _check_line(out.stdout)


def _also():
    with open(os.path.join(ROOT, "bench_also.json")) as fh:
        return _strict(fh.read())


def _check_roofline(r, envs):
    assert r["bound"] in ("hbm", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    # the line carries 5 significant digits
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]
    h, v = r["hbm"], r["valu"]
    assert h["peak"] == 8000.0 and h["bytes_per_launch"] == 84 * envs
    assert abs(h["achieved"] - h["bytes_per_launch"] / r["launch_us_events"] / 1e3) / h["achieved"] < 1e-4
    assert v["peak"] == 157.3 and abs(v["achieved"] - v["flops_per_launch"] / r["launch_us_events"] / 1e6) / v["achieved"] < 1e-4
    assert r["frac"] == max(h["frac"], v["frac"])
    assert r["traffic"] is None or r["traffic"] >= 0.9 * h["bytes_per_launch"]


def test_the_drivers_exact_command_prints_one_capped_line():
    """The driver's argv, unchanged (`--gpus 1 --steps 20 --warmup 5`: secondary workloads and CPU baseline included): ONE
    stdout line of at most 4 096 bytes with every contract key, roofline.configs filled; the headline is the 262 144-env
    workload and the per-step time does not degrade into launch latency."""
    d = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["timing_protocol"] == 3
    assert "workload" in d["config"] and "model" not in d["config"] and "262 144" in d["config"]["workload"]
    assert d["config"]["envs_per_gpu"] == 262144 and d["config"]["integrator"] == "rk4"
    assert d["repeats"] >= 3 and d["repeats"] * 20 * d["ms_per_step"] >= 40.0      # >= ~50 ms timed by the wall clock alone
    assert d["event_repeats"] == d["repeats"] and d["ms_per_step"] < d["ms_per_step_with_events"] < 1.15 * d["ms_per_step"]   # (the events' own price)
    # value = envs * steps / time
    assert abs(d["value"] - d["config"]["total_envs"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    _check_roofline(r, 262144)
    assert r["bound"] == "valu"       # four acceleration evaluations per env step
    assert r["launches_per_step"] == 2 and r["one_launch_us"] > r["launch_us_events"] and len(r["note"]) <= 120
    assert r["configs_cols"] == ["us_events", "hbm_frac", "valu_frac", "launches_per_step"]
    for w in ("msj-4096-euler", "upper-body-8192-euler", "upper-body-8192-rk4", "msj-262144-euler", "msj-2097152-euler", "fused-env-2097152"):
        us, hbm, valu, launches = r["configs"][w]
        assert us > 0 and hbm > 0 and launches >= 1, w
    # north_star's HBM evidence readable off the line alone: the Euler shard and the 2 M Euler batch
    assert r["configs"]["msj-262144-euler"][1] > 0.4 and r["configs"]["msj-2097152-euler"][1] > 0.4
    assert set(r["traffic_over_algorithmic_rows"]) == {"msj-262144-euler", "msj-2097152-euler"}
    assert all(0.9 < x < 1.5 for x in r["traffic_over_algorithmic_rows"].values())
    # ... and as SCALARS of `roofline` (the driver's record keeps the scalar fields): both fractions from the events and from the
    # wall clock `value` is built on, the one-launch figure, the three HBM-bound rows, traffic over algorithmic bytes of the headline
    assert r["hbm_frac"] == r["hbm"]["frac"] and r["valu_frac"] == r["valu"]["frac"] == r["frac"]
    assert 0.8 * r["frac"] < r["frac_wall"] < r["frac"] and r["frac_wall"] == r["valu_frac_wall"]
    assert abs(r["frac_wall"] - 3036 * 262144 / (d["ms_per_step"] * 1e-3) / 157.3e12) < 2e-4 * r["frac_wall"]
    assert abs(r["hbm_frac_wall"] - 84 * 262144 / (d["ms_per_step"] * 1e-3) / 8e12) < 2e-4 * r["hbm_frac_wall"]
    assert r["one_launch_frac"] == r["one_launch"]["frac"] < r["frac"]
    assert r["euler_262144_hbm_frac"] == r["configs"]["msj-262144-euler"][1] > 0.4
    assert r["euler_2097152_hbm_frac"] == r["configs"]["msj-2097152-euler"][1] > 0.4
    assert r["fused_env_2097152_hbm_frac"] == r["configs"]["fused-env-2097152"][1] > 0.4
    assert 0.9 < r["traffic_over_algorithmic"] < 1.5 and r["traffic_stale"] in (False, True)
    # wall time per step (barrier + sync around 20 launches + the amortised statistics reduction) close to the device-event time per
    # step: 6-9 % apart on the boxes of rounds 4-5 (HSA_ENABLE_INTERRUPT=0 changes nothing: the waits poll already); 20 % = a bug
    assert d["ms_per_step"] * 1e3 < 1.20 * r["launch_us_events"]
    assert d["ms_per_step_median"] * 1e3 < 1.12 * r["launch_us_events"]      # (the mean carries the outlier regions; the median must stay close)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["python_env_processes"]["processes"] >= 1 and c["python_env_processes"]["value"] > 0
    assert d["value"] > 1e7          # BASELINE.json's target for one MI355X
    assert d["sanity"]["finite"] and d["collective"] is None
    # the long form went to the side file
    names = [a["workload"] for a in _also()["also"]]
    assert "ppo-65536-fused" in names and "fused-rollout-4096" in names


def test_bench_default_run_carries_every_config_with_sanity_fields():
    d = _run(["--no-cpu-baseline"])
    assert d["steps"] == 400 and "262 144" in d["config"]["workload"] and d["cpu_baseline"] is None
    also = _also()["also"]
    names = [a["workload"] for a in also]
    for w in ("msj-4096-euler", "msj-262144-euler", "msj-2097152-euler", "upper-body-8192-euler",
              "upper-body-8192-rk4", "fused-env-2097152", "ppo-65536-fused", "ppo-65536-torch"):
        assert w in names, names
    ppo = {a["workload"]: a for a in also if a["workload"].startswith("ppo-")}
    assert ppo["ppo-65536-fused"]["value"] > 3 * ppo["ppo-65536-torch"]["value"]      # the consumer on the matrix cores
    cfg = d["roofline"]["configs"]                # the driver's record keeps `roofline`: the secondary workloads in compact form
    by = {a["workload"]: a for a in also}
    for w, row in cfg.items():
        assert row[0] > 0 and row[1] > 0 and by[w]["finite"] is True, w
        assert abs(row[0] - by[w]["launch_us_events"]) < 1e-3 * row[0]
    for a in also:
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
    total = 16 + 5 + 20 * (2 * d["repeats"] + 2)               # decorrelation + warm-up + the regions (R by the wall clock, R with events; rehearsal and sizing region)
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
    assert c["ok"] and c["allreduce_calls"] == (16 + 10 + 250 * 7) // 100 + 1 and c["every_steps"] == 100     # --repeats 3: 3 + 3 regions + the rehearsal, no sizing region
    assert c["expected"] == 4096.0 * (16 + 10 + 250 * 7)


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
    d = _check_line(out.stdout)          # the same capped line with the same keys at N > 1
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 2 * 262144 and d["scaling"] == "weak"
    c = d["collective"]
    assert c["ok"] and c["world_size"] == 2 and c["backend"] == "gloo"
    assert c["allreduce_calls"] == (16 + 5 + 20 * 7) // 100 + 1       # 3 + 3 timed regions + the rehearsal (--repeats given: no sizing region)
    assert c["n_env_steps_allreduced"] == 2 * 262144.0 * (16 + 5 + 20 * 7)
    assert d["cpu_baseline"] is None and d["roofline"]["configs"] == {}
    assert d["roofline"]["one_launch"] is None          # N > 1: the one-launch leg runs with --one-launch only


def test_bare_gpus_2_starts_its_own_ranks_over_gloo():
    """`python3 bench.py --gpus 2 --steps 20 --warmup 5` WITHOUT a launcher (the driver's N = 1 command with another N): the
    parent starts torch.distributed.run itself as a child before any GPU call, relays rank 0's one line, exits 0.  Two ranks
    share the box's one GPU, collectives over gloo (throughput means nothing here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ROBOY_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _check_line(out.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["total_envs"] == 2 * 262144
    c = d["collective"]
    assert c["ok"] and c["world_size"] == 2 and c["backend"] == "gloo"
    assert c["n_env_steps_allreduced"] == c["expected"] == 2 * 262144.0 * (16 + 5 + 20 * (2 * d["repeats"] + 2))
    assert d["cpu_baseline"] is None and "starting" in out.stderr
