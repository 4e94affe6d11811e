"""bench.py's contract line (CPU): the formatter that writes the one stdout line keeps a synthetic worst case under
LINE_CAP = 4 096 bytes, with every contract key, as strict JSON (a parser that rejects NaN / Infinity reads it)."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "repeats", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "collective", "sanity")
LONG = math.pi * 1e9 / 7.0          # a float whose repr takes 17-18 characters


def strict_loads(text):
    def refuse(token):
        raise ValueError("non-finite JSON constant %r" % token)
    return json.loads(text, parse_constant=refuse)


def _roofline(n_envs, us, chains=1):
    return {"bound": "valu", "achieved": LONG / 1e8, "peak": 157.3, "unit": "TFLOP/s", "frac": LONG / 1e10, "traffic": LONG,
            "traffic_source": "profiles/r5_a/hbm_traffic_pmc.json", "launch_us_events": us, "launches_per_step": chains,
            "envs_per_launch": n_envs // chains,
            "hbm": {"achieved": LONG / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": LONG / 1e10, "frac_of_measured_copy_6.29TBps": 0.2,
                    "bytes_per_env_step": 84, "bytes_per_launch": 84 * n_envs},
            "valu": {"achieved": LONG / 1e8, "peak": 157.3, "unit": "TFLOP/s", "frac": LONG / 1e10,
                     "frac_of_measured_fma_rate_109TFLOPs": 0.6, "flops_per_env_step": 3036, "flops_per_launch": 3036 * n_envs,
                     "source": "profiles/flops_per_env_step.json (instrumented restatement, oracle/flop_count.cpp)"}}


def worst_case(world=8, nan=False):
    n = 262144
    head = {"workload": "msj-262144-rk4", "label": bench.WORKLOADS["msj-262144-rk4"][5], "envs_per_gpu": n, "integrator": "rk4",
            "substeps": 1, "steps": 20, "warmup": 5, "repeats": 256, "value": LONG * 10, "ms_per_step": LONG / 1e11,
            "ms_per_step_min": LONG / 1e11, "ms_per_step_max": LONG / 1e11, "ms_per_step_median": LONG / 1e11,
            "ms_per_step_with_events": LONG / 1e11, "event_repeats": 256,
            "timed_device_ms": LONG / 1e7, "launch_us_events": LONG / 1e8, "roofline": _roofline(n, LONG / 1e8, 2),
            "kernel": "msj_step_env_per_lane", "stats": [-LONG * 1e3, LONG * 1e6, 896794624.0, LONG, LONG, LONG, 8.0 * n * 3421, LONG],
            "collective": {"backend": "rccl (torch.distributed 'nccl')", "world_size": world, "allreduce_calls": 1234,
                           "payload_bytes": 64, "every_steps": 100, "us_per_allreduce": LONG / 1e8,
                           "n_env_steps_allreduced": 8.0 * n * 3421, "expected": 8.0 * n * 3421, "ok": True},
            "finite": True, "feasible_frac": LONG / 1e10}
    if nan:
        head["roofline"]["traffic"] = float("nan")
        head["feasible_frac"] = float("inf")
        head["stats"][0] = float("nan")
    also = []
    for w in bench.CONFIG_ROWS + ("msj-2097152-rk4", "fused-rollout-4096", "fused-env-UpperBodyRobot-8192"):
        also.append({"workload": w, "label": "x" * 120, "value": LONG, "ms_per_step": LONG / 1e11, "launch_us_events": LONG / 1e8,
                     "steps": 300, "repeats": 10, "kernel": "tree_split_step", "roofline": _roofline(65536, LONG / 1e8, 2),
                     "finite": True, "feasible_frac": 0.99})
    also.append({"workload": "ppo-65536-fused", "value": LONG, "unit": "timesteps/s", "rollout_ms": 3.0, "update_ms": 20.0,
                 "steps": 256, "finite": True, "feasible_frac": 0.9})
    one = {"us_events": LONG / 1e8, "ms_per_step": LONG / 1e11, "value": LONG * 10, "frac": LONG / 1e10, "hbm_frac": LONG / 1e10, "repeats": 64}
    cpu = {"value": LONG / 100, "unit": "env-steps/s", "cores": 128, "kind": "port",
           "sample": "262144 envs x 1234 steps of msj-262144-rk4, C fp32 restatement (oracle/roboy_oracle.c), OpenMP over envs",
           "by_threads": {str(t): {"value": LONG / 100, "envs": n, "steps": 1234} for t in (1, 64, 128)},
           "value_1_core": LONG / 1000, "host_cores_available": 128,
           "python_env_processes": {"value": LONG / 1e4, "processes": 64, "per_process": LONG / 1e6, "seconds_each": 4.0,
                                    "cores": 64, "what": "y" * 200}}
    return head, also, one, cpu


@pytest.mark.parametrize("nan", [False, True])
def test_worst_case_line_stays_under_the_cap_and_parses_strictly(nan):
    head, also, one, cpu = worst_case(nan=nan)
    text = bench.format_line(bench.build_line(head, also, one, cpu, 8, "MsjRobot", True))
    assert "\n" not in text and len(text.encode()) <= bench.LINE_CAP == 4096, len(text)
    d = strict_loads(text)
    for key in CONTRACT_KEYS:
        assert key in d, key
    assert "also" not in d
    r = d["roofline"]
    assert set(r["configs"]) == set(bench.CONFIG_ROWS) and all(len(row) == len(r["configs_cols"]) == 4 for row in r["configs"].values())
    assert set(r["traffic_over_algorithmic_rows"]) == set(bench.TRAFFIC_ROWS)
    # the scalars the driver's record keeps: north_star's claims readable without the nested objects
    for key in ("frac", "frac_wall", "hbm_frac", "valu_frac", "hbm_frac_wall", "valu_frac_wall", "one_launch_frac", "one_launch_hbm_frac",
                "euler_262144_hbm_frac", "euler_2097152_hbm_frac", "fused_env_2097152_hbm_frac", "traffic_over_algorithmic", "traffic_stale",
                "valu_frac_of_measured_fma_rate"):
        assert key in r and not isinstance(r[key], (dict, list)), key
    assert r["euler_262144_hbm_frac"] == r["configs"]["msj-262144-euler"][1] and r["fused_env_2097152_hbm_frac"] == r["configs"]["fused-env-2097152"][1]
    assert abs(r["frac_wall"] - 3036 * 262144 / (head["ms_per_step"] * 1e-3) / 157.3e12) < 1e-4 * r["frac_wall"]
    if not nan:
        assert abs(r["traffic_over_algorithmic"] - LONG / (84 * 262144)) < 1e-4 * r["traffic_over_algorithmic"]
    assert r["hbm"]["frac"] > 0 and r["valu"]["frac"] > 0 and r["one_launch_us"] > 0 and len(r["note"]) <= 120
    # value and ms_per_step survive unrounded (the driver recomputes one from the other); counts survive exactly
    assert d["value"] == head["value"] and d["ms_per_step"] == head["ms_per_step"]
    assert d["collective"]["n_env_steps_allreduced"] == 8 * 262144 * 3421 == d["sanity"]["allreduced_stats"][6]
    assert d["timing_protocol"] == 3 and d["ms_per_step_median"] > 0 and d["ms_per_step_with_events"] > 0 and d["event_repeats"] == 256
    assert len(d["cpu_baseline"]["by_threads"]) == 3 and all(isinstance(v, float) for v in d["cpu_baseline"]["by_threads"].values())
    if nan:
        assert r["traffic"] is None and d["sanity"]["feasible_frac"] is None and d["sanity"]["allreduced_stats"][0] is None


def test_minimal_line_without_secondary_workloads():
    head, _, _, _ = worst_case()
    head["collective"] = None
    d = strict_loads(bench.format_line(bench.build_line(head, [], None, None, 1, "MsjRobot", False)))
    assert d["roofline"]["configs"] == {} and d["roofline"]["one_launch"] is None and d["cpu_baseline"] is None
    assert d["collective"] is None and d["config"]["launch"] == "eager per-step launches" and d["n_gpus"] == 1


def test_strict_parser_rejects_what_the_lenient_one_accepts():
    with pytest.raises(ValueError):
        strict_loads('{"a": NaN}')
    assert bench.sanitize({"a": float("nan"), "b": [float("-inf"), 1.5]}) == {"a": None, "b": [None, 1.5]}
    assert bench.sig(123456.789) == 123460.0 and bench.sig(896794624.0) == 896794624 and bench.sig(float("nan")) is None


# ---- `python bench.py --gpus N` without a launcher starts its own ranks (bench.self_launch) ----

_FAKE_RANKS = r"""
import os, sys, time
mode = sys.argv[1]
if mode == "ok":
    print("NCCL version banner that a library wrote to stdout")
    sys.stderr.write("rank noise on stderr\n")
    print('{"metric":"m","value":1.5,"n_gpus":2}')
elif mode == "fail":
    sys.stderr.write("bench.py needs a GPU\n")
    sys.exit(7)
elif mode == "silent":
    pass
elif mode == "hang":
    import subprocess
    child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
    open(sys.argv[2], "w").write("%d %d" % (os.getpid(), child.pid))
    time.sleep(600)
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                                         # a zombie still answers kill(pid, 0)
        with open("/proc/%d/stat" % pid) as fh:
            return fh.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_launcher_argv_is_the_contracts_command():
    cmd = bench.launcher_argv(8, 29555, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[4:10] == ["--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29555"]
    assert cmd[10] == os.path.join(ROOT, "bench.py") and cmd[11:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert 1024 < bench.free_port() < 65536


@pytest.mark.parametrize("mode,rc,line", [("ok", 0, True), ("fail", 7, False), ("silent", 5, False)])
def test_self_launch_relays_one_line_and_the_exit_code(monkeypatch, capfd, mode, rc, line):
    monkeypatch.setattr(bench, "launcher_argv", lambda n, port, argv: [sys.executable, "-c", _FAKE_RANKS, mode])
    assert bench.self_launch(2, ["--gpus", "2"]) == rc
    out, err = capfd.readouterr()
    lines = [l for l in out.splitlines() if l.strip()]
    if line:
        assert lines == ['{"metric":"m","value":1.5,"n_gpus":2}']
        assert "NCCL version banner" in err and "rank noise" in err      # the ranks' stdout noise is not ours
    else:
        assert lines == []


def test_self_launch_kills_the_whole_child_tree_on_its_timeout(monkeypatch, tmp_path, capfd):
    import time
    pidfile = str(tmp_path / "pids")
    monkeypatch.setattr(bench, "launcher_argv", lambda n, port, argv: [sys.executable, "-c", _FAKE_RANKS, "hang", pidfile])
    t0 = time.time()
    assert bench.self_launch(2, ["--gpus", "2"], timeout=3.0) == 124
    assert time.time() - t0 < 30
    pids = [int(p) for p in open(pidfile).read().split()]
    time.sleep(0.5)
    assert len(pids) == 2 and not any(_alive(p) for p in pids)
    assert "did not finish" in capfd.readouterr().err


def test_bare_gpus_2_starts_two_ranks_instead_of_a_usage_exit():
    """Without a GPU the ranks stop at 'bench.py needs a GPU' - but they were started (round 5: exit 1 with a usage message)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-also"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert "starting" in out.stderr and "torch.distributed.run" in out.stderr
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and "needs a GPU" in out.stderr and out.stdout.strip() == ""
