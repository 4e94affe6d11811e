#!/usr/bin/env python3
"""PPO iteration timing (rollout / update) of the fused path at two batch sizes: bench.py's run_ppo_iteration on its own."""
import sys
import torch
sys.path.insert(0, "/root/repo")
import bench
for n in (65536, 262144):
    r = bench.run_ppo_iteration(torch, n, True)
    print(r["workload"], "rollout_ms %.2f update_ms %.2f timesteps/s %.3e finite %s" % (r["rollout_ms"], r["update_ms"], r["value"], r["finite"]))
