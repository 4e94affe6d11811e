"""GPU parity of the fused policy step (gym_roboy_amd/csrc/mlp_policy.hip, through its C ABI) against a plain PyTorch
fp32 statement of the same network (gym_roboy_amd/ppo.py: MlpPolicy, evaluated on the CPU in float64 as the referee).
Tolerances: mean / value 2e-5 (two tanh layers of fp32 fmaf chains with v_exp_f32 / v_rcp_f32 against float64),
log-probability 1e-4."""
import ctypes
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _policy(obs_dim, act_dim, seed):
    import torch
    from gym_roboy_amd.ppo import MlpPolicy
    torch.manual_seed(seed)
    p = MlpPolicy(obs_dim, act_dim)
    with torch.no_grad():                     # not the near-zero last layer of a fresh policy: every weight counts
        for q in p.parameters():
            q.add_(0.3 * torch.randn_like(q))
    return p


def _run(policy, obs, step=0, seed=5, offset=0, deterministic=False, step_base=None):
    import torch
    from gym_roboy_amd.ppo import FusedPolicyStep
    f = FusedPolicyStep(policy.cuda(), seed=seed)
    o = torch.as_tensor(obs, dtype=torch.float32, device="cuda").contiguous()
    n, ad = o.shape[0], f.act_dim
    act = torch.empty(n, ad, device="cuda"); logp = torch.empty(n, device="cuda"); val = torch.empty(n, device="cuda")
    mean = torch.empty(n, ad, device="cuda")
    f.act_into(o, act, logp, val, step=step, mean=mean, sample_offset=offset, deterministic=deterministic, step_base=step_base)
    torch.cuda.synchronize()
    return act.cpu().numpy(), logp.cpu().numpy(), val.cpu().numpy(), mean.cpu().numpy()


@pytest.mark.parametrize("obs_dim,act_dim,n", [(9, 8, 1000), (9, 8, 64), (9, 8, 1), (60, 38, 333), (3, 1, 129), (95, 64, 70), (10, 33, 65)])
def test_mean_value_and_log_probability_match_the_torch_policy(obs_dim, act_dim, n):
    import torch
    policy = _policy(obs_dim, act_dim, obs_dim + act_dim)
    rng = np.random.default_rng(n)
    obs = rng.uniform(-2.0, 2.0, (n, obs_dim)).astype(np.float32)
    ref = _policy(obs_dim, act_dim, obs_dim + act_dim).double()
    with torch.no_grad():
        d = ref.dist(torch.from_numpy(obs).double())
        mean_ref, std_ref = d.mean.numpy(), d.stddev.numpy()
        val_ref = ref.value(torch.from_numpy(obs).double()).numpy()
    act, logp, val, mean = _run(policy, obs)
    assert np.abs(mean - mean_ref).max() < 2e-5 * max(1.0, np.abs(mean_ref).max())
    assert np.abs(val - val_ref).max() < 2e-5 * max(1.0, np.abs(val_ref).max())
    with torch.no_grad():
        logp_ref = d.log_prob(torch.from_numpy(act).double()).sum(-1).numpy()
    assert np.abs(logp - logp_ref).max() < 1e-4 * max(1.0, np.abs(logp_ref).max())
    a_det, lp_det, _, _ = _run(policy, obs, deterministic=True)
    assert np.array_equal(a_det, mean)
    assert np.allclose(lp_det, -np.log(std_ref[0]).sum() - 0.5 * act_dim * math.log(2 * math.pi), atol=1e-5)


def test_noise_is_standard_normal_keyed_by_sample_and_step_and_independent_of_sharding():
    import torch
    policy = _policy(9, 8, 1)
    n = 200_000
    obs = np.random.default_rng(0).uniform(-1, 1, (n, 9)).astype(np.float32)
    act, _, _, mean = _run(policy, obs, step=3)
    std = np.exp(policy.log_std.detach().cpu().numpy())
    eps = (act - mean) / std
    assert abs(eps.mean()) < 0.01 and abs(eps.var() - 1.0) < 0.01 and abs((eps ** 4).mean() - 3.0) < 0.1
    c = np.corrcoef(eps.T)
    assert np.abs(c - np.eye(8)).max() < 0.01                        # dimensions uncorrelated
    assert abs(np.corrcoef(eps[:-1, 0], eps[1:, 0])[0, 1]) < 0.01     # neighbouring samples too
    a2, _, _, _ = _run(policy, obs, step=4)
    assert np.abs(a2 - act).max() > 0.1                               # another step: other noise
    # the same step through a device-side step base (what a captured launch uses)
    base = torch.tensor([3], dtype=torch.int32, device="cuda")
    a3, _, _, _ = _run(policy, obs, step=0, step_base=base)
    assert np.array_equal(a3, act)
    # two shards with their sample offsets reproduce the single batch bit for bit
    h = 77_777
    lo, _, _, _ = _run(policy, obs[:h], step=3)
    hi, _, _, _ = _run(policy, obs[h:], step=3, offset=h)
    assert np.array_equal(np.concatenate([lo, hi]), act)


def test_ppo_rollout_with_the_fused_policy_step_trains():
    """PPO with fused_policy=True (graph mode): rollouts come from the fused kernel, stored log-probabilities agree
    with what the torch policy assigns to the same actions, and an update runs."""
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from gym_roboy_amd.ppo import PPO
    env = RoboyVecEnv(MsjRobot(), 1024, seed=2)
    agent = PPO(env, n_steps=16, use_graphs=True, fused_policy=True, seed=3)
    roll = agent.collect()
    with torch.no_grad():
        lp = agent.policy.dist(roll["obs"]).log_prob(roll["act"]).sum(-1)
        v = agent.policy.value(roll["obs"])
    assert torch.isfinite(roll["act"]).all() and (lp - roll["logp"]).abs().max() < 1e-3
    assert (v - roll["val"]).abs().max() < 1e-4
    first = roll["act"].clone()
    agent.update(roll)
    roll2 = agent.collect()                                           # replay: fresh noise, updated weights
    assert (roll2["act"] - first).abs().max() > 1e-3
    with torch.no_grad():
        lp2 = agent.policy.dist(roll2["obs"]).log_prob(roll2["act"]).sum(-1)
    assert (lp2 - roll2["logp"]).abs().max() < 1e-3
    env.close()
