"""GPU parity of the fused policy step (gym_roboy_amd/csrc/mlp_policy.hip, through its C ABI) against a plain PyTorch
fp32 statement of the same network (gym_roboy_amd/ppo.py: MlpPolicy, evaluated on the CPU in float64 as the referee).
Tolerances: mean / value 2e-5 (two tanh layers of fp32 fmaf chains with v_exp_f32 / v_rcp_f32 against float64),
log-probability 1e-4."""
import ctypes
import math

import os

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


# ---- the PPO minibatch gradient (csrc/mlp_train.hip) against torch autograd in float64 ----
def _minibatch(policy64, obs_dim, act_dim, B, seed, cliprange):
    """A minibatch in which every branch of the two clipped losses occurs: actions around the policy's mean, old
    log-probabilities off by up to +-0.5 (ratios on both sides of the clip range), old values near and far."""
    import torch
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand(B, obs_dim, generator=g, dtype=torch.float64) * 4 - 2
    with torch.no_grad():
        d = policy64.dist(obs)
        act = d.mean + d.stddev * torch.randn(B, act_dim, generator=g, dtype=torch.float64)
        logp = d.log_prob(act).sum(-1)
        v = policy64.value(obs)
    logp_old = logp + (torch.rand(B, generator=g, dtype=torch.float64) - 0.5)
    adv = torch.randn(B, generator=g, dtype=torch.float64)
    val_old = v + (torch.rand(B, generator=g, dtype=torch.float64) - 0.5) * 6 * cliprange
    # keep every sample clear of the clip boundaries: there fp32 and fp64 may take different branches, and ONE such sample
    # of a large minibatch is a relative error of 1e-3 in a gradient whose terms mostly cancel (seen at B = 200 000)
    ratio = (logp - logp_old).exp()
    near = ((ratio - (1 - cliprange)).abs() < 1e-4) | ((ratio - (1 + cliprange)).abs() < 1e-4)
    logp_old = torch.where(near, logp_old + 0.01, logp_old)
    near_v = ((v - val_old).abs() - cliprange).abs() < 1e-4
    val_old = torch.where(near_v, val_old + 0.01, val_old)
    ret = v + torch.randn(B, generator=g, dtype=torch.float64)
    return obs, act, adv, logp_old, val_old, ret


def _torch_loss(policy, obs, act, adv, logp_old, val_old, ret, cliprange, vf_coef, ent_coef):
    import torch
    d = policy.dist(obs)
    logp = d.log_prob(act).sum(-1)
    ratio = (logp - logp_old).exp()
    pg = torch.max(-adv * ratio, -adv * ratio.clamp(1 - cliprange, 1 + cliprange)).mean()
    v = policy.value(obs)
    v_clip = val_old + (v - val_old).clamp(-cliprange, cliprange)
    vf = 0.5 * torch.max((v - ret) ** 2, (v_clip - ret) ** 2).mean()
    ent = d.entropy().sum(-1).mean()
    return pg - ent_coef * ent + vf_coef * vf, pg, vf


@pytest.mark.parametrize("obs_dim,act_dim,B", [(9, 8, 1000), (9, 8, 64), (9, 8, 37), (9, 8, 1), (9, 8, 2), (9, 8, 20000), (60, 38, 777), (3, 1, 200), (30, 8, 129),
                                                (31, 8, 100), (32, 8, 100), (10, 33, 100), (63, 40, 70),
                                                # more than one 64-sample tile per wave (1 024 waves): the prefetching form's two buffers
                                                # alternate (9 observations; 29: the last that fits the LDS), the plain small instance loops (30),
                                                # the general one (40)
                                                (9, 8, 200000), (29, 8, 140000), (30, 8, 140000), (40, 12, 140000)])
def test_ppo_minibatch_gradient_matches_torch_autograd(obs_dim, act_dim, B):
    import torch
    from gym_roboy_amd import _policy_native as pn
    from gym_roboy_amd.ppo import FusedPolicyGrad
    want_form = (2 if obs_dim <= 29 else 1) if act_dim <= 8 and obs_dim <= 31 else 0
    if os.environ.get("ROBOY_POLICY_PREFETCH", "1")[0] != "0":
        assert pn.load().rp_grad_form(obs_dim, act_dim) == want_form
    cliprange, vf_coef, ent_coef = 0.2, 0.5, 0.1
    policy = _policy(obs_dim, act_dim, 11 + obs_dim)
    ref = _policy(obs_dim, act_dim, 11 + obs_dim).double()
    mb = _minibatch(ref, obs_dim, act_dim, B, B, cliprange)
    loss, pg_ref, vf_ref = _torch_loss(ref, *mb, cliprange, vf_coef, ent_coef)
    loss.backward()
    policy = policy.cuda()
    fg = FusedPolicyGrad(policy)
    dev = [t.float().cuda().contiguous() for t in mb]
    pg, vf = fg.run(*dev, cliprange, vf_coef, ent_coef)
    torch.cuda.synchronize()
    assert abs(pg.item() - pg_ref.item()) < 1e-4 * max(1.0, abs(pg_ref.item()))
    assert abs(vf.item() - vf_ref.item()) < 1e-4 * max(1.0, abs(vf_ref.item()))
    worst = 0.0
    for (name, p), (_, q) in zip(policy.named_parameters(), ref.named_parameters()):
        got, want = p.grad.detach().cpu().double(), q.grad
        scale = max(want.abs().max().item(), 1e-6)
        err = (got - want).abs().max().item() / scale
        worst = max(worst, err)
        assert err < 5e-4, (name, err, scale)
    assert worst < 5e-4
    # the same minibatch addressed through row indices into larger "rollout" tensors: bit-identical gradient
    g_direct = fg._g.clone()
    perm = torch.randperm(3 * B, device="cuda")[:B]
    big = [torch.zeros(3 * B, *t.shape[1:], device="cuda") for t in dev]
    for t_big, t in zip(big, dev):
        t_big[perm] = t
    fg.run(big[0], big[1], dev[2], big[3], big[4], big[5], cliprange, vf_coef, ent_coef, index=perm)
    torch.cuda.synchronize()
    assert torch.equal(fg._g, g_direct)


def test_gradient_kernel_refuses_policies_beyond_its_lds_budget_loudly():
    from gym_roboy_amd.ppo import FusedPolicyGrad
    with pytest.raises(RuntimeError, match="too large"):
        FusedPolicyGrad(_policy(63, 64, 0).cuda())
    with pytest.raises(RuntimeError, match="obs_dim <= 63"):
        FusedPolicyGrad(_policy(64, 8, 0).cuda())


def test_ppo_with_fused_rollout_and_fused_update_learns_like_the_torch_path():
    """Same seed, same env: PPO(fused_policy, fused_update) and the all-torch PPO see different noise streams, so they
    are compared statistically - both must improve the mean reward on the toy horizon - and the fused update must
    reproduce torch's update when both are fed the SAME rollout."""
    import copy
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from gym_roboy_amd.ppo import PPO
    env = RoboyVecEnv(MsjRobot(), 512, seed=1)
    a = PPO(env, n_steps=16, seed=4, reward_scale=0.01, fused_update=True)
    b = PPO(env, n_steps=16, seed=4, reward_scale=0.01, fused_policy=False, fused_update=False)
    assert a._fgrad is not None and b._fgrad is None
    b.policy.load_state_dict(copy.deepcopy(a.policy.state_dict()))
    roll = a.collect()
    n = roll["obs"].shape[0] * roll["obs"].shape[1]
    orders = [torch.randperm(n, device="cuda") for _ in range(a.noptepochs)]      # the same sample orders for both
    a.update({k: v.clone() for k, v in roll.items()}, sample_orders=orders)
    b.update({k: v.clone() for k, v in roll.items()}, sample_orders=orders)
    for (n, p), (_, q) in zip(a.policy.named_parameters(), b.policy.named_parameters()):
        assert (p - q).abs().max().item() < 2e-4, n            # 16 clip + Adam steps apart by rounding only
    # and on its own sample order (the device-side permutation) it keeps training
    stats = a.update({k: v.clone() for k, v in roll.items()})
    assert all(np.isfinite(v) for v in stats.values())
    env.close()


def test_device_permutation_equals_the_host_statement_and_is_a_bijection():
    import ctypes as c
    import torch
    from gym_roboy_amd import _policy_native as pn
    lib = pn.load()
    for n, key in ((1, 5), (2, 6), (37, 7), (4096, 8), (100003, 9), (1 << 20, (3 << 32) | 4)):
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        pn.check(lib.rp_perm_dev(key, n, 0, n, c.c_void_p(out.data_ptr()), c.c_void_p(torch.cuda.current_stream().cuda_stream)))
        host = np.empty(n, np.int64)
        pn.check(lib.rp_perm_host(key, n, 0, n, host.ctypes.data_as(c.c_void_p)))
        got = out.cpu().numpy()
        assert np.array_equal(got, host)
        assert np.array_equal(np.sort(got), np.arange(n))
    # a slice of the order is the slice of the whole
    part = torch.empty(1000, dtype=torch.int64, device="cuda")
    pn.check(lib.rp_perm_dev(9, 100003, 5000, 1000, c.c_void_p(part.data_ptr()), c.c_void_p(torch.cuda.current_stream().cuda_stream)))
    whole = np.empty(100003, np.int64)
    pn.check(lib.rp_perm_host(9, 100003, 0, 100003, whole.ctypes.data_as(c.c_void_p)))
    assert np.array_equal(part.cpu().numpy(), whole[5000:6000])


@pytest.mark.parametrize("B,total", [(1, 10), (2, 10), (1000, 5000), (70001, 300000)])
def test_minibatch_advantage_statistics_match_torch(B, total):
    import torch
    from gym_roboy_amd.ppo import FusedPolicyGrad
    fg = FusedPolicyGrad(_policy(9, 8, 1).cuda())
    g = torch.Generator().manual_seed(B)
    adv = (torch.randn(total, generator=g) * 3.0 + 1.5).cuda()
    idx = torch.randperm(total, generator=g)[:B].cuda()
    for _ in range(2):                                         # the scratch block is left ready for the next call
        st = fg.minibatch_adv_stats(adv, idx).cpu()
        sel = adv[idx].double()
        mean = sel.mean().item()
        inv = 1.0 / ((sel.std().item() if B > 1 else float("nan")) + 1e-8) if B > 1 else 1e8
        assert abs(st[0].item() - mean) < 1e-5 * max(1.0, abs(mean))
        assert abs(st[1].item() - inv) < 1e-4 * inv


@pytest.mark.parametrize("obs_dim,act_dim", [(9, 8), (60, 38)])
def test_fused_clip_and_adam_equal_torch_clip_and_adam(obs_dim, act_dim):
    """FusedAdam.step() over the gradient vector against clip_grad_norm_ + torch.optim.Adam.step() on a copy of the
    policy, for gradients that are clipped (large) and not (small), over several steps (bias corrections)."""
    import copy
    import torch
    from gym_roboy_amd.ppo import FusedAdam, FusedPolicyGrad
    policy = _policy(obs_dim, act_dim, 3).cuda()
    ref = copy.deepcopy(policy)
    fg = FusedPolicyGrad(policy)
    fa = FusedAdam(fg, 2.5e-4, eps=1e-5, max_grad_norm=0.5)
    opt = torch.optim.Adam(ref.parameters(), lr=2.5e-4, eps=1e-5)
    names = [n for n, _ in policy.named_parameters()]
    gen = torch.Generator(device="cuda").manual_seed(1)
    ent_coef = 0.1
    for step, mag in enumerate((5.0, 1e-3, 0.3, 2.0, 1e-2)):
        fg._g.copy_(torch.randn(fg._g.shape, device="cuda", generator=gen) * mag)
        for (name, p), (_, q) in zip(fg._named.items(), [(n, dict(ref.named_parameters())[_torch_name(n)]) for n in fg._named]):
            q.grad = fg._views[name].detach().clone()
        dict(ref.named_parameters())["log_std"].grad -= ent_coef                   # the entropy bonus FusedAdam adds itself
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
        fa.step(ent_coef)
        torch.cuda.synchronize()
        for name, p in fg._named.items():
            q = dict(ref.named_parameters())[_torch_name(name)]
            assert (p.detach() - q.detach()).abs().max().item() < 2e-6 * (step + 1), (name, step)
    # the module's parameters are views of the flat buffer: a state_dict round trip keeps them attached
    sd = copy.deepcopy(policy.state_dict())
    policy.load_state_dict(sd)
    assert policy.log_std.data_ptr() == fa.params[fg._layout["log_std"][0]:].data_ptr()


def _torch_name(n):
    """'pi_w1' -> 'pi.0.weight' ..."""
    if n == "log_std":
        return n
    net, kind = n.split("_")
    return "%s.%d.%s" % (net, {"1": 0, "2": 2, "3": 4}[kind[1]], "weight" if kind[0] == "w" else "bias")


def test_fused_gae_equals_the_torch_loop():
    import torch
    from gym_roboy_amd.ppo import gae, gae_fused
    g = torch.Generator().manual_seed(0)
    T, N = 37, 1000
    rew, val = torch.randn(T, N, generator=g).cuda(), torch.randn(T, N, generator=g).cuda()
    done = (torch.rand(T, N, generator=g) < 0.05).float().cuda()
    last = torch.randn(N, generator=g).cuda()
    a0, r0 = gae(rew, val, done, last, 0.99, 0.95)
    a1, r1 = gae_fused(rew, val, done, last, 0.99, 0.95)
    torch.cuda.synchronize()
    assert (a0 - a1).abs().max().item() < 1e-5 and (r0 - r1).abs().max().item() < 1e-5


def test_ppo_on_the_upper_body_with_the_fused_kernels():
    """The 60 -> 38 policy of the joint-tree robot goes through the general instances of the kernels."""
    import torch
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from gym_roboy_amd.ppo import PPO
    env = RoboyVecEnv(UpperBodyRobot(), 256, seed=5)
    agent = PPO(env, n_steps=8, use_graphs=True, fused_policy=True, fused_update=True, seed=1, reward_scale=0.01)
    roll = agent.collect()
    with torch.no_grad():
        lp = agent.policy.dist(roll["obs"]).log_prob(roll["act"]).sum(-1)
    assert (lp - roll["logp"]).abs().max() < 2e-3
    before = [p.detach().clone() for p in agent.policy.parameters()]
    stats = agent.update(roll)
    assert all(np.isfinite(v) for v in stats.values())
    assert any((p.detach() - b).abs().max() > 0 for p, b in zip(agent.policy.parameters(), before))
    env.close()


def _fused_rank(rank, world, port, out_dir):
    import os
    import torch
    import torch.distributed as dist
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from gym_roboy_amd.ppo import PPO
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    env = RoboyVecEnv(MsjRobot(), 256, seed=0, env_id_offset=256 * rank)
    agent = PPO(env, n_steps=8, seed=5, dist=dist, reward_scale=0.01, fused_policy=True, fused_update=True)
    roll = agent.collect()
    agent.update(roll)
    torch.cuda.synchronize()
    torch.save({"params": [p.detach().cpu() for p in agent.policy.parameters()], "act": roll["act"].cpu()},
               os.path.join(out_dir, "r%d.pt" % rank))
    env.close()
    dist.destroy_process_group()


def test_two_ranks_with_the_fused_kernels_stay_in_step(tmp_path):
    """Two ranks (both on the one GPU, gradients averaged over gloo) with fused_policy / fused_update: different
    exploration noise and env shards per rank, identical parameters after the update (the gradient views of
    FusedPolicyGrad go through average_gradients like torch's)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_fused_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert (a["act"] - b["act"]).abs().max() > 0.1                     # own noise per rank
    for p, q in zip(a["params"], b["params"]):
        assert torch.equal(p, q)
    assert all(torch.isfinite(p).all() for p in a["params"])
