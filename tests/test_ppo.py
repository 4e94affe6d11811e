"""The PPO consumer on CPU: GAE against a hand computation,
learning on a toy vec env, checkpoint round-trip, and 2-rank gloo gradient
averaging (the N > 1 path; RCCL on the GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from gym_roboy_amd._gymcompat import spaces
from gym_roboy_amd.ppo import PPO, MlpPolicy, average_gradients, gae

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ToyVecEnv:
    """obs = target in [-1,1]^2; reward = -|action - target|^2; episodes of 8 steps."""
    def __init__(self, n, seed=0):
        self.n, self.rng, self.t = n, np.random.default_rng(seed), 0
        self.observation_space = spaces.Box(low=-1, high=1, shape=(2,), dtype="float32")
        self.action_space = spaces.Box(low=-1, high=1, shape=(2,), dtype="float32")
    def reset(self):
        self.target = self.rng.uniform(-1, 1, (self.n, 2)).astype(np.float32)
        return self.target
    def step(self, a):
        r = -np.sum((np.asarray(a) - self.target) ** 2, axis=1).astype(np.float32)
        self.t += 1
        done = np.full(self.n, self.t % 8 == 0)
        if done[0]:
            self.reset()
        return self.target, r, done, [{}] * self.n


def test_gae_matches_hand_computation():
    r = torch.tensor([[1.0], [2.0], [3.0]]); v = torch.tensor([[0.5], [0.4], [0.3]])
    d = torch.tensor([[0.0], [1.0], [0.0]]); last = torch.tensor([0.2])
    adv, ret = gae(r, v, d, last, gamma=0.9, lam=0.8)
    d2 = 3.0 + 0.9 * 0.2 - 0.3
    d1 = 2.0 - 0.4                       # step 1 ended the episode: no bootstrap, no carry-over
    d0 = 1.0 + 0.9 * 0.4 - 0.5 + 0.9 * 0.8 * d1
    assert torch.allclose(adv.squeeze(), torch.tensor([d0, d1, d2]))
    assert torch.allclose(ret, adv + v)


def test_ppo_learns_the_toy_task_and_checkpoints(tmp_path):
    env = ToyVecEnv(64)
    agent = PPO(env, n_steps=32, device="cpu", learning_rate=3e-3, ent_coef=0.0, seed=1)
    first = agent.collect()["rew"].mean().item()
    agent.learn(total_timesteps=64 * 32 * 30)
    last = agent.collect()["rew"].mean().item()
    assert last > first + 0.2, (first, last)
    path = str(tmp_path / "model.pkl")
    agent.save(path)
    other = PPO(ToyVecEnv(64), n_steps=32, device="cpu", seed=2).load(path)
    obs = torch.randn(5, 2)
    assert torch.equal(agent.policy.pi(obs), other.policy.pi(obs))
    assert other.num_timesteps == agent.num_timesteps


def _rank(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    pol = MlpPolicy(3, 2)
    x = torch.arange(12, dtype=torch.float32).reshape(4, 3)[rank * 2:rank * 2 + 2]   # each rank: half the batch
    pol.pi(x).pow(2).mean().backward()
    average_gradients(pol, dist)
    torch.save([p.grad.clone() for p in pol.pi.parameters()], os.path.join(out_dir, "g%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch_gradient(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    torch.manual_seed(0)
    pol = MlpPolicy(3, 2)
    x = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    pol.pi(x).pow(2).mean().backward()
    for a, b, p in zip(g0, g1, pol.pi.parameters()):
        assert torch.equal(a, b)
        assert torch.allclose(a, p.grad, atol=1e-6)


def test_adam_state_moves_between_the_two_optimiser_forms():
    """A checkpoint of the torch path resumed on the fused path and back (PPO.load): torch.optim.Adam's per-parameter
    moments <-> the flat moment vectors of FusedAdam, laid out like the gradient vector of include/roboy_policy.h."""
    import torch
    from gym_roboy_amd import _policy_native as pn
    from gym_roboy_amd.ppo import MlpPolicy, adam_state_flat_to_torch, adam_state_torch_to_flat
    torch.manual_seed(0)
    policy = MlpPolicy(9, 8)
    opt = torch.optim.Adam(policy.parameters(), lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        a, logp, v = policy.act(torch.randn(32, 9))
        d = policy.dist(torch.randn(32, 9))
        (d.log_prob(torch.randn(32, 8)).sum() + policy.value(torch.randn(32, 9)).sum()).backward()
        opt.step()
    layout, n = pn.grad_layout(9, 8)
    m, v = torch.zeros(n), torch.zeros(n)
    t = adam_state_torch_to_flat(opt.state_dict(), policy, layout, m, v)
    assert t == 3
    off, shape = layout["vf_w2"]
    assert torch.equal(m[off:off + 64 * 64].view(64, 64), opt.state[policy.vf[2].weight]["exp_avg"])
    off, shape = layout["log_std"]
    assert torch.equal(v[off:off + 8], opt.state[policy.log_std]["exp_avg_sq"])
    back = adam_state_flat_to_torch({"m": m, "v": v, "t": t}, policy, layout, opt.state_dict())
    opt2 = torch.optim.Adam(policy.parameters(), lr=1e-3)
    opt2.load_state_dict(back)
    for p in policy.parameters():
        assert torch.equal(opt2.state[p]["exp_avg"], opt.state[p]["exp_avg"]) and float(opt2.state[p]["step"]) == 3.0
    assert adam_state_torch_to_flat(torch.optim.Adam(policy.parameters()).state_dict(), policy, layout, m, v) is None
    with pytest.raises(ValueError):
        adam_state_torch_to_flat(opt.state_dict(), MlpPolicy(5, 8), pn.grad_layout(5, 8)[0], m, v)
    # ADVICE (round 4): an optimiser that stepped only SOME of its parameters is refused (its moments under one step count would
    # resume the others with zero moments and a wrong bias correction)
    partial = opt.state_dict()
    del partial["state"][0]
    with pytest.raises(ValueError, match="partial optimiser state"):
        adam_state_torch_to_flat(partial, policy, layout, m, v)
    # ADVICE (round 5): ... but a FROZEN parameter (requires_grad = False: torch.optim.Adam never creates state for it) is not a
    # partial state - a checkpoint of a policy with a fixed log_std converts, its moments stay zero
    frozen = MlpPolicy(9, 8)
    frozen.log_std.requires_grad_(False)
    opt3 = torch.optim.Adam(frozen.parameters(), lr=1e-3)
    opt3.zero_grad()
    (frozen.dist(torch.randn(16, 9)).log_prob(torch.randn(16, 8)).sum() + frozen.value(torch.randn(16, 9)).sum()).backward()
    opt3.step()
    assert frozen.log_std not in opt3.state
    m3, v3 = torch.zeros(n), torch.zeros(n)
    assert adam_state_torch_to_flat(opt3.state_dict(), frozen, layout, m3, v3) == 1
    off, shape = layout["log_std"]
    assert not m3[off:off + 8].any() and not v3[off:off + 8].any() and m3.abs().sum() > 0


def test_an_explicit_chain_request_that_cannot_be_honoured_warns():
    """ADVICE (round 4): PPO(rollout_chains=2) over an env that cannot step sub-ranges (or without the fused policy step) runs one
    chain - and says so."""
    from gym_roboy_amd.ppo import PPO

    class Agent:                      # _pick_chains reads these three attributes only
        _pick_chains = PPO._pick_chains
        CHAIN_BATCH, CHAIN_MAX_OBS = PPO.CHAIN_BATCH, PPO.CHAIN_MAX_OBS
    a = Agent()
    a._fused, a._chains_arg, a.env = None, 2, object()
    with pytest.warns(RuntimeWarning, match="fused policy step is off"):
        assert a._pick_chains(65536) == 1
    a._fused = type("F", (), {"obs_dim": 9})()
    with pytest.warns(RuntimeWarning, match="whole batches only"):
        assert a._pick_chains(65536) == 1
    a.env = type("E", (), {"step_range_dev": None, "range_capable": lambda self: True})()
    with pytest.warns(RuntimeWarning, match="fewer than 512"):
        assert a._pick_chains(256) == 1
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert a._pick_chains(65536) == 2
        a._chains_arg = None
        assert a._pick_chains(256) == 1 and a._pick_chains(PPO.CHAIN_BATCH) == 2      # the library's own choice: silent
        a._chains_arg = 1
        assert a._pick_chains(1 << 20) == 1


@pytest.mark.gpu
def test_ppo_runs_on_the_device_env_without_host_copies():
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(MsjRobot(), 256, seed=0)
    agent = PPO(env, n_steps=16, device="cuda", ent_coef=0.1, reward_scale=0.01)
    logs = []
    agent.learn(total_timesteps=256 * 16 * 3, log=logs.append)
    assert len(logs) == 3 and all(np.isfinite(l["loss"]) for l in logs)
    assert agent.num_timesteps == 256 * 16 * 3
    st = env.stats()
    assert st["n_env_steps"] == 256 * 16 * 3
    env.close()


@pytest.mark.gpu
def test_graph_mode_rollout_is_a_valid_rollout():
    """The captured rollout (policy, sampling, env kernel through the C ABI, GAE) produces what
    the eager loop produces from the same policy: same env transitions for the same actions,
    log-probs and values of the policy, GAE of the rewards."""
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from gym_roboy_amd.ppo import gae
    T, N = 16, 512
    env = RoboyVecEnv(MsjRobot(), N, seed=3)
    agent = PPO(env, n_steps=T, device="cuda", ent_coef=0.1, reward_scale=0.01, seed=4, use_graphs=True,
                fused_policy=False, fused_update=False)     # the torch path: the exact statement of a rollout
    assert agent.use_graphs
    roll = {k: v.clone() for k, v in agent.collect().items()}
    assert all(bool(torch.isfinite(v).all()) for v in roll.values())
    assert float(roll["act"].std()) > 0.5                       # sampled, not the mean
    # replay the recorded actions through a second env with eager launches
    env2 = RoboyVecEnv(MsjRobot(), N, seed=3)
    obs = torch.as_tensor(env2.reset(), device="cuda")
    assert torch.equal(obs, roll["obs"][0])
    with torch.no_grad():
        for t in range(T):
            d = agent.policy.dist(roll["obs"][t])
            assert torch.allclose(d.log_prob(roll["act"][t]).sum(-1), roll["logp"][t], rtol=1e-5, atol=1e-5)
            assert torch.allclose(agent.policy.value(roll["obs"][t]), roll["val"][t], rtol=1e-5, atol=1e-6)
            o, r, dn, _ = env2.step(roll["act"][t].clamp(-1, 1).contiguous())
            assert torch.equal(r * 0.01, roll["rew"][t]) and torch.equal(dn.float(), roll["done"][t])
            if t + 1 < T:
                assert torch.equal(o, roll["obs"][t + 1])
        adv, ret = gae(roll["rew"], roll["val"], roll["done"], agent.policy.value(o), 0.99, 0.95)
    assert torch.allclose(adv, roll["adv"], rtol=1e-5, atol=1e-6) and torch.allclose(ret, roll["ret"], rtol=1e-5, atol=1e-6)
    env.close(); env2.close()


@pytest.mark.gpu
def test_graph_mode_learns_counts_and_resumes(tmp_path):
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(MsjRobot(), 256, seed=0)
    agent = PPO(env, n_steps=16, device="cuda", ent_coef=0.1, reward_scale=0.01, use_graphs=True)
    logs = []
    agent.learn(total_timesteps=256 * 16 * 4, log=logs.append)
    assert len(logs) == 4 and all(np.isfinite(l["loss"]) for l in logs)
    assert agent.num_timesteps == 256 * 16 * 4
    assert env.stats()["n_env_steps"] == 256 * 16 * 4          # replayed steps are reported to the env
    first = [l["mean_reward"] for l in logs]
    assert len(set(first)) == 4                                 # every replay is a new rollout (RNG advances)
    path = str(tmp_path / "model.pkl")
    agent.save(path)
    before = {k: v.clone() for k, v in agent.policy.state_dict().items()}
    agent.load(path)                                            # resume from the checkpoint, rollout graph stays valid
    agent.learn(total_timesteps=256 * 16, log=logs.append)
    assert np.isfinite(logs[-1]["loss"])
    assert any(not torch.equal(before[k], v) for k, v in agent.policy.state_dict().items())
    env.close()


@pytest.mark.gpu
def test_checkpoints_resume_across_the_two_optimiser_forms(tmp_path):
    """model.pkl written by the torch path (fused_update=False) resumes on the fused path with Adam's moments and step count,
    and the other way round (PPO.load used to restart them silently)."""
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(MsjRobot(), 512, seed=1)
    kw = dict(n_steps=8, device="cuda", ent_coef=0.1, reward_scale=0.01, seed=1)
    a = PPO(env, fused_policy=False, fused_update=False, **kw)
    a.learn(total_timesteps=512 * 8 * 2)
    pa = str(tmp_path / "torch.pkl")
    a.save(pa)
    b = PPO(env, **kw)
    assert b._fgrad is not None
    b.load(pa)
    assert b._fadam.t == a.opt.state[a.policy.log_std]["step"] and float(b._fadam.m.abs().sum()) > 0
    off, shape = b._fgrad._layout["pi_w1"]
    assert torch.equal(b._fadam.v[off:off + 64 * 9].view(64, 9), a.opt.state[a.policy.pi[0].weight]["exp_avg_sq"])
    b.learn(total_timesteps=512 * 8)
    pb = str(tmp_path / "fused.pkl")
    b.save(pb)
    c = PPO(env, fused_policy=False, fused_update=False, **kw)
    c.load(pb)
    assert float(c.opt.state[c.policy.log_std]["step"]) == b._fadam.t
    off, shape = b._fgrad._layout["vf_b2"]
    assert torch.equal(c.opt.state[c.policy.vf[2].bias]["exp_avg"], b._fadam.m[off:off + 64])
    c.learn(total_timesteps=512 * 8)
    env.close()


@pytest.mark.gpu
def test_graph_mode_on_a_robot_whose_kernels_are_built_at_run_time():
    """An 8-tendon ball-joint robot that is not the reference's MsjRobot, above 65 536 envs: its env-per-lane kernels
    are specialised with hiprtc.  The build (a compilation and a module load) must happen before the rollout is
    captured - rb_env_configure does it, and PPO's warm-up asks again - and the captured rollout must run them."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from random_robots import random_ball_joint_robot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    robot, _ = random_ball_joint_robot(1)
    env = RoboyVecEnv(robot, 65536 + 256, seed=2)
    agent = PPO(env, n_steps=4, device="cuda", ent_coef=0.1, reward_scale=0.01, seed=1, use_graphs=True)
    roll = agent.collect()                                       # builds the graph
    assert env.sim.specialization() == "jit"
    assert all(bool(torch.isfinite(v).all()) for v in roll.values())
    first = roll["rew"].clone()
    roll = agent.collect()                                       # replays it
    assert bool(torch.isfinite(roll["rew"]).all()) and not torch.equal(first, roll["rew"])
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robot_name,n", [("msj", 2048 + 256), ("msj", 70000), ("upper", 20000)])
def test_two_chain_rollout_equals_the_one_chain_rollout(robot_name, n):
    """PPO's graph rollout as two chains of (policy step, env step) launches over the two halves of the batch
    (rb_env_step_range_dev on two streams) against the same rollout as one chain over the whole batch: the policy's noise is
    keyed by the global sample index and envs are independent, so every rollout tensor is the same bit for bit - over two
    replays (fresh noise, carried observations) and with a ragged second half."""
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    rolls = {}
    for chains in (1, 2):
        env = RoboyVecEnv(MsjRobot() if robot_name == "msj" else UpperBodyRobot(), n, seed=5)
        if robot_name == "upper":
            env.sim.select_kernel(1)                  # one wave per 64 envs: the joint-tree form that steps sub-ranges
        agent = PPO(env, n_steps=6, device="cuda", ent_coef=0.1, reward_scale=0.01, seed=3, use_graphs=True, rollout_chains=chains)
        agent.collect()
        assert agent.rollout_chains == chains
        roll = agent.collect()
        rolls[chains] = {k: v.clone() for k, v in roll.items()}
        assert env.stats()["n_env_steps"] == 2 * 6 * n
        env.close()
    for k in rolls[1]:
        assert torch.equal(rolls[1][k], rolls[2][k]), k
    assert bool(torch.isfinite(rolls[2]["adv"]).all())


@pytest.mark.gpu
def test_train_then_play_back(tmp_path, capsys):
    """train_parallel.py -> model.pkl -> visualize_agent.py, the reference's two drivers."""
    from gym_roboy_amd import train_parallel, visualize_agent
    out = str(tmp_path / "results")
    train_parallel.main(["256", out, "--rounds", "1", "--steps-per-round", str(256 * 128)])
    assert os.path.exists(os.path.join(out, "model.pkl"))
    total = visualize_agent.main([os.path.join(out, "model.pkl"), "--steps", "20", "--pause", "0"])
    assert np.isfinite(total)
    assert "reward" in capsys.readouterr().out


def _train_rank(rank, world, port, out_dir):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gym_roboy_amd import train_parallel
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        agent = train_parallel.main(["256", os.path.join(out_dir, "results"), "--rounds", "1", "--steps-per-round", str(256 * 32 * 3),
                                     "--n-steps", "32", "--backend", "gloo"])
    torch.cuda.synchronize()
    torch.save({"params": [p.detach().cpu() for p in agent.policy.parameters()], "graphs": agent.use_graphs,
                "graph_built": agent._rollout_graph is not None, "stdout": buf.getvalue(),
                "local_steps": agent.env.stats()["n_env_steps"]}, os.path.join(out_dir, "rank%d.pt" % rank))


@pytest.mark.gpu
def test_two_rank_training_replays_graphs_and_reports_the_statistics_of_all_ranks(tmp_path):
    """train_parallel with two ranks (sharing the one GPU, collectives over gloo): every rank replays its own captured
    rollout graph, the parameters stay identical across the ranks, and the statistics rank 0 prints are the sum over
    both ranks' envs (reference: one process per env, /root/reference/gym_roboy/train_parallel.py:19-35)."""
    import re
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_train_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert a["graphs"] and b["graphs"] and a["graph_built"] and b["graph_built"]
    for p, q in zip(a["params"], b["params"]):
        assert torch.equal(p, q)
    assert a["local_steps"] == b["local_steps"] == 256 * 32 * 3
    m = re.search(r"'n_env_steps': ([0-9.]+)", a["stdout"])
    assert m and float(m.group(1)) == 2 * 256 * 32 * 3          # both ranks' env steps
    assert "episode statistics" not in b["stdout"]             # one report, by rank 0
