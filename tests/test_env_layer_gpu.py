"""GPU parity of the fused env layer (rb_env_step_dev) against the reference's
formulas.

The host model below replays, for every env, exactly what the reference does
around a simulator (``RoboyEnv.step`` roboy_env.py:51-70 inside a
``SubprocVecEnv`` worker that calls ``env.reset()`` on done): it uses the
plain physics kernel for the state (same kernels, so q/qd/goal must match the
fused kernel BIT FOR BIT), ``oracle/philox_np.py`` for the goal stream and
``gym_roboy_amd/envs/reward.py`` (pinned to the reference by tests/golden) in
float64 for reward/done.  Reward is float arithmetic: rtol 2e-5 / atol 2e-4 on
values up to |r| ~ 33 (fp32 exp of a fp32 norm).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


from host_env_model import HipStepper, HostEnvModel


@pytest.mark.parametrize("n,integrator,auto_reset,vel_penalty,form", [
    (777, "euler", True, False, 1), (777, "euler", True, True, 1), (777, "euler", False, False, 1), (777, "euler", False, True, 1),
    (777, "rk4", True, True, 1),
    # above 65 536 envs the fused kernel switches to its rolled-loop form (256-thread workgroups, LDS set-points)
    (70001, "euler", True, True, 1), (70001, "rk4", True, False, 1),
    # the two-lanes-per-env form (round 5): chosen explicitly, and by the library (form 0: nothing selected on the env's
    # handle) for MsjRobot above 8 192 and up to 24 576 envs (Euler) / 32 768 envs (RK4); up to 8 192 envs its choice is eight lanes per env
    (777, "rk4", True, True, 5), (777, "euler", False, True, 5), (777, "rk4", True, False, 0), (9001, "euler", True, False, 0),
    (20001, "rk4", True, True, 0), (32768, "rk4", True, True, 0),
    # eight lanes per env (round 5): lane k rescales and evaluates tendon k, lane 0 of the group accounts; ragged batch, both integrators
    (777, "rk4", True, True, 2), (777, "euler", False, False, 2), (3001, "rk4", True, False, 2)])
def test_fused_env_step_matches_host_replay(msj_robot, n, integrator, auto_reset, vel_penalty, form):
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    seed, max_len = 5, 12
    vec = RoboyVecEnv(msj_robot, n, seed=seed, joint_vel_penalty=vel_penalty, auto_reset=auto_reset,
                      max_episode_length=max_len, integrator=integrator)
    if form:
        vec.sim.select_kernel(form)                   # 1: one env per lane, 2: eight lanes per env, 5: two lanes per env; 0: the library's choice
    host = HostEnvModel(msj_robot, HipStepper(msj_robot, n, seed, integrator=integrator, kernel={0: 2 if n <= 8192 else 5, 1: 1, 2: 2, 5: 5}[form]), n, seed, max_len,
                        vel_penalty, True, auto_reset)
    obs0 = vec.reset()
    # vec.__init__ drew goal 0 (configure), reset() drew goal 1: mirror RoboyEnv(...) then reset()
    host.goal = host.draw(np.ones(n, bool))
    assert np.array_equal(obs0[:, :6], np.zeros((n, 6), np.float32))
    assert np.array_equal(obs0[:, 6:], host.goal)
    rng = np.random.default_rng(0)
    n_done = 0
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        if t % 7 == 3:
            a[: n // 2] = 0.0
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(a)
        clear = margin > 1e-5
        assert np.array_equal(done[clear], h_done[clear])
        same = done == h_done
        assert np.array_equal(obs[same], h_obs[same].astype(np.float32))
        np.testing.assert_allclose(rew[same], h_rew[same], rtol=2e-5, atol=2e-4)
        assert same.all(), "done flags diverged on a borderline env; host replay is no longer in step"
        n_done += int(done.sum())
    assert n_done >= n   # every env hit the 12-step episode limit at least once
    st = vec.stats()
    assert st["n_env_steps"] == 40 * n
    assert st["n_episodes"] == n_done
    got = np.array([st[k] for k in ("sum_return", "sum_return_sq", "n_episodes", "sum_length", "n_goal_reached",
                                    "n_infeasible_steps", "n_env_steps", "sum_reward")])
    np.testing.assert_allclose(got, host.stats, rtol=1e-4, atol=1e-2)
    vec.close(); host.stepper.close()


def test_fused_env_step_in_pair_form_on_kernarg_constants_and_the_other_mirror_plane(msj_robot):
    """MsjRobot turned by 90 degrees about z (mirror plane y-z, constants not the baked table's): the two-lanes-per-env env
    kernel's kernarg instance against the host replay over the plain step in the same form - bit for bit."""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from test_mirror_pairs import _rotated_msj
    desc = _rotated_msj()

    class Turned(type(msj_robot)):
        @classmethod
        def get_description(cls):
            return desc
    robot = Turned()
    n, seed, max_len = 1500, 8, 10
    vec = RoboyVecEnv(robot, n, seed=seed, joint_vel_penalty=True, auto_reset=True, max_episode_length=max_len, integrator="rk4")
    vec.sim.select_kernel(5)
    host = HostEnvModel(robot, HipStepper(robot, n, seed, integrator="rk4", kernel=5), n, seed, max_len, True, True, True)
    obs0 = vec.reset()
    host.goal = host.draw(np.ones(n, bool))
    assert np.array_equal(obs0[:, 6:], host.goal)
    rng = np.random.default_rng(1)
    n_done = 0
    for t in range(25):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(a)
        assert np.array_equal(done, h_done) or (margin[done != h_done] < 1e-5).all()
        assert np.array_equal(done, h_done)
        assert np.array_equal(obs, h_obs.astype(np.float32))
        np.testing.assert_allclose(rew, h_rew, rtol=2e-5, atol=2e-4)
        n_done += int(done.sum())
    assert n_done >= 2 * n and vec.stats()["n_episodes"] == n_done
    vec.close(); host.stepper.close()


def _heavy_robot():
    """MsjRobot whose actuators are so heavy that one step does not move it:
    lets a test park the state exactly on the goal."""
    from gym_roboy_amd.envs.robots import MsjRobot, RobotDescription, msj_platform_spec
    spec = msj_platform_spec()
    for j in spec["joints"]:
        j["armature"] = 1.0e6
    desc = RobotDescription(spec)

    class HeavyMsjRobot(MsjRobot):
        @classmethod
        def get_description(cls):
            return desc
    return HeavyMsjRobot()


def test_reaching_the_goal_gives_bonus_and_done():
    """roboy_env.py:104-107 / test_roboy_env.py:60-68: at the goal with zero
    action the reward is the maximum, -exp(0) + 1000 = 999."""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    n = 64
    vec = RoboyVecEnv(_heavy_robot(), n, seed=1, auto_reset=False)
    obs = vec.reset()
    goal = obs[:, 6:9].copy()
    vec.sim.set_state(goal, np.zeros((n, 3), np.float32))
    obs, rew, done, _ = vec.step(np.zeros((n, 8), np.float32))
    assert np.abs(obs[:, :3] - goal).max() < 1e-4
    assert done.all()
    np.testing.assert_allclose(rew, 999.0, atol=1e-2)
    assert not np.array_equal(obs[:, 6:9], vec.step(np.zeros((n, 8), np.float32))[0][:, 6:9])  # goal was resampled
    st = vec.stats()
    assert st["n_goal_reached"] >= n
    # moving at the goal is not "reached" (test_roboy_env.py:71-79)
    obs = vec.reset()
    goal = obs[:, 6:9].copy()
    fast = np.tile(vec.robot.get_joint_vels_space().high, (n, 1)).astype(np.float32)
    vec.sim.set_state(goal, fast)
    _, rew, done, _ = vec.step(np.zeros((n, 8), np.float32))
    assert not done.any() and np.all(rew < 0)
    vec.close()


def test_torch_tensor_path_is_zero_copy_and_equal(msj_robot):
    import torch
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    n = 2048
    a = np.random.default_rng(3).uniform(-1, 1, (n, 8)).astype(np.float32)
    v1 = RoboyVecEnv(msj_robot, n, seed=9)
    v2 = RoboyVecEnv(msj_robot, n, seed=9)
    v1.reset(); v2.reset()
    o1, r1, d1, _ = v1.step(a)
    o2, r2, d2, _ = v2.step(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(o1, o2.cpu().numpy())
    assert np.array_equal(r1, r2.cpu().numpy())
    assert np.array_equal(d1, d2.cpu().numpy())
    v1.close(); v2.close()


def test_vec_env_surface_of_stable_baselines(msj_robot):
    """step_async/step_wait, seed, get_attr, render: what PPO2 calls on the reference's SubprocVecEnv."""
    from gym_roboy_amd.envs import RoboyVecEnv
    a = RoboyVecEnv(msj_robot, 64, seed=5)
    b = RoboyVecEnv(msj_robot, 64, seed=5)
    a.reset(); b.reset()
    acts = np.random.default_rng(0).uniform(-1, 1, (64, 8)).astype(np.float32)
    a.step_async(acts)
    oa, ra, da, _ = a.step_wait()
    ob, rb_, db, _ = b.step(acts)
    assert np.array_equal(oa, ob) and np.array_equal(ra, rb_) and np.array_equal(da, db)
    with pytest.raises(RuntimeError):
        a.step_wait()
    assert a.seed() == [None] * 64 and a.seed(5) == [None] * 64
    with pytest.raises(NotImplementedError):
        a.seed(6)
    assert a.get_attr("num_envs") == [64] * 64 and a.get_attr("num_envs", indices=[0, 3]) == [64, 64]
    with pytest.raises(NotImplementedError):
        a.env_method("reset")
    assert a.render() is None
    a.close(); b.close()


@pytest.mark.parametrize("n_tendons,n,integrator", [(6, 500, "euler"), (12, 70001, "euler"), (4, 300, "rk4")])
def test_fused_env_layer_for_other_tendon_counts(msj_robot, n_tendons, n, integrator):
    """RoboyVecEnv over a ball-joint robot with another tendon count: the run-time-count form of the
    fused kernel against the host replay (plain step kernel + numpy Philox + reward.py)."""
    from test_physics_gpu import _ball_joint_robot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    robot = _ball_joint_robot(msj_robot, n_tendons, 7)
    seed, max_len = 2, 9
    vec = RoboyVecEnv(robot, n, seed=seed, max_episode_length=max_len, integrator=integrator, joint_vel_penalty=True)
    host = HostEnvModel(robot, HipStepper(robot, n, seed, integrator=integrator), n, seed, max_len, True, True, True)
    obs0 = vec.reset()
    host.goal = host.draw(np.ones(n, bool))
    assert np.array_equal(obs0[:, 6:], host.goal)
    rng = np.random.default_rng(4)
    n_done = 0
    for t in range(20):
        a = rng.uniform(-1.2, 1.2, (n, n_tendons)).astype(np.float32)      # some outside the box: clamped
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(np.clip(a, -1, 1))
        same = done == h_done
        assert same.all() or (margin[~same] < 1e-5).all()
        assert same.all()
        assert np.array_equal(obs, h_obs.astype(np.float32))
        np.testing.assert_allclose(rew, h_rew, rtol=2e-5, atol=2e-4)
        n_done += int(done.sum())
    assert n_done >= 2 * n and vec.stats()["n_episodes"] == n_done
    vec.close(); host.stepper.close()
