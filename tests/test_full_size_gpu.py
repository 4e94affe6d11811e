"""BASELINE.json's full batch sizes (262 144 envs RK4; 2 097 152 envs Euler),
checked through size-independent properties - the oracle cannot run 2M envs in
seconds, so: a strided sample of the full batch against the oracle, env
permutation equivariance, shard invariance, determinism, the rest equilibrium,
and the boundary behaviour, all on the full-size launch configuration."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _states(desc, n, seed):
    rng = np.random.default_rng(seed)
    q = rng.uniform(0.98 * desc.q_lo, 0.98 * desc.q_hi, (n, desc.n_q)).astype(np.float32)
    qd = rng.uniform(-desc.qd_max, desc.qd_max, (n, desc.n_q)).astype(np.float32)
    sp = rng.uniform(-0.3, 0.3, (n, desc.n_t)).astype(np.float32)
    return q, qd, sp


@pytest.mark.parametrize("n,integrator", [(262144, "rk4"), (2097152, "euler")])
def test_full_batch_sample_matches_oracle_and_is_permutation_equivariant(msj_robot, n, integrator):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    desc = msj_robot.get_description()
    q, qd, sp = _states(desc, n, 1)
    sim = HipBatchSimulation(msj_robot, n, integrator=integrator)
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    # (a) every 509th env against the fp64 oracle (509 is prime: all lanes / waves / workgroup slots get hit)
    idx = np.arange(0, n, 509)
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1)
    assert np.abs(q1[idx] - qo).max() < 2e-5 and np.abs(qd1[idx] - qdo).max() < 2e-5
    assert np.mean(f1[idx] == fo) > 0.999
    # (b) determinism: the same launch again gives the same bits
    sim.set_state(q, qd)
    q2, qd2, f2 = sim.forward_step_command(sp)
    assert np.array_equal(q1, q2) and np.array_equal(qd1, qd2) and np.array_equal(f1, f2)
    # (c) envs are independent: permuting the batch permutes the result, bit for bit
    perm = np.random.default_rng(2).permutation(n)
    sim.set_state(q[perm], qd[perm])
    q3, qd3, f3 = sim.forward_step_command(sp[perm])
    assert np.array_equal(q3, q1[perm]) and np.array_equal(qd3, qd1[perm]) and np.array_equal(f3, f1[perm])
    sim.close()


@pytest.mark.parametrize("n,integrator", [(262144, "rk4"), (200001, "rk4"), (524288, "euler")])
def test_rollout_in_two_chains_equals_one_launch_per_step(msj_robot, n, integrator):
    """Above a batch threshold rb_rollout_dev's graphs step the two halves of the batch as two independent chains of launches
    (two streams: one half's launch gaps and load / store phases under the other's arithmetic).  Envs are independent: the states
    are those of one launch per step (the eager path), bit for bit - also with a ragged second half."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    ring, steps = 4, 72
    outs = []
    for use_graph in (False, True):
        sim = HipBatchSimulation(msj_robot, n, integrator=integrator, seed=11)
        d_ring = sim.malloc(4 * ring * n * 8)
        for r in range(ring):
            sim.fill_actions_dev(d_ring + 4 * r * n * 8, r)
        sim.rollout_dev(d_ring, ring, steps, 0.3, use_graph=use_graph)
        sim.rollout_dev(d_ring, ring, 16, 0.3, use_graph=use_graph)        # a second graph (another chunk size) on the same handle
        sim.synchronize()
        outs.append(sim.read_state())
        sim.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.abs(outs[0][0]).max() > 0.01


def test_joint_tree_rollout_in_two_chains_equals_one_launch_per_step():
    """The same for the env-per-lane joint-tree kernel (env-major rows: the second chain starts at a row offset)."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    robot = UpperBodyRobot()
    n, ring, steps = 65536 + 100, 2, 24
    outs = []
    for use_graph in (False, True):
        sim = HipBatchSimulation(robot, n, integrator="euler", seed=3)
        assert sim.info()["kernel"] == 1 and sim.rollout_chains() == int(os.environ.get("ROBOY_SIM_CHAINS", "2"))
        d_ring = sim.malloc(4 * ring * n * sim.n_t)
        for r in range(ring):
            sim.fill_actions_dev(d_ring + 4 * r * n * sim.n_t, r)
        sim.rollout_dev(d_ring, ring, steps, 0.3, use_graph=use_graph)
        sim.synchronize()
        outs.append(sim.read_state())
        sim.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.abs(outs[0][0]).max() > 0.01


def test_full_batch_shards_reproduce_the_whole(msj_robot):
    """configs[4]: 2 097 152 envs = 8 shards of 262 144; two of the shards here,
    driven by the device action stream, against the same rows of one big batch."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    n_shard, seed = 262144, 4
    whole = HipBatchSimulation(msj_robot, 2 * n_shard, seed=seed)
    parts = [HipBatchSimulation(msj_robot, n_shard, seed=seed, env_id_offset=r * n_shard) for r in (0, 1)]
    def run(sim):
        d = sim.malloc(4 * sim.n_envs * 8)
        for t in range(12):
            sim.fill_actions_dev(d, t)
            sim.step_dev(d, 0.3)
        sim.synchronize()
        return sim.read_state()
    qw, qdw, fw = run(whole)
    for r, p in enumerate(parts):
        qp, qdp, fp = run(p)
        sl = slice(r * n_shard, (r + 1) * n_shard)
        assert np.array_equal(qw[sl], qp) and np.array_equal(qdw[sl], qdp) and np.array_equal(fw[sl], fp)
        p.close()
    assert np.abs(qw).max() > 0.01
    whole.close()


def test_full_batch_rest_equilibrium_and_boundary(msj_robot):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    n = 2097152
    sim = HipBatchSimulation(msj_robot, n)
    d = sim.malloc(4 * n * 8)
    sim.upload(d, np.zeros((n, 8), np.float32))
    sim.rollout_dev(d, 1, 20, 1.0, use_graph=True)
    sim.synchronize()
    q, qd, f = sim.read_state()
    assert not q.any() and not qd.any() and f.all()          # exact equilibrium, every env
    sim.upload(d, np.tile(msj_robot.get_action_space().low, (n, 1)))
    sim.rollout_dev(d, 1, 40, 1.0, use_graph=False)
    sim.synchronize()
    q, qd, f = sim.read_state()
    assert not f.any()                                        # all pushed into a limit and stay there
    assert np.all(np.abs(q) <= np.float32(0.6) + 1e-6)
    assert np.array_equal(q[0], q[-1]) and np.array_equal(q[0], q[n // 2])   # identical inputs, identical outputs
    sim.close()


# ----------------------------------------------------------------------------
# BASELINE.json configs[3]: the upper body (20 DOF / 38 tendons) at its stated 8 192 envs
@pytest.fixture(scope="module")
def upper_body():
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    return UpperBodyRobot()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
@pytest.mark.parametrize("kernel", [1, 3, 4, 6])      # env-per-lane / octets / several waves per env group (the library's choice at 8 192 envs) / its lean two-part form
def test_upper_body_full_batch_sample_determinism_permutation(upper_body, integrator, kernel):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    desc = upper_body.get_description()
    n = 8192
    q, qd, sp = _states(desc, n, 7)
    sim = HipBatchSimulation(upper_body, n, integrator=integrator)
    assert sim.info()["kernel"] == 4
    sim.select_kernel(kernel)
    assert sim.info()["kernel"] == kernel
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    assert np.isfinite(q1).all() and np.isfinite(qd1).all()
    # (a) every 13th env (630 envs: every wave slot of a workgroup, both dispatch generations) vs the fp64 oracle
    idx = np.arange(0, n, 13)
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1,
                                            threads=8)
    assert np.abs(q1[idx] - qo).max() < 2e-5, np.abs(q1[idx] - qo).max()
    assert np.abs(qd1[idx] - qdo).max() < 2e-5, np.abs(qd1[idx] - qdo).max()
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1[idx] != fo) & ~near)
    # (b) determinism
    sim.set_state(q, qd)
    q2, qd2, f2 = sim.forward_step_command(sp)
    assert np.array_equal(q1, q2) and np.array_equal(qd1, qd2) and np.array_equal(f1, f2)
    # (c) permutation equivariance, bit for bit
    perm = np.random.default_rng(8).permutation(n)
    sim.set_state(q[perm], qd[perm])
    q3, qd3, f3 = sim.forward_step_command(sp[perm])
    assert np.array_equal(q3, q1[perm]) and np.array_equal(qd3, qd1[perm]) and np.array_equal(f3, f1[perm])
    sim.close()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_upper_body_between_the_split_form_and_a_wave_per_simd(upper_body, integrator):
    """16 384 < n <= 32 768 envs: the library's choice is the lean two-part split form (rb_kernel 6: two workgroups per CU share
    its LDS) - a ragged 32 731-env batch against the fp64 oracle (every 509th env), deterministic, and the fused env layer's
    states bit-identical to the plain step's in the same form."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from oracle.c_oracle import COracle
    desc = upper_body.get_description()
    n = 32768 - 37
    q, qd, sp = _states(desc, n, 11)
    sim = HipBatchSimulation(upper_body, n, integrator=integrator)
    assert sim.info()["kernel"] == 6
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    assert sim.info()["kernel"] == 6 and np.isfinite(q1).all() and np.isfinite(qd1).all()
    idx = np.concatenate([np.arange(0, n, 509), np.arange(n - 40, n)])          # ... and the ragged last group
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1, threads=8)
    assert np.abs(q1[idx] - qo).max() < 2e-5 and np.abs(qd1[idx] - qdo).max() < 2e-5
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1[idx] != fo) & ~near)
    sim.set_state(q, qd)
    q2, qd2, f2 = sim.forward_step_command(sp)
    assert np.array_equal(q1, q2) and np.array_equal(qd1, qd2) and np.array_equal(f1, f2)
    # the same envs in the five-wave form (two generations) and as one wave per 64 envs: rounding apart
    for other in (4, 1):
        sim.select_kernel(other)
        sim.set_state(q, qd)
        q3, qd3, _ = sim.forward_step_command(sp)
        assert np.abs(q3 - q1).max() < 1e-5 and np.abs(qd3 - qd1).max() < 1e-5
    sim.close()
    # fused env layer in the same form: one step from the same state with actions = set-points / 0.3
    vec = RoboyVecEnv(upper_body, n, seed=1, auto_reset=False, integrator=integrator)
    assert vec.sim.info()["kernel"] == 6
    vec.reset()
    vec.sim.set_state(q, qd)
    act = np.clip(sp / np.float32(0.3), -1, 1).astype(np.float32)
    obs, rew, done, _ = vec.step(act)
    ref = HipBatchSimulation(upper_body, n, integrator=integrator)
    ref.set_state(q, qd)
    from gym_roboy_amd.envs import reward as rw
    one = np.ones(desc.n_t, np.float32)
    box = upper_body.get_action_space()
    qr, qdr, fr = ref.forward_step_command(rw.rescale_between_boxes(act, -one, one, box.low, box.high).astype(np.float32))
    assert np.array_equal(obs[:, :desc.n_q], qr) and np.array_equal(obs[:, desc.n_q:2 * desc.n_q], qdr)
    assert np.isfinite(rew).all()
    vec.close(); ref.close()


def test_upper_body_full_batch_shards_and_rest_equilibrium(upper_body):
    """8 192 envs = 2 shards of 4 096 (env_id_offset), driven by the device action stream; then the rest pose."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    n_shard, seed = 4096, 6
    whole = HipBatchSimulation(upper_body, 2 * n_shard, seed=seed)
    parts = [HipBatchSimulation(upper_body, n_shard, seed=seed, env_id_offset=r * n_shard) for r in (0, 1)]

    def run(sim):
        d = sim.malloc(4 * sim.n_envs * sim.n_t)
        for t in range(6):
            sim.fill_actions_dev(d, t)
            sim.step_dev(d, 0.3)
        sim.synchronize()
        return sim.read_state()
    qw, qdw, fw = run(whole)
    for r, p in enumerate(parts):
        qp, qdp, fp = run(p)
        sl = slice(r * n_shard, (r + 1) * n_shard)
        assert np.array_equal(qw[sl], qp) and np.array_equal(qdw[sl], qdp) and np.array_equal(fw[sl], fp)
        p.close()
    assert np.abs(qw).max() > 0.01 and np.isfinite(qw).all()
    whole.forward_reset_command()
    q, qd, f = whole.forward_step_command(np.zeros((2 * n_shard, whole.n_t), np.float32))
    # fp32 gravity terms cancel to ~1e-8 rad (tests/test_tree_robot_gpu.py), every env alike
    assert np.abs(q).max() < 1e-6 and np.abs(qd).max() < 1e-5 and f.all()
    assert np.array_equal(q[0], q[-1]) and np.array_equal(q[0], q[4097])
    whole.close()


@pytest.mark.parametrize("robot_name,n,integrator,steps", [("msj", 4096, "euler", 100000), ("msj", 262144, "rk4", 4000),
                                                           ("upper", 8192, "euler", 20000), ("upper", 20480, "rk4", 2000)])
def test_long_random_action_rollouts_stay_finite_and_inside_the_limits(msj_robot, upper_body, robot_name, n, integrator, steps):
    """Soak: tens of thousands of env steps under i.i.d. random set-points (a ring of 64 Philox slabs, hipGraph replay, the library's
    own kernel form and chains at each size) - every state stays finite, inside the joint limits and the speed limits, the
    feasible fraction stays high (limits are touched and left again, nothing sticks), and the statistics counter saw every step."""
    import torch
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    robot = msj_robot if robot_name == "msj" else upper_body
    desc = robot.get_description()
    sim = HipBatchSimulation(robot, n, integrator=integrator, seed=7)
    st = torch.cuda.Stream()
    sim.set_stream(st.cuda_stream)
    ring_n = 64
    ring = torch.empty(ring_n * n * desc.n_t, dtype=torch.float32, device="cuda")
    for r in range(ring_n):
        sim.fill_actions_dev(ring.data_ptr() + 4 * r * n * desc.n_t, r)
    done = 0
    while done < steps:
        k = min(1024, steps - done)
        sim.rollout_dev(ring.data_ptr(), ring_n, k, 0.3, use_graph=True)
        done += k
    sim.synchronize()
    q, qd, feas = sim.read_state()
    assert np.isfinite(q).all() and np.isfinite(qd).all()
    assert np.all(q >= desc.q_lo.astype(np.float32) - 1e-6) and np.all(q <= desc.q_hi.astype(np.float32) + 1e-6)
    assert np.all(np.abs(qd) <= desc.qd_max.astype(np.float32) * (1 + 1e-6))
    assert feas.mean() > 0.9, feas.mean()
    assert np.abs(q).max() > 0.05                                  # (and the robots did move)
    import ctypes
    from gym_roboy_amd import _native as nat
    out = (ctypes.c_double * 8)()
    nat.check(sim._lib.rb_env_stats(sim.handle, out, 0))
    assert out[6] == float(n) * steps                             # slot 6: env steps issued since the handle was created
    sim.close()


@pytest.mark.parametrize("case", ["graphs", "eager-head-graphs-tail", "upper-body"])
def test_consumers_behind_the_join_see_the_other_chains_writes(msj_robot, case):
    """The chains' fork / join events carry no system-scope fence (RB_CHAIN_EVENT_FLAGS, csrc/roboy_sim.hip): what orders the other
    chain's writes before the consumers behind the join is the packets' own agent-scope release / acquire
    (profiles/r6_a/chain_fence_scopes.log).  A batch whose second half starts at a block index that is not a multiple of 8, so
    that the kernel that reads the state right behind the join (pack_state_kernel: another block -> XCD map than the range
    launches) runs on other XCDs than the writers - a stale line in an XCD-private L2 would show against the same rollout stepped
    as one chain.  Bit for bit, in every shape a rollout call takes:
      graphs                   8 steps: the chains' graphs alone (150 rollouts)
      eager-head-graphs-tail   22 steps: one ring turn of plain launches on both chains, 16-step graphs, the join, two trailing
                               whole-batch launches on the handle's stream (they read what the other chain wrote)
      upper-body               the joint-tree chains (one wave per 64 envs, env-major rows) with a trailing partial turn"""
    import torch
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    if case == "upper-body":
        from gym_roboy_amd.envs.robots import UpperBodyRobot
        robot, n, integrator, steps, iters = UpperBodyRobot(), 65536 + 192, "euler", 18, 25
    else:
        robot, n, integrator = msj_robot, 262144 + 768, "rk4"
        steps, iters = (8, 150) if case == "graphs" else (22, 60)
    n_t = robot.get_description().n_t
    st = torch.cuda.Stream()
    sims = []
    for chains in (2, 1):
        s = HipBatchSimulation(robot, n, integrator=integrator, seed=1)
        s.set_stream(st.cuda_stream)
        s.set_rollout_chains(chains)
        sims.append(s)
    assert sims[0].rollout_chains() == 2 and sims[1].rollout_chains() == 1
    ring = torch.empty(4 * n * n_t, dtype=torch.float32, device="cuda")
    for r in range(4):
        sims[0].fill_actions_dev(ring.data_ptr() + 4 * r * n * n_t, r)
    for it in range(iters):
        outs = []
        for s in sims:
            s.rollout_dev(ring.data_ptr(), 4, steps, 0.3, use_graph=True)
            outs.append(s.read_state())
        for a, b in zip(*outs):
            assert np.array_equal(a, b), "rollout %d: the reader behind the join saw stale state" % it
        if case == "upper-body" and it % 8 == 7:      # keep the trees inside their boxes: restart from the reset state
            for s in sims:
                s.forward_reset_command()
    for s in sims:
        s.close()
