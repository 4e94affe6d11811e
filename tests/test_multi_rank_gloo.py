"""N > 1 path on CPU: two gloo ranks own contiguous env shards, replay the env
layer over the C oracle, all-reduce the 8-double statistics vector exactly as
bench.py does over RCCL, and must reproduce the single-process result of the
whole batch (streams are keyed by the global env id, so sharding is invisible)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL, STEPS, MAX_LEN, SEED = 301, 25, 8, 11


def _rollout(first, count):
    from gym_roboy_amd.envs.robots import MsjRobot
    from host_env_model import COracleStepper, HostEnvModel
    from oracle import philox_np as ph
    robot = MsjRobot()
    model = HostEnvModel(robot, COracleStepper(robot, count), count, SEED, MAX_LEN, False, True, True,
                         env_id_offset=first)
    ids = np.arange(first, first + count, dtype=np.uint64)
    last = None
    for t in range(STEPS):
        last = model.step(ph.actions(SEED, ids, t, 8))
    return model.stats.copy(), last[0]


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    from gym_roboy_amd.sharding import allreduce_stats, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_bounds(N_TOTAL, world, rank)
    stats, obs = _rollout(first, count)
    t = torch.from_numpy(stats.copy())
    allreduce_stats(t, dist)
    # time-like quantities are reduced with MAX, as bench.py does
    wall = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    np.save(os.path.join(out_dir, "stats_%d.npy" % rank), t.numpy())
    np.save(os.path.join(out_dir, "obs_%d.npy" % rank), obs)
    np.save(os.path.join(out_dir, "wall_%d.npy" % rank), wall.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_batch_exactly():
    from gym_roboy_amd.sharding import shard_bounds
    for n in (0, 1, 7, 8, 301, 2097152):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    assert shard_bounds(2097152, 8, 3) == (3 * 262144, 262144)
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def test_two_gloo_ranks_reproduce_the_single_process_batch(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole_stats, whole_obs = _rollout(0, N_TOTAL)
    s0, s1 = (np.load(tmp_path / ("stats_%d.npy" % r)) for r in (0, 1))
    assert np.array_equal(s0, s1)                        # every rank holds the global sum
    np.testing.assert_allclose(s0, whole_stats, rtol=1e-12)
    assert s0[6] == N_TOTAL * STEPS and s0[2] > 0
    obs = np.concatenate([np.load(tmp_path / ("obs_%d.npy" % r)) for r in (0, 1)])
    assert np.array_equal(obs, whole_obs)                # sharding is invisible, bit for bit
    assert np.load(tmp_path / "wall_0.npy")[0] == 2.0    # max over ranks

    from gym_roboy_amd.sharding import summarize
    d = summarize(s0)
    assert d["mean_length"] <= MAX_LEN and 0 <= d["goal_rate"] <= 1
