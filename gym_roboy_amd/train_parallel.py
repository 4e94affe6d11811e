"""Training driver: PPO over a GPU-resident batch of RoboyEnvs.

Counterpart of ``/root/reference/gym_roboy/train_parallel.py`` (same CLI:
``python -m gym_roboy_amd.train_parallel <num_envs> [results_dir]``): where the
reference builds ``num_cpu`` ROS-backed envs in ``num_cpu`` processes (:19-29),
this builds one ``RoboyVecEnv`` of ``num_envs`` envs per GPU; like the
reference it trains in rounds of 100 000 timesteps and saves ``model.pkl``
after each (:16,33-35; the reference loops forever, here ``--rounds`` bounds it).
Launch with ``python -m torch.distributed.run --nproc-per-node N ...`` for N
GPUs: env shards by ``env_id_offset``, gradients averaged over RCCL.
"""
import argparse
import os

TRAINING_STEPS_BETWEEN_BACKUPS = 100000


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("num_envs", type=int, help="environments per GPU (the reference's num_cpu)")
    ap.add_argument("results_dir", nargs="?", default="./training_results")
    ap.add_argument("--rounds", type=int, default=1, help="rounds of 100 000 timesteps (reference: unbounded)")
    ap.add_argument("--steps-per-round", type=int, default=TRAINING_STEPS_BETWEEN_BACKUPS)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-graphs", action="store_true",
                    help="launch every kernel from Python instead of replaying HIP graphs (single-rank runs use graphs)")
    ap.add_argument("--torch-policy", action="store_true",
                    help="policy step and minibatch gradient through torch (autograd, rocBLAS) instead of the fused "
                         "matrix-core kernels of include/roboy_policy.h")
    args = ap.parse_args(argv)

    import torch
    from .envs.robots import MsjRobot
    from .envs.vec_env import RoboyVecEnv
    from .ppo import PPO

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    results = os.path.abspath(args.results_dir)
    model_file = os.path.join(results, "model.pkl")
    if rank == 0:
        os.makedirs(results, exist_ok=True)

    env = RoboyVecEnv(MsjRobot(), args.num_envs, seed=args.seed, device=local_rank,
                      env_id_offset=rank * args.num_envs)
    more_exploration = 0.1                      # train_parallel.py:30
    agent = PPO(env, ent_coef=more_exploration, device="cuda", dist=dist, seed=args.seed, reward_scale=0.01,
                use_graphs=(world == 1 and not args.no_graphs), fused_policy=not args.torch_policy,
                fused_update=not args.torch_policy)
    if os.path.exists(model_file):
        agent.load(model_file)                  # resume from the last backup
    for _ in range(args.rounds):
        agent.learn(total_timesteps=args.steps_per_round,
                    log=(lambda s: print({k: round(v, 4) for k, v in s.items()})) if rank == 0 else None)
        if rank == 0:
            agent.save(model_file)
            print("episode statistics:", env.stats())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
