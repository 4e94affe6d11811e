"""Training driver: PPO over a GPU-resident batch of RoboyEnvs.

Counterpart of ``/root/reference/gym_roboy/train_parallel.py`` (same CLI:
``python -m gym_roboy_amd.train_parallel <num_envs> [results_dir]``): where the
reference builds ``num_cpu`` ROS-backed envs in ``num_cpu`` processes (:19-29),
this builds one ``RoboyVecEnv`` of ``num_envs`` envs per GPU; like the
reference it trains in rounds of 100 000 timesteps and saves ``model.pkl``
after each (:16,33-35; the reference loops forever, here ``--rounds`` bounds it).
Launch with ``python -m torch.distributed.run --nproc-per-node N ...`` for N
GPUs: env shards by ``env_id_offset``; every rank replays its own captured
rollout graph (policy step, env step, GAE are rank-local), the gradient
vector is averaged over RCCL once per minibatch and the episode statistics are
summed over the ranks (``sharding.allreduce_stats``) before rank 0 prints them.
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
                    help="launch every kernel from Python instead of replaying HIP graphs")
    ap.add_argument("--n-steps", type=int, default=128, help="rollout length per update (stable_baselines' default)")
    ap.add_argument("--backend", default=os.environ.get("ROBOY_TRAIN_BACKEND", "nccl"), choices=("nccl", "gloo"),
                    help="nccl = RCCL over xGMI, one rank per GPU; gloo = rehearsal with the ranks sharing the visible GPUs")
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
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    results = os.path.abspath(args.results_dir)
    model_file = os.path.join(results, "model.pkl")
    if rank == 0:
        os.makedirs(results, exist_ok=True)

    env = RoboyVecEnv(MsjRobot(), args.num_envs, seed=args.seed, device=local_rank,
                      env_id_offset=rank * args.num_envs)
    more_exploration = 0.1                      # train_parallel.py:30
    agent = PPO(env, n_steps=args.n_steps, ent_coef=more_exploration, device="cuda", dist=dist, seed=args.seed,
                reward_scale=0.01, use_graphs=not args.no_graphs, fused_policy=not args.torch_policy,
                fused_update=not args.torch_policy)
    if os.path.exists(model_file):
        agent.load(model_file)                  # resume from the last backup
    for _ in range(args.rounds):
        agent.learn(total_timesteps=args.steps_per_round,
                    log=(lambda s: print({k: round(v, 4) for k, v in s.items()})) if rank == 0 else None)
        # episode statistics of ALL ranks' envs: the 8-double block of every rank, summed (RCCL; gloo: through the host)
        from .sharding import STAT_KEYS, allreduce_stats, summarize
        local = env.stats()
        block = torch.tensor([local[k] for k in STAT_KEYS], dtype=torch.float64,
                             device="cuda" if (dist is not None and args.backend == "nccl") else "cpu")
        allreduce_stats(block, dist)
        if rank == 0:
            agent.save(model_file)
            print("episode statistics (all ranks):", summarize(block.cpu()))
    if world > 1:
        dist.destroy_process_group()
    return agent


if __name__ == "__main__":
    main()
