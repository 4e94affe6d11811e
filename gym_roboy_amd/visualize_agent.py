"""Play a trained agent back in one environment and print its rewards.

Counterpart of ``/root/reference/gym_roboy/visualize_agent.py``
(``python -m gym_roboy_amd.visualize_agent <model.pkl>``): the reference loads a
stable_baselines ``PPO2`` model, steps a single ROS-backed ``RoboyEnv`` with it
and logs each reward with a 30 ms pause (:18-25,28-43).  Here the model is the
checkpoint written by ``gym_roboy_amd.train_parallel`` and the env is a
``RoboyEnv`` over the in-process ``HipSimulationClient``.
"""
import argparse
import time


class Logger:
    """Prints the step reward and paces the loop (reference :18-25)."""

    def __init__(self, pause_secs: float = 0.03):
        self.pause_secs = pause_secs
        self.total = 0.0

    def log(self, step: int, reward: float):
        self.total += reward
        print("step %4d  reward %10.4f  return %12.4f" % (step, reward, self.total))
        if self.pause_secs > 0:
            time.sleep(self.pause_secs)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("model", help="model.pkl written by gym_roboy_amd.train_parallel")
    ap.add_argument("--steps", type=int, default=400, help="the reference loops forever; one episode by default")
    ap.add_argument("--pause", type=float, default=0.03)
    args = ap.parse_args(argv)

    import numpy as np
    import torch
    from .envs import RoboyEnv
    from .envs.robots import MsjRobot
    from .envs.simulations import HipSimulationClient
    from .ppo import MlpPolicy

    env = RoboyEnv(simulation_client=HipSimulationClient(robot=MsjRobot()))
    policy = MlpPolicy(env.observation_space.shape[0], env.action_space.shape[0])
    policy.load_state_dict(torch.load(args.model, map_location="cpu")["policy"])
    logger = Logger(args.pause)
    obs = env.reset()
    for step in range(args.steps):
        with torch.no_grad():
            action, _, _ = policy.act(torch.as_tensor(obs, dtype=torch.float32)[None], deterministic=True)
        obs, reward, done, _ = env.step(np.clip(action[0].numpy(), -1.0, 1.0).astype(np.float32))
        logger.log(step, reward)
        if done:
            obs = env.reset()
    return logger.total


if __name__ == "__main__":
    main()
