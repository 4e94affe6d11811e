"""``RoboyEnv``: the single-env ``gym.GoalEnv`` in front of a ``SimulationClient``.

Drop-in for ``/root/reference/gym_roboy/envs/roboy_env.py:10-134``: same
constructor, attributes (``step_num``, ``reward_range``, ``action_space``,
``observation_space``, ``_goal_state``, ``_last_state`` ...) and behaviour,
including the quirks listed in SURVEY.md §8c: observations are float64, the
step counter starts at 1 and an episode is done when it exceeds 400, reaching
``done`` only resamples the goal (no simulator reset, no counter reset).  The
arithmetic lives in ``reward.py`` so the batched device env layer can be
checked against the very same functions.

This class is plumbing (one Python call per env step, ~10^4 steps/s at best);
the throughput path is ``RoboyVecEnv`` / ``HipBatchSimulation``.
"""
from typing import Tuple

import numpy as np

from .._gymcompat import BaseGoalEnv, spaces
from . import reward as rw
from .robots import RobotState, RoboyRobot
from .simulations import SimulationClient


class RoboyEnv(BaseGoalEnv):

    def __init__(self, simulation_client: SimulationClient = None, seed: int = None,
                 joint_vel_penalty: bool = False,
                 is_agent_getting_bonus_for_reaching_goal: bool = True):
        if simulation_client is None:
            # gym.make('msj-control-v1') with no kwargs: the reference raises a
            # TypeError here (its register() passes none, gym_roboy/__init__.py:3-6);
            # the drop-in default is the in-process HIP client.
            from .robots import MsjRobot
            from .simulations import HipSimulationClient
            simulation_client = HipSimulationClient(robot=MsjRobot())
        self.seed(seed)
        self._simulation_client = simulation_client
        self._joint_vel_penalty = joint_vel_penalty
        self._is_agent_getting_bonus_for_reaching_goal = is_agent_getting_bonus_for_reaching_goal
        self._robot = robot = simulation_client.robot
        self._last_state = None   # type: RobotState
        self._goal_state = None   # type: RobotState

        angles, vels = robot.get_joint_angles_space(), robot.get_joint_vels_space()
        self._GOAL_JOINT_VEL = robot.new_zero_state().joint_angles
        self._MAX_DISTANCE_JOINT_ANGLE = _l2_distance(angles.low, angles.high)
        self._MAX_DISTANCE_JOINT_VELS = _l2_distance(vels.low, vels.high)
        self._PENALTY_FOR_TOUCHING_BOUNDARY = 1
        self._BONUS_FOR_REACHING_GOAL = 1000
        self._MAX_EPISODE_LENGTH = 400

        self.reward_range = self._create_reward_range(robot=robot)
        self.action_space = spaces.Box(low=-1, high=1, shape=robot.get_action_space().shape,
                                       dtype="float32")
        self.observation_space = spaces.Box(
            low=np.concatenate((angles.low, vels.low, angles.low)),
            high=np.concatenate((angles.high, vels.high, angles.high)),
            dtype="float32")
        self._set_new_goal()
        self.step_num = 1

    # ------------------------------------------------------------- gym API
    def step(self, action):
        assert self.action_space.contains(action)
        setpoints = _rescale_from_one_space_to_other(
            input_val=np.array(action), input_space=self.action_space,
            output_space=self._robot.get_action_space()).tolist()

        new_state = self._simulation_client.forward_step_command(setpoints)
        self.step_num += 1
        self._last_state = new_state
        obs = self._make_obs(robot_state=new_state)
        info = {}
        reward = self.compute_reward(current_state=new_state, goal_state=self._goal_state, info=info)
        done = self._did_reach_goal(current_state=new_state, goal_state=self._goal_state) \
            or self._reached_max_steps()
        if done:
            self._set_new_goal()
        return obs, reward, done, info

    def reset(self):
        self._simulation_client.forward_reset_command()
        self._last_state = self._simulation_client.read_state()
        self.step_num = 1
        self._set_new_goal()
        return self._make_obs(robot_state=self._last_state)

    def render(self, mode="human"):
        pass

    def seed(self, seed=None):
        np.random.seed(seed)

    def compute_reward(self, current_state: RobotState, goal_state: RobotState, info=None):
        robot = self._robot
        angles, vels = robot.get_joint_angles_space(), robot.get_joint_vels_space()
        reward = rw.compute_reward(
            q=current_state.joint_angles, qd=current_state.joint_vels,
            feasible=current_state.is_feasible,
            goal_q=goal_state.joint_angles, goal_qd=goal_state.joint_vels,
            angle_box=(angles.low, angles.high), vel_box=(vels.low, vels.high),
            max_dist_angle=self._MAX_DISTANCE_JOINT_ANGLE,
            max_dist_vel=self._MAX_DISTANCE_JOINT_VELS,
            joint_vel_penalty=self._joint_vel_penalty,
            goal_bonus_enabled=False,
            boundary_penalty=self._PENALTY_FOR_TOUCHING_BOUNDARY)
        # the bonus goes through _did_reach_goal so the banner prints as in the reference
        if self._did_reach_goal(current_state=current_state, goal_state=goal_state) and \
                self._is_agent_getting_bonus_for_reaching_goal:
            reward = reward + self._BONUS_FOR_REACHING_GOAL
        assert self.reward_range[0] <= reward <= self.reward_range[1], \
            "'{}' not between '{}' and '{}'".format(reward, self.reward_range[0], self.reward_range[1])
        return float(reward)

    # ------------------------------------------------------------ internals
    def _create_reward_range(self, robot: RoboyRobot) -> Tuple[float, float]:
        angles, vels = robot.get_joint_angles_space(), robot.get_joint_vels_space()
        best = robot.new_state(joint_angle=angles.high, joint_vel=vels.high, is_feasible=True)
        worst = robot.new_state(joint_angle=angles.low, joint_vel=vels.low, is_feasible=False)
        # evaluated before reward_range exists, so without the range assert
        self.reward_range = (-float("inf"), float("inf"))
        max_reward = self.compute_reward(current_state=best, goal_state=best)
        min_reward = self.compute_reward(current_state=worst, goal_state=best)
        return min_reward, max_reward

    def _reached_max_steps(self) -> bool:
        return self.step_num > self._MAX_EPISODE_LENGTH

    def _make_obs(self, robot_state: RobotState):
        return np.concatenate([robot_state.joint_angles, robot_state.joint_vels,
                               self._goal_state.joint_angles])

    def _set_new_goal(self, goal_joint_angle=None):
        """Use the given goal, or ask the simulator for a random feasible one."""
        if goal_joint_angle is None:
            goal_joint_angle = self._simulation_client.get_new_goal_joint_angles()
        self._goal_state = self._robot.new_state(joint_angle=goal_joint_angle,
                                                 joint_vel=self._GOAL_JOINT_VEL,
                                                 is_feasible=True)

    def _did_reach_goal(self, current_state: RobotState, goal_state: RobotState) -> bool:
        reached = bool(rw.did_reach_goal(
            current_state.joint_angles, current_state.joint_vels,
            goal_state.joint_angles, goal_state.joint_vels,
            self._MAX_DISTANCE_JOINT_ANGLE, self._MAX_DISTANCE_JOINT_VELS))
        if reached:
            print("#############GOAL REACHED#############")
        return reached


def _l2_distance(joint_angle1, joint_angle2):
    # numpy scalar of the inputs' dtype, as in the reference: the goal
    # thresholds derived from it (:127,:130) are therefore float32 values
    return rw.l2_distance(joint_angle1, joint_angle2)


def _rescale_from_one_space_to_other(input_val: np.ndarray, input_space, output_space) -> np.ndarray:
    """Map a point of ``input_space`` to the point of ``output_space`` that is
    equally far from the bounds (reference ``roboy_env.py:143-158``)."""
    if not isinstance(input_val, np.ndarray):
        raise TypeError('type of argument "input_val" must be numpy.ndarray')
    assert input_space.shape == output_space.shape
    assert input_space.contains(input_val)
    return rw.rescale_between_boxes(input_val, input_space.low, input_space.high,
                                    output_space.low, output_space.high)
