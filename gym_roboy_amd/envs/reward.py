"""Reward / termination arithmetic of the env layer, batched.

Pure numpy functions over a leading env axis, so the single-env ``RoboyEnv``
and the host-side checks of the fused device env layer share one statement of
the reference's formulas (``/root/reference/gym_roboy/envs/roboy_env.py``):

* ``l2_distance``           - ``_l2_distance``            (:137-140)
* ``rescale_between_boxes`` - ``_rescale_from_one_space_to_other`` (:143-158)
* ``normalize``             - ``RoboyRobot._normalize_between_minus1_and1`` (roboy_robot.py:93-95)
* ``did_reach_goal``        - ``_did_reach_goal``         (:125-134)
* ``compute_reward``        - ``compute_reward``          (:92-112)
"""
import numpy as np


def l2_distance(a, b):
    diff = np.subtract(a, b)
    diff = np.where(np.isnan(diff), 0, diff)   # inf - inf counts as 0 (:139)
    if diff.ndim == 1:   # the reference's exact call, so single-env values match it bit for bit
        return np.linalg.norm(diff, ord=2)
    return np.sqrt(np.sum(diff * diff, axis=-1))


def rescale_between_boxes(x, in_low, in_high, out_low, out_high):
    """Point of the output box that is as far from its bounds as ``x`` is from
    the input box's: ``slope * (x - in_high) + out_high`` in that order."""
    slope = (out_high - out_low) / (in_high - in_low)
    return slope * (x - in_high) + out_high


def normalize(val, max_val, min_val):
    return (2 * val - max_val - min_val) / (max_val - min_val)


def did_reach_goal(q, qd, goal_q, goal_qd, max_dist_angle, max_dist_vel):
    angles_close = l2_distance(q, goal_q) < max_dist_angle / 200
    vels_close = l2_distance(qd, goal_qd) < max_dist_vel / 5
    return np.logical_and(angles_close, vels_close)


def compute_reward(q, qd, feasible, goal_q, goal_qd, angle_box, vel_box, max_dist_angle,
                   max_dist_vel, joint_vel_penalty, goal_bonus_enabled,
                   boundary_penalty=1.0, goal_bonus=1000.0):
    """Reward for states [..., n_q]; ``angle_box``/``vel_box`` are (low, high)."""
    qn = normalize(q, angle_box[1], angle_box[0])
    gn = normalize(goal_q, angle_box[1], angle_box[0])
    reward = -np.exp(l2_distance(qn, gn))
    if joint_vel_penalty:
        vn = normalize(qd, vel_box[1], vel_box[0])
        gvn = normalize(goal_qd, vel_box[1], vel_box[0])
        dv = vn - gvn
        speed = np.linalg.norm(dv) if dv.ndim == 1 else np.sqrt(np.sum(dv * dv, axis=-1))
        reward = (speed + 1) * (reward - np.exp(reward))
    reward = np.where(np.asarray(feasible, dtype=bool), reward, reward - abs(boundary_penalty))
    if goal_bonus_enabled:
        reached = did_reach_goal(q, qd, goal_q, goal_qd, max_dist_angle, max_dist_vel)
        reward = np.where(reached, reward + goal_bonus, reward)
    return reward
