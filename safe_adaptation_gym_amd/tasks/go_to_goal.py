from safe_adaptation_gym_amd.tasks.task import GOAL_KEEPOUT, GOAL_PLACEMENT, Task


class GoToGoal(Task):
  """Reference tasks/go_to_goal.py.  Device side: 3-D progress reward, goal-met resample (sag_device.hpp, task
  switch); the goal's yaw draw and the reset-time resample (go_to_goal.py:50-80) are the native sampler's."""
  NAME, TASK_ID, HAS_GOAL = 'go_to_goal', 3, True
  GOAL_SIZE = 0.3
  GOAL_KEEPOUT = GOAL_KEEPOUT

  def setup_placements(self):
    return {'goal': (GOAL_PLACEMENT, self.GOAL_KEEPOUT)}

  @property
  def obstacles(self):
    return [9, 10, 0, 1]


class GoToGoalScarce(GoToGoal):
  """tasks/go_to_goal_scarce.py: progress reward only within 1.5 goal sizes."""
  NAME, TASK_ID = 'go_to_goal_scarce', 6


class GoToGoalMotor(GoToGoal):
  """tasks/go_to_goal_motor.py:12-16: motor-x gear x10."""
  NAME, TASK_ID = 'go_to_goal_motor', 5
  GEAR = 0.3 * 10.

  def modify_tree(self, rs):
    return [(('actuator', 'x'), ('gear', f'{self.GEAR} 0 0 0 0 0'))]


class GoToGoalDamping(GoToGoal):
  """tasks/go_to_goal_damping.py:12-17: slide damping x0.1."""
  NAME, TASK_ID = 'go_to_goal_damping', 4
  DAMPING = 0.01 * 0.1

  def modify_tree(self, rs):
    return [(('joint', 'y'), ('damping', self.DAMPING)), (('joint', 'x'), ('damping', self.DAMPING))]


class CatchGoal(GoToGoal):
  """tasks/catch_goal.py: the goal orbits its origin (= the goal position after reset); radii / timer live in the
  device task state."""
  NAME, TASK_ID = 'catch_goal', 0
  MIN_RADIUS, MAX_RADIUS, SAMPLE_POINTS = 0.2, 1.0, 10


class Unsupervised(GoToGoal):
  """tasks/unsupervised.py: reward = [circle_reward, goal_reward]."""
  NAME, TASK_ID, REWARD_DIM = 'unsupervised', 13, 2

  @property
  def obstacles(self):
    return [5, 6, 0, 1]
