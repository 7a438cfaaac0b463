"""Task descriptors.

In the reference a Task object computes rewards in Python against MujocoBridge
(tasks/task.py:14-97).  Here the per-step logic (compute_reward / set_mocaps /
on-goal reset) runs on the GPU, selected by `task_id`; what stays on the host is
what the reference does at World construction and reset time: placements, obstacle
counts, extents, the yaw / goal / button draws (in the reference's draw order,
SURVEY App. B) and the per-task dynamics variants."""
import numpy as np

from safe_adaptation_gym_amd import consts, utils


class Task:
  NAME = None        # snake_case registry key (benchmark.TASKS)
  TASK_ID = None     # enum sag_task
  BOX_KIND = 0       # enum sag_box_kind
  NUM_BUTTONS = 0
  HAS_GOAL = False
  GEAR = 0.3         # point motor-x gear (point.xml:36)
  DAMPING = 0.01     # point slide damping (point.xml:15-16)
  REWARD_DIM = 1

  def __init__(self):
    self._ctrl_scale = None
    self._bound = None

  # -- reference surface (tasks/task.py) ---------------------------------------
  def setup_placements(self):
    raise NotImplementedError

  @property
  def obstacles(self):
    """[hazards, vases, gremlins, pillars] counts (tasks/task.py:63-70)."""
    return [4, 5, 0, 1]

  @property
  def placement_extents(self):
    return consts.PLACEMENT_EXTENTS

  @property
  def arena_radius(self):
    return self.placement_extents[2] * np.sqrt(2.)

  def ctrl_scale(self, rs, control_size):
    # one Cauchy draw per Task instance, cached (tasks/task.py:85-89)
    if self._ctrl_scale is None:
      self._ctrl_scale = rs.standard_cauchy(control_size)
    return self._ctrl_scale

  def constraint_bound(self, rs, max_bound):
    if self._bound is None:
      self._bound = rs.uniform(0., max_bound)
    return self._bound

  def modify_tree(self, rs):
    return None

  # -- host halves of build_world_config / reset ---------------------------------
  def draw_world_config(self, layout, rs):
    """Consume the yaw draws of the task's build_world_config in the reference's
    order; may edit the layout (HaulBox).  Returns {body name: yaw}."""
    raise NotImplementedError

  def reset(self, layout, placements, rs, state):
    """Host half of task.reset(): draws that need the env's RandomState.
    `state` is a dict of task-state fields carried into the device record."""
    raise NotImplementedError
