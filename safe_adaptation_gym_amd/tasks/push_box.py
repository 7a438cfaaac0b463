from safe_adaptation_gym_amd.tasks.go_to_goal import GoToGoal


class PushBox(GoToGoal):
  """Reference tasks/push_box.py: a free box (main body + four corner columns) to be pushed onto the goal."""
  NAME, TASK_ID, BOX_KIND, BOX_YAW = 'push_box', 10, 1, True
  BOX_SIZE = 0.2
  BOX_KEEPOUT = 0.5
  BOX_DENSITY = 0.001

  def setup_placements(self):
    placements = super().setup_placements()
    placements.update({'box': (None, self.BOX_KEEPOUT)})
    return placements

  @property
  def obstacles(self):
    return [2, 3, 0, 1]

  @property
  def placement_extents(self):
    return [-1.75, -1.75, 1.75, 1.75]


class PushBoxScarce(PushBox):
  """tasks/push_box_scarce.py: box placed in a fixed rectangle, keepout x1.1."""
  NAME, TASK_ID = 'push_box_scarce', 11

  def setup_placements(self):
    placements = GoToGoal.setup_placements(self)
    placements.update({'box': ([(-2.25, -2.25, 2.25, 2.25)], self.BOX_KEEPOUT * 1.1)})
    return placements


class HaulBox(PushBox):
  """tasks/haul_box.py: box spawned 3 box sizes ahead of the robot (world +x, haul_box.py:17-18) and tied to it by a
  length-limited tendon."""
  NAME, TASK_ID = 'haul_box', 7
  BOX_AT_ROBOT = PushBox.BOX_SIZE * 3.


class RollRod(PushBox):
  """tasks/roll_rod.py: the object is a cylinder lying on its side; no yaw draw."""
  NAME, TASK_ID, BOX_KIND, BOX_YAW = 'roll_rod', 12, 2, False
  ROD_LENGTH, ROD_RADIUS = 0.3, 0.08
  BOX_KEEPOUT = 0.7
  BOX_SIZE = 0.25

  @property
  def placement_extents(self):
    return -1.75, -1.75, 1.75, 1.75


class DribbleBall(PushBox):
  """tasks/dribble_ball.py: the object is a sphere; no yaw draw."""
  NAME, TASK_ID, BOX_KIND, BOX_YAW = 'dribble_ball', 2, 3, False
  SPHERE_RADIUS = 0.14
  BOX_KEEPOUT = 0.2
  BOX_SIZE = SPHERE_RADIUS

  @property
  def placement_extents(self):
    return -1.75, -1.75, 1.75, 1.75
