from safe_adaptation_gym_amd.tasks.task import Task


class PressButtons(Task):
  """Reference tasks/press_buttons.py.  The NORMAL / BUTTON_CHANGE state machine and the 5-tick timer run on the
  device; the first goal button is the sampler's rs.choice (press_buttons.py:71-77)."""
  NAME, TASK_ID = 'press_buttons', 8
  NUM_BUTTONS = 4
  BUTTON_RESET = 1
  BUTTONS_KEEPOUT = 0.2
  BUTTON_SIZE = 0.1
  BUTTON_TICKING_DELAY = 5
  _RECT = (-1.35, -1.35, 1.35, 1.35)

  def setup_placements(self):
    return {f'buttons{i}': ([self._RECT], self.BUTTONS_KEEPOUT) for i in range(self.NUM_BUTTONS)}

  @property
  def obstacles(self):
    return [6, 8, 0, 0]


class PressButtonsScarce(PressButtons):
  """tasks/press_buttons_scarce.py: reward only on a press."""
  NAME, TASK_ID = 'press_buttons_scarce', 9
  _RECT = (-1.75, -1.75, 1.75, 1.75)


class Collect(PressButtons):
  """tasks/collect.py: six goal buttons, each collected once; no host draws at reset."""
  NAME, TASK_ID = 'collect', 1
  NUM_BUTTONS = 6
  BUTTON_RESET = 2
  _RECT = (-1.5, -1.5, 1.5, 1.5)

  @property
  def placement_extents(self):
    return [-2.25, -2.25, 2.25, 2.25]
