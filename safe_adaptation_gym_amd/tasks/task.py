"""Task descriptors.

In the reference a Task object computes rewards in Python against MujocoBridge
(tasks/task.py:14-97).  Here the per-step logic (compute_reward / set_mocaps /
on-goal reset) runs on the GPU, selected by `TASK_ID`; what stays on the host is what the
reference reads from the Task at World construction and reset time: `obstacles`,
`placement_extents`, `setup_placements()` and the attributes below.  `descriptor()` turns them into the
`sag_task_desc` that the native sampler (csrc/sag_sampler.cpp) draws layouts from, in the reference's
draw order - so a subclass that overrides them changes the worlds it gets.  What a subclass
can NOT do is redefine the per-step logic from Python: overriding compute_reward & co raises."""
import numpy as np

from safe_adaptation_gym_amd import consts

# the device-side goal resampling (tasks/go_to_goal.py:59-80 inside step()) has these built in
GOAL_PLACEMENT = [(-1.5, -1.5, 1.5, 1.5)]
GOAL_KEEPOUT = 0.4
MAX_COUNTS = {'hazards': 9, 'vases': 10, 'pillars': 2, 'buttons': 6}   # record capacity (include/sag.h)
_DEVICE_SIDE = ('compute_reward', 'compute_cost', 'set_mocaps', 'reset', 'build_world_config', 'compute_info')


class Task:
  NAME = None        # snake_case registry key (benchmark.TASKS)
  TASK_ID = None     # enum sag_task: the device-side per-step logic
  BOX_KIND = 0       # enum sag_box_kind
  BOX_YAW = False    # build_world_config draws a yaw for the task object (push_box.py:28-72)
  BOX_AT_ROBOT = 0.  # > 0 (haul_box.py:17-18): after sampling, the object sits this far ahead (world +x) of the robot
  NUM_BUTTONS = 0
  BUTTON_RESET = 0   # task.reset: 1 = rs.choice(NUM_BUTTONS) + ticking delay (press_buttons.py:71-77), 2 = all active (collect.py)
  BUTTON_TICKING_DELAY = 5
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

  # -- what the native sampler and the kernels are told about this task -----------------------
  def descriptor(self):
    """dict with the fields of `sag_task_desc` (include/sag.h), built from this object's own surface."""
    cls = type(self).__name__
    if self.TASK_ID is None:
      raise TypeError(f'{cls}: no TASK_ID - the per-step logic (reward, mocaps, on-goal reset) runs on the device and a task '
                      'must name the one it uses (subclass one of the 14 reference tasks)')
    for name in _DEVICE_SIDE:
      if hasattr(type(self), name):
        raise NotImplementedError(f'{cls}.{name}: the per-step / reset logic of a task runs on the device (selected by TASK_ID = '
                                  f'{self.TASK_ID}); it cannot be overridden from Python')
    obs = list(self.obstacles)
    if len(obs) != 4 or obs[2] != 0:
      raise ValueError(f'{cls}.obstacles = {obs}: [hazards, vases, gremlins, pillars] with no gremlins (no reference task spawns them)')
    for kind, cnt in zip(('hazards', 'vases', 'pillars'), (obs[0], obs[1], obs[3])):
      if not 0 <= cnt <= MAX_COUNTS[kind]:
        raise ValueError(f'{cls}: {cnt} {kind} (the record holds 0..{MAX_COUNTS[kind]})')
    ext = [float(v) for v in self.placement_extents]
    pl = dict(self.setup_placements())
    d = {'task_id': int(self.TASK_ID), 'n_hazards': obs[0], 'n_vases': obs[1], 'n_pillars': obs[3], 'has_goal': 0,
         'box_kind': int(self.BOX_KIND), 'box_yaw': int(bool(self.BOX_YAW)), 'box_at_robot': int(self.BOX_AT_ROBOT > 0),
         'n_buttons': 0, 'button_reset': int(self.BUTTON_RESET), 'button_timer': int(self.BUTTON_TICKING_DELAY),
         'extents': ext, 'goal_keepout': GOAL_KEEPOUT, 'box_keepout': 0., 'button_keepout': 0., 'box_offset': float(self.BOX_AT_ROBOT),
         'box_rect': [0.] * 4, 'button_rect': [0.] * 4, 'gear': float(self.GEAR), 'damping': float(self.DAMPING)}

    def one_rect(key, rects):
      if rects is None:
        return [0.] * 4
      if len(rects) != 1:
        raise NotImplementedError(f'{cls}.setup_placements()[{key!r}]: {len(rects)} rectangles (the sampler draws from one)')
      return [float(v) for v in rects[0]]

    keys = list(pl)
    if 'goal' in pl:
      rects, keepout = pl.pop('goal')
      if [tuple(r) for r in (rects or [])] != GOAL_PLACEMENT or keepout != GOAL_KEEPOUT:
        raise NotImplementedError(f'{cls}: goal placement {rects}, keepout {keepout} - the in-step goal resampling on the device is built '
                                  f'for {GOAL_PLACEMENT}, {GOAL_KEEPOUT}')
      d['has_goal'] = 1
    if bool(d['has_goal']) != bool(self.HAS_GOAL):
      raise ValueError(f'{cls}: HAS_GOAL = {self.HAS_GOAL} but setup_placements() has {"a" if d["has_goal"] else "no"} goal')
    if 'box' in pl:
      rects, keepout = pl.pop('box')
      d['box_rect'], d['box_keepout'] = one_rect('box', rects), float(keepout)
    if ('box' in keys) != bool(self.BOX_KIND):
      raise ValueError(f'{cls}: BOX_KIND = {self.BOX_KIND} but setup_placements() has {"a" if "box" in keys else "no"} box')
    btn = [k for k in pl if k.startswith('buttons')]
    if btn:
      if btn != [f'buttons{i}' for i in range(len(btn))] or len(btn) > MAX_COUNTS['buttons']:
        raise ValueError(f'{cls}: button placements {btn} (buttons0 .. buttons{MAX_COUNTS["buttons"] - 1}, in order)')
      first = pl[btn[0]]
      if any(pl[k] != first for k in btn):
        raise NotImplementedError(f'{cls}: the buttons share one rectangle and keepout')
      d['n_buttons'], d['button_rect'], d['button_keepout'] = len(btn), one_rect('buttons', first[0]), float(first[1])
      for k in btn:
        pl.pop(k)
    if d['n_buttons'] != self.NUM_BUTTONS:
      raise ValueError(f'{cls}: NUM_BUTTONS = {self.NUM_BUTTONS} but setup_placements() places {d["n_buttons"]}')
    if pl:
      raise NotImplementedError(f'{cls}.setup_placements(): {sorted(pl)} - the device world has a goal, a task object and buttons')
    order = [k for k in keys if k in ('goal', 'box')] + [k for k in keys if k.startswith('buttons')]
    if keys != order or ('goal' in keys and 'box' in keys and keys.index('goal') > keys.index('box')):
      raise NotImplementedError(f'{cls}.setup_placements(): order {keys} (the sampler draws goal, box, buttons in this order)')
    # what the library itself would refuse (button_timer outside the device's 3-bit field or different from the constant it
    # re-arms with, negative / non-finite keep-outs, degenerate rectangles, a goal together with buttons): here, with the
    # class name, instead of a late "record exceeds capacities" - or silently different worlds after the first goal_met
    from .. import _native
    msg = _native.task_desc_check(d)
    if msg:
      raise ValueError(f'{cls}: {msg}')
    return d
