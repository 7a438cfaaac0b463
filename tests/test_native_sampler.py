"""The native reset path (csrc/sag_sampler.cpp, exact numpy-legacy MT19937) against the
reference-generated fixtures and against the package's Python World.  No GPU."""
import numpy as np
import pytest

import golden_util as gu
from safe_adaptation_gym_amd import _native as nat
from safe_adaptation_gym_amd import benchmark
from safe_adaptation_gym_amd.robot import Robot
from world import World

RESETS = gu.load_json_gz('resets.json.gz')


def _check(rec_f, rec_i, state, snap, rots, robot_rot, task):
  lay = snap['layout']
  np.testing.assert_array_equal(rec_f[nat.F_ROBOT:nat.F_ROBOT + 2], np.float32(lay['robot']))
  assert rec_f[nat.F_ROBOT + 2] == np.float32(robot_rot)
  for name, xy in lay.items():
    if name.startswith('hazards'):
      o = nat.F_HAZARDS + 2 * int(name[7:])
    elif name.startswith('vases'):
      o = nat.F_VASES + 6 * int(name[5:])
    elif name.startswith('pillars'):
      o = nat.F_PILLARS + 2 * int(name[7:])
    elif name.startswith('buttons'):
      o = nat.F_BUTTONS + 2 * int(name[7:])
    elif name == 'goal':
      o = nat.F_GOAL
    elif name == 'box':
      o = nat.F_BOX
    else:
      continue
    np.testing.assert_array_equal(rec_f[o:o + 2], np.float32(xy), err_msg=name)
  # yaw draws: robot, obstacles in layout order, then the task's; vases and the box keep theirs
  names = [n for n in snap['layout_order']]
  obst = [n for n in names if n[:4] in ('haza', 'vase', 'pill')]
  yaw = dict(zip(obst, rots[1:1 + len(obst)]))
  for n, v in yaw.items():
    if n.startswith('vases'):
      assert rec_f[nat.F_VASES + 6 * int(n[5:]) + 2] == np.float32(v)
  rs = np.random.RandomState()
  rs.set_state(state)
  assert gu.rs_probe(rs) == snap['rs_probe'], 'generator position after the reset draws'
  ts = snap['task_state']
  if ts.get('_goal_button'):
    assert rec_i[nat.I_GOAL_BUTTON] == int(ts['_goal_button'][7:])


@pytest.mark.parametrize('rec', RESETS, ids=lambda r: f"{r['robot']}-{r['task']}-{r['seed']}")
def test_native_sampler_matches_reference(rec):
  tid = gu.TASK_ID[rec['task']]
  rf, ri, st, states = nat.sample_layouts(rec['robot'], [rec['seed']], tid, want_rng=True)
  assert st[0] == 0
  # the product path: the descriptor comes from the Python Task object (envs.py), not from the library's table
  rf_d, ri_d, st_d = nat.sample_layouts(rec['robot'], [rec['seed']], 0, descs=[benchmark.TASKS[rec['task']]().descriptor()])
  np.testing.assert_array_equal(rf_d, rf); np.testing.assert_array_equal(ri_d, ri)
  _check(rf[0], ri[0], states[0], rec['first'], rec['rots'], rec['robot_rot'], rec['task'])
  assert list(ri[0, 1:6]) == [rec['obstacles'][0], rec['obstacles'][1], rec['obstacles'][3],
                              benchmark.TASKS[rec['task']].NUM_BUTTONS, benchmark.TASKS[rec['task']].BOX_KIND]
  rf2, ri2, st2, states2 = nat.sample_layouts(rec['robot'], [rec['seed'] + 1], tid, first_episode=False,
                                              want_rng=True)
  _check(rf2[0], ri2[0], states2[0], rec['second'], rec['second']['rots'], rec['second']['robot_rot'],
         rec['task'])


@pytest.mark.parametrize('robot,task', [('point', 'go_to_goal'), ('car', 'push_box'),
                                        ('doggo', 'press_buttons'), ('point', 'haul_box'),
                                        ('point', 'catch_goal'), ('point', 'collect')])
def test_native_sampler_equals_python_world(robot, task):
  """Whole records, 64 seeds, multi-threaded: identical to World.record()."""
  n = 64
  seeds = 1000 + np.arange(n)
  rf, ri, st = nat.sample_layouts(robot, seeds, gu.TASK_ID[task], env_id0=5, nthreads=4)
  assert not st.any()
  rb = Robot(f'xmls/{robot}.xml')
  for j in range(n):
    w = World(np.random.RandomState(int(seeds[j])), benchmark.TASKS[task](), rb)
    w.sample_layout()
    w.reset()
    pf, pi = w.record(env_id=5 + j)
    np.testing.assert_array_equal(ri[j], pi)
    np.testing.assert_array_equal(rf[j], pf)


def test_native_sampler_config_and_cauchy_scale():
  ref = gu.load_json('rng.json')['666']
  rf, ri, st = nat.sample_layouts('point', [666], 3, config={'robot_ctrl_range_scale': 0.5, 'hazards_size': 0.3})
  # standard_cauchy draws come first: scale = cauchy * 0.5 + 1 (world.py:72-73)
  rs = np.random.RandomState(666)
  want = rs.standard_cauchy(2) * 0.5 + 1.0
  np.testing.assert_allclose(rf[0, nat.F_CTRL_SCALE:nat.F_CTRL_SCALE + 2], want, rtol=1e-6)
  assert rf[0, nat.F_HAZARD_SIZE] == np.float32(0.3) and rf[0, nat.F_KEEPOUT + 1] == np.float32(0.3)
  with pytest.raises(KeyError):
    nat.sample_layouts('point', [1], 3, config={'hazard_size': 1})


def test_native_sampler_impossible_layout_reports_failure():
  rf, ri, st = nat.sample_layouts('doggo', [0], 3, config={'hazards_size': 2.0, 'vases_size': 2.0, 'pillars_size': 2.0})
  assert st[0] < 0


# ---- the tasks/* plugin surface (tasks/task.py:14-97) reaches the sampler ----------------------------------------
@pytest.mark.parametrize('name', sorted(benchmark.TASKS))
def test_task_class_descriptor_equals_library_table(name):
  """What each of the 14 Python Task classes says about itself (obstacles, placement_extents, setup_placements(),
  attributes) is exactly the library's own descriptor for that task id."""
  t = benchmark.TASKS[name]()
  assert t.descriptor() == nat.task_desc_default(t.TASK_ID)


def test_task_subclass_changes_the_layouts():
  """a subclass that overrides the reference's Task surface changes the worlds."""
  from safe_adaptation_gym_amd.tasks.go_to_goal import GoToGoal
  from safe_adaptation_gym_amd.tasks.push_box import PushBox

  class Sparse(GoToGoal):
    @property
    def obstacles(self):
      return [3, 3, 0, 0]

    @property
    def placement_extents(self):
      return [-1.0, -1.5, 2.5, 1.5]

  rf, ri, st = nat.sample_layouts('point', 100 + np.arange(32), 0, descs=[Sparse().descriptor()])
  assert not st.any()
  assert (ri[:, nat.I_NH] == 3).all() and (ri[:, nat.I_NV] == 3).all() and (ri[:, nat.I_NP] == 0).all()
  assert (ri[:, nat.I_TASK] == GoToGoal.TASK_ID).all()
  hz = rf[:, nat.F_HAZARDS:nat.F_HAZARDS + 6].reshape(-1, 3, 2)
  assert hz[..., 0].min() >= -1.0 + 0.2 - 1e-6 and hz[..., 0].max() <= 2.5 - 0.2 + 1e-6 and hz[..., 0].max() > 2.0
  assert np.abs(hz[..., 1]).max() <= 1.5 - 0.2 + 1e-6
  assert (rf[:, nat.F_HAZARDS + 6:nat.F_HAZARDS + 18] == 0).all(), 'unused hazard slots stay empty'

  class FarBox(PushBox):
    BOX_KEEPOUT = 0.3

    def setup_placements(self):
      p = super().setup_placements()
      p['box'] = ([(1.0, -0.5, 1.6, 0.5)], self.BOX_KEEPOUT)
      return p

  rf, ri, st = nat.sample_layouts('car', 7 + np.arange(16), 0, descs=[FarBox().descriptor()])
  assert not st.any()
  assert (rf[:, nat.F_BOX] >= 1.3 - 1e-6).all() and (np.abs(rf[:, nat.F_BOX + 1]) <= 0.2 + 1e-6).all()
  assert np.allclose(rf[:, nat.F_KEEPOUT + 4], 0.3)


def test_task_surface_limits_raise():
  from safe_adaptation_gym_amd.tasks.go_to_goal import GoToGoal
  from safe_adaptation_gym_amd.tasks.task import Task

  class NoId(Task):
    def setup_placements(self):
      return {}

  with pytest.raises(TypeError, match='TASK_ID'):
    NoId().descriptor()

  class OwnReward(GoToGoal):
    def compute_reward(self, *a):
      return 0.

  with pytest.raises(NotImplementedError, match='runs on the device'):
    OwnReward().descriptor()

  class TooMany(GoToGoal):
    @property
    def obstacles(self):
      return [12, 3, 0, 1]

  with pytest.raises(ValueError, match='hazards'):
    TooMany().descriptor()

  class MovedGoal(GoToGoal):
    def setup_placements(self):
      return {'goal': ([(-1, -1, 1, 1)], 0.4)}

  with pytest.raises(NotImplementedError, match='goal placement'):
    MovedGoal().descriptor()


def test_descriptor_validation_names_the_field():
  """what the device cannot serve is refused when the descriptor is BUILT, by name - button_timer outside its
  3-bit field (or different from the constant the device re-arms with), negative / non-finite keep-outs, degenerate
  rectangles, a goal together with buttons - at Task.descriptor() and again at the C ABI."""
  from safe_adaptation_gym_amd.tasks.press_buttons import PressButtons
  from safe_adaptation_gym_amd.tasks.push_box import PushBox

  class SlowButtons(PressButtons):
    BUTTON_TICKING_DELAY = 9

  with pytest.raises(ValueError, match='SlowButtons: button_timer'):
    SlowButtons().descriptor()

  class NegativeKeepout(PushBox):
    BOX_KEEPOUT = -0.1

  with pytest.raises(ValueError, match='NegativeKeepout: box_keepout'):
    NegativeKeepout().descriptor()

  class BackwardsRect(PushBox):
    def setup_placements(self):
      p = super().setup_placements()
      p['box'] = ([(1.0, -0.5, 0.5, 0.5)], self.BOX_KEEPOUT)
      return p

  with pytest.raises(ValueError, match='BackwardsRect: box_rect'):
    BackwardsRect().descriptor()

  good = PressButtons().descriptor()
  assert nat.task_desc_check(good) is None
  for field, value, word in [('button_timer', 7, 'button_timer'), ('button_keepout', float('nan'), 'button_keepout'),
                             ('has_goal', 1, 'has_goal'), ('button_rect', [0.5, 0.0, 0.5, 1.0], 'button_rect'),
                             ('extents', [0.0, 0.0, float('inf'), 1.0], 'extents'), ('n_buttons', 7, 'n_buttons')]:
    bad = dict(good, **{field: value})
    assert word in nat.task_desc_check(bad)
    with pytest.raises(nat.SagError, match=word):
      nat.sample_layouts('point', [1, 2], 0, descs=[bad])
