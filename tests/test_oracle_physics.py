"""Analytic known answers for the rigid-body specification (oracle/sag_oracle.c).
The reference delegates dynamics to MuJoCo (absent here), so these pin the
integrator to the MJCF's physics, not to MuJoCo's numbers."""
import numpy as np
import pytest

import golden_util as gu
from oracle_lib import (F_GOAL, F_HAZARDS, F_PILLARS, F_ROBOT, F_VASES, I_NH, I_NP, I_NV, Oracle)


@pytest.fixture(scope='module')
def oracle():
  return Oracle()


def empty_world(task='go_to_goal', **kw):
  names = ['robot', 'goal']
  rf, ri = gu.base_record(task, names, {'robot': 0.4})
  rf[F_GOAL:F_GOAL + 2] = [50., 50.]  # far away: never met
  ri[I_NH] = ri[I_NV] = ri[I_NP] = 0
  return rf, ri


def run(oracle, rf, ri, action, steps, noise=(0., 0.)):
  e = oracle.env(rf, ri)
  outs = []
  for _ in range(steps):
    outs.append(oracle.step(e, 0, action, noise=noise, tape=np.zeros(8, np.uint32)))
  return e, outs


def test_terminal_forward_speed(oracle):
  # motor force saturates at gear*0.05 = 0.015 N; slide damping 0.01 -> 1.5 m/s
  rf, ri = empty_world()
  rf[F_ROBOT + 2] = 0.7
  e, _ = run(oracle, rf, ri, [1.0, 0.0], 400)
  f, _ = oracle.record(e)
  v = np.hypot(f[F_ROBOT + 3], f[F_ROBOT + 4])
  assert abs(v - 1.5) < 1e-3
  # heading unchanged, motion along it
  assert abs(f[F_ROBOT + 2] - 0.7) < 1e-9
  assert abs(np.arctan2(f[F_ROBOT + 4], f[F_ROBOT + 3]) - 0.7) < 1e-6


def test_saturation_threshold(oracle):
  # |ctrl| >= 0.05 saturates the motor: 0.05 and 1.0 give the same motion
  rf, ri = empty_world()
  a, _ = run(oracle, rf, ri, [0.05, 0.0], 50)
  b, _ = run(oracle, rf, ri, [1.0, 0.0], 50)
  np.testing.assert_allclose(oracle.record(a)[0], oracle.record(b)[0], atol=1e-12)
  c, _ = run(oracle, rf, ri, [0.025, 0.0], 400)
  f, _ = oracle.record(c)
  assert abs(np.hypot(f[F_ROBOT + 3], f[F_ROBOT + 4]) - 0.75) < 1e-3


def test_terminal_yaw_rate(oracle):
  # velocity servo saturated: torque 0.3*0.05 = 0.015, hinge damping 0.005 -> 3 rad/s
  rf, ri = empty_world()
  e, _ = run(oracle, rf, ri, [0.0, 1.0], 400)
  f, _ = oracle.record(e)
  # (the COM offset makes the origin orbit, whose slide damping takes a little: 2.998)
  assert abs(f[F_ROBOT + 5] - 3.0) < 5e-3


@pytest.mark.parametrize('task,expect', [('go_to_goal_motor', 15.0), ('go_to_goal_damping', 15.0)])
def test_variant_dynamics(oracle, task, expect):
  rf, ri = empty_world(task)
  e, _ = run(oracle, rf, ri, [1.0, 0.0], 4000)
  f, _ = oracle.record(e)
  assert abs(np.hypot(f[F_ROBOT + 3], f[F_ROBOT + 4]) - expect) < 0.05


def test_free_decay_rate(oracle):
  # zero control: m dv/dt = -d v  (implicit Euler: v <- v / (1 + h d/m) per substep)
  rf, ri = empty_world()
  rf[F_ROBOT + 3] = 1.0
  e, _ = run(oracle, rf, ri, [0.0, 0.0], 10)
  f, _ = oracle.record(e)
  m = 4 / 3 * np.pi * 1e-3 + 1e-3
  want = (1 + 0.004 * 0.01 / m) ** (-50)
  assert abs(f[F_ROBOT + 3] - want) < 2e-3  # COM offset couples x/yaw slightly


def test_pillar_stops_robot_and_costs(oracle):
  rf, ri = empty_world()
  ri[I_NP] = 1
  rf[F_PILLARS:F_PILLARS + 2] = [1.0, 0.0]
  e, outs = run(oracle, rf, ri, [1.0, 0.0], 200)
  f, _ = oracle.record(e)
  # arrow tip (0.15 ahead) rests on the pillar surface (x = 0.8), penetration < 1 mm
  assert 0.65 - 1e-3 < f[F_ROBOT] < 0.65 + 1e-3
  assert abs(f[F_ROBOT + 3]) < 1e-3
  assert outs[-1].cost == 1 and outs[0].cost == 0
  # accelerometer ~ 0 at rest against the pillar (drive force balanced by contact)
  assert abs(outs[-1].obs[48]) < 0.3


def test_vase_is_pushed_then_stops_by_floor_friction(oracle):
  rf, ri = empty_world()
  ri[I_NV] = 1
  rf[F_VASES:F_VASES + 2] = [0.5, 0.0]
  e, outs = run(oracle, rf, ri, [1.0, 0.0], 60)
  f, _ = oracle.record(e)
  assert f[F_VASES] > 0.6, 'vase should have been pushed along +x'
  assert any(o.cost for o in outs), 'touching a vase is a cost event'
  assert f[F_ROBOT] > 0.3, 'a vase (8 mg) barely slows the robot'
  # stop pushing, back off: the vase comes to rest
  e2, _ = run(oracle, f, _ri(oracle, e), [-1.0, 0.0], 100)
  f2, _ = oracle.record(e2)
  assert np.hypot(f2[F_VASES + 3], f2[F_VASES + 4]) < 1e-3
  assert abs(f2[F_VASES + 5]) < 1e-2


def _ri(oracle, e):
  return oracle.record(e)[1]


def test_hazard_is_not_a_collider(oracle):
  rf, ri = empty_world()
  ri[I_NH] = 1
  rf[F_HAZARDS:F_HAZARDS + 2] = [0.5, 0.0]
  a, outs = run(oracle, rf, ri, [1.0, 0.0], 40)
  ri[I_NH] = 0
  b, _ = run(oracle, rf, ri, [1.0, 0.0], 40)
  np.testing.assert_allclose(oracle.record(a)[0][:6], oracle.record(b)[0][:6], atol=0)
  assert any(o.cost for o in outs) and not outs[0].cost


def test_long_random_rollout_stays_finite_and_deterministic(oracle):
  ep = gu.load_json_gz('episodes.json.gz')[3]
  assert ep['task'] == 'go_to_goal'
  rf, ri = gu.episode_init_record(ep)
  rs = np.random.RandomState(0)
  acts = rs.uniform(-1, 1, (1500, 2))
  finals = []
  for _ in range(2):
    e = oracle.env(rf, ri)
    for a in acts:
      o = oracle.step(e, 0, a, key=(5, 6))
      assert not o.done
    finals.append(oracle.record(e)[0])
  assert np.isfinite(finals[0]).all()
  np.testing.assert_array_equal(finals[0], finals[1])
  assert np.abs(finals[0][:2]).max() < 50


def test_f32_oracle_tracks_f64(oracle):
  """Precision, not logic: the float build of the same source stays within 1e-4 over one step."""
  o32 = Oracle(f32=True)
  ep = gu.load_json_gz('episodes.json.gz')[3]
  rf, ri = gu.episode_init_record(ep)
  rf = rf.astype(np.float32)
  e64, e32 = oracle.env(rf, ri), o32.env(rf, ri)
  a = oracle.step(e64, 0, [1.0, 0.5], noise=[0, 0], tape=np.zeros(8, np.uint32))
  b = o32.step(e32, 0, [1.0, 0.5], noise=[0, 0], tape=np.zeros(8, np.uint32))
  np.testing.assert_allclose(oracle.record(e64)[0], o32.record(e32)[0], rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(np.array(a.obs[:60]), np.array(b.obs[:60]), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------
# PushBox family: the task object as a planar free body (specification)
# ---------------------------------------------------------------------------
from oracle_lib import F_BOX, F_KEEPOUT, I_BOX_KIND, I_TASK  # noqa: E402


def box_world(task, kind, box_xy, box_yaw=0.0):
  rf, ri = empty_world(task)
  ri[I_BOX_KIND] = kind
  rf[F_BOX:F_BOX + 3] = [box_xy[0], box_xy[1], box_yaw]
  rf[F_KEEPOUT + 4] = 0.5
  return rf, ri


def test_push_box_is_pushed_not_a_cost(oracle):
  rf, ri = box_world('push_box', 1, (0.6, 0.0))
  e, outs = run(oracle, rf, ri, [1.0, 0.0], 80)
  f, _ = oracle.record(e)
  assert f[F_BOX] > 0.7, 'box pushed along +x'
  assert abs(f[F_BOX + 1]) < 0.05
  assert not any(o.cost for o in outs), 'the box is not an obstacle (consts.OBSTACLES)'
  # robot stays behind the box face (arrow tip .15 ahead of the origin, box half .2)
  assert f[F_BOX] - f[F_ROBOT] > 0.34
  # reward = -d(robot-box) - d(box-goal): pushing towards a far goal at (50,50) is progress
  assert sum(o.reward[0] for o in outs[40:]) > 0


def test_rod_rolls_easier_than_it_slides(oracle):
  # rod axis along local y: a push along x rolls it, a push along y must slide it (mu 1.2)
  rf, ri = box_world('roll_rod', 2, (0.45, 0.0))
  e, _ = run(oracle, rf, ri, [1.0, 0.0], 60)
  rolled = oracle.record(e)[0][F_BOX] - 0.45
  rf, ri = box_world('roll_rod', 2, (0.0, 0.65))
  rf[F_ROBOT + 2] = np.pi / 2
  e, _ = run(oracle, rf, ri, [1.0, 0.0], 60)
  slid = oracle.record(e)[0][F_BOX + 1] - 0.65
  assert rolled > 0.2 and slid > 0.05


def test_ball_keeps_rolling_after_the_kick(oracle):
  rf, ri = box_world('dribble_ball', 3, (0.5, 0.0))
  e, _ = run(oracle, rf, ri, [1.0, 0.0], 40)
  f, ri2 = oracle.record(e)
  v_kick = f[F_BOX + 3]
  assert v_kick > 0.5
  e, _ = run(oracle, f, ri2, [-1.0, 0.0], 10)   # robot backs off
  f2, _ = oracle.record(e)
  # rolling resistance .05 N / r: deceleration .357 g / 1.4 = 2.5 m/s^2 -> .5 m/s lost in .2 s
  assert 0 < f2[F_BOX + 3] < v_kick
  assert abs((v_kick - f2[F_BOX + 3]) - 2.5 * 0.2) < 0.25


def test_haul_tendon_limits_the_distance(oracle):
  rf, ri = box_world('haul_box', 1, (0.6, 0.0))
  rf[F_ROBOT + 2] = np.pi  # drive away from the box
  e, _ = run(oracle, rf, ri, [1.0, 0.0], 150)
  f, _ = oracle.record(e)
  d = np.hypot(f[F_BOX] - f[F_ROBOT], f[F_BOX + 1] - f[F_ROBOT + 1])
  L = np.hypot(d, 0.1)
  assert f[F_ROBOT] < -0.5, 'robot keeps moving'
  assert L < 0.76, f'tendon stretched to {L}'
  assert f[F_BOX] < 0.55, 'box is hauled along'


def test_box_spawned_over_a_pillar_is_pushed_clear_only_when_flagged_awake(oracle):
  """HaulBox places its box at robot + .6 with no keep-out check (haul_box.py:17-18): it may sit inside a pillar.
  With the install-time flag SAG_I_AWAKE (what sag_set_layout's overlap test sets: ADVICE r3) the box takes part in the
  first forward evaluation and is pushed clear of the pillar, as MuJoCo separates such a pair; a box that came to rest
  there WITHOUT the flag stays asleep until something active touches it (the sleeping-body rule)."""
  from oracle_lib import I_AWAKE
  rf, ri = box_world('haul_box', 1, (0.6, 0.0))
  ri[I_NP] = 1
  rf[F_PILLARS:F_PILLARS + 2] = [0.85, 0.0]        # pillar r .2: 5 cm inside the box's main geom (half .2)
  ri[I_AWAKE] = 1 << 10
  e, _ = run(oracle, rf, ri, [0.0, 0.0], 40)
  f, i = oracle.record(e)
  assert f[F_BOX] < 0.6 - 0.04, 'the box is pushed off the pillar'
  assert np.hypot(f[F_BOX] - 0.85, f[F_BOX + 1]) >= 0.2 + 0.2 - 2e-3
  assert i[I_AWAKE] == 0, 'the flag lasts for the first substep'
  ri[I_AWAKE] = 0
  e, _ = run(oracle, rf, ri, [0.0, 0.0], 40)
  assert oracle.record(e)[0][F_BOX] == pytest.approx(0.6, abs=1e-7)


# ---------------------------------------------------------------------------
# Car robot: planar reduction of car.xml (specification)
# ---------------------------------------------------------------------------
F_EXT = 144


def car_run(oracle, rf, ri, action, steps):
  e = oracle.env(rf, ri)
  outs = []
  for _ in range(steps):
    outs.append(oracle.step(e, 1, action, noise=(0., 0.), tape=np.zeros(8, np.uint32)))
  return e, outs


def test_car_straight_line_force_balance(oracle):
  """Full throttle: the wheels do not saturate their friction, the rear ball does (its joint
  damping .001 N m s is large for a 2.6 g ball), so the terminal state is the balance
  2 (tau - d w_wheel) / r = mu N_caster, moving along the body's -y with no lateral slip."""
  rf, ri = empty_world()
  rf[F_ROBOT + 2] = 0.3
  e, outs = car_run(oracle, rf, ri, [1.0, 1.0], 250)
  f, _ = oracle.record(e)
  yaw = f[F_ROBOT + 2]
  v = f[F_ROBOT + 3:F_ROBOT + 5]
  fwd = np.array([np.sin(yaw), -np.cos(yaw)])          # -y_body in world axes
  speed = np.dot(v, fwd)
  assert 0.80 < speed < 0.84, 'converged floor friction: 0.82 m/s (one sweep of six elements gave 0.77)'
  assert abs(np.dot(v, [np.cos(yaw), np.sin(yaw)])) < 5e-3, 'no sideways slip'
  assert abs(yaw - 0.3) < 0.01, 'converged floor friction: no heading drift (VERDICT r2: <= 0.01 rad over 250 steps)'
  import ctypes as C
  k = (C.c_double * 9)(); oracle.lib.sago_car_constants(k)
  drive = sum((0.02 - 0.001 * f[F_EXT + i]) / 0.05 for i in range(2))
  assert abs(drive - 1.0 * k[8]) < 0.02 * k[8], 'wheel drive balances the sliding caster'
  assert 0.05 * f[F_EXT] > speed, 'driven wheels slip forward a little'
  assert 0 < 0.05 * f[F_EXT + 2] < speed, 'the ball under-rotates (it slides)'
  q = f[F_EXT + 5:F_EXT + 9]
  assert abs(np.linalg.norm(q) - 1) < 1e-9
  o = np.array(outs[-1].obs[:72])
  np.testing.assert_allclose(o[63:72].reshape(3, 3) @ o[63:72].reshape(3, 3).T, np.eye(3), atol=1e-9)
  assert abs(o[50] - 9.81) < 1e-12 and abs(o[52] + speed) < 1e-3   # accelerometer z, velocimeter y


def test_car_differential_drive_turns_in_place(oracle):
  rf, ri = empty_world()
  e, _ = car_run(oracle, rf, ri, [1.0, -1.0], 100)
  f, _ = oracle.record(e)
  assert abs(f[F_ROBOT + 5]) > 1.0, 'opposite wheel torques spin the car'
  assert np.hypot(f[F_ROBOT + 3], f[F_ROBOT + 4]) < 0.3
  assert np.hypot(f[F_ROBOT], f[F_ROBOT + 1]) < 0.2
  assert f[F_EXT] > 0 > f[F_EXT + 1]


def test_car_wheels_resist_sideways_motion(oracle):
  rf, ri = empty_world()
  rf[F_ROBOT + 3] = 0.5    # along body x (yaw 0): sideways for the wheels
  e, _ = car_run(oracle, rf, ri, [0.0, 0.0], 6)
  f, _ = oracle.record(e)
  assert abs(f[F_ROBOT + 3]) < 0.05
  # along the rolling direction, wheels and ball already rolling (v_y + r w = 0): it coasts,
  # slowed only by the joint damping (the wheels' armature is a 0.2 kg-equivalent flywheel)
  rf[F_ROBOT + 3], rf[F_ROBOT + 4] = 0.0, -0.5
  rf[F_EXT], rf[F_EXT + 1], rf[F_EXT + 2] = 10.0, 10.0, 10.0
  e, _ = car_run(oracle, rf, ri, [0.0, 0.0], 1)
  f, _ = oracle.record(e)
  # joint damping (2 x .001 x 10 / .05 = .4 N) + sliding caster (.14 N) on .03 + .2 kg: -2.3 m/s^2
  assert abs(f[F_ROBOT + 4] - (-0.5 + 2.3 * 0.08)) < 0.05
  rf[F_ROBOT + 3], rf[F_ROBOT + 4] = 0.5, 0.0
  e, _ = car_run(oracle, rf, ri, [0.0, 0.0], 1)
  assert abs(oracle.record(e)[0][F_ROBOT + 3]) < 0.05, 'sideways: stopped within one step'


def test_car_stops_at_pillar_with_cost(oracle):
  rf, ri = empty_world()
  ri[I_NP] = 1
  rf[F_ROBOT + 2] = np.pi / 2         # -y_body = +x world
  rf[F_PILLARS:F_PILLARS + 2] = [1.0, 0.0]
  e, outs = car_run(oracle, rf, ri, [1.0, 1.0], 150)
  f, _ = oracle.record(e)
  # front bumper face at .175 ahead of the origin, pillar surface at x = .8
  assert 0.8 - 0.175 - 2e-3 < f[F_ROBOT] < 0.8 - 0.175 + 2e-3
  assert outs[-1].cost == 1 and outs[0].cost == 0
  assert np.isfinite(oracle.record(e)[0]).all()


def test_car_constants(oracle):
  import ctypes as C
  out = (C.c_double * 9)()
  oracle.lib.sago_car_constants(out)
  m, Io, ox, oy, Iw, Ib, NL, NR, NC = list(out)
  assert abs(m - 0.030445) < 1e-5 and abs(ox) < 1e-15 and abs(oy - 0.0074) < 1e-4
  assert abs(Iw - 2.5245e-4) < 1e-7 and abs(Ib - 2.618e-6) < 1e-8
  assert abs(NL + NR + NC - m * 9.81) < 1e-12 and NL == NR and NC > NL
