#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE - runs ONLY in the build container.

Imports the *unmodified* reference package from /root/reference under
import-name-only stubs for its missing third-party deps (dm_control, gym,
xmltodict - none of them installed here, MuJoCo itself is absent) and drives
the reference's pure-NumPy code with a pose-provider object standing in for
``MujocoBridge`` (poses / contacts / sensor values are *inputs*; the stubs
carry no arithmetic).  Outputs are written as small fixtures under
``tests/golden/``; nothing of the reference (source or bytecode) is copied.

What is pinned:
  lidar.npz      SafeAdaptationGym._lidar            safe_adaptation_gym.py:174-223
  resets.json    make/seed/set_task/_build_world     safe_adaptation_gym.py:85-118,165-172,
                 World.__init__/sample_layout        world.py:36-137,172-217
                 task.reset (goal resample, button)  tasks/go_to_goal.py:50-80, press_buttons.py:63-84
  episodes.json  SafeAdaptationGym.step with scripted poses: noise draw order, reward,
                 cost (hazard + contact count), lidar grouping, goal-met resample, button
                 state machine, CatchGoal mocap     safe_adaptation_gym.py:56-83, world.py:139-165,219-231,
                                                    tasks/*.py compute_reward/set_mocaps
  sampler.json   benchmark.TASKS order, TaskSampler  benchmark/__init__.py:14-20,61-84, task_sampler.py:10-19

Usage:  python oracle/gen_golden.py   (re-creates tests/golden/*)
"""
import json
import os
import sys
import types
from types import SimpleNamespace
from unittest import mock

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests',
                   'golden')


# ----------------------------------------------------------------------------
# import-name stubs (no arithmetic)
# ----------------------------------------------------------------------------
class PhysicsError(RuntimeError):
  pass


def install_stubs():
  def mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m

  class _Box:

    def __init__(self, low, high, shape=None, dtype=None):
      self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

  class _Env:
    pass

  gym = mod('gym', Env=_Env)
  gym.spaces = mod('gym.spaces', Box=_Box)
  gym.core = mod('gym.core', ActType=object, ObsType=object)
  mod('xmltodict')
  dmc = mod('dm_control')
  dmc.mujoco = mod('dm_control.mujoco')
  dmc.mjcf = mock.MagicMock()  # HaulBox/RollRod/DribbleBall build XML strings only
  sys.modules['dm_control.mjcf'] = dmc.mjcf
  dmc.rl = mod('dm_control.rl')
  dmc.rl.control = mod('dm_control.rl.control', PhysicsError=PhysicsError)
  dmc.utils = mod('dm_control.utils')

  def tolerance(x, bounds=(0., 0.), margin=0., sigmoid='gaussian',
                value_at_margin=0.1):
    # dm_control.utils.rewards.tolerance restated for margin == 0 only (the
    # only way the reference calls it: go_to_goal_scarce.py:26-32,
    # push_box_scarce.py:33-39): indicator(lower <= x <= upper).
    assert margin == 0.
    lower, upper = bounds
    return float(np.where(np.logical_and(lower <= x, x <= upper), 1.0, 0.0))

  dmc.utils.rewards = mod('dm_control.utils.rewards', tolerance=tolerance)
  # render.py imports dm_control.mjcf only
  sys.path.insert(0, REF)


install_stubs()
import safe_adaptation_gym  # noqa: E402
from safe_adaptation_gym import benchmark, consts, tasks  # noqa: E402
from safe_adaptation_gym import utils as ref_utils  # noqa: E402
from safe_adaptation_gym.benchmark import task_sampler  # noqa: E402
from safe_adaptation_gym.safe_adaptation_gym import SafeAdaptationGym  # noqa: E402

ROBOTS = {
    'point': dict(nu=2, z_height=0.1),
    'car': dict(nu=2, z_height=0.1),
    'doggo': dict(nu=12, z_height=0.22),
}

# yaw draws are embedded in XML strings by the reference; record them instead
# of parsing (instrumentation of the reference, not a change of behaviour).
_ROT_LOG = []
_orig_random_rot = ref_utils.random_rot


def _logging_random_rot(rs):
  v = _orig_random_rot(rs)
  _ROT_LOG.append(float(v))
  return v


ref_utils.random_rot = _logging_random_rot


# ----------------------------------------------------------------------------
# pose provider standing in for MujocoBridge (poses are inputs)
# ----------------------------------------------------------------------------
class _Physics:

  def __init__(self, bridge):
    self._b = bridge

  def step(self, nstep=1):
    self._b._advance(nstep)

  def forward(self):
    self._b.n_forward += 1


class _GeomUser(dict):
  """physics.named.model.geom_user with nuser_geom=1: every entry is a float array
  of shape (1,), and assignment writes into it (so `= [GROUP_GOAL]` works and
  `entry == GROUP_GOAL` is truthy), as in MuJoCo's named indexer."""

  def __setitem__(self, k, v):
    dict.__setitem__(self, k, np.asarray(v, float).reshape(1))


class PoseProvider:
  """Serves exactly the calls listed in SURVEY 8b from scripted data."""

  def __init__(self, robot):
    self.robot = robot
    self.physics = _Physics(self)
    self.nu = robot.nu
    self.actuator_ctrlrange = np.stack(
        [-np.ones(robot.nu), np.ones(robot.nu)], -1)
    self.user_groups = _GeomUser()
    self.geom_rgba = {}
    self.site_rgba = {}
    self.pos = {}
    self.mat = np.eye(3)
    self.contact_names = []  # list of (geom1, geom2)
    self.sensors = {}
    self.com = np.zeros(3)
    self.vel = np.zeros(3)
    self.time = 0.
    self.dt = 0.004
    self.n_forward = 0
    self.script = None  # callable(bridge, nstep) advancing poses
    self.last_ctrl = None
    self.world_config = None

  # -- reset-time ------------------------------------------------------------
  def rebuild(self, config):
    self.world_config = config
    self.time = 0.
    self.pos = {}
    # geom 'robot' carries the default user value 0 (= GROUP_INACTIVE)
    self.user_groups = _GeomUser()
    self.user_groups['robot'] = consts.GROUP_INACTIVE
    self.pos['robot'] = np.r_[np.asarray(config['robot_xy'], float),
                              config['robot_z_height']]
    th = config['robot_rot']
    self.mat = np.array([[np.cos(th), -np.sin(th), 0.],
                         [np.sin(th), np.cos(th), 0.], [0., 0., 1.]])
    self.rot0 = th

  def place(self, layout, groups, z):
    for name, xy in layout.items():
      if name == 'robot':
        continue
      self.pos[name] = np.r_[np.asarray(xy, float), z[name]]
      self.user_groups[name] = groups[name]

  # -- accessors -------------------------------------------------------------
  def _advance(self, nstep):
    self.time += nstep * self.dt
    if self.script is not None:
      self.script(self, nstep)

  def get_sensor(self, name):
    return self.sensors[name]

  def robot_pos(self):
    return self.pos['robot']

  def robot_mat(self):
    return self.mat

  def robot_vel(self):
    return self.vel

  def body_com(self, name):
    return self.com

  def body_pos(self, name):
    return self.pos[name]

  def set_body_pos(self, name, pos):
    pos = np.asarray(pos, float)
    self.pos[name][:pos.size] = pos

  def set_mocap_pos(self, name, pos):
    raise AssertionError('no task spawns gremlins')

  def set_control(self, action):
    self.last_ctrl = np.array(action, float)

  def robot_contacts(self, group_geom_names):
    # same counting rule as mujoco_bridge.py:177-191 over the scripted list
    robot_geoms = self.robot.geom_names
    count = 0
    for g1, g2 in self.contact_names:
      part = (g1 in robot_geoms) or (g2 in robot_geoms)
      grp = any(g1.startswith(p) or g2.startswith(p) for p in group_geom_names)
      count += int(part and grp)
    return count


def make_robot(name):
  r = ROBOTS[name]
  geoms = {
      'point': {'robot', 'pointarrow'},
      'car': {
          'robot', 'back_bumper', 'back_connector', 'front_bumper',
          'front_connector', 'left', 'right', 'rear'
      },
      'doggo': {'robot', 'robot2'} | {
          f'{p}_{i}' for p in ('aux', 'hip', 'ankle') for i in range(1, 5)
      },
  }[name]
  hinge_pos, hinge_vel, bq, bav = [], [], [], []
  if name == 'car':
    bq, bav = ['ballquat_rear'], ['ballangvel_rear']
  if name == 'doggo':
    js = [f'hip_{i}_z' for i in range(1, 5)] + [
        f'hip_{i}_y' for i in range(1, 5)
    ] + [f'ankle_{i}' for i in range(1, 5)]
    hinge_pos = ['jointpos_' + j for j in js]
    hinge_vel = ['jointvel_' + j for j in js]
  return SimpleNamespace(
      name=name,
      nu=r['nu'],
      z_height=r['z_height'],
      geom_names=geoms,
      hinge_pos_names=hinge_pos,
      hinge_vel_names=hinge_vel,
      ballquat_names=bq,
      ballangvel_names=bav)


BODY_Z = {
    'hazards': 2e-2,
    'vases': 0.1 - 4e-5,
    'pillars': 0.5,
    'goal': 0.3 / 2. + 1e-2,
    'buttons': 0.1,
    'box': 0.2,
}
GROUPS = {
    'hazards': consts.GROUP_OBSTACLES,
    'vases': consts.GROUP_OBSTACLES,
    'pillars': consts.GROUP_OBSTACLES,
    'goal': consts.GROUP_GOAL,
    'buttons': consts.GROUP_OBJECTS,
    'box': consts.GROUP_OBJECTS,
}


def _kind(name):
  for k in BODY_Z:
    if name.startswith(k):
      return k
  raise KeyError(name)


def box_z(task):
  if isinstance(task, tasks.RollRod):
    return task.ROD_RADIUS
  if isinstance(task, tasks.DribbleBall):
    return task.SPHERE_RADIUS
  return task.BOX_SIZE


def make_env(robot_name, task_name, seed, config=None):
  """safe_adaptation_gym.make() (__init__.py:6-24) with the pose provider in
  place of MujocoBridge/Robot (whose constructors need MuJoCo)."""
  env = object.__new__(SafeAdaptationGym)
  env._world = None
  env.base_config = config
  env._rgb_observation = False
  env.robot = make_robot(robot_name)
  env._render_lidars_and_collision = False
  env._render_options = {}
  bridge = PoseProvider(env.robot)
  bridge.dt = {'point': 0.004, 'car': 0.008, 'doggo': 0.012}[robot_name]
  env.mujoco_bridge = bridge
  env._observation_space = None
  env._sensors_names = list(SafeAdaptationGym.BASE_SENSORS)
  if robot_name == 'doggo':
    env._sensors_names += SafeAdaptationGym.DOGGO_EXTRA_SENSORS
  _set_default_sensors(env)

  # reference's rebuild() places bodies from the XML; our stand-in needs the
  # layout, which lives on the World.  Patch _build_world to place them
  # between rebuild() and World.reset(), keeping the reference's call order.
  def _build_world():
    cfg = env._world.sample_layout()
    bridge.rebuild(cfg)
    layout = env._world._layout
    task = env._world.task
    z = {}
    g = {}
    for n in layout:
      if n == 'robot':
        continue
      k = _kind(n)
      z[n] = box_z(task) if k == 'box' else BODY_Z[k]
      g[n] = GROUPS[k]
    bridge.place(layout, g, z)
    env._world.reset(bridge)

  env._build_world = _build_world
  env.seed(seed)
  if task_name is not None:
    env.set_task(benchmark.TASKS[task_name]())
  return env


def _set_default_sensors(env):
  b = env.mujoco_bridge
  for s in env._sensors_names:
    b.sensors[s] = np.zeros(1 if s.startswith('touch') else 3)
  for s in env.robot.hinge_vel_names:
    b.sensors[s] = np.zeros(1)
  for s in env.robot.hinge_pos_names:
    b.sensors[s] = np.zeros(1)
  for s in env.robot.ballangvel_names:
    b.sensors[s] = np.zeros(3)
  for s in env.robot.ballquat_names:
    b.sensors[s] = np.array([1., 0., 0., 0.])


def _quat2mat(q):
  # standard unit-quaternion -> rotation matrix (what mju_quat2Mat computes;
  # utils.py:110-116 delegates to MuJoCo, absent here).
  w, x, y, z = q
  return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z),
                    2 * (x * z + w * y)],
                   [2 * (x * y + w * z), w * w - x * x + y * y - z * z,
                    2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x),
                    w * w - x * x - y * y + z * z]])


ref_utils.quat2mat = _quat2mat


def rs_dump(rs):
  st = rs.get_state()
  return {
      'key': [int(x) for x in st[1]],
      'pos': int(st[2]),
      'has_gauss': int(st[3]),
      'cached_gaussian': float(st[4])
  }


def rs_probe(rs):
  """Fingerprint of the RandomState position that does not disturb it."""
  st = rs.get_state()
  c = np.random.RandomState()
  c.set_state(st)
  return float(c.random_sample())


# ----------------------------------------------------------------------------
# 1. lidar
# ----------------------------------------------------------------------------
def single_bin(obs):
  """Recover the bin index of a single-object lidar reading."""
  m = obs.max()
  idx = np.flatnonzero(obs == m)
  if len(idx) == 1:
    return int(idx[0])
  # alias == 0: bins (b-1, b) carry the same value; b is the one whose cyclic
  # predecessor also carries it.
  for i in idx:
    if (i - 1) % 16 in idx:
      return int(i)
  raise AssertionError


def gen_lidar():
  rs = np.random.RandomState(20241008)
  env = make_env('point', None, 0)
  b = env.mujoco_bridge
  cases = []

  def rotz(t):
    return np.array([[np.cos(t), -np.sin(t), 0.], [np.sin(t), np.cos(t), 0.],
                     [0., 0., 1.]])

  def rand_rot(rs, tilt):
    # yaw * small tilt about a random horizontal axis (car / doggo base)
    t = rs.uniform(0, 2 * np.pi)
    ax = rs.uniform(0, 2 * np.pi)
    a = np.array([np.cos(ax), np.sin(ax), 0.])
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(tilt) * K + (1 - np.cos(tilt)) * K @ K
    return rotz(t) @ R

  def run(robot_pos, mat, positions):
    b.pos['robot'] = np.asarray(robot_pos, float)
    b.mat = np.asarray(mat, float)
    obs = env._lidar([np.asarray(p, float) for p in positions])
    bins = []
    for p in positions:
      o1 = env._lidar([np.asarray(p, float)])
      bins.append(single_bin(o1) if o1.max() > 0 else -1)
    return obs, bins

  # (a) exact-axis / diagonal cases, identity rotation, robot at origin z=0.1
  axis_pts = []
  for k in range(8):
    ang = k * np.pi / 4
    axis_pts.append([round(np.cos(ang)) * 1.0, round(np.sin(ang)) * 1.0])
  for p in axis_pts:
    cases.append(([0., 0., 0.1], np.eye(3), [p]))
    cases.append(([0.5, -0.25, 0.1], np.eye(3),
                  [[p[0] * 2 + 0.5, p[1] * 2 - 0.25]]))
  # object exactly on the robot (dist 0) and beyond max range
  cases.append(([0., 0., 0.1], np.eye(3), [[0., 0.]]))
  cases.append(([0., 0., 0.1], np.eye(3), [[5., 0.], [0., 5.0001], [3., 4.]]))
  # 3-vector positions (z truncated, :206-207)
  cases.append(([0.3, 0.2, 0.1], rotz(0.7), [[1., 1., 0.16], [-1., 0.5, 0.5]]))
  # (b) random planar cases (Point robot: R = Rz)
  for _ in range(120):
    k = rs.randint(1, 22)
    rp = np.r_[rs.uniform(-2, 2, 2), 0.1]
    pts = rs.uniform(-2.5, 2.5, (k, 2))
    cases.append((rp, rotz(rs.uniform(0, 2 * np.pi)), pts))
  # float32-representable inputs (what the device state holds)
  for _ in range(60):
    k = rs.randint(1, 22)
    rp = np.r_[rs.uniform(-2, 2, 2), 0.1].astype(np.float32).astype(float)
    pts = rs.uniform(-2.5, 2.5, (k, 2)).astype(np.float32).astype(float)
    th = float(np.float32(rs.uniform(0, 2 * np.pi)))
    cases.append((rp, rotz(th), pts))
  # (c) tilted base (car/doggo): -robot_z * R[2,:2] enters ego_xy
  for _ in range(60):
    k = rs.randint(1, 12)
    rp = np.r_[rs.uniform(-2, 2, 2), rs.uniform(0.05, 0.3)]
    pts = rs.uniform(-2.5, 2.5, (k, 2))
    cases.append((rp, rand_rot(rs, rs.uniform(0, 0.5)), pts))
  # (d) empty list
  cases.append(([0., 0., 0.1], np.eye(3), []))

  K = 22
  n = len(cases)
  robot_pos = np.zeros((n, 3))
  robot_mat = np.zeros((n, 3, 3))
  pts = np.full((n, K, 2), np.nan)
  cnt = np.zeros(n, np.int32)
  obs = np.zeros((n, 16))
  bins = np.full((n, K), -2, np.int32)
  for i, (rp, m, ps) in enumerate(cases):
    o, bb = run(rp, m, ps)
    robot_pos[i] = rp
    robot_mat[i] = m
    cnt[i] = len(ps)
    for j, p in enumerate(ps):
      pts[i, j] = np.asarray(p, float)[:2]
      bins[i, j] = bb[j]
    obs[i] = o
  np.savez_compressed(
      os.path.join(OUT, 'lidar.npz'),
      robot_pos=robot_pos,
      robot_mat=robot_mat,
      points=pts,
      count=cnt,
      obs=obs,
      bins=bins)
  print('lidar cases', n)


# ----------------------------------------------------------------------------
# 2. resets: make -> set_task -> layout -> task.reset, all robots x tasks
# ----------------------------------------------------------------------------
def task_state(task):
  d = {}
  for k in ('_last_goal_distance', '_last_box_distance',
            '_last_box_goal_distance', '_goal_button', '_current_radius',
            '_next_radius'):
    if hasattr(task, k):
      v = getattr(task, k)
      d[k] = None if v is None else (v if isinstance(v, str) else float(v))
  if hasattr(task, '_state'):
    d['_state'] = task._state.name
  if hasattr(task, '_goal_button_timer'):
    d['_button_timer'] = task._goal_button_timer.time
  if hasattr(task, '_timer'):
    d['_catch_timer'] = task._timer.time
  if hasattr(task, '_origin') and task._origin is not None:
    d['_origin'] = [float(x) for x in task._origin]
  if hasattr(task, '_active_buttons'):
    d['_active_buttons'] = sorted(task._active_buttons)
  if hasattr(task, 'goal'):  # Unsupervised wraps a GoToGoal
    d['_last_goal_distance'] = (None if task.goal._last_goal_distance is None
                                else float(task.goal._last_goal_distance))
  return d


def snapshot(env):
  w = env._world
  b = env.mujoco_bridge
  return {
      'layout': {k: [float(x) for x in v] for k, v in w._layout.items()},
      'layout_order': list(w._layout.keys()),
      'body_pos': {k: [float(x) for x in v] for k, v in b.pos.items()},
      'groups': {k: int(np.ravel(v)[0]) for k, v in b.user_groups.items()},
      'task_state': task_state(w.task),
      'rs_probe': rs_probe(env.rs),
  }


def gen_resets():
  out = []
  for robot in ('point', 'car', 'doggo'):
    for task_name in benchmark.TASKS:
      for seed in (666, 667, 12345):
        if robot != 'point' and seed == 12345:
          continue
        _ROT_LOG.clear()
        env = make_env(robot, task_name, seed)
        w = env._world
        rec = {
            'robot': robot,
            'task': task_name,
            'seed': seed,
            'ctrl_range_scale':
                [float(x) for x in np.ravel(w._robot_ctrl_range_scale)],
            'bound': float(w.bound),
            'placements_margin': float(w.config.placements_margin),
            'keepouts': {k: float(v[1]) for k, v in w._placements.items()},
            'robot_rot': float(env.mujoco_bridge.rot0),
            'rots': list(_ROT_LOG),
            'modify_tree': repr(w.task.modify_tree(env.rs)),
            'obstacles': [int(x) for x in w.task.obstacles],
            'extents': [float(x) for x in w.task.placement_extents],
            'first': snapshot(env),
        }
        # a second episode: reset() without seed -> seed+1, fresh RandomState
        _ROT_LOG.clear()
        env.reset()
        rec['second'] = snapshot(env)
        rec['second']['robot_rot'] = float(env.mujoco_bridge.rot0)
        rec['second']['rots'] = list(_ROT_LOG)
        out.append(rec)
  import gzip
  with gzip.open(os.path.join(OUT, 'resets.json.gz'), 'wt') as f:
    json.dump(out, f)
  print('reset records', len(out))


# ----------------------------------------------------------------------------
# 3. scripted episodes through the reference's step()
# ----------------------------------------------------------------------------
def rotz(t):
  return np.array([[np.cos(t), -np.sin(t), 0.], [np.sin(t), np.cos(t), 0.],
                   [0., 0., 1.]])


def run_episode(robot, task_name, seed, n_steps, script_seed):
  """Drive the reference step() with a scripted pose trajectory.

  Script: the robot (and box, if any) chases the current target with a noisy
  pursuit so goals are met several times; vases get nudged; contacts are
  scripted from geometric proximity so cost / button logic is exercised."""
  env = make_env(robot, task_name, seed)
  b = env.mujoco_bridge
  w = env._world
  task = w.task
  srs = np.random.RandomState(script_seed)
  nstep = {'point': 5, 'car': 10, 'doggo': 12}[robot]
  yaw = [b.rot0]
  robot_geom = 'robot'

  def target_xy():
    if isinstance(task, tasks.PushBox):
      return b.pos['box'][:2].copy()
    if isinstance(task, tasks.Collect):
      act = sorted(task._active_buttons)
      if not act:
        return np.zeros(2)
      return b.pos[act[0]][:2].copy()
    if isinstance(task, tasks.PressButtons):
      return b.pos[task._goal_button][:2].copy()
    return b.pos['goal'][:2].copy()

  def script(bridge, n):
    rp = bridge.pos['robot']
    tgt = target_xy()
    d = tgt - rp[:2]
    dist = np.linalg.norm(d) + 1e-9
    speed = 0.12 if isinstance(task, tasks.PushBox) else 0.25
    stepv = d / dist * min(speed, dist * 0.9) + srs.normal(0, 0.01, 2)
    rp[:2] += stepv
    yaw[0] = np.arctan2(stepv[1], stepv[0]) + srs.normal(0, 0.05)
    bridge.mat = rotz(yaw[0])
    # kinematically consistent rigid-body data for the Point robot: origin
    # velocity v0, yaw rate wz, subtree COM at R (off,0,0), COM velocity
    # v0 + wz x (R off)  (point.xml:18-19 masses: sphere 4/3 pi r^3, box 1e-3)
    wz = srs.normal(0, 1)
    off = 1e-4 / (4. / 3. * np.pi * 1e-3 + 1e-3)
    v0 = np.r_[stepv / (n * bridge.dt), 0.]
    lever = bridge.mat @ np.array([off, 0., 0.])
    bridge.com = rp + lever
    bridge.vel = v0 + np.cross([0., 0., wz], lever)
    bridge.v0 = v0
    bridge.wz = wz
    if 'box' in bridge.pos:
      bp = bridge.pos['box']
      g = bridge.pos['goal'][:2] - bp[:2]
      gd = np.linalg.norm(g) + 1e-9
      if np.linalg.norm(bp[:2] - rp[:2]) < 0.45:
        bp[:2] += g / gd * min(0.15, gd * 0.9) + srs.normal(0, 0.005, 2)
    # nudge one vase now and then (free bodies move when touched)
    vs = [k for k in bridge.pos if k.startswith('vases')]
    if vs and srs.uniform() < 0.2:
      k = vs[srs.randint(len(vs))]
      bridge.pos[k][:2] += srs.normal(0, 0.02, 2)
    # scripted contacts from proximity
    contacts = []
    for name, p in bridge.pos.items():
      if name == 'robot':
        continue
      dd = np.linalg.norm(p[:2] - rp[:2])
      if name.startswith('vases') and dd < 0.2:
        contacts.append((robot_geom, name))
      if name.startswith('pillars') and dd < 0.3:
        contacts.append((name, robot_geom))
      if name.startswith('buttons') and dd < 0.2:
        contacts.append(('pointarrow' if robot == 'point' else robot_geom,
                         name))
      if name == 'box' and dd < 0.3:
        contacts.append((robot_geom, 'box'))
    contacts.append(('floor', 'vases0'))  # resting contact, not the robot
    bridge.contact_names = contacts
    s = bridge.sensors
    s['accelerometer'] = np.r_[srs.normal(0, 1, 2), 9.81]
    s['velocimeter'] = bridge.mat.T @ bridge.v0
    s['gyro'] = np.r_[0., 0., bridge.wz]
    s['magnetometer'] = bridge.mat.T @ np.array([0., -0.5, 0.])

  b.script = script
  steps = []
  rec = {
      'robot': robot,
      'task': task_name,
      'seed': seed,
      'nstep': nstep,
      'dt': b.dt,
      'init': snapshot(env),
      'init_obs': [float(x) for x in env.observation],
      'init_robot_rot': float(b.rot0),
      'rs_state': rs_dump(env.rs),
      'keepouts': {k: float(v[1]) for k, v in w._placements.items()},
      'obstacles': [int(x) for x in task.obstacles],
  }
  names = [k for k in b.pos.keys()]
  rec['names'] = names
  prev_layout = rec['init']['layout']
  for t in range(n_steps):
    action = srs.uniform(-1, 1, env.robot.nu)
    obs, reward, done, info = env.step(action)
    snap = snapshot(env)
    assert list(b.pos.keys()) == names
    st = {
        'action': [float(x) for x in action],
        'ctrl': [float(x) for x in b.last_ctrl],
        'time': float(b.time),
        # body positions AFTER the step (goal already moved if resampled)
        'pos': [[float(x) for x in b.pos[k]] for k in names],
        'groups': [snap['groups'][k] for k in names],
        'robot_mat': [float(x) for x in b.mat.ravel()],
        'robot_vel': [float(x) for x in b.vel],
        'robot_v0': [float(x) for x in b.v0],
        'robot_yaw': float(yaw[0]),
        'robot_com': [float(x) for x in b.com],
        'contacts': [list(c) for c in b.contact_names],
        'sensors': {
            k: [float(x) for x in np.ravel(v)]
            for k, v in b.sensors.items()
            if k in ('accelerometer', 'velocimeter', 'gyro', 'magnetometer')
        },
        'obs': [float(x) for x in obs],
        'reward': [float(x) for x in np.ravel(reward)],
        'done': bool(done),
        'cost': float(info['cost']),
        'bound': float(info['bound']),
        'task_state': snap['task_state'],
        'rs_probe': snap['rs_probe'],
    }
    if snap['layout'] != prev_layout:
      st['layout'] = snap['layout']
      st['layout_order'] = snap['layout_order']
      prev_layout = snap['layout']
    steps.append(st)
  rec['steps'] = steps
  return rec


def gen_episodes():
  out = []
  plan = [('point', t, 666 + i, 100, 1000 + i)
          for i, t in enumerate(benchmark.TASKS)]
  plan += [('car', 'push_box', 700, 60, 2000), ('car', 'go_to_goal', 701, 60,
                                                 2001),
           ('doggo', 'press_buttons', 702, 60, 2002),
           ('doggo', 'catch_goal', 703, 40, 2003)]
  for robot, task_name, seed, n, ss in plan:
    rec = run_episode(robot, task_name, seed, n, ss)
    met = sum(1 for s in rec['steps'] if max(s['reward']) > 0.5)
    cost = sum(s['cost'] for s in rec['steps'])
    print(f'episode {robot}/{task_name}: goal-ish events {met}, cost steps {cost}')
    out.append(rec)
  import gzip
  with gzip.open(os.path.join(OUT, 'episodes.json.gz'), 'wt') as f:
    json.dump(out, f)


# ----------------------------------------------------------------------------
# 4. benchmark registry / TaskSampler
# ----------------------------------------------------------------------------
def gen_sampler():
  out = {'tasks_order': list(benchmark.TASKS.keys())}
  out['tasks_class'] = {k: v.__name__ for k, v in benchmark.TASKS.items()}
  for name in ('multitask', 'task_adaptation'):
    for seed in (666, 7):
      bm = benchmark.make(name, batch_size=12, seed=seed)
      tr = [n for n, _ in bm.train_tasks]
      te = [n for n, _ in bm.test_tasks]
      tr2 = [n for n, _ in bm.train_tasks]
      out[f'{name}_{seed}'] = {'train': tr, 'test': te, 'train2': tr2}
  with open(os.path.join(OUT, 'sampler.json'), 'w') as f:
    json.dump(out, f)
  print('sampler ok', out['tasks_order'])


# ----------------------------------------------------------------------------
# 5. RNG draw primitives (legacy MT19937 normal / uniform / cauchy / choice)
# ----------------------------------------------------------------------------
def gen_rng():
  out = {}
  for seed in (0, 666, 4242):
    rs = np.random.RandomState(seed)
    out[str(seed)] = {
        'normal2': [float(x) for x in rs.normal(size=2)],
        'uniform': float(rs.uniform(-1.1, 1.1)),
        'normal12': [float(x) for x in rs.normal(size=12)],
        'choice4': int(rs.choice(4)),
        'cauchy2': [float(x) for x in rs.standard_cauchy(2)],
    }
  with open(os.path.join(OUT, 'rng.json'), 'w') as f:
    json.dump(out, f)


if __name__ == '__main__':
  os.makedirs(OUT, exist_ok=True)
  gen_lidar()
  gen_resets()
  gen_episodes()
  gen_sampler()
  gen_rng()
  print('done ->', os.path.abspath(OUT))
