"""Builds batches of layouts with the package's host sampler (reference semantics,
seed base 666) for the parity tests and the bench."""
import numpy as np

from safe_adaptation_gym_amd import _native as nat
from safe_adaptation_gym_amd import benchmark
from safe_adaptation_gym_amd.robot import Robot
from world import World


def sample_records_native(robot_name, task_names, n, seed=666, config=None):
  """Same through the native sampler; task_names may be one name or a per-env list."""
  if isinstance(task_names, str):
    tids = benchmark.TASKS[task_names].TASK_ID
  else:
    tids = np.array([benchmark.TASKS[t].TASK_ID for t in task_names], np.int32)
  rf, ri, st = nat.sample_layouts(robot_name, seed + np.arange(n), tids, config=config)
  assert not st.any()
  return rf, ri


def sample_records(robot_name, task_name, n, seed=666, config=None):
  """n records: env i sampled with RandomState(seed + i) exactly as make() would."""
  robot = Robot(f'xmls/{robot_name}.xml')
  rf = np.zeros((n, nat.REC_FLOATS), np.float32)
  ri = np.zeros((n, nat.REC_INTS), np.int32)
  for i in range(n):
    rs = np.random.RandomState(seed + i)
    task = benchmark.TASKS[task_name]() if isinstance(task_name, str) else benchmark.TASKS[task_name[i]]()
    w = World(rs, task, robot, config)
    w.sample_layout()
    w.reset()
    rf[i], ri[i] = w.record(env_id=i)
  return rf, ri


def pursuit_actions(rf, ri, rng, p_random=0.2, robot='point'):
  """A policy that drives Point robots towards their current target with noise, so
  goals are met and obstacles are hit: u0 = forward, u1 = turn rate command."""
  n = len(rf)
  x, y, yaw = rf[:, nat.F_ROBOT], rf[:, nat.F_ROBOT + 1], rf[:, nat.F_ROBOT + 2]
  tx, ty = rf[:, nat.F_GOAL].copy(), rf[:, nat.F_GOAL + 1].copy()
  nb = ri[:, nat.I_NB]
  for i in np.flatnonzero(nb > 0):
    task = ri[i, nat.I_TASK]
    if task == 1:  # collect: lowest active button
      m = int(ri[i, nat.I_ACTIVE_MASK])
      b = (m & -m).bit_length() - 1 if m else 0
    else:
      b = ri[i, nat.I_GOAL_BUTTON]
    tx[i], ty[i] = rf[i, nat.F_BUTTONS + 2 * b], rf[i, nat.F_BUTTONS + 2 * b + 1]
  # box tasks: drive at the box (tests place the goal just beyond it, see goal_beyond_box)
  for i in np.flatnonzero(ri[:, nat.I_BOX_KIND] > 0):
    tx[i], ty[i] = rf[i, nat.F_BOX], rf[i, nat.F_BOX + 1]
    if ri[i, nat.I_TASK] == 7:  # haul_box: the box follows on its tether; head for the goal
      tx[i], ty[i] = rf[i, nat.F_GOAL], rf[i, nat.F_GOAL + 1]
  heading = yaw if robot == 'point' else yaw - np.pi / 2   # the car drives along its -y axis
  ang = np.arctan2(ty - y, tx - x) - heading
  ang = (ang + np.pi) % (2 * np.pi) - np.pi
  if robot == 'point':
    a = np.stack([np.where(np.abs(ang) < 1.0, 1.0, 0.2), np.clip(2.0 * ang, -1, 1)], -1)
  else:
    # wheel torques saturate at |u| = .02 (car.xml:7): steer by easing one wheel; left faster = CCW
    turn = np.clip(1.5 * ang, -1, 1)
    a = 0.02 * np.stack([1 + 2 * np.minimum(turn, 0), 1 - 2 * np.maximum(turn, 0)], -1)
  rnd = rng.uniform(-1, 1, (n, 2))
  pick = rng.uniform(size=n) < p_random
  a[pick] = rnd[pick]
  return a.astype(np.float32)


def goal_beyond_box(rf, ri, dist=0.55):
  """Test set-up for the PushBox family: move every goal to `dist` beyond the box on the
  robot -> box line, so a straight push meets it (and the on-goal resample, RNG draws and
  PushBox.reset run many times in a short rollout)."""
  rf = rf.copy()
  for i in np.flatnonzero(ri[:, nat.I_BOX_KIND] > 0):
    r = rf[i, nat.F_ROBOT:nat.F_ROBOT + 2]
    b = rf[i, nat.F_BOX:nat.F_BOX + 2]
    d = (b - r) / (np.linalg.norm(b - r) + 1e-9)
    rf[i, nat.F_GOAL:nat.F_GOAL + 2] = b + dist * d
  return rf
