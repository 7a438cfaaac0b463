"""Helpers that turn the golden fixtures (tests/golden/, generated from the
reference by oracle/gen_golden.py) into sag.h records."""
import gzip
import json
import os

import numpy as np

from oracle_lib import (F_ACTION_NOISE, F_BOX, F_BUTTONS, F_CATCH, F_CTRL_SCALE, F_DAMP,
                        F_GEAR, F_GOAL, F_HAZARD_SIZE, F_HAZARDS, F_KEEPOUT, F_LAST,
                        F_PILLAR_SIZE, F_PILLARS, F_ROBOT, F_ROBOT0, F_VASE_SIZE, F_VASES,
                        I_ACTIVE_MASK, I_BOX_KIND, I_BTN_STATE, I_BTN_TIMER, I_CATCH_TIMER,
                        I_ENV_ID, I_GOAL_BUTTON, I_NB, I_NH, I_NP, I_NV, I_STEP, I_TASK,
                        REC_FLOATS, REC_INTS)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TASKS = [
    'catch_goal', 'collect', 'dribble_ball', 'go_to_goal', 'go_to_goal_damping',
    'go_to_goal_motor', 'go_to_goal_scarce', 'haul_box', 'press_buttons',
    'press_buttons_scarce', 'push_box', 'push_box_scarce', 'roll_rod', 'unsupervised'
]
TASK_ID = {n: i for i, n in enumerate(TASKS)}
BOX_KIND = {
    'push_box': 1, 'push_box_scarce': 1, 'haul_box': 1, 'roll_rod': 2, 'dribble_ball': 3
}
OBSTACLE_PREFIXES = ['hazards', 'vases', 'gremlins', 'pillars']
# geoms of the robot XMLs (robot.py:20-22: every geom except `floor`); a contact counts when one side is one of these
ROBOT_GEOMS = {
    'point': {'robot', 'pointarrow'},
    'car': {'robot', 'back_bumper', 'back_connector', 'front_bumper', 'front_connector', 'left', 'right', 'rear'},
    'doggo': {'robot', 'robot2'} | {f'{p}_{i}' for p in ('aux', 'hip', 'ankle') for i in range(1, 5)},
}
ROBOT_ID = {'point': 0, 'car': 1, 'doggo': 2}
ROBOT_NU = {'point': 2, 'car': 2, 'doggo': 12}
ROBOT_OBS = {'point': 60, 'car': 72, 'doggo': 104}
# observation columns that the scripted reference episodes pin: everything except what MuJoCo's forward pass
# produces (accelerometer; Doggo's 8 touch forces) - the scripts leave those sensors at arbitrary / zero values
PINNED_SENSOR_COLS = {
    'point': list(range(50, 60)),
    'car': list(range(50, 72)),       # + ballangvel_rear (3), ballquat_rear as 3x3 (9)
    'doggo': list(range(51, 60)) + list(range(68, 104)),   # + 12 joint rates, 12 x (sin, cos)
}
# at reset the scripted bridge holds zeros in the four base sensors (placeholders), so only the extra columns
# (identity ballquat; zero joint rates; sin 0 / cos 0 pairs) say something about the layout there
INIT_PINNED_COLS = {'point': [], 'car': list(range(60, 72)), 'doggo': list(range(68, 104))}
EPISODE_KEYS = None


def episode_keys():
  """(robot, task) of every reference episode in tests/golden/episodes.json.gz (14 Point, 2 Car, 2 Doggo)."""
  global EPISODE_KEYS
  if EPISODE_KEYS is None:
    EPISODE_KEYS = [(e['robot'], e['task']) for e in load_json_gz('episodes.json.gz')]
  return EPISODE_KEYS


def set_robot_planar(rf, robot, yaw, wz):
  """The scripted episodes move the robots as planar rigid bodies at their XML height: for Doggo that is the base
  pose z = .22, quaternion about z, joints at 0, angular velocity (0, 0, wz) in the base frame."""
  if robot == 'doggo':
    E = 144
    rf[E] = 0.22
    rf[E + 1:E + 5] = [np.cos(0.5 * yaw), 0.0, 0.0, np.sin(0.5 * yaw)]
    rf[E + 5] = 0.0
    rf[E + 6:E + 9] = [0.0, 0.0, wz]


def load_json_gz(name):
  with gzip.open(os.path.join(GOLDEN, name), 'rt') as f:
    return json.load(f)


def load_json(name):
  with open(os.path.join(GOLDEN, name)) as f:
    return json.load(f)


def _names(names, prefix):
  out = [n for n in names if n.startswith(prefix)]
  return sorted(out, key=lambda n: int(n[len(prefix):]))


def set_task_state(rf, ri, ts):
  if ts.get('_last_goal_distance') is not None:
    rf[F_LAST] = ts['_last_goal_distance']
  if ts.get('_last_box_distance') is not None:
    rf[F_LAST + 1] = ts['_last_box_distance']
  if ts.get('_last_box_goal_distance') is not None:
    rf[F_LAST + 2] = ts['_last_box_goal_distance']
  if ts.get('_goal_button'):
    ri[I_GOAL_BUTTON] = int(ts['_goal_button'][len('buttons'):])
  if '_state' in ts:
    ri[I_BTN_STATE] = 1 if ts['_state'] == 'NORMAL' else 0
  if '_button_timer' in ts:
    ri[I_BTN_TIMER] = ts['_button_timer']
  if '_catch_timer' in ts:
    ri[I_CATCH_TIMER] = ts['_catch_timer']
  if ts.get('_current_radius') is not None:
    rf[F_CATCH + 2] = ts['_current_radius']
    rf[F_CATCH + 3] = ts['_next_radius']
  if '_origin' in ts:
    rf[F_CATCH:F_CATCH + 2] = ts['_origin']
  if '_active_buttons' in ts:
    m = 0
    for n in ts['_active_buttons']:
      m |= 1 << int(n[len('buttons'):])
    ri[I_ACTIVE_MASK] = m


def set_poses(rf, names, pos, yaw=None, v0=None, wz=None):
  """Write body positions (dict or list aligned with names) into a record."""
  if not isinstance(pos, dict):
    pos = dict(zip(names, pos))
  rf[F_ROBOT:F_ROBOT + 2] = pos['robot'][:2]
  if yaw is not None:
    rf[F_ROBOT + 2] = yaw
  if v0 is not None:
    rf[F_ROBOT + 3:F_ROBOT + 5] = v0[:2]
  if wz is not None:
    rf[F_ROBOT + 5] = wz
  for k, n in enumerate(_names(names, 'hazards')):
    rf[F_HAZARDS + 2 * k:F_HAZARDS + 2 * k + 2] = pos[n][:2]
  for k, n in enumerate(_names(names, 'vases')):
    rf[F_VASES + 6 * k:F_VASES + 6 * k + 2] = pos[n][:2]
  for k, n in enumerate(_names(names, 'pillars')):
    rf[F_PILLARS + 2 * k:F_PILLARS + 2 * k + 2] = pos[n][:2]
  for k, n in enumerate(_names(names, 'buttons')):
    rf[F_BUTTONS + 2 * k:F_BUTTONS + 2 * k + 2] = pos[n][:2]
  if 'box' in pos:
    rf[F_BOX:F_BOX + 2] = pos['box'][:2]


def base_record(task, names, keepouts, env_id=0, robot='point'):
  rf = np.zeros(REC_FLOATS, np.float64)
  ri = np.zeros(REC_INTS, np.int32)
  ri[I_TASK] = TASK_ID[task]
  ri[I_NH] = len(_names(names, 'hazards'))
  ri[I_NV] = len(_names(names, 'vases'))
  ri[I_NP] = len(_names(names, 'pillars'))
  ri[I_NB] = len(_names(names, 'buttons'))
  ri[I_BOX_KIND] = BOX_KIND.get(task, 0)
  ri[I_BTN_STATE] = 1
  ri[I_ENV_ID] = env_id
  rf[F_GEAR] = 3.0 if task == 'go_to_goal_motor' else 0.3
  rf[F_DAMP] = 0.001 if task == 'go_to_goal_damping' else 0.01
  rf[F_ACTION_NOISE] = 0.01
  rf[F_CTRL_SCALE:F_CTRL_SCALE + 12] = 1.0
  rf[F_HAZARD_SIZE], rf[F_VASE_SIZE], rf[F_PILLAR_SIZE] = 0.2, 0.1, 0.2
  rf[F_KEEPOUT] = keepouts['robot']
  rf[F_KEEPOUT + 1] = keepouts.get('hazards0', 0.2)
  rf[F_KEEPOUT + 2] = keepouts.get('vases0', 0.15)
  rf[F_KEEPOUT + 3] = keepouts.get('pillars0', 0.3)
  rf[F_KEEPOUT + 4] = keepouts.get('box', 0.5)
  rf[F_CATCH + 2], rf[F_CATCH + 3] = 1.0, 0.2  # catch_goal.py:15-16
  if robot != 'doggo':
    rf[144 + 5] = 1.0  # car rear-ball quaternion w (ignored by point)
  return rf, ri


def episode_init_record(ep):
  names = ep['names']
  rf, ri = base_record(ep['task'], names, ep['keepouts'], robot=ep['robot'])
  set_poses(rf, names, ep['init']['body_pos'], yaw=ep['init_robot_rot'])
  set_robot_planar(rf, ep['robot'], ep['init_robot_rot'], 0.0)
  rf[F_ROBOT0:F_ROBOT0 + 3] = rf[F_ROBOT:F_ROBOT + 3]
  if 'goal' in ep['init']['body_pos']:
    rf[F_GOAL:F_GOAL + 2] = ep['init']['body_pos']['goal'][:2]
  set_task_state(rf, ri, ep['init']['task_state'])
  return rf, ri


def contact_inputs(robot, contacts):
  """(cost contact count, button mask) by the rule of mujoco_bridge.py:177-191."""
  geoms = ROBOT_GEOMS[robot]
  count, mask = 0, 0
  for g1, g2 in contacts:
    part = g1 in geoms or g2 in geoms
    if part and any(g1.startswith(p) or g2.startswith(p) for p in OBSTACLE_PREFIXES):
      count += 1
    for g in (g1, g2):
      if part and g.startswith('buttons'):
        mask |= 1 << int(g[len('buttons'):])
  return count, mask


def rs_from_dump(d):
  rs = np.random.RandomState()
  rs.set_state(('MT19937', np.array(d['key'], np.uint32), d['pos'], d['has_gauss'],
                d['cached_gaussian']))
  return rs


def rs_copy(rs):
  c = np.random.RandomState()
  c.set_state(rs.get_state())
  return c


def rs_words(rs, n):
  """n raw MT19937 words (advances rs)."""
  return rs.randint(0, 2**32, size=n, dtype=np.uint32) if n else np.zeros(0, np.uint32)


def rs_probe(rs):
  return float(rs_copy(rs).random_sample())
