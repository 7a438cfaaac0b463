"""Pin the CPU oracle (oracle/sag_oracle.c) against fixtures generated from the
reference's own NumPy code (oracle/gen_golden.py -> tests/golden/)."""
import numpy as np
import pytest

import golden_util as gu
from oracle_lib import (F_GOAL, F_LAST, I_FLAGS, Oracle)


@pytest.fixture(scope='module')
def oracle():
  return Oracle()


def test_lidar_matches_reference(oracle):
  """safe_adaptation_gym.py:174-223 on 260 cases incl. exact-axis, tilted base, empty."""
  z = np.load(gu.GOLDEN + '/lidar.npz')
  n = len(z['count'])
  for i in range(n):
    k = int(z['count'][i])
    obs, bins = oracle.lidar(z['robot_pos'][i], z['robot_mat'][i], z['points'][i, :k])
    # values: NumPy's matmul/BLAS may round the 3-term dot products differently by an
    # ulp; the bin indices (below) are what must be bit-exact.
    np.testing.assert_allclose(obs, z['obs'][i], rtol=0, atol=1e-14, err_msg=f'case {i}')
    assert np.array_equal(obs > 0, z['obs'][i] > 0)
    for j in range(k):
      if z['bins'][i, j] >= 0:
        assert bins[j] == z['bins'][i, j], (i, j)


EPISODES = None


def episodes():
  global EPISODES
  if EPISODES is None:
    EPISODES = gu.load_json_gz('episodes.json.gz')
  return EPISODES


@pytest.mark.parametrize('robot,task', gu.episode_keys(), ids=lambda v: v)
def test_episode_matches_reference(oracle, robot, task):
  """Replay the reference's step() (scripted poses) through the oracle with physics
  off: noise draw order, reward, cost, lidar grouping, goal resample, button / catch
  state machines and RNG consumption must all match - for the 14 Point episodes and the Car
  (push_box, go_to_goal) and Doggo (press_buttons, catch_goal) ones, whose observations also pin
  the 72- and 104-column sensor order of safe_adaptation_gym.py:225-237."""
  ep = [e for e in episodes() if e['robot'] == robot and e['task'] == task][0]
  rid, nu, od = gu.ROBOT_ID[robot], gu.ROBOT_NU[robot], gu.ROBOT_OBS[robot]
  cols = gu.PINNED_SENSOR_COLS[robot]
  names = ep['names']
  rf, ri = gu.episode_init_record(ep)
  e = oracle.env(rf, ri)
  rs = gu.rs_from_dump(ep['rs_state'])
  # first observation (reset): lidar + non-acceleration sensors
  out = oracle.observe(e, rid)
  obs0 = np.array(out.obs[:od])
  assert len(ep['init_obs']) == od
  np.testing.assert_allclose(obs0[:48], ep['init_obs'][:48], rtol=0, atol=1e-12)
  icols = gu.INIT_PINNED_COLS[robot]
  np.testing.assert_allclose(obs0[icols], np.array(ep['init_obs'])[icols], rtol=0, atol=1e-12)
  n_met = 0
  for t, st in enumerate(ep['steps']):
    noise = rs.normal(size=nu)  # safe_adaptation_gym.py:63-65
    ctrl = np.clip(np.array(st['action']) + 0.01 * noise, -1, 1)
    np.testing.assert_allclose(ctrl, st['ctrl'], rtol=0, atol=1e-15)
    tape = gu.rs_words(gu.rs_copy(rs), 512)
    rf, ri = oracle.record(e)
    gu.set_poses(rf, names, st['pos'], yaw=st['robot_yaw'], v0=st['robot_v0'],
                 wz=st['sensors']['gyro'][2])
    gu.set_robot_planar(rf, robot, st['robot_yaw'], st['sensors']['gyro'][2])
    e = oracle.env(rf, ri)
    cc, mask = gu.contact_inputs(robot, st['contacts'])
    out = oracle.step(e, rid, st['action'], noise=noise, tape=tape, nstep=0,
                      ext_contacts=cc, ext_btn_mask=mask)
    gu.rs_words(rs, out.tape_used)
    rf, ri = oracle.record(e)
    assert ri[I_FLAGS] == 0
    # outputs
    nr = len(st['reward'])
    np.testing.assert_allclose(np.array(out.reward[:nr]), st['reward'], rtol=0, atol=1e-12,
                               err_msg=f'step {t}')
    assert out.cost == int(st['cost']), t
    assert out.done == int(st['done'])
    assert len(st['obs']) == od
    obs = np.array(out.obs[:od])
    np.testing.assert_allclose(obs[:48], st['obs'][:48], rtol=0, atol=1e-12, err_msg=f'step {t}')
    # sensors: the accelerometer (and Doggo's touch forces) are physics-derived (not pinned); the rest is
    # kinematics, in the reference's column order
    np.testing.assert_allclose(obs[cols], np.array(st['obs'])[cols], rtol=0, atol=1e-12, err_msg=f'step {t}')
    # state after
    assert gu.rs_probe(rs) == st['rs_probe'], f'RNG position diverged at step {t}'
    ts = st['task_state']
    if ts.get('_last_goal_distance') is not None:
      assert abs(rf[F_LAST] - ts['_last_goal_distance']) < 1e-12
    if ts.get('_last_box_distance') is not None:
      assert abs(rf[F_LAST + 1] - ts['_last_box_distance']) < 1e-12
      assert abs(rf[F_LAST + 2] - ts['_last_box_goal_distance']) < 1e-12
    if 'goal' in names:
      g = st['pos'][names.index('goal')]
      np.testing.assert_allclose(rf[F_GOAL:F_GOAL + 2], g[:2], rtol=0, atol=1e-12)
    rf2, ri2 = rf.copy(), ri.copy()
    gu.set_task_state(rf2, ri2, ts)
    np.testing.assert_array_equal(ri2, ri)
    np.testing.assert_allclose(rf2, rf, rtol=0, atol=1e-12)
    n_met += out.goal_met
  assert n_met > (3 if robot == 'point' else 0), 'fixture should exercise goal-met events'
