"""HIP path (libsag.so through the C ABI) vs the CPU oracle and the golden fixtures.
Run on an MI355X: python -m pytest tests -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import batch_util as bu
import golden_util as gu
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

OBS_TOL = 2e-5     # fp32 state -> fp64 lidar closeness, sensors
STATE_TOL = 1e-4   # one-step qpos/qvel: HIP fp32 vs oracle fp64 from the same fp32 state
REW_TOL = 2e-6


@pytest.fixture(scope='module')
def nat():
  from safe_adaptation_gym_amd import _native
  if _native.device_count() < 1:
    pytest.fail('no HIP device visible: the GPU tests need an MI355X')
  return _native


@pytest.fixture(scope='module')
def oracle():
  return Oracle()


@pytest.fixture(scope='module')
def oracle32():
  return Oracle(f32=True)


# ----------------------------------------------------------------------------------
# BASELINE config 2: lidar + hazard cost kernel, 4096 envs
# ----------------------------------------------------------------------------------
def _lidar_inputs(n, K, seed):
  rs = np.random.RandomState(seed)
  robot = np.concatenate([rs.uniform(-2, 2, (n, 2)), rs.uniform(0, 2 * np.pi, (n, 1))], 1).astype(np.float32)
  pts = rs.uniform(-2.5, 2.5, (n, K, 2)).astype(np.float32)
  grp = rs.randint(0, 4, (n, K)).astype(np.uint8)
  haz = (grp == 1) & (rs.uniform(size=(n, K)) < 0.5)
  grp = (grp | (haz.astype(np.uint8) << 7)).astype(np.uint8)
  # make hazard hits common: pull some hazards next to the robot
  near = haz & (rs.uniform(size=(n, K)) < 0.1)
  pts[near] = (robot[:, None, :2] + rs.uniform(-0.25, 0.25, (n, K, 2)).astype(np.float32))[near]
  return robot, pts, grp


@pytest.mark.parametrize('team', [None, 0, 2, 4, 8, 16])
@pytest.mark.parametrize('K', [21, 13, 30])
def test_lidar_cost_kernel_bitexact_bins_and_flags(nat, oracle, K, team, monkeypatch):
  """Every kernel behind sag_lidar_cost on the same inputs.  A lane per pose (team 0): K <= 21 runs the register-resident
  kernel (k_lidar_cost_reg), more points per env the LDS-staged one; teams of 2 / 4 / 8 / 16 lanes per pose
  (k_lidar_cost_team): 16 is what this batch size (BASELINE config 2's 4096) selects by itself (team None), 8 and 4 what
  larger batches get."""
  n = 4096
  robot, pts, grp = _lidar_inputs(n, K, 1)
  if team is not None:
    monkeypatch.setenv('SAG_LIDAR_TEAM', str(team))
  ctx = nat.Context('point', n)
  lidar, bins, cost = ctx.lidar_cost(robot, pts, grp)
  o_lidar, o_bins, o_cost = oracle.lidar_cost(robot, pts, grp)
  np.testing.assert_array_equal(bins, o_bins)          # bit-exact bin indices
  np.testing.assert_array_equal(cost, o_cost)          # bit-exact cost flags
  assert 0.05 < cost.mean() < 0.95
  np.testing.assert_allclose(lidar, o_lidar, rtol=0, atol=1e-6)
  ctx.close()


@pytest.mark.parametrize('team', [0, 2, 4, 8, 16])
@pytest.mark.parametrize('K', [21, 30])
def test_lidar_cost_device_entry_partial_block_and_unaligned_buffers(nat, K, team, monkeypatch):
  """sag_lidar_cost_device (the entry bench.py times): n not a multiple of the 64-env block (nor of a team kernel's 16 /
  4 poses) and buffers that start 4 bytes into their allocation (the kernels' linear 16-byte copies need their unaligned
  branch) give the same bins, values and flags as the host-buffer entry; one timed launch is reported as one launch; a
  rejected call (K beyond the LDS staging) reserves no timing slot."""
  n = 1000 + 37
  robot, pts, grp = _lidar_inputs(n, K, 3)
  ctx = nat.Context('point', 64)
  monkeypatch.setenv('SAG_LIDAR_TEAM', '0')     # (read at every launch) the host-buffer entry on the lane-per-pose kernels ...
  lidar0, bins0, cost0 = ctx.lidar_cost(robot, pts, grp)
  monkeypatch.setenv('SAG_LIDAR_TEAM', str(team))   # ... against the device entry on each kernel form
  off = 4
  bufs = {}
  for name, arr in (('robot', robot.astype(np.float32)), ('pts', pts.astype(np.float32)), ('grp', grp.astype(np.uint8))):
    d = ctx.dev_alloc(arr.nbytes + 64)
    ctx.dev_upload(C.c_void_p(d.value + off), np.ascontiguousarray(arr))
    bufs[name] = d
  d_lid, d_bins, d_cost = ctx.dev_alloc(n * 48 * 4 + 64), ctx.dev_alloc(n * K * 4 + 64), ctx.dev_alloc(n + 64)
  p = lambda d: C.c_void_p(d.value + off)
  ctx.enable_timing(True)
  ctx.kernel_time_ms(reset=True)
  with pytest.raises(nat.SagError):
    ctx.lidar_cost_device(n, 101, p(bufs['robot']), p(bufs['pts']), p(bufs['grp']), p(d_lid), p(d_bins), p(d_cost))
  ctx.lidar_cost_device(n, K, p(bufs['robot']), p(bufs['pts']), p(bufs['grp']), p(d_lid), p(d_bins), p(d_cost))
  ctx.wait()
  ms, cnt = ctx.kernel_time_ms(reset=True)
  assert cnt == 1 and 0 < ms < 100, (ms, cnt)
  ctx.enable_timing(False)
  lidar = ctx.dev_download(p(d_lid), (n, 48), np.float32)
  bins = ctx.dev_download(p(d_bins), (n, K), np.int32)
  cost = ctx.dev_download(p(d_cost), (n,), np.uint8)
  np.testing.assert_array_equal(bins, bins0)
  np.testing.assert_array_equal(cost, cost0)
  np.testing.assert_array_equal(lidar, lidar0)
  ctx.close()


def test_lidar_cost_kernel_golden_planar_cases(nat):
  """Reference-generated cases (tests/golden/lidar.npz) that a planar robot can express."""
  z = np.load(gu.GOLDEN + '/lidar.npz')
  ctx = nat.Context('point', 64)
  checked = 0
  for i in range(len(z['count'])):
    m, k = z['robot_mat'][i], int(z['count'][i])
    planar = abs(m[2, 2] - 1) < 1e-15 and k > 0
    if not planar:
      continue
    yaw = np.arctan2(m[1, 0], m[0, 0])
    f32 = lambda a: np.asarray(a, np.float32)
    if not (np.array_equal(f32(z['robot_pos'][i][:2]), z['robot_pos'][i][:2]) and
            np.array_equal(f32(z['points'][i, :k]), z['points'][i, :k]) and float(np.float32(yaw)) == yaw):
      continue  # inputs not exactly representable in the fp32 ABI
    lidar, bins, _ = ctx.lidar_cost(np.r_[z['robot_pos'][i][:2], yaw][None], z['points'][i, :k][None],
                                    np.ones((1, k), np.uint8))
    want = z['bins'][i, :k]
    ok = want >= 0
    np.testing.assert_array_equal(bins[0][ok], want[ok], err_msg=f'case {i}')
    np.testing.assert_allclose(lidar[0, :16], z['obs'][i], rtol=0, atol=1e-6)
    checked += 1
  assert checked >= 20
  ctx.close()


def test_lidar_cost_empty_and_ragged(nat, oracle):
  ctx = nat.Context('point', 8)
  robot = np.zeros((3, 3), np.float32)
  lidar, bins, cost = ctx.lidar_cost(robot, np.zeros((3, 0, 2), np.float32), np.zeros((3, 0), np.uint8))
  assert not lidar.any() and not cost.any()
  # ragged: inactive slots (group 0) are ignored and report bin -1
  pts = np.array([[[1, 0], [0, 1], [9, 9]]] * 3, np.float32)
  grp = np.array([[1, 0, 0], [1, 2, 0], [0, 0, 0]], np.uint8)
  lidar, bins, cost = ctx.lidar_cost(robot, pts, grp)
  o = oracle.lidar_cost(robot, pts, grp)
  np.testing.assert_array_equal(bins, o[1])
  np.testing.assert_allclose(lidar, o[0], atol=1e-7)
  assert (bins[2] == -1).all() and bins[0, 0] == 0 and bins[1, 1] == 4
  ctx.close()


def _mat2quat(m):
  """Rotation matrix -> unit quaternion (w, x, y, z), largest-component branch."""
  t = np.trace(m)
  if t > 0:
    s = np.sqrt(t + 1.0) * 2
    q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
  else:
    i = int(np.argmax(np.diag(m)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
    q = [0.0] * 4
    q[0] = (m[k, j] - m[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (m[j, i] + m[i, j]) / s
    q[1 + k] = (m[k, i] + m[i, k]) / s
  return np.array(q) / np.linalg.norm(q)


def test_tilted_lidar_reference_cases_on_device(nat):
  """The 60 tilted-base cases of tests/golden/lidar.npz (reference _lidar with a pitched / rolled robot_mat:
  e = (d @ R)[:2] picks up -z R[2,:2], safe_adaptation_gym.py:197-216) through the device's Doggo lidar: the
  base pose goes in as position + quaternion (sag_set_state), the points as hazards + pillars, and
  sag_observe's obstacle lidar is compared with the fixture directly.  Tolerance: the ABI stores fp32 (pose,
  quaternion, points: relative 6e-8, i.e. <= 3e-6 bins and 1e-6 of closeness at these ranges); the lidar value is
  continuous across bin boundaries, so no case needs excluding."""
  z = np.load(gu.GOLDEN + '/lidar.npz')
  idx = [i for i in range(len(z['count'])) if abs(z['robot_mat'][i][2, 2] - 1) > 1e-15 and 0 < z['count'][i] <= 11]
  assert len(idx) == 60
  n = len(idx)
  rf = np.zeros((n, nat.REC_FLOATS), np.float32)
  ri = np.zeros((n, nat.REC_INTS), np.int32)
  E = nat.F_ROBOT_EXT
  for j, i in enumerate(idx):
    b_rf, b_ri = gu.base_record('go_to_goal', [], {'robot': 0.4}, env_id=j, robot='doggo')
    m, k, pts = z['robot_mat'][i], int(z['count'][i]), z['points'][i]
    b_rf[0:2] = z['robot_pos'][i][:2]
    b_rf[2] = np.arctan2(m[1, 0], m[0, 0])
    b_rf[E] = z['robot_pos'][i][2]
    b_rf[E + 1:E + 5] = _mat2quat(m)
    nh = min(k, 9)
    b_rf[nat.F_HAZARDS:nat.F_HAZARDS + 2 * nh] = pts[:nh].ravel()
    b_rf[nat.F_PILLARS:nat.F_PILLARS + 2 * (k - nh)] = pts[nh:k].ravel()
    b_rf[nat.F_GOAL:nat.F_GOAL + 2] = [3.0, 3.0]
    b_ri[nat.I_NH], b_ri[nat.I_NP] = nh, k - nh
    rf[j], ri[j] = b_rf, b_ri
  ctx = nat.Context('doggo', n)
  ctx.set_state(rf, ri)
  obs = ctx.observe()
  want = np.stack([z['obs'][i] for i in idx])
  np.testing.assert_allclose(obs[:, :16], want, rtol=0, atol=OBS_TOL)
  assert (want > 0).sum() > 100 and not obs[:, 16:32].any()   # (objects group empty)
  ctx.close()


# ----------------------------------------------------------------------------------
# full step: lockstep against the oracle
# ----------------------------------------------------------------------------------
LOCKSTEP_TASKS = ['go_to_goal', 'go_to_goal_scarce', 'go_to_goal_motor', 'go_to_goal_damping',
                  'catch_goal', 'unsupervised', 'press_buttons', 'press_buttons_scarce', 'collect',
                  'push_box', 'push_box_scarce', 'haul_box', 'roll_rod', 'dribble_ball']


def _flags_agree(dev, orc, margin, tol=1e-5):
  """Flags must be equal unless the oracle says the decision sits within tol of its threshold."""
  bad = (dev != orc) & (margin > tol)
  return int(bad.sum()), int((dev != orc).sum())


def _lidar_e2e_bound(nat, d_rf, o_rf):
  """Per-env bound on |lidar(device state) - lidar(oracle state)| derived from the ACTUAL state difference of
  the two (first order, x1.5): a body at distance D seen with closeness s = (5 - D) / 5 contributes
  s * (16 / 2 pi) * (|d rel. position| / D + |d yaw|) through the bin fraction and |d rel. position| / 5 through
  the closeness; the lidar is a max over bodies, so the env's bound is the max over its bodies.  Replaces a
  constant (1e-3), which is wrong in both directions: far too loose for a body 3 m away, too tight for the box a
  car is tethered to at 0.6 m (the error scales like 1 / D)."""
  n = len(d_rf)
  cols = ([(nat.F_HAZARDS + 2 * k, nat.F_HAZARDS + 2 * k + 1) for k in range(nat.MAX_HAZARDS)] +
          [(nat.F_VASES + 6 * k, nat.F_VASES + 6 * k + 1) for k in range(nat.MAX_VASES)] +
          [(nat.F_PILLARS + 2 * k, nat.F_PILLARS + 2 * k + 1) for k in range(nat.MAX_PILLARS)] +
          [(nat.F_BUTTONS + 2 * k, nat.F_BUTTONS + 2 * k + 1) for k in range(nat.MAX_BUTTONS)] +
          [(nat.F_BOX, nat.F_BOX + 1), (nat.F_GOAL, nat.F_GOAL + 1)])
  d_rob = np.hypot(d_rf[:, 0] - o_rf[:, 0], d_rf[:, 1] - o_rf[:, 1]).astype(np.float64)
  d_yaw = np.abs(d_rf[:, 2] - o_rf[:, 2]).astype(np.float64)
  bound = np.zeros(n)
  for cx, cy in cols:
    D = np.hypot(o_rf[:, cx] - o_rf[:, 0], o_rf[:, cy] - o_rf[:, 1]).astype(np.float64)
    d_rel = d_rob + np.hypot(d_rf[:, cx] - o_rf[:, cx], d_rf[:, cy] - o_rf[:, cy])
    # + what the fp32 record hides of the oracle's fp64 state (half an ulp per coordinate): it matters when the robot
    # sits within millimetres of a body's centre (a goal being met), where the bearing amplifies everything by 1 / D
    d_rel = d_rel + 1.2e-7 * (np.abs(o_rf[:, 0]) + np.abs(o_rf[:, 1]) + np.abs(o_rf[:, cx]) + np.abs(o_rf[:, cy]))
    s = np.clip(5.0 - D, 0, None) / 5.0
    with np.errstate(divide='ignore', invalid='ignore'):
      b = s * (16 / (2 * np.pi)) * (np.where(d_rel > 0, d_rel / D, 0.0) + d_yaw) + d_rel / 5.0
    bound = np.maximum(bound, np.nan_to_num(b, nan=np.inf))
  # + the f(state) tolerance on either side (the oracle's record is its fp64 state rounded to fp32)
  return 1.5 * bound + 2 * OBS_TOL


CAR_TASKS = ['go_to_goal', 'push_box', 'press_buttons', 'unsupervised', 'catch_goal', 'haul_box']
MIXED = 'multitask'  # per-env task ids drawn by the benchmark's TaskSampler (BASELINE config 4 shape)


@pytest.mark.parametrize('robot,task', [('point', t) for t in LOCKSTEP_TASKS] + [('car', t) for t in CAR_TASKS] +
                         [('point', MIXED), ('car', MIXED)])
def test_step_lockstep_vs_oracle(nat, oracle, oracle32, robot, task):
  _lockstep(nat, oracle, oracle32, robot, task, int(os.environ.get('SAG_LOCKSTEP_SCALE', '1')))


# The soak at its real size (8 x the envs: 1536 envs x 160 steps per case): the two cases that were marginal
# against the old constant end-to-end lidar bound (car tethered to / pushing the box at < 1 m) and the two
# multitask mixes
@pytest.mark.parametrize('robot,task', [('car', 'haul_box'), ('car', MIXED), ('point', MIXED), ('car', 'push_box')])
def test_step_lockstep_soak(nat, oracle, oracle32, robot, task):
  _lockstep(nat, oracle, oracle32, robot, task, 8)


def _lockstep(nat, oracle, oracle32, robot, task, scale):
  """Every step: take the device state, advance the device AND both oracle builds (fp64 =
  the specification, fp32 = same source in the device's precision) from it with identical
  action / noise / random tape, compare outputs and next state, continue from the device.

  Stated tolerance: after one step from identical fp32 state, qpos/qvel agree within
  1e-4 (abs + rel) with the fp64 oracle and 2e-5 with the fp32 oracle, except for
  contact-onset threshold events (a penetration within float rounding of zero at a substep
  boundary switches a stiff contact on one substep earlier), allowed on <= 0.05 % of
  env-steps.  Discrete outputs (goal_met, done, RNG words consumed, task ints) are exact;
  cost flags are exact except where the oracle reports the decision within 1e-5 of its
  threshold.  Observations are checked as a function of the device's own post-step state (tight) and end to
  end against the oracle's within the bound that the actual state difference implies (_lidar_e2e_bound)."""
  n, T = 192 * scale, 160
  rid = {'point': 0, 'car': 1}[robot]
  od = 60 if robot == 'point' else 72
  if task == MIXED:
    # heterogeneous batch: task of env i from benchmark.make('multitask') (task_sampler.py:15-19)
    from safe_adaptation_gym_amd import benchmark
    names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=666).train_tasks]
    assert len(set(names)) >= 10
    rf, ri = bu.sample_records_native(robot, names, n, seed=666)
  else:
    rf, ri = bu.sample_records(robot, task, n, seed=666)
  if task not in ('haul_box',):
    rf = bu.goal_beyond_box(rf, ri)
  ctx = nat.Context(robot, n, seed=1234)
  ctx.set_layout(rf, ri)
  rng = np.random.RandomState(7)
  mt = np.random.RandomState(99)
  obs0 = ctx.observe()
  rf, ri = ctx.get_state()
  arr = oracle.make_batch(rf, ri)
  np.testing.assert_allclose(obs0, oracle.observe_batch(arr, rid, od), rtol=0, atol=OBS_TOL)
  n_met = n_cost = n_near = viol64 = viol32 = acc_bad = acc_e2e = 0
  bad64_env, bad32_env = np.zeros(n, int), np.zeros(n, int)
  for t in range(T):
    rf, ri = ctx.get_state()
    arr, arr32 = oracle.make_batch(rf, ri), oracle32.make_batch(rf, ri)
    act = bu.pursuit_actions(rf, ri, rng, robot=robot)
    noise = mt.normal(size=(n, 2)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
    d_obs, d_rew, d_cost, d_done, d_met, d_used = ctx.step(act, noise, tape)
    o_obs, o_rew, o_cost, o_done, o_met, o_used, o_margin = oracle.step_batch_full(arr, rid, act, noise, tape, obs_dim=od)
    o32 = oracle32.step_batch_full(arr32, rid, act, noise, tape, obs_dim=od)
    d_rf, d_ri = ctx.get_state()
    o_rf, o_ri = oracle.batch_records(arr)
    o32_rf, _ = oracle32.batch_records(arr32)
    # state: rows outside the tolerance are counted, not hidden
    tol64 = np.full(d_rf.shape[1], STATE_TOL)
    tol32 = np.full(d_rf.shape[1], 2e-5)
    # rest capture (REST_W = 1e-4 rad/s) and contacts at zero depth between resting bodies are
    # decided at float rounding: free-body spin rates may differ by that scale
    for wf in [46] + [81 + 6 * k + 5 for k in range(10)]:
      tol32[wf] = 2e-4
      tol64[wf] = 2e-4
    if robot == 'car':
      # the rear ball (2.6 g, I = 2.6e-6 kg m^2, joint damping 1e-3: time constant I / d = 2.6 ms < h) and the wheels
      # are stiff spinning parts: the fp32 oracle (world-frame friction, divisions) and the device (body-frame
      # constants, reciprocals) round differently by an ulp per operation, amplified here ~100x; values are O(10) rad/s
      for wf in range(144, 149):
        tol32[wf] = 2e-4
    if task in ('dribble_ball', MIXED):
      # the ball's spin (unobservable; I = 4.5e-8 kg m^2) is set by friction torques of a stiff,
      # underdamped contact (solref .018 .2): fp32 rounding is amplified ~1000x there
      tol64[46] = tol32[46] = 5e-3
    bad64 = (np.abs(d_rf - o_rf) > tol64 + tol64 * np.abs(o_rf)).any(1)
    bad32 = (np.abs(d_rf - o32_rf) > tol32 + tol32 * np.abs(o32_rf)).any(1)
    viol64 += int(bad64.sum())
    viol32 += int(bad32.sum())
    bad64_env += bad64; bad32_env += bad32
    ok = ~bad64
    # discrete outputs
    np.testing.assert_array_equal(d_done, o_done)
    np.testing.assert_array_equal(d_met[ok], o_met[ok], err_msg=f'goal_met step {t}')
    np.testing.assert_array_equal(d_used[ok], o_used[ok])
    # cost flags: exact vs the fp64 oracle unless the decision sits at a threshold - the hazard
    # distance within 1e-5 of its radius (reported by the oracle) or a contact whose
    # penetration changes sign with the arithmetic (then the fp32 oracle sides with the device)
    hard, soft = _flags_agree(d_cost[ok], o_cost[ok], o_margin[ok])
    if hard:
      mism = ok & (d_cost != o_cost) & (o_margin > 1e-5)
      assert (d_cost[mism] == o32[2][mism]).all(), f'cost flag mismatch away from any threshold at step {t}'
      soft += hard
    n_near += soft
    np.testing.assert_array_equal(d_ri[ok], o_ri[ok], err_msg=f'task ints step {t}')
    # observation = f(post-step state): check f tightly on the device's own post-step state,
    # and the end-to-end values loosely (they inherit the one-step state tolerance; the lidar
    # amplifies bearing errors by 16/2pi per radian)
    f_obs = oracle.observe_batch(oracle.make_batch(d_rf, d_ri), rid, od)
    np.testing.assert_allclose(d_obs[:, :48], f_obs[:, :48], rtol=0, atol=OBS_TOL, err_msg=f'lidar step {t}')
    np.testing.assert_allclose(d_obs[:, 50:], f_obs[:, 50:], rtol=0, atol=1e-5, err_msg=f'sensors step {t}')
    e2e = _lidar_e2e_bound(nat, d_rf, o_rf)
    worst = np.abs(d_obs[ok, :48] - o_obs[ok, :48]).max(1) - e2e[ok]
    assert (worst <= 0).all(), f'lidar e2e step {t}: {worst.max():.3g} above the bound implied by the state difference'
    np.testing.assert_allclose(d_obs[ok, 50:], o_obs[ok, 50:], rtol=2e-4, atol=2e-4, err_msg=f'sensors e2e step {t}')
    # accelerometer = forward dynamics at the post-step state incl. stiff contact forces
    # (k = 2770 /s^2 per metre of penetration, b = 105 /s per m/s): checked as a function of
    # the device's own post-step state; end to end it is only counted
    f_acc = oracle.step_batch_full(oracle.make_batch(d_rf, d_ri), rid, act, noise, tape, nstep=0, obs_dim=od)[0][:, 48:50]
    acc_bad += int((np.abs(d_obs[:, 48:50] - f_acc) > 2e-2 + 2e-3 * np.abs(f_acc)).any(1).sum())
    acc_e2e += int((np.abs(d_obs[:, 48:50] - o_obs[:, 48:50]) > 2e-2 + 5e-3 * np.abs(o_obs[:, 48:50])).any(1).sum())
    np.testing.assert_allclose(d_rew[ok], o_rew[ok], rtol=0, atol=2e-4, err_msg=f'reward step {t}')
    n_met += int(d_met.sum())
    n_cost += int(d_cost.sum())
  assert n_met > (5 if ri[:, 5].any() else 20), 'the rollout should exercise goal-met events'
  assert n_cost > 20, 'the rollout should exercise cost events'
  assert n_near <= 0.001 * n * T
  # Budgets per TASK, at ~2-3x the measured counts (profiles/r04_lockstep_counts.txt; round 3: Point tasks <= 3 of 30 720
  # env-steps, Car tasks <= 9, car/haul_box 29 against the fp32 build, dribble_ball 105 of 30 720 against the fp32 build):
  #   dribble_ball   the ball's contact is stiff and underdamped (k h^2 = 1.4) and amplifies rounding: 1 % for the Point; the
  #                  Car (2x the substep, k h^2 = 5.7; it keeps the ball in SUSTAINED contact against its bumper) 12 % -
  #                  measured 92 of the 1920 env-steps of the Car mix's 12 dribble_ball envs, nearly all from one env
  #   car            8 geoms and a 2x longer step, proportionally more contact onsets per env-step: 0.05 %, haul_box 0.15 %
  #   point          0.02 %
  # A mixed batch is budgeted task by task - the sum over its envs of their task's rate (round 3 gave the Car mix one
  # rate, 0.4 %, which its dribble_ball envs alone filled to 85 %: the verdict's "budgets at 2 - 3 x measured does not hold
  # there") - and the counts of the mix's dribble_ball envs are logged beside the rest.
  def task_frac(name):
    if name == 'dribble_ball': return 0.12 if robot == 'car' else 0.01
    if robot == 'car': return 0.0015 if name == 'haul_box' else 0.0005
    return 0.0002
  env_names = names if task == MIXED else [task] * n
  budget = T * sum(task_frac(nm) for nm in env_names)
  is_ball = np.array([nm == 'dribble_ball' for nm in env_names])
  if task == MIXED:
    rest = T * sum(task_frac(nm) for nm in env_names if nm != 'dribble_ball')
    _log_lockstep(f'{robot}/{task}: of which its {int(is_ball.sum())} dribble_ball envs {int(bad64_env[is_ball].sum())} / {int(bad32_env[is_ball].sum())} (fp64 / fp32), '
                  f'the other {int((~is_ball).sum())} envs {int(bad64_env[~is_ball].sum())} / {int(bad32_env[~is_ball].sum())} against a budget of {rest:.0f}')
    assert bad64_env[~is_ball].sum() <= rest and bad32_env[~is_ball].sum() <= rest
  _log_lockstep(f'{robot}/{task}: {n} envs x {T} steps resynchronised every step | env-steps outside the stated tolerance vs fp64 oracle {viol64} '
                f'({viol64 / (n * T):.2e}), vs fp32 oracle {viol32} ({viol32 / (n * T):.2e}), budget {budget / (n * T):.1e} | cost flags within 1e-5 of a '
                f'threshold {n_near} | accelerometer off on identical state {acc_bad}, end to end {acc_e2e} | goal-met events {n_met}, cost events {n_cost}')
  assert viol64 <= budget, f'{viol64} env-steps outside the fp64 tolerance'
  assert viol32 <= budget, f'{viol32} env-steps outside the fp32 tolerance'
  assert acc_bad <= 0.0005 * n * T, f'{acc_bad} accelerometer readings off on identical state'
  assert acc_e2e <= 0.005 * n * T, f'{acc_e2e} accelerometer readings off end to end'
  ctx.close()


# ----------------------------------------------------------------------------------
# free-running comparison: no resynchronisation, the drift is observed and bounded
# ----------------------------------------------------------------------------------
LOCKSTEP_LOG = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'r04_lockstep_counts.txt')


def _log_lockstep(line):
  """What the lockstep budgets are budgets OF: the measured counts, one line per case (copied to profiles/ at round end)."""
  print(line)
  try:
    os.makedirs(os.path.dirname(LOCKSTEP_LOG), exist_ok=True)
    with open(LOCKSTEP_LOG, 'a') as f:
      f.write(line + '\n')
  except OSError:
    pass


DRIFT_LOG = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'r04_free_running_drift.txt')


@pytest.mark.parametrize('robot,task', [('point', 'go_to_goal'), ('car', 'push_box'), ('doggo', 'go_to_goal')])
def test_free_running_drift(nat, oracle, robot, task, request):
  """Device and fp64 oracle start from the same state and run 200 steps on the same actions, noise and random
  tapes WITHOUT ever being resynchronised (the lockstep tests bound the one-step error only).  Contact dynamics
  amplify rounding - a vase touched one substep earlier ends somewhere else, and a legged robot under random
  torques is outright chaotic - so the yardstick is the oracle itself: the fp64 oracle run twice, once with its
  state rounded to fp32 after every step (what the ABI's fp32 records do to any implementation).  Asserted:
  (a) the device's median drift is within 3x that reference level (+ 1 mm); (b) Point / Car, whose dynamics are
  not chaotic between contacts, keep the median env within 1 mm over 200 steps; (c) the streams of discrete outputs
  stay statistically identical (cost-flag and goal-met agreement).  Figures go to gpurun_out/ and DESIGN.md."""
  rid = {'point': 0, 'car': 1, 'doggo': 2}[robot]
  nu, od = gu.ROBOT_NU[robot], gu.ROBOT_OBS[robot]
  # Doggo: 2048 envs (at 64 envs the standard error of the cost rate was half the rate); the oracle
  # legs run on the host's cores (envs are independent)
  n, T = (2048 if robot == 'doggo' else 192), 200
  if robot == 'doggo' and os.environ.get('SAG_HOSTEMU'):
    pytest.skip('2048 Doggo envs x 200 steps take hours in the host emulator; its Doggo coverage is the lockstep and long-run tests')
  oracle.lib.sago_set_threads(min(16, len(os.sched_getaffinity(0))))
  request.addfinalizer(lambda: oracle.lib.sago_set_threads(1))   # (also when an assertion below fails)
  rf, ri = bu.sample_records_native(robot, task, n, seed=4242)
  ctx = nat.Context(robot, n, seed=99)
  ctx.set_layout(rf, ri)
  rf, ri = ctx.get_state()
  arr = oracle.make_batch(rf, ri)
  arr_r = oracle.make_batch(rf, ri)   # the oracle itself with its state rounded to fp32 after every step (as the ABI stores it)
  rng, mt = np.random.RandomState(3), np.random.RandomState(4)
  dev_cost, orc_cost, dev_met, orc_met, dpos, rpos, ref_cost, ref_met = [], [], [], [], [], [], [], []
  start = rf[:, :2].copy()
  for t in range(T):
    if robot == 'doggo':
      act = mt.uniform(-1, 1, size=(n, nu)).astype(np.float32)
    else:
      act = bu.pursuit_actions(ctx.get_state()[0], ri, rng, robot=robot)
    noise = mt.normal(size=(n, nu)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
    d = ctx.step(act, noise, tape)
    o = oracle.step_batch_full(arr, rid, act, noise, tape, obs_dim=od)
    o_r = oracle.step_batch_full(arr_r, rid, act, noise, tape, obs_dim=od)
    ref_cost.append(o_r[2]); ref_met.append(o_r[4])
    r_rf, r_ri = oracle.batch_records(arr_r)
    arr_r = oracle.make_batch(r_rf, r_ri)
    d_rf = ctx.get_state()[0]
    o_rf = oracle.batch_records(arr)[0]
    dpos.append(np.hypot(d_rf[:, 0] - o_rf[:, 0], d_rf[:, 1] - o_rf[:, 1]))
    rpos.append(np.hypot(r_rf[:, 0] - o_rf[:, 0], r_rf[:, 1] - o_rf[:, 1]))
    dev_cost.append(d[2]); orc_cost.append(o[2]); dev_met.append(d[4]); orc_met.append(o[4])
  dpos, rpos = np.array(dpos), np.array(rpos)
  dev_cost, orc_cost, dev_met, orc_met = (np.array(x) for x in (dev_cost, orc_cost, dev_met, orc_met))
  q = lambda t: np.quantile(dpos[t], [0.5, 0.9, 0.99])
  cost_agree = (dev_cost == orc_cost).mean()
  # cost flags come in runs (an env inside a hazard stays there for many steps): the sampling unit of the rate
  # comparison is the env, not the env-step
  per_env = dev_cost.mean(0) - orc_cost.mean(0)
  cost_se = per_env.std(ddof=1) / np.sqrt(n)
  ref_cost, ref_met = np.array(ref_cost), np.array(ref_met)
  ref_rate = ref_cost.mean()
  ref_agree = (ref_cost == orc_cost).mean()   # the yardstick for the agreement: the oracle against its own fp32-rounded twin
  travel = [float(np.hypot(x[:, 0] - start[:, 0], x[:, 1] - start[:, 1]).mean()) for x in (d_rf, o_rf, r_rf)]
  met_agree = (dev_met == orc_met).mean()
  close = (dpos[-1] < 1e-3).mean()
  line = (f'{robot}/{task}: {n} envs x {T} steps free-running vs fp64 oracle | robot position drift [m] median/p90/p99 at '
          f'step 50: {q(49)[0]:.2e}/{q(49)[1]:.2e}/{q(49)[2]:.2e}, step 100: {q(99)[0]:.2e}/{q(99)[1]:.2e}/{q(99)[2]:.2e}, '
          f'step 200: {q(199)[0]:.2e}/{q(199)[1]:.2e}/{q(199)[2]:.2e} | envs within 1 mm at step 200: {close:.3f} | '
          f'cost-flag agreement per env-step {cost_agree:.5f} (device rate {dev_cost.mean():.4f}, oracle {orc_cost.mean():.4f}, its fp32-rounded '
          f'twin {ref_rate:.4f}; standard error of the per-env rate difference {cost_se:.4f}) | '
          f'goal-met agreement {met_agree:.5f} (device {int(dev_met.sum())}, oracle {int(orc_met.sum())}, twin {int(ref_met.sum())} events) | '
          f'mean distance travelled in {T} steps: device {travel[0]:.3f}, oracle {travel[1]:.3f}, twin {travel[2]:.3f} m | '
          f'cost-flag agreement of the twin with the oracle {ref_agree:.5f} | reference level - the '
          f'fp64 oracle against ITSELF with its state rounded to fp32 after every step: median/p90 at step 200 '
          f'{np.median(rpos[-1]):.2e}/{np.quantile(rpos[-1], 0.9):.2e}')
  print(line)
  try:
    os.makedirs(os.path.dirname(DRIFT_LOG), exist_ok=True)
    with open(DRIFT_LOG, 'a') as f:
      f.write(line + '\n')
  except OSError:
    pass
  assert np.isfinite(dpos).all()
  assert q(199)[0] <= 3 * np.median(rpos[-1]) + 1e-3, 'drift beyond what fp32 state storage alone causes in the oracle'
  if robot != 'doggo':
    assert q(199)[0] < 1e-3, 'the median env should not separate from its oracle twin'
  # Doggo is chaotic: after ~100 steps the flag streams of ANY two runs are independent samples of one distribution
  # (two independent streams of rate p agree on 1 - 2 p (1 - p) of the env-steps), so the bar is the agreement the
  # oracle reaches with its own fp32-rounded twin, not a constant
  assert cost_agree > (ref_agree - 0.01 if robot == 'doggo' else 0.99)
  if robot == 'doggo':
    assert abs(per_env.mean()) < 3 * cost_se, f'cost rates differ by {per_env.mean():.5f}: more than 3 standard errors ({cost_se:.5f}) of the env-to-env scatter'
  else:
    assert abs(per_env.mean()) < max(0.01, 4 * cost_se + 0.002), 'cost rates differ beyond the env-to-env scatter'
  assert met_agree > 0.995
  ctx.close()


# ----------------------------------------------------------------------------------
# golden episodes (reference step() with scripted poses) through the device
# ----------------------------------------------------------------------------------
GOAL_FAMILY = ['go_to_goal', 'go_to_goal_scarce', 'go_to_goal_motor', 'go_to_goal_damping',
               'catch_goal', 'unsupervised']


# Car / Doggo reference episodes (oracle/gen_golden.py:729-732).  push_box (car) and press_buttons (doggo) depend on
# contacts for more than `cost` (button presses): the reference's own contact list of every step goes in through
# sag_set_ext_contacts, so reward, goal-met, button state machine and RNG position are all checked against the
# reference (round 2 copied the task state from the fixture and skipped those).
DEVICE_EPISODES = gu.episode_keys()   # all 18: the 14 Point tasks, car go_to_goal / push_box, doggo catch_goal / press_buttons


@pytest.mark.parametrize('robot,task', DEVICE_EPISODES, ids=lambda v: v)
def test_golden_episode_on_device(nat, robot, task):
  """Reference-generated episodes replayed on the device with the physics off (nstep = 0): the poses and the
  contact list of every step are the fixture's (sag_set_state, sag_set_ext_contacts: what MuJoCo computed in the
  reference), everything else is computed by the kernels and checked against the reference directly: reward, cost,
  goal-met, lidar (with the reference's grouping), goal resampling, button / collect state, RNG consumption; for every
  robot the observation pins the reference's column order (60 / 72 / 104, safe_adaptation_gym.py:225-237)."""
  from oracle_lib import F_GOAL, F_LAST
  ep = [e for e in gu.load_json_gz('episodes.json.gz') if e['robot'] == robot and e['task'] == task][0]
  names = ep['names']
  nu, od = gu.ROBOT_NU[robot], gu.ROBOT_OBS[robot]
  cols, icols = gu.PINNED_SENSOR_COLS[robot], gu.INIT_PINNED_COLS[robot]
  rf, ri = gu.episode_init_record(ep)
  ctx = nat.Context(robot, 1)
  ctx.set_state(rf[None].astype(np.float32), ri[None])
  rs = gu.rs_from_dump(ep['rs_state'])
  obs0 = ctx.observe()[0]
  assert obs0.shape == (od,) and len(ep['init_obs']) == od
  np.testing.assert_allclose(obs0[:48], ep['init_obs'][:48], rtol=0, atol=OBS_TOL)
  np.testing.assert_allclose(obs0[icols], np.array(ep['init_obs'])[icols], rtol=0, atol=1e-6)
  n_met = n_cost = 0
  for t, st in enumerate(ep['steps']):
    noise = rs.normal(size=nu)
    tape = gu.rs_words(gu.rs_copy(rs), 4096)
    rf, ri = ctx.get_state()
    rf = rf[0].astype(np.float64)
    # poses only: the task state (goal, last distances, button state, timers) stays the device's own
    gu.set_poses(rf, names, st['pos'], yaw=st['robot_yaw'], v0=st['robot_v0'], wz=st["sensors"]["gyro"][2])
    gu.set_robot_planar(rf, robot, st['robot_yaw'], st['sensors']['gyro'][2])
    ctx.set_state(rf[None].astype(np.float32), ri)
    cc, mask = gu.contact_inputs(robot, st['contacts'])
    ctx.set_ext_contacts([cc], [mask])
    obs, rew, cost, done, met, used = ctx.step(np.array([st['action']], np.float32), noise[None], tape[None], nstep=0)
    gu.rs_words(rs, int(used[0]))
    assert gu.rs_probe(rs) == st['rs_probe'], f'RNG position diverged at step {t}'
    nr = len(st['reward'])
    np.testing.assert_allclose(rew[0, :nr], st['reward'], rtol=0, atol=3e-6, err_msg=f'reward step {t}')
    assert int(cost[0]) == int(st['cost']), f'cost step {t}'
    assert int(done[0]) == int(st['done'])
    np.testing.assert_allclose(obs[0, :48], st['obs'][:48], rtol=0, atol=OBS_TOL, err_msg=f'lidar step {t}')
    np.testing.assert_allclose(obs[0, cols], np.array(st['obs'])[cols], rtol=0, atol=1e-5, err_msg=f'sensors step {t}')
    # task state after the step = the reference's
    rf2, ri2 = ctx.get_state()
    ts = st['task_state']
    exp_f, exp_i = rf2[0].astype(np.float64).copy(), ri2[0].copy()
    gu.set_task_state(exp_f, exp_i, ts)
    np.testing.assert_array_equal(ri2[0], exp_i, err_msg=f'task ints step {t}')
    np.testing.assert_allclose(rf2[0], exp_f, rtol=0, atol=2e-6, err_msg=f'task floats step {t}')
    if 'goal' in names:
      g = st['pos'][names.index('goal')]
      np.testing.assert_allclose(rf2[0, F_GOAL:F_GOAL + 2], g[:2], rtol=0, atol=1e-6, err_msg=f'goal step {t}')
    n_met += int(met[0]); n_cost += int(cost[0])
  assert n_met > 0, 'the fixture should exercise goal-met events'
  ctx.close()


# ----------------------------------------------------------------------------------
# API behaviour
# ----------------------------------------------------------------------------------
def test_state_roundtrip_reset_and_errors(nat):
  n = 130  # not a multiple of the wavefront
  rf, ri = bu.sample_records('point', 'go_to_goal', n)
  ctx = nat.Context('point', n)
  with pytest.raises(nat.SagError, match='before sag_set_layout'):
    ctx.step(np.zeros((n, 2), np.float32))
  ctx.set_layout(rf, ri)
  g_rf, g_ri = ctx.get_state()
  exp = rf.copy()
  exp[:, 38] = np.hypot(rf[:, 0] - rf[:, 32], rf[:, 1] - rf[:, 33])  # task.reset installs `last`
  np.testing.assert_allclose(g_rf, exp, rtol=0, atol=1e-6)
  np.testing.assert_array_equal(g_ri, ri)
  a = np.random.RandomState(0).uniform(-1, 1, (n, 2)).astype(np.float32)
  out1 = [ctx.step(a) for _ in range(5)]
  ctx.reset()
  r_rf, r_ri = ctx.get_state()
  np.testing.assert_array_equal(r_rf, g_rf)
  # sag_reset starts a NEW episode of the same layout: only the episode nonce of the device generator moves
  # (the reference reseeds its RandomState on every reset, safe_adaptation_gym.py:97-101)
  e_ri = g_ri.copy(); e_ri[:, nat.I_EPISODE] += 1
  np.testing.assert_array_equal(r_ri, e_ri)
  out2 = [ctx.step(a) for _ in range(5)]
  # ... so the action noise of episode 1 differs from episode 0 in (nearly) every env
  assert (np.abs(out2[0][0][:, 48:] - out1[0][0][:, 48:]).max(1) > 0).mean() > 0.8
  # a checkpoint restore (sag_set_state) replays bit for bit: counter-based generator, same counter
  ctx.set_state(g_rf, g_ri)
  out3 = [ctx.step(a) for _ in range(5)]
  for x, y in zip(out1, out3):
    for u, v in zip(x[:5], y[:5]):
      np.testing.assert_array_equal(u, v)
  # another key (env.seed()): other noise from the same state
  ctx.set_state(g_rf, g_ri)
  ctx.set_seed(12345)
  out4 = ctx.step(a)
  assert (np.abs(out4[0][:, 48:] - out1[0][0][:, 48:]).max(1) > 0).mean() > 0.8
  # partial reset by env id: those envs move to the next episode, the others are untouched
  before = ctx.get_state()
  some = np.array([7, 128, 0], np.int32)
  ctx.reset(some)
  after = ctx.get_state()
  rest = np.setdiff1d(np.arange(n), some)
  np.testing.assert_array_equal(after[0][rest], before[0][rest])
  np.testing.assert_array_equal(after[1][rest], before[1][rest])
  np.testing.assert_array_equal(after[0][some], g_rf[some])
  np.testing.assert_array_equal(after[1][some][:, nat.I_EPISODE], 2)
  # subset set_state / get_state by env id
  ids = np.array([5, 129, 64], np.int32)
  s_rf, s_ri = ctx.get_state(ids)
  s_rf[:, 0] += 1.0
  ctx.set_state(s_rf, s_ri, ids)
  np.testing.assert_array_equal(ctx.get_state(ids)[0], s_rf)
  with pytest.raises(nat.SagError):
    ctx.get_state(np.array([n], np.int32))
  bad = ri.copy(); bad[0, 2] = 11
  with pytest.raises(nat.SagError, match='capacit'):
    ctx.set_layout(rf, bad)
  ctx.close()
  with pytest.raises(nat.SagError):
    nat.Context('point', 4, device=99)


def test_box_spawned_over_an_obstacle_is_flagged_and_pushed_clear(nat, oracle):
  """HaulBox places its box at robot + .6 with no keep-out check (haul_box.py:17-18), so it can sit
  inside a vase or a pillar.  sag_set_layout flags the overlapping free bodies awake (SAG_I_AWAKE, bounding circles), the
  first substep pushes them apart as MuJoCo would, and device and oracle agree step by step from the installed state;
  sag_reset restores the flags with the layout; sag_set_state takes them as given (0 = asleep: the body stays)."""
  n = 6
  rf, ri = bu.sample_records('point', 'haul_box', n, seed=11)
  for e in range(n):
    rf[e, 65:69] = 50.0; ri[e, 3] = 1   # one pillar (r .2) ...
    ri[e, 2] = 1                        # ... and one vase
    rf[e, 81:87] = 0
    box = rf[e, 41:43]
    if e < 3:
      rf[e, 65:67] = box + [0.25, 0.0]    # the pillar 5 cm inside the box's main geom
      rf[e, 81:83] = box + [50.0, 0.0]
    else:
      rf[e, 81:83] = box + [0.0, 0.28]    # the vase (half .1) 2 cm inside it
  ctx = nat.Context('point', n, seed=3)
  ctx.set_layout(rf, ri)
  rf0, ri0 = ctx.get_state()
  assert (ri0[:3, 15] == 1 << 10).all() and (ri0[3:, 15] == (1 << 10 | 1)).all(), ri0[:, 15]
  zero = np.zeros((n, 2), np.float32)
  tape = np.zeros((n, 64), np.uint32)
  for t in range(12):
    s_rf, s_ri = ctx.get_state()
    arr = oracle.make_batch(s_rf, s_ri)
    ctx.step(zero, zero, tape)
    oracle.step_batch_full(arr, 0, zero, zero, tape)
    d_rf, d_ri = ctx.get_state()
    o_rf, o_ri = oracle.batch_records(arr)
    np.testing.assert_allclose(d_rf, o_rf, rtol=STATE_TOL, atol=STATE_TOL, err_msg=f'step {t}')
    np.testing.assert_array_equal(d_ri, o_ri)
    assert (d_ri[:, 15] == 0).all(), 'the flags last for the first substep'
  moved = np.hypot(*(d_rf[:3, 41:43] - rf0[:3, 41:43]).T)
  assert (moved > 0.02).all(), f'a box is pushed off the pillar it was spawned in: {moved}'
  moved = np.hypot(*(d_rf[3:, 81:83] - rf0[3:, 81:83]).T)
  assert (moved > 0.004).all(), f'box and vase are pushed apart (the vase, 8 mg against 96, is the one that moves): {moved}'
  ctx.reset()
  assert (ctx.get_state()[1][:, 15] == ri0[:, 15]).all(), 'sag_reset restores the install-time flags'
  asleep = ri0.copy(); asleep[:, 15] = 0
  ctx.set_state(rf0, asleep)
  for _ in range(5):
    ctx.step(zero, zero, tape)
  np.testing.assert_array_equal(ctx.get_state()[0][:, 41:44], rf0[:, 41:44])   # a record restored as asleep stays asleep
  ctx.close()


def test_ext_contacts_do_not_outlive_the_state_they_were_given_for(nat):
  """contact results supplied with sag_set_ext_contacts are for the NEXT step of THAT state: sag_set_state,
  sag_set_layout and sag_reset drop them."""
  n = 4
  rf, ri = bu.sample_records('point', 'go_to_goal', n, seed=5)
  rf[:, 47:65] = 50.0   # hazards far away: no geometric cost
  ctx = nat.Context('point', n, seed=3)
  ctx.set_layout(rf, ri)
  zero = np.zeros((n, 2), np.float32)
  tape = np.zeros((n, 64), np.uint32)
  ctx.set_ext_contacts([1] * n, [0] * n)
  assert ctx.step(zero, zero, tape, nstep=0)[2].all(), 'the supplied contact count raises the cost'
  for drop in (lambda: ctx.set_state(*ctx.get_state()), lambda: ctx.set_layout(rf, ri), lambda: ctx.reset()):
    ctx.set_ext_contacts([1] * n, [0] * n)
    drop()
    assert not ctx.step(zero, zero, tape, nstep=0)[2].any()
  ctx.close()


def test_physics_error_is_data(nat):
  """Non-finite state -> (reward -10, done, cost 0) for that env only (safe_adaptation_gym.py:73-75)."""
  n = 70
  rf, ri = bu.sample_records('point', 'go_to_goal', n)
  ctx = nat.Context('point', n)
  ctx.set_layout(rf, ri)
  s_rf, s_ri = ctx.get_state()
  s_rf[3, 3] = np.nan
  s_rf[66, 0] = np.inf
  ctx.set_state(s_rf, s_ri)
  obs, rew, cost, done, met, _ = ctx.step(np.zeros((n, 2), np.float32))
  assert done[3] == 1 and done[66] == 1 and done.sum() == 2
  assert rew[3, 0] == -10 and rew[66, 0] == -10 and cost[3] == 0
  ok = np.ones(n, bool); ok[[3, 66]] = False
  assert np.isfinite(obs[ok]).all() and np.isfinite(rew[ok]).all()
  ctx.close()


def test_env_api_shapes_and_reference_surface(nat):
  import safe_adaptation_gym_amd as sag
  env = sag.make('point', 'go_to_goal', seed=666, n_envs=100)
  assert env.action_space.shape == (2,) and env.observation_space.shape == (60,)
  obs = env.reset()
  assert obs.shape == (100, 60) and obs.dtype == np.float32
  a = np.stack([env.action_space.sample() for _ in range(100)])
  obs, reward, done, info = env.step(a)
  assert obs.shape == (100, 60) and reward.shape == (100,) and done.shape == (100,)
  assert set(info) >= {'cost', 'bound'} and (info['bound'] == 25).all()
  assert np.all((obs[:, :48] >= 0) & (obs[:, :48] <= 1))
  env.set_task(sag.tasks.Unsupervised)
  obs, reward, done, info = env.step(a)
  assert reward.shape == (100, 2)
  from safe_adaptation_gym_amd import benchmark
  bm = benchmark.make('multitask', batch_size=3, seed=1)
  for name, task in bm.train_tasks:
    obs = env.reset(options={'task': task})
    assert obs.shape == (100, 60)
  # the tasks/* plugin surface: a subclass that overrides `obstacles` changes the worlds the env builds
  class Sparse(sag.tasks.GoToGoal):
    @property
    def obstacles(self):
      return [3, 3, 0, 0]
  env.set_task(Sparse)
  rf, ri = env.get_state()
  assert (ri[:, nat.I_NH] == 3).all() and (ri[:, nat.I_NV] == 3).all() and (ri[:, nat.I_NP] == 0).all()
  obs, reward, done, info = env.step(a)
  assert np.isfinite(obs).all() and (obs[:, 16:32] == 0).all(), 'no task object: the middle lidar stays empty'
  class OwnReward(sag.tasks.GoToGoal):
    def compute_reward(self, *args):
      return 0.
  with pytest.raises(NotImplementedError):
    env.set_task(OwnReward)
  env.close()


def test_early_fork_equals_single_launch(nat, monkeypatch):
  """Large batches start the quiet kernel before the compaction that feeds the busy kernel (sag_api.hip,
  `early_fork`: k_compact and k_step_quiet then run side by side on the tstate words).  Forced on here at a
  batch of a few hundred wavefronts: results must still equal the single launch bit for bit."""
  n, T = 20011, 70
  rf, ri = bu.sample_records_native('point', 'go_to_goal', n, seed=4100)
  ctxs = []
  for split, fork in (('0', '0'), ('1', '1')):
    monkeypatch.setenv('SAG_SPLIT', split)
    monkeypatch.setenv('SAG_EARLY_FORK', fork)
    c = nat.Context('point', n, seed=78)
    c.set_layout(rf, ri)
    ctxs.append(c)
  rng = np.random.RandomState(4)
  cost_events = 0
  for t in range(T):
    s_rf, s_ri = ctxs[0].get_state()
    act = bu.pursuit_actions(s_rf, s_ri, rng, robot='point')
    outs = [c.step(act) for c in ctxs]
    for a, b in zip(outs[0][:5], outs[1][:5]):
      np.testing.assert_array_equal(a, b, err_msg=f'step {t}')
    cost_events += int(outs[0][2].sum())
  sa, sb = ctxs[0].get_state(), ctxs[1].get_state()
  np.testing.assert_array_equal(sa[0], sb[0])
  np.testing.assert_array_equal(sa[1], sb[1])
  assert cost_events > 0
  for c in ctxs:
    c.close()


@pytest.mark.parametrize('robot,task', [('point', 'go_to_goal'), ('point', MIXED), ('car', 'push_box'),
                                        ('point', 'haul_box')])
def test_split_launch_equals_single_launch(nat, monkeypatch, robot, task):
  """QUIET + BUSY launches (envs classified at the end of the previous step) must reproduce the
  single full-physics launch bit for bit: state, observations, rewards, flags, over a rollout
  long enough for goals, contacts and resting bodies to occur."""
  n, T = 1500, 120   # not a multiple of 64 nor of the 256-env neighbourhood
  if task == MIXED:
    from safe_adaptation_gym_amd import benchmark
    names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=5).train_tasks]
    rf, ri = bu.sample_records_native(robot, names, n, seed=4000)
  else:
    rf, ri = bu.sample_records_native(robot, task, n, seed=4000)
  rf = bu.goal_beyond_box(rf, ri) if task != 'haul_box' else rf
  ctxs = []
  # single launch; split with one busy list (what a batch this small gets); split with the busy list kept by KIND of contact (what
  # batches get whose busy launch needs more than one round of resident wavefronts: k_compact) - the order of the list and the
  # grouping of envs into wavefronts must not show in any result
  for flag, kinds_min in (('0', None), ('1', None), ('1', '0')):
    monkeypatch.setenv('SAG_SPLIT', flag)
    if kinds_min is None:
      monkeypatch.delenv('SAG_BUSY_KINDS_MIN', raising=False)
      monkeypatch.delenv('SAG_BUSY_KINDS', raising=False)
    else:
      monkeypatch.setenv('SAG_BUSY_KINDS_MIN', kinds_min)
      monkeypatch.setenv('SAG_BUSY_KINDS', '1')     # (the default keeps kinds for the Car only)
    c = nat.Context(robot, n, seed=77)
    c.set_layout(rf, ri)
    ctxs.append(c)
  monkeypatch.delenv('SAG_BUSY_KINDS_MIN', raising=False)
  monkeypatch.delenv('SAG_BUSY_KINDS', raising=False)
  rng = np.random.RandomState(3)
  busy_share = []
  for t in range(T):
    s_rf, s_ri = ctxs[0].get_state()
    act = bu.pursuit_actions(s_rf, s_ri, rng, robot=robot)
    outs = [c.step(act) for c in ctxs]          # counter-based noise: same key, env id, step
    sa = ctxs[0].get_state()
    for other, c in zip(outs[1:], ctxs[1:]):
      for a, b in zip(outs[0][:5], other[:5]):
        np.testing.assert_array_equal(a, b, err_msg=f'step {t}')
      sb = c.get_state()
      np.testing.assert_array_equal(sa[0], sb[0], err_msg=f'state step {t}')
      np.testing.assert_array_equal(sa[1], sb[1])
    assert ctxs[1].busy_count() == ctxs[2].busy_count()
    if t in (40, 41, 80):
      # state installed from outside between steps (partial reset to the installed layout, then a
      # set_state of a few envs): the busy lists and hot records of the split form must follow
      ids = rng.choice(n, size=97, replace=False).astype(np.int32)
      for c in ctxs:
        c.reset(ids)
      some = np.sort(rng.choice(n, size=33, replace=False)).astype(np.int32)
      p_rf, p_ri = ctxs[0].get_state(some)
      p_rf[:, 0] += 0.01
      for c in ctxs:
        c.set_state(p_rf, p_ri, some)
  assert outs[0][2].sum() > 0
  for c in ctxs:
    c.close()


def test_busy_lists_by_kind_at_scale_equal_single_launch(nat, monkeypatch):
  """The path large Car batches take (busy lists by kind of contact, hundreds of compaction blocks, thousands of busy wavefronts per
  kind, the gate on the busy count of the step before) against the single launch, bit for bit: 600 000 Car / push_box envs, pursuit
  actions for 24 steps.  The gate is lowered to 8192 busy envs (production: 64 per resident slot = 131 072, i.e. batches beyond ~1.5 M
  envs) so that it opens during the rollout as it does in production: the first steps run on one list, the later ones on eight."""
  if os.environ.get('SAG_HOSTEMU'):
    pytest.skip('600 000 envs: a GPU-sized case (the host emulator runs the same lists at 1500 envs)')
  n, T = 600_000, 24
  from safe_adaptation_gym_amd import benchmark
  tid = benchmark.TASKS['push_box'].TASK_ID
  rf, ri, st = nat.sample_layouts('car', 7000 + np.arange(n, dtype=np.int64), tid, nthreads=min(16, len(os.sched_getaffinity(0))))
  assert not st.any()
  rf = bu.goal_beyond_box(rf, ri)
  ctxs = []
  for split, kmin in (('0', None), ('1', '8192')):
    monkeypatch.setenv('SAG_SPLIT', split)
    if kmin is None:
      monkeypatch.delenv('SAG_BUSY_KINDS_MIN', raising=False)
    else:
      monkeypatch.setenv('SAG_BUSY_KINDS_MIN', kmin)
    c = nat.Context('car', n, seed=71)
    c.set_layout(rf, ri)
    ctxs.append(c)
  monkeypatch.delenv('SAG_BUSY_KINDS_MIN', raising=False)
  rng = np.random.RandomState(5)
  busy = []
  for t in range(T):
    s_rf, s_ri = ctxs[0].get_state()
    act = bu.pursuit_actions(s_rf, s_ri, rng, robot='car')
    outs = [c.step(act) for c in ctxs]
    for a, b in zip(outs[0][:5], outs[1][:5]):
      np.testing.assert_array_equal(a, b, err_msg=f'step {t}')
    busy.append(ctxs[1].busy_count())
  sb = ctxs[1].get_state()
  s_rf, s_ri = ctxs[0].get_state()
  np.testing.assert_array_equal(s_rf, sb[0])
  np.testing.assert_array_equal(s_ri, sb[1])
  assert busy[0] == n, 'after an install every env is busy'
  assert sum(b > 8192 for b in busy[1:]) >= 8 and sum(0 < b <= 8192 for b in busy[1:]) >= 3, \
      f'the rollout must run on one list first and on the lists by kind later: busy envs per step {busy}'
  assert outs[0][2].sum() > 0
  for c in ctxs:
    c.close()


def test_device_buffers_env_api(nat):
  """make(..., device_buffers=True) - the reference's make / reset / step surface with the results left
  in HBM (DeviceArray views, __cuda_array_interface__) and actions taken from HBM: the same numbers as the NumPy path
  bit for bit, sharded or not, and a step costs what its kernels cost (no PCIe round trip, no host synchronisation)."""
  import time
  import safe_adaptation_gym_amd as sag
  n = 4096
  host = sag.make('point', 'go_to_goal', seed=3, n_envs=n)
  dev = sag.make('point', 'go_to_goal', seed=3, n_envs=n, device_buffers=True)
  # (the host emulation of the device sources runs one kernel at a time: no shard threads there)
  sh = sag.make('point', 'go_to_goal', seed=3, n_envs=n, device_buffers=True, devices=[0] if os.environ.get('SAG_HOSTEMU') else [0, 0, 0])
  o_h, o_d, o_s = host.reset(), dev.reset(), sh.reset()
  if not isinstance(o_s, list):
    o_s = [o_s]
  assert isinstance(o_d, nat.DeviceArray) and o_d.shape == (n, 60) and o_d.__cuda_array_interface__['data'][0] == o_d.ptr
  np.testing.assert_array_equal(o_h, o_d.numpy())
  np.testing.assert_array_equal(o_h, np.concatenate([x.numpy() for x in o_s]))
  rng = np.random.RandomState(0)
  c = dev._ctx[0]
  d_act = c.dev_alloc(n * 2 * 4)
  for t in range(12):
    a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    h = host.step(a)
    if t % 2:   # actions from HBM (what a policy on the GPU hands over) ...
      c.dev_upload(d_act, a)
      d = dev.step(nat.DeviceArray(c, d_act.value, (n, 2), np.float32))
    else:       # ... or from the host
      d = dev.step(a)
    s = sh.step(a)
    if not isinstance(s[0], list):
      s = ([s[0]], [s[1]], [s[2]], {k: ([v] if k != 'bound' else v) for k, v in s[3].items()})
    for key, hv, dv, sv in [('obs', h[0], d[0], s[0]), ('reward', h[1], d[1], s[1]), ('done', h[2], d[2], s[2]),
                            ('cost', h[3]['cost'], d[3]['cost'], s[3]['cost']), ('goal_met', h[3]['goal_met'], d[3]['goal_met'], s[3]['goal_met'])]:
      np.testing.assert_array_equal(hv.astype(np.float32), dv.numpy().astype(np.float32), err_msg=f'{key} step {t}')
      np.testing.assert_array_equal(hv.astype(np.float32), np.concatenate([x.numpy() for x in sv]).astype(np.float32), err_msg=f'{key} (shards) step {t}')
    assert d[1].shape == (n,) and d[1].strides == (8,) and d[3]['bound'].shape == (n,)
  # a step through the reference's API costs what its kernels cost
  act = nat.DeviceArray(c, d_act.value, (n, 2), np.float32)
  for _ in range(50):
    dev.step(act, sync=False)
  dev.wait()
  c.enable_timing(True); c.kernel_time_ms(reset=True)
  K = 400
  t0 = time.perf_counter()
  for _ in range(K):
    dev.step(act, sync=False)
  dev.wait()
  wall_ms = (time.perf_counter() - t0) / K * 1e3
  kernel_ms = c.kernel_time_ms(reset=True)[0]
  c.enable_timing(False)
  t0 = time.perf_counter()
  for _ in range(100):
    host.step(a)
  host_ms = (time.perf_counter() - t0) / 100 * 1e3
  print(f'env.step at {n} envs: device buffers {wall_ms:.4f} ms per step (kernels {kernel_ms:.4f} ms), NumPy path {host_ms:.4f} ms')
  if not os.environ.get('SAG_HOSTEMU'):
    assert wall_ms <= 1.2 * kernel_ms + 0.005, f'{wall_ms:.4f} ms per step against {kernel_ms:.4f} ms of kernels'
  c.dev_free(d_act)
  for e in (host, dev, sh):
    e.close()


def test_car_env_api(nat):
  """BASELINE config 3 through the front end: Car / push_box, obs 72 (48 lidar + 12 + 3 + 9)."""
  import safe_adaptation_gym_amd as sag
  env = sag.make('car', 'push_box', seed=666, n_envs=70)
  assert env.observation_space.shape == (72,) and env.action_space.shape == (2,)
  obs = env.reset()
  assert obs.shape == (70, 72)
  np.testing.assert_allclose(obs[:, 63:72].reshape(-1, 3, 3), np.broadcast_to(np.eye(3), (70, 3, 3)), atol=1e-7)
  for _ in range(20):
    obs, reward, done, info = env.step(np.full((70, 2), 0.02, np.float32))
  assert np.isfinite(obs).all() and not done.any() and reward.shape == (70,)
  R = obs[:, 63:72].reshape(-1, 3, 3)
  np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), (70, 3, 3)), atol=1e-5)
  assert (np.abs(obs[:, 60:63]).max(1) > 0.1).mean() > 0.9, 'the rear ball spins once the car moves'
  env.close()


# ----------------------------------------------------------------------------------
# Doggo (3-D articulated; fp64 robot solve on the device, fp32 planar world)
# ----------------------------------------------------------------------------------
DOGGO_TASKS = ['go_to_goal', 'push_box', 'press_buttons', 'haul_box', 'unsupervised', 'collect', MIXED]


DOGGO_C4 = 'multitask-4096'   # BASELINE config 4 at one rank's size


@pytest.mark.parametrize('task', DOGGO_TASKS + [DOGGO_C4])
def test_doggo_lockstep_vs_oracle(nat, oracle, task):
  """Same protocol as test_step_lockstep_vs_oracle for the Doggo robot (BASELINE config 4 shape
  for 'multitask').  Stated tolerance after one step (12 substeps) from identical fp32 state:
  robot qpos within 2e-5, qvel within 2e-3 abs + 2e-3 rel of the fp64 oracle (the device solves the
  robot in fp64 but stores fp32 and keeps the planar world in fp32; 12 substeps of stiff
  soft-contact dynamics amplify that rounding), free bodies within the planar STATE_TOL;
  threshold events (a sphere's penetration changing sign with the rounding) are counted and
  bounded.  goal_met / done / task ints / RNG words exact on rows inside the tolerance.
  `multitask-4096` is BASELINE config 4 at its stated size per rank (4096 envs, ~290 per task, 14 steps: VERDICT r3
  item 2), the oracle on every host core.  Cost flags that differ away from any threshold are counted over the WHOLE
  run (round 3 allowed one per step) and budgeted at 2 - 3 x the measured count (profiles/r04_lockstep_counts.txt)."""
  n, T = (4096, 14) if task == DOGGO_C4 else (64, 30)
  from safe_adaptation_gym_amd import benchmark
  if task == DOGGO_C4:
    oracle.lib.sago_set_threads(os.cpu_count() or 1)
  if task in (MIXED, DOGGO_C4):
    names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=666).train_tasks]
  else:
    names = task
  rf, ri = bu.sample_records_native('doggo', names, n, seed=666)
  ctx = nat.Context('doggo', n, seed=4321)
  ctx.set_layout(rf, ri)
  mt = np.random.RandomState(5)
  obs0 = ctx.observe()
  rf, ri = ctx.get_state()
  np.testing.assert_allclose(obs0, oracle.observe_batch(oracle.make_batch(rf, ri), 2, 104), rtol=0, atol=OBS_TOL)
  E = 144
  pos_f = [0, 1, 2, E] + list(range(E + 1, E + 5)) + list(range(E + 9, E + 22))
  viol = n_rows = 0
  touched = flag_mism = flag_near = overflow = 0
  for t in range(T):
    rf, ri = ctx.get_state()
    arr = oracle.make_batch(rf, ri)
    act = mt.uniform(-1, 1, size=(n, 12)).astype(np.float32)
    if t < 3:
      act[:] = 0   # let the robots land first
    noise = mt.normal(size=(n, 12)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
    d_obs, d_rew, d_cost, d_done, d_met, d_used = ctx.step(act, noise, tape)
    o_obs, o_rew, o_cost, o_done, o_met, o_used, o_margin = oracle.step_batch_full(arr, 2, act, noise, tape, obs_dim=104)
    d_rf, d_ri = ctx.get_state()
    o_rf, o_ri = oracle.batch_records(arr)
    tol = np.full(d_rf.shape[1], 2e-3)
    tol[pos_f] = 2e-5
    for k in range(10):
      tol[81 + 6 * k:81 + 6 * k + 6] = STATE_TOL
      tol[81 + 6 * k + 5] = 2e-4
    tol[41:47] = STATE_TOL; tol[46] = 5e-3
    bad = (np.abs(d_rf - o_rf) > tol + tol * np.abs(o_rf)).any(1)
    viol += int(bad.sum()); n_rows += n
    ok = ~bad
    np.testing.assert_array_equal(d_done, o_done)
    assert not d_done.any()
    np.testing.assert_array_equal(d_met[ok], o_met[ok])
    np.testing.assert_array_equal(d_used[ok], o_used[ok])
    np.testing.assert_array_equal(d_ri[ok], o_ri[ok])
    flag_mism += int((ok & (d_cost != o_cost) & (o_margin > 1e-5)).sum())
    flag_near += int((ok & (d_cost != o_cost) & (o_margin <= 1e-5)).sum())
    overflow += int((d_ri[:, 13] & 4 != 0).sum())
    # observation as a function of the device's own post-step state (lidar, kinematic sensors)
    f_obs = oracle.observe_batch(oracle.make_batch(d_rf, d_ri), 2, 104)
    np.testing.assert_allclose(d_obs[:, :48], f_obs[:, :48], rtol=0, atol=OBS_TOL, err_msg=f'lidar step {t}')
    np.testing.assert_allclose(d_obs[:, 51:60], f_obs[:, 51:60], rtol=0, atol=2e-5, err_msg=f'vel/gyro/mag step {t}')
    np.testing.assert_allclose(d_obs[:, 68:], f_obs[:, 68:], rtol=0, atol=2e-5, err_msg=f'joint sensors step {t}')
    # end to end (accelerometer and touch come from the stiff contact solve: relative tolerance)
    np.testing.assert_allclose(d_obs[ok, :48], o_obs[ok, :48], rtol=0, atol=1e-3)
    np.testing.assert_allclose(d_obs[ok, 60:68], o_obs[ok, 60:68], rtol=2e-2, atol=2e-3, err_msg=f'touch step {t}')
    np.testing.assert_allclose(d_rew[ok], o_rew[ok], rtol=0, atol=2e-4, err_msg=f'reward step {t}')
    touched += int((d_obs[:, 60:68] > 0).any(1).sum())
  # cost flags differing away from any threshold, over the whole run: a contact whose penetration changes sign with the
  # rounding (fp32 planar world on the device, fp64 in the oracle).  Measured (profiles/r04_lockstep_counts.txt): 0 of
  # 57 344 env-steps at config 4's size, 0 - 1 of 1920 in the small cases; budget 1e-4 of the env-steps, at least 2.
  # State: 10 of 57 344 outside the tolerance at config 4's size (budget 5e-4), 0 - 3 of 1920 in the small cases (0.5 %).
  flag_budget = max(2, int(1e-4 * n_rows))
  viol_frac = 0.0005 if task == DOGGO_C4 else 0.005
  _log_lockstep(f'doggo/{task}: {n} envs x {T} steps resynchronised every step | env-steps outside the stated tolerance vs fp64 oracle {viol} '
                f'({viol / n_rows:.2e}), budget {viol_frac:.1e} | cost flags differing away from a threshold {flag_mism} ({flag_mism / n_rows:.2e}; budget {flag_budget}), '
                f'within 1e-5 of one {flag_near} | env-steps with floor touch {touched} | env-steps with a row overflow {overflow}')
  assert touched > 0.8 * n * (T - 3), 'the robots should stand on the floor'
  assert viol <= viol_frac * n_rows, f'{viol} of {n_rows} env-steps outside the stated tolerance'
  assert flag_mism <= flag_budget, f'{flag_mism} cost flags differ away from a threshold'
  if task == DOGGO_C4:
    oracle.lib.sago_set_threads(1)
  ctx.close()


def test_doggo_capsule_and_cylinder_contacts_on_device(nat, oracle):
  """VERDICT r3 item 1 on the device: the analytic contact cases of tests/test_oracle_doggo.py (a vase corner against the
  SHAFT of ankle_1 - which no end sphere reaches -, a pillar against two shafts, a vase face against the torso
  cylinders, the flat end cap, the merged knee rows) evaluated by k_doggo_physics on the same states: cost flags as
  stated, accelerometer and touch columns equal to the oracle's (the forward evaluation behind an observation is solved
  cold on both sides, so sag_observe and an nstep = 0 step agree too)."""
  from test_oracle_doggo import contact_scenarios
  sc = contact_scenarios()
  rf = np.stack([v[0] for v in sc.values()]); ri = np.stack([v[1] for v in sc.values()])
  want = np.array([v[2] for v in sc.values()], np.uint8)
  n = len(sc)
  ctx = nat.Context('doggo', n, seed=1)
  ctx.set_layout(rf.astype(np.float32), ri)
  rf32, ri32 = ctx.get_state()
  arr = oracle.make_batch(rf32, ri32)
  zero = np.zeros((n, 12), np.float32)
  tape = np.zeros((n, 64), np.uint32)
  d_obs, _, d_cost, d_done, _, _ = ctx.step(zero, zero, tape, nstep=0)
  o_obs, _, o_cost, _, _, _, _ = oracle.step_batch_full(arr, 2, zero, zero, tape, obs_dim=104, nstep=0)
  np.testing.assert_array_equal(d_cost, want)
  np.testing.assert_array_equal(o_cost, want)
  assert not d_done.any()
  np.testing.assert_allclose(d_obs[:, 48:51], o_obs[:, 48:51], rtol=2e-3, atol=2e-3, err_msg='accelerometer')
  np.testing.assert_allclose(d_obs[:, 60:68], o_obs[:, 60:68], rtol=2e-3, atol=1e-4, err_msg='touch')
  lying = list(sc).index('lying on its left side on the knees of legs 1 and 2')
  assert d_obs[lying, 60] > 0.01 and d_obs[lying, 61] > 0.01 and not d_obs[lying, 62:68].any()   # half of each merged knee row
  np.testing.assert_allclose(ctx.observe()[:, 48:68], d_obs[:, 48:68], rtol=0, atol=1e-6)      # observe == step(nstep = 0)
  ctx.close()


def test_doggo_random_contact_geometry_on_device(nat, oracle):
  """The capsule / cylinder narrowphase on configurations no rollout visits: 2048 worlds of the multitask mix, the robot
  in a random pose (height .12 - .30, tilted up to .5 rad, every joint anywhere inside its range) and the world's vases,
  pillars, buttons and task object scattered over the metre around it, at any yaw - shafts against corners, cylinders
  against faces, axes through boxes.  One forward evaluation (a step without substeps) on the device and in the oracle
  from the same state: cost flags equal except where a penetration changes sign with the rounding (counted; the planar
  narrowphase is fp32 on the device, fp64 in the oracle), accelerometer and touch within the solver's tolerance on the
  envs whose rows fit the device's budget (the others are flagged, bit 2 of SAG_I_FLAGS, and counted)."""
  from safe_adaptation_gym_amd import benchmark
  n = 256 if os.environ.get('SAG_HOSTEMU') else 2048
  names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=77).train_tasks]
  rf, ri = bu.sample_records_native('doggo', names, n, seed=77)
  rng = np.random.RandomState(3)
  E = 144
  lo = np.radians([-10, -75, -75, -10, -75, -75, -30, -10, 0, -75, -10, 0, -75])
  hi = np.radians([30, 15, 0, 30, 15, 0, 30, 30, 135, 0, 30, 135, 0])
  for e in range(n):
    yaw, tilt, ax = rng.uniform(0, 2 * np.pi), rng.uniform(0, 0.5), rng.uniform(0, 2 * np.pi)
    qy = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    qt = np.array([np.cos(tilt / 2), np.sin(tilt / 2) * np.cos(ax), np.sin(tilt / 2) * np.sin(ax), 0])
    w0, x0, y0, z0 = qy; w1, x1, y1, z1 = qt
    q = np.array([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                  w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1])
    rf[e, 2] = yaw
    rf[e, E] = rng.uniform(0.12, 0.30)
    rf[e, E + 1:E + 5] = q
    rf[e, E + 5:E + 9] = 0
    rf[e, E + 9:E + 22] = rng.uniform(lo, hi)
    rf[e, E + 22:E + 35] = 0
    base = rf[e, 0:2]

    def near(rmax):
      a, r = rng.uniform(0, 2 * np.pi), rmax * np.sqrt(rng.uniform(0.0, 1.0))   # uniform over the disc
      return base + r * np.array([np.cos(a), np.sin(a)])
    for k in range(ri[e, 2]):
      rf[e, 81 + 6 * k:81 + 6 * k + 2] = near(1.0); rf[e, 81 + 6 * k + 2] = rng.uniform(0, 2 * np.pi); rf[e, 81 + 6 * k + 3:81 + 6 * k + 6] = 0
    for k in range(ri[e, 3]):
      rf[e, 65 + 2 * k:67 + 2 * k] = near(1.2)
    for k in range(ri[e, 4]):
      rf[e, 69 + 2 * k:71 + 2 * k] = near(1.0)
    if ri[e, 5]:
      rf[e, 41:43] = near(1.0); rf[e, 43] = rng.uniform(0, 2 * np.pi); rf[e, 44:47] = 0
  ctx = nat.Context('doggo', n, seed=9)
  ctx.set_layout(rf, ri)
  ctx.set_state(rf, ri)
  s_rf, s_ri = ctx.get_state()
  arr = oracle.make_batch(s_rf, s_ri)
  zero = np.zeros((n, 12), np.float32)
  tape = np.zeros((n, 64), np.uint32)
  oracle.lib.sago_set_threads(os.cpu_count() or 1)
  d_obs, _, d_cost, d_done, _, _ = ctx.step(zero, zero, tape, nstep=0)
  o_obs, _, o_cost, o_done, _, _, _ = oracle.step_batch_full(arr, 2, zero, zero, tape, obs_dim=104, nstep=0)
  oracle.lib.sago_set_threads(1)
  overflow = (ctx.get_state()[1][:, 13] & 4) != 0
  ok = ~overflow
  flag_mism = int((d_cost[ok] != o_cost[ok]).sum())
  acc_bad = (np.abs(d_obs[:, 48:51] - o_obs[:, 48:51]) > 0.05 + 0.02 * np.abs(o_obs[:, 48:51])).any(1) & ok
  touch_bad = (np.abs(d_obs[:, 60:68] - o_obs[:, 60:68]) > 0.01 + 0.02 * np.abs(o_obs[:, 60:68])).any(1) & ok
  _log_lockstep(f'doggo/random contact geometry: {n} states, one forward evaluation | cost rate device {d_cost.mean():.3f} oracle {o_cost.mean():.3f} | '
                f'flags differing {flag_mism} | accelerometer outside 2 % {int(acc_bad.sum())}, touch outside 2 % {int(touch_bad.sum())} | '
                f'states beyond the row budget (flagged, not compared) {int(overflow.sum())}')
  assert 0.2 < o_cost.mean() < 0.9, 'the scatter should produce both touching and free robots'
  np.testing.assert_array_equal(d_done, o_done)
  assert flag_mism <= max(2, int(0.003 * n)), f'{flag_mism} cost flags differ'
  assert acc_bad.sum() <= max(2, int(0.01 * n)) and touch_bad.sum() <= max(2, int(0.01 * n))
  assert overflow.mean() < 0.25
  ctx.close()


def test_doggo_launch_order_does_not_change_results(nat, monkeypatch):
  """k_doggo_physics takes the envs longest-first, pairing them differently from step to step (and, inside a cost class,
  from run to run): an env's arithmetic depends on its own rows only (the PGS path is chosen per env), so the results
  are those of the plain order bit for bit - including the haul_box envs whose rows exceed the 32-lane path."""
  n, T = (96, 12) if os.environ.get('SAG_HOSTEMU') else (1024, 25)
  rf, ri = bu.sample_records_native('doggo', 'haul_box', n, seed=666)
  rng = np.random.RandomState(8)
  acts = rng.uniform(-1, 1, (T, n, 12)).astype(np.float32)
  out = {}
  for sched in ('1', '0'):
    monkeypatch.setenv('SAG_DOGGO_SCHED', sched)
    ctx = nat.Context('doggo', n, seed=77, has_box=True)
    ctx.set_layout(rf, ri)
    obs = []
    for t in range(T):
      obs.append(ctx.step(acts[t], None, None)[0].copy())
    out[sched] = (np.stack(obs), *ctx.get_state())
    ctx.close()
  heavy = int(((out['1'][2][:, 13] & 4) != 0).sum())
  np.testing.assert_array_equal(out['1'][0], out['0'][0])
  np.testing.assert_array_equal(out['1'][1], out['0'][1])
  np.testing.assert_array_equal(out['1'][2], out['0'][2])
  print(f'{n} haul_box envs x {T} steps: longest-first == plain order bit for bit ({heavy} envs beyond the row budget at the end)')


def test_doggo_env_api(nat):
  """make('doggo', ...): obs 104, 12 actions; zero action lets the robot settle on its feet."""
  import safe_adaptation_gym_amd as sag
  env = sag.make('doggo', 'go_to_goal', seed=3, n_envs=96)
  obs = env.reset()
  assert obs.shape == (96, 104) and env.action_space.shape == (12,)
  for _ in range(40):
    obs, reward, done, info = env.step(np.zeros((96, 12), np.float32))
  assert np.isfinite(obs).all() and not done.any()
  touch = obs[:, 60:68]
  assert (touch.sum(1) > 0.3).mean() > 0.95      # m g = 0.39 N carried by the feet
  assert (np.abs(obs[:, 68:80]).max(1) < 0.5).mean() > 0.9   # joint rates have died down
  env.close()


def test_parity_rng_mode_matches_reference_draw_order(nat, oracle):
  """make(parity_rng=True): noise and in-step draws come from per-env RandomState in the
  reference's order; stepping the oracle with the same generators gives the same result."""
  import safe_adaptation_gym_amd as sag
  n = 16
  env = sag.make('point', 'go_to_goal', seed=666, n_envs=n, parity_rng=True)
  env.reset(seed=666)
  rf, ri = env.get_state()
  arr = oracle.make_batch(rf, ri)
  rss = [np.random.RandomState() for _ in range(n)]
  for r, e in zip(rss, env.rs):
    r.set_state(e.get_state())
  rng = np.random.RandomState(3)
  for t in range(60):
    rf, ri = env.get_state()
    act = bu.pursuit_actions(rf, ri, rng, p_random=0.05)
    obs, rew, done, info = env.step(act)
    noise = np.stack([r.normal(size=2) for r in rss]).astype(np.float32)
    tape = np.stack([gu.rs_words(gu.rs_copy(r), 256) for r in rss])
    arr = oracle.make_batch(rf, ri)
    o = oracle.step_batch_full(arr, 0, act, noise, tape)
    for r, u in zip(rss, o[5]):
      gu.rs_words(r, int(u))
    np.testing.assert_array_equal(info['goal_met'], o[4].astype(bool))
    np.testing.assert_allclose(rew, o[1][:, 0], rtol=0, atol=1e-5)
    for r, e in zip(rss, env.rs):
      assert gu.rs_probe(r) == gu.rs_probe(e)
  env.close()


# ----------------------------------------------------------------------------------
# rgb_observation: device ray caster vs the oracle's fp64 statement of the image
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize('robot,task', [('point', 'go_to_goal'), ('point', 'press_buttons'), ('point', 'push_box'),
                                        ('car', 'dribble_ball'), ('car', 'collect'), ('doggo', 'go_to_goal'),
                                        ('doggo', 'roll_rod'), ('point', 'unsupervised'),
                                        ('doggo', 'haul_box')])   # (the last one: BASELINE config 5's own pair)
def test_rgb_observation_matches_oracle(nat, oracle, robot, task):
  """Every pixel is a hard decision (surface hit, checker square, 8-bit rounding) evaluated in fp64
  on both sides: images agree exactly except where libm differences flip a decision - bounded to
  0.1 % of the pixels, never by more than the two candidate colours of an edge."""
  n = 24
  rid = {'point': 0, 'car': 1, 'doggo': 2}[robot]
  rf, ri = bu.sample_records_native(robot, task, n, seed=31)
  ctx = nat.Context(robot, n, seed=5, max_buttons=nat.MAX_BUTTONS, has_box=True)
  ctx.set_layout(rf, ri)
  rng = np.random.RandomState(1)
  for rounds in range(2):
    img = ctx.render_rgb()
    assert img.shape == (n, 64, 64, 3) and img.dtype == np.uint8
    rf, ri = ctx.get_state()
    ref = np.stack([oracle.render_rgb(oracle.env(rf[k], ri[k]), rid) for k in range(n)])
    diff = (img.astype(int) - ref.astype(int))
    bad = np.abs(diff).max(-1) > 0
    assert bad.mean() <= 1e-3, f'{bad.sum()} pixels differ'
    assert len(np.unique(img.reshape(-1, 3), axis=0)) > 20          # not a flat image
    assert (img[:, :8].astype(int).mean() != img[:, -8:].astype(int).mean())
    for _ in range(5):   # move, then render again
      ctx.step(rng.uniform(-1, 1, (n, ctx.info['nu'])).astype(np.float32))
  ctx.close()


@pytest.mark.parametrize('robot,task,camera', [('point', 'go_to_goal', 'fixedfar'), ('car', 'push_box', 'track'),
                                               ('doggo', 'press_buttons', 'fixednear'), ('point', 'collect', 'vision')])
def test_human_view_matches_oracle(nat, oracle, robot, task, camera):
  """SURVEY 8f rank 4 (render.py, safe_adaptation_gym.py:109-111,239-257): the scene's other cameras, a non-square
  image, and the overlays of render_lidars_and_collision (three lidar rings whose alpha follows the last
  observation, the cost sphere) - device ray caster against the oracle's statement of the same image."""
  n, W, H = 12, 96, 64
  rid = {'point': 0, 'car': 1, 'doggo': 2}[robot]
  cam = nat.Context.CAMERAS[camera]
  rf, ri = bu.sample_records_native(robot, task, n, seed=77)
  ctx = nat.Context(robot, n, seed=5)
  ctx.set_layout(rf, ri)
  rng = np.random.RandomState(2)
  cost_seen = 0
  for rounds in range(3):
    for _ in range(4):
      out = ctx.step(rng.uniform(-1, 1, (n, ctx.info['nu'])).astype(np.float32))
    if rounds == 1:   # put some robots onto a hazard so that the cost indicator is drawn
      s_rf, s_ri = ctx.get_state()
      s_rf[:n // 2, 0:2] = s_rf[:n // 2, nat.F_HAZARDS:nat.F_HAZARDS + 2]
      ctx.set_state(s_rf, s_ri)
      out = ctx.step(np.zeros((n, ctx.info['nu']), np.float32), nstep=0)
    obs, cost = out[0], out[2]
    cost_seen += int(cost.sum())
    img = ctx.render(camera, W, H, overlays=True)
    assert img.shape == (n, H, W, 3) and img.dtype == np.uint8
    rf2, ri2 = ctx.get_state()
    ref = np.stack([oracle.render(oracle.env(rf2[k], ri2[k]), rid, cam, W, H, True, obs[k, :48], cost[k]) for k in range(n)])
    bad = np.abs(img.astype(int) - ref.astype(int)).max(-1) > 0
    assert bad.mean() <= 1e-3, f'{bad.sum()} pixels differ'
    plain = ctx.render(camera, W, H, overlays=False)
    if camera != 'vision':   # (the rings float above the robot's own camera, out of its view)
      assert (plain != img).any(), 'the overlays must be visible'
  assert cost_seen > 0
  ctx.close()


def test_render_env_api(nat):
  """env.render() / make(render_options=..., render_lidar_and_collision=...) mirror the reference's call
  (safe_adaptation_gym/__init__.py:6-24, safe_adaptation_gym.py:109-111)."""
  import safe_adaptation_gym_amd as sag
  env = sag.make('car', 'go_to_goal', seed=3, n_envs=4, render_options=dict(camera_id='fixedfar', height=48, width=80),
                 render_lidar_and_collision=True)
  env.reset()
  env.step(np.zeros((4, 2), np.float32))
  img = env.render()
  assert img.shape == (4, 48, 80, 3) and img.dtype == np.uint8 and len(np.unique(img.reshape(-1, 3), axis=0)) > 10
  assert env.render(mode='rgb_array', camera_id='track', height=32, width=32).shape == (4, 32, 32, 3)
  with pytest.raises(KeyError):
    env.render(camera_id='nope')
  env.close()


def _shards_equal_one_context(devices, robot='point', task='go_to_goal', n=300, steps=12):
  import safe_adaptation_gym_amd as sag
  nu = gu.ROBOT_NU[robot]
  if task == 'multitask':   # heterogeneous batch: one Task object per env, in the benchmark's TaskSampler order
    from safe_adaptation_gym_amd import benchmark
    names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=666).train_tasks]
    mk = lambda dev: sag.make(robot, None, seed=11, n_envs=n, devices=dev)
    one, two = mk([0]), mk(devices)
    for e in (one, two):
      e.set_task([benchmark.TASKS[nm]() for nm in names])
  else:
    one = sag.make(robot, task, seed=11, n_envs=n, devices=[0])
    two = sag.make(robot, task, seed=11, n_envs=n, devices=devices)
  o1, o2 = one.reset(), two.reset()
  np.testing.assert_array_equal(o1, o2)
  rng = np.random.RandomState(0)
  for _ in range(steps):
    a = rng.uniform(-1, 1, (n, nu)).astype(np.float32)
    r1, r2 = one.step(a), two.step(a)
    np.testing.assert_array_equal(r1[0], r2[0])
    np.testing.assert_array_equal(r1[1], r2[1])
    np.testing.assert_array_equal(r1[2], r2[2])
    np.testing.assert_array_equal(r1[3]['cost'], r2[3]['cost'])
  s1, s2 = one.get_state(), two.get_state()
  np.testing.assert_array_equal(s1[0], s2[0])
  np.testing.assert_array_equal(s1[1], s2[1])
  one.close(); two.close()


@pytest.mark.parametrize('robot', ['point', 'car'])
def test_maximum_capacity_worlds_lockstep(nat, oracle, robot):
  """Edge case "maximum sizes": every slot of the record in use - 9 hazards, 10 vases, 2 pillars, 6 buttons (Collect's)
  on a larger field - declared through the tasks/* surface (a subclass overriding `obstacles` / `placement_extents`), drawn
  by the native sampler, stepped on the device and compared with the oracle from identical state every step."""
  from safe_adaptation_gym_amd.tasks.press_buttons import Collect

  class Crowded(Collect):
    @property
    def obstacles(self):
      return [nat.MAX_HAZARDS, nat.MAX_VASES, 0, nat.MAX_PILLARS]

    @property
    def placement_extents(self):
      return [-3.0, -3.0, 3.0, 3.0]

  n, T = 96, 60
  rf, ri, st = nat.sample_layouts(robot, 4000 + np.arange(n), 0, descs=[Crowded().descriptor()])
  assert not st.any()
  assert (ri[:, nat.I_NH] == nat.MAX_HAZARDS).all() and (ri[:, nat.I_NV] == nat.MAX_VASES).all()
  assert (ri[:, nat.I_NP] == nat.MAX_PILLARS).all() and (ri[:, nat.I_NB] == nat.MAX_BUTTONS).all()
  rid, nu, od = {'point': (0, 2, 60), 'car': (1, 2, 72)}[robot]
  ctx = nat.Context(robot, n, seed=77)
  ctx.set_layout(rf, ri)
  rng, mt = np.random.RandomState(1), np.random.RandomState(2)
  obs0 = ctx.observe()
  rf, ri = ctx.get_state()
  np.testing.assert_allclose(obs0, oracle.observe_batch(oracle.make_batch(rf, ri), rid, od), rtol=0, atol=2e-5)
  assert (obs0[:, :48] > 0).mean() > 0.3, 'a crowded field shows up in all three lidar groups'
  bad_rows = n_cost = 0
  for t in range(T):
    rf, ri = ctx.get_state()
    arr = oracle.make_batch(rf, ri)
    act = bu.pursuit_actions(rf, ri, rng, robot=robot)
    noise = mt.normal(size=(n, nu)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
    d = ctx.step(act, noise, tape)
    o = oracle.step_batch_full(arr, rid, act, noise, tape, obs_dim=od)
    d_rf, d_ri = ctx.get_state()
    o_rf, o_ri = oracle.batch_records(arr)
    bad = (np.abs(d_rf - o_rf) > 1e-4 + 1e-4 * np.abs(o_rf)).any(1)
    bad_rows += int(bad.sum())
    ok = ~bad
    np.testing.assert_array_equal(d[3], o[3])                      # done
    np.testing.assert_array_equal(d[4][ok], o[4][ok])              # goal met (a button collected)
    np.testing.assert_array_equal(d_ri[ok], o_ri[ok])              # task state incl. the collected-button mask
    near = o[6] <= 1e-5                                            # cost decision within 1e-5 of a threshold
    np.testing.assert_array_equal(d[2][ok & ~near], o[2][ok & ~near])
    np.testing.assert_allclose(d[1][ok], o[1][ok], rtol=0, atol=2e-4)
    n_cost += int(d[2].sum())
  assert n_cost > 50, 'nine hazards should be driven through'
  assert bad_rows <= 0.002 * n * T, f'{bad_rows} env-steps outside the stated tolerance'
  ctx.close()


@pytest.mark.parametrize('robot', ['point', 'car', 'doggo'])
def test_empty_world_lockstep(nat, oracle, robot):
  """Edge case "empty input": a GoToGoal world with no obstacle at all (obstacles = [0, 0, 0, 0] through the tasks/* surface).
  The obstacle and object lidar groups stay zero, no cost is ever raised, and the step still matches the oracle."""
  from safe_adaptation_gym_amd.tasks.go_to_goal import GoToGoal

  class Empty(GoToGoal):
    @property
    def obstacles(self):
      return [0, 0, 0, 0]

  n, T = 64, 25
  rf, ri, st = nat.sample_layouts(robot, 9000 + np.arange(n), 0, descs=[Empty().descriptor()])
  assert not st.any() and not ri[:, [nat.I_NH, nat.I_NV, nat.I_NP, nat.I_NB]].any()
  rid, nu, od = {'point': (0, 2, 60), 'car': (1, 2, 72), 'doggo': (2, 12, 104)}[robot]
  ctx = nat.Context(robot, n, seed=78)
  ctx.set_layout(rf, ri)
  rng, mt = np.random.RandomState(1), np.random.RandomState(2)
  tol = 2e-3 if robot == 'doggo' else 1e-4
  for t in range(T):
    rf, ri = ctx.get_state()
    arr = oracle.make_batch(rf, ri)
    act = mt.uniform(-1, 1, size=(n, nu)).astype(np.float32) if robot == 'doggo' else bu.pursuit_actions(rf, ri, rng, robot=robot)
    noise = mt.normal(size=(n, nu)).astype(np.float32)
    tape = mt.randint(0, 2**32, size=(n, 64), dtype=np.uint32)
    d = ctx.step(act, noise, tape)
    o = oracle.step_batch_full(arr, rid, act, noise, tape, obs_dim=od)
    assert not d[0][:, :32].any(), 'obstacle and object lidar groups of an empty world'
    assert (d[0][:, 32:48] > 0).any(1).all(), 'the goal is always in lidar range'
    assert not d[2].any() and not o[2].any()
    np.testing.assert_array_equal(d[4], o[4])
    np.testing.assert_allclose(d[1], o[1], rtol=0, atol=2e-4)
    np.testing.assert_allclose(ctx.get_state()[0], oracle.batch_records(arr)[0], rtol=tol, atol=tol)
  ctx.close()


def test_two_device_shards_equal_one_context(nat):
  """make(..., devices=[0, 1]): the batch is split into contiguous shards, one context / stream / host thread per
  GPU, no collective.  Shard-concatenated results must equal a single-context run bit for bit (same global env ids
  -> same layouts and the same counter-based noise).  Needs two visible GPUs."""
  if nat.device_count() < 2:
    pytest.skip('needs 2 GPUs (the 8-GPU node of the driver runs it)')
  _shards_equal_one_context([0, 1])   # odd split: 150 + 150 envs, not multiples of the wavefront


@pytest.mark.parametrize('robot,task,n,steps', [('point', 'go_to_goal', 300, 12), ('car', 'push_box', 130, 8), ('doggo', 'multitask', 70, 4)])
def test_shards_on_one_device_equal_one_context(nat, robot, task, n, steps):
  """The same sharding path on the 1-GPU box: devices=[0, 0, 0] = three contexts, streams and host threads on ONE GPU
  (what differs from the multi-GPU case is only the device ordinal).  Ragged shards (300 = 100 x 3, 130 = 44 + 43 + 43,
  70 = 24 + 23 + 23), global env ids as RNG stream ids, the multitask order of BASELINE config 4 for the Doggo."""
  if os.environ.get('SAG_HOSTEMU'):
    pytest.skip('the host emulator (tests/hostemu) runs one kernel at a time: no concurrent host threads')
  _shards_equal_one_context([0, 0, 0], robot, task, n, steps)


def test_rgb_observation_env_api(nat):
  """make(..., rgb_observation=True): reset/step return [N, 64, 64, 3] uint8, the declared space."""
  import safe_adaptation_gym_amd as sag
  env = sag.make('point', 'go_to_goal', seed=2, n_envs=8, rgb_observation=True)
  assert env.observation_space.shape == (64, 64, 3) and env.observation_space.dtype == np.uint8
  obs = env.reset()
  assert obs.shape == (8, 64, 64, 3) and obs.dtype == np.uint8
  obs2, reward, done, info = env.step(np.ones((8, 2), np.float32))
  assert obs2.shape == obs.shape and reward.shape == (8,)
  sky = obs[:, 0].reshape(-1, 3).astype(int)
  assert (sky[:, 2] > sky[:, 0]).all()        # top row: sky, blue dominates
  env.close()


def test_doggo_long_run_is_stable(nat):
  """300 steps (3600 substeps) of random torques on the config-4 task mix: every state stays
  finite, no env reports a physics failure, the base stays between the floor and a jump height,
  the joints stay near their ranges (soft limits), quaternions stay normalised."""
  from safe_adaptation_gym_amd import benchmark
  n = 256
  names = [nm for nm, _ in benchmark.make('multitask', batch_size=n, seed=7).train_tasks]
  rf, ri = bu.sample_records_native('doggo', names, n, seed=99)
  ctx = nat.Context('doggo', n, seed=11)
  ctx.set_layout(rf, ri)
  rng = np.random.RandomState(0)
  E = 144
  for t in range(300):
    obs, rew, cost, done, met, _ = ctx.step(rng.uniform(-1, 1, (n, 12)).astype(np.float32))
    assert not done.any(), f'physics failure at step {t}'
  rf, ri = ctx.get_state()
  assert np.isfinite(rf).all() and np.isfinite(obs).all()
  z = rf[:, E]
  assert (z > 0.02).all() and (z < 0.6).all(), (z.min(), z.max())
  np.testing.assert_allclose(np.linalg.norm(rf[:, E + 1:E + 5], axis=1), 1.0, atol=1e-5)
  q = np.rad2deg(rf[:, E + 9:E + 22])
  assert q.min() > -110 and q.max() < 170, (q.min(), q.max())
  assert np.abs(rf[:, :2]).max() < 4.0     # nobody left the arena at speed
  ctx.close()


def test_doggo_cooperative_linear_algebra(nat, oracle):
  """The wave-cooperative Doggo routines (32 lanes per env: kinematics, composite inertias, mass
  matrix, RNEA bias, Cholesky, smooth solve, explicit inverse) against the oracle's serial fp64
  ones on tumbling mid-air states: same numbers up to summation order."""
  n = 37   # odd: the last wavefront is half filled
  rng = np.random.RandomState(12)
  rf, ri = bu.sample_records_native('doggo', 'go_to_goal', n, seed=1)
  E = 144
  for k in range(n):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    rf[k, E] = rng.uniform(1, 2); rf[k, E + 1:E + 5] = q
    rf[k, E + 5] = rng.normal(); rf[k, E + 6:E + 9] = rng.normal(size=3)
    rf[k, 3:5] = rng.normal(size=2)
    rf[k, E + 9:E + 22] = np.deg2rad(rng.uniform(-20, 20, 13)); rf[k, E + 22:E + 35] = rng.normal(size=13)
  ctx = nat.Context('doggo', n, seed=3)
  ctx.set_state(rf, ri)
  M, bias, qacc, Minv = ctx.debug_doggo_coop()
  rf, ri = ctx.get_state()
  for k in range(n):
    oM, obias, _, _, oqacc = oracle.doggo_debug(oracle.env(rf[k], ri[k]))
    np.testing.assert_allclose(M[k], oM, rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(bias[k], obias, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(qacc[k], oqacc, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Minv[k] @ oM, np.eye(19), atol=1e-7)
  ctx.close()
