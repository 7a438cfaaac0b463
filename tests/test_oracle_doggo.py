"""Known-answer tests of the Doggo specification (oracle/sag_oracle_doggo.inc).  MuJoCo is
absent, so the model is pinned by physics it must satisfy: mass from the geoms, symmetric
positive-definite mass matrix, free fall, energy conservation without contacts, momentum,
standing equilibrium on the floor, joint limits, mirror symmetry."""
import numpy as np
import pytest

import oracle_lib as ol
from golden_util import base_record

DOGGO = 2
DT = 0.012
EXT = ol.F_ROBOT_EXT


@pytest.fixture(scope='module')
def oracle():
  return ol.Oracle()


def doggo_record(x=0.0, y=0.0, yaw=0.0, z=0.22, task='go_to_goal', names=('robot', 'goal')):
  rf, ri = base_record(task, list(names), {'robot': 0.4})
  rf[ol.F_ROBOT:ol.F_ROBOT + 3] = [x, y, yaw]
  rf[EXT:EXT + 40] = 0     # (base_record presets the car's ball quaternion)
  rf[EXT] = z
  rf[EXT + 1:EXT + 5] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
  rf[ol.F_GOAL:ol.F_GOAL + 2] = [5.0, 5.0]
  return rf, ri


def capsule_mass(a, b, r=0.032, dens=5.0):
  L = np.linalg.norm(np.subtract(b, a))
  return dens * (np.pi * r * r * L + 4 / 3 * np.pi * r**3)


def test_total_mass_matches_the_geoms(oracle):
  """doggo.xml:15,48 two cylinders r .075 x .2 density .5; 12 capsules r .032 density 5."""
  rf, ri = doggo_record()
  M, bias, sph, mass, qacc = oracle.doggo_debug(oracle.env(rf, ri))
  cyl = 0.5 * np.pi * 0.075**2 * 0.2
  legs = (capsule_mass((.1, 0, 0), (.2, .1, 0)) + capsule_mass((0, 0, 0), (.098, .0566, -.05)) +
          capsule_mass((0, 0, 0), (-.1176, -.0679, -.1)))
  assert mass == pytest.approx(2 * cyl + 4 * legs, rel=1e-12)
  np.testing.assert_allclose(np.diag(M)[:3], mass, rtol=1e-12)


def test_mass_matrix_is_symmetric_positive_definite(oracle):
  rng = np.random.RandomState(0)
  for _ in range(5):
    rf, ri = doggo_record(yaw=rng.uniform(0, 6))
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    rf[EXT + 1:EXT + 5] = q
    rf[EXT + 9:EXT + 22] = rng.uniform(-0.5, 0.5, 13)
    M = oracle.doggo_debug(oracle.env(rf, ri))[0]
    np.testing.assert_allclose(M, M.T, atol=1e-15)
    assert np.linalg.eigvalsh(M).min() > 0


def test_kinetic_energy_matches_the_body_sum(oracle):
  """qd^T M qd / 2 == sum over spheres is not available; instead: M is invariant under a world
  rotation about z applied to the base (the robot is the same mechanism)."""
  rf, ri = doggo_record(yaw=0.0)
  rf[EXT + 9:EXT + 22] = np.linspace(-0.3, 0.3, 13)
  M0 = oracle.doggo_debug(oracle.env(rf, ri))[0]
  rf2, _ = doggo_record(yaw=1.1)
  rf2[EXT + 9:EXT + 22] = rf[EXT + 9:EXT + 22]
  M1 = oracle.doggo_debug(oracle.env(rf2, ri))[0]
  c, s = np.cos(1.1), np.sin(1.1)
  T = np.eye(19); T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]   # world linear velocity rotates
  np.testing.assert_allclose(T.T @ M1 @ T, M0, atol=1e-12)


def test_free_fall(oracle):
  """High above the floor, no contact: the base accelerates at -g while the springs move the legs;
  the centre of mass follows the parabola exactly (semi-implicit Euler: z_n = z0 - g h^2 n(n+1)/2)."""
  rf, ri = doggo_record(z=5.0)
  rf[EXT + 9:EXT + 22] = np.deg2rad([0, -10, -20, 0, -10, -20, 0, 0, 0, -20, 0, 0, -20])  # springs relaxed
  e = oracle.env(rf, ri)
  n = 20
  oracle.doggo_substeps(e, np.zeros(12), n, DT)
  out, _ = oracle.record(e)
  assert out[EXT] == pytest.approx(5.0 - 9.81 * DT * DT * n * (n + 1) / 2, abs=1e-9)
  assert out[EXT + 5] == pytest.approx(-9.81 * DT * n, abs=1e-9)
  np.testing.assert_allclose(out[EXT + 22:EXT + 35], 0, atol=1e-9)   # joints stay put
  np.testing.assert_allclose(out[ol.F_ROBOT:ol.F_ROBOT + 2], 0, atol=1e-12)


def test_energy_is_conserved_without_contacts(oracle):
  """Tumbling with swinging legs in free flight: kinetic + gravitational + spring energy changes
  only by the first-order error of semi-implicit Euler (for the free-fall part exactly
  -m g^2 h T / 2), i.e. proportionally to h.  A wrong Coriolis / gravity / spring term would leave
  an h-independent drift."""
  def drift(h, n):
    rf, ri = doggo_record(z=2.0)
    rf[EXT + 6:EXT + 9] = [0.8, -0.5, 0.3]                       # base angular velocity
    rf[EXT + 22:EXT + 35] = np.linspace(-1, 1, 13) * 0.5         # joint rates
    rf[EXT + 9:EXT + 22] = np.deg2rad([5, -20, -30, 5, -20, -30, 0, 5, 40, -30, 5, 40, -30])
    e = oracle.env(rf, ri)
    e0 = oracle.doggo_energy(e)
    oracle.doggo_substeps(e, np.zeros(12), n, h)
    return oracle.doggo_energy(e) - e0
  d1, d2, d3 = drift(0.002, 50), drift(0.0005, 200), drift(0.000125, 800)
  assert d1 / d2 == pytest.approx(4.0, rel=0.02)
  assert d2 / d3 == pytest.approx(4.0, rel=0.02)
  mass = oracle.doggo_debug(oracle.env(*doggo_record()))[3]
  # the free-fall term dominates; the rest (rotation, legs) is of the same order or smaller
  assert abs(d3 + 0.5 * mass * 9.81**2 * 0.000125 * 0.1) < 2e-5


def test_linear_momentum_in_free_flight(oracle):
  """Motors, springs and joint limits are internal forces: the horizontal momentum M[0:2,:] qd is
  constant up to the integrator's O(h) error, the vertical one changes by -m g T."""
  def change(h, n):
    rf, ri = doggo_record(z=50.0)
    rf[ol.F_ROBOT + 3:ol.F_ROBOT + 5] = [0.3, -0.2]
    rf[EXT + 6:EXT + 9] = [0.5, 0.4, -0.3]
    rf[EXT + 9:EXT + 22] = np.deg2rad([5, -20, -30, 5, -20, -30, 0, 5, 40, -30, 5, 40, -30])
    e = oracle.env(rf, ri)

    def momentum(e):
      M = oracle.doggo_debug(e)[0]
      r = oracle.record(e)[0]
      qd = np.r_[r[ol.F_ROBOT + 3:ol.F_ROBOT + 5], r[EXT + 5], r[EXT + 6:EXT + 9], r[EXT + 22:EXT + 35]]
      return (M @ qd)[:3]
    p0 = momentum(e)
    ctrl = np.array([1, -1, 1, -1, 0.5, 0.5, -0.5, -0.5, 1, 1, -1, -1.0])
    oracle.doggo_substeps(e, ctrl, n, h)
    return momentum(e) - p0, np.abs(p0[:2]).max()
  d1, scale = change(0.002, 60)
  d2, _ = change(0.00025, 480)
  mass = oracle.doggo_debug(oracle.env(*doggo_record()))[3]
  assert np.abs(d1[:2]).max() < 0.03 * scale
  assert np.abs(d2[:2]).max() < np.abs(d1[:2]).max() / 5     # first order in h
  assert d2[2] == pytest.approx(-mass * 9.81 * 0.12, rel=1e-3)


def test_settles_standing_on_the_floor(oracle):
  """Dropped from the reset pose (z = .22, joints 0) with zero control the robot comes to rest on
  its feet: velocities vanish, the feet carry the weight (touch sensors sum to m g), the base stays
  above the floor and level."""
  rf, ri = doggo_record()
  e = oracle.env(rf, ri)
  for _ in range(60):
    out = oracle.step(e, DOGGO, np.zeros(12), noise=np.zeros(12))
  r = oracle.record(e)[0]
  mass = oracle.doggo_debug(e)[3]
  qd = np.r_[r[ol.F_ROBOT + 3:ol.F_ROBOT + 5], r[EXT + 5:EXT + 9], r[EXT + 22:EXT + 35]]
  assert np.abs(qd).max() < 3e-2, qd   # no joint damping in doggo.xml: the legs creep to rest
  assert 0.05 < r[EXT] < 0.25
  q = r[EXT + 1:EXT + 5]
  assert abs(q[1]) < 0.05 and abs(q[2]) < 0.05              # level
  touch = np.array(out.obs[60:68])
  assert touch.sum() == pytest.approx(mass * 9.81, rel=0.01)
  assert out.cost == 0 and out.done == 0


def test_mirror_symmetry(oracle):
  """The mechanism is symmetric about the x-z plane (legs 1<->4, 2<->3): mirrored controls give the
  mirrored motion."""
  # small torques, short horizon: no joint limit is reached (limit rows are solved in a fixed order)
  a = 0.03 * np.array([0.5, -0.2, 0.3, 0.1, 0.4, 0.6, 0.3, 0.2, -0.5, -0.1, -0.7, -0.6])   # hip_z 1-4, hip_y 1-4, ankle 1-4
  # mirror: leg 1<->4, 2<->3; hip_z axes are (0,0,1) on the left, (0,0,-1) on the right, so the
  # mirrored hip_z angle keeps its sign; hip_y axis (0,1,0) and the ankle axes mirror into each other
  # with the same sign convention
  perm = [3, 2, 1, 0]
  b = np.r_[a[0:4][perm], a[4:8][perm], a[8:12][perm]]
  outs = []
  for ctrl in (a, b):
    rf, ri = doggo_record(z=50.0)   # free flight: contact rows are solved in a fixed (unmirrored) order
    e = oracle.env(rf, ri)
    rf[EXT + 9:EXT + 22] = np.deg2rad([5, -20, -30, 5, -20, -30, 0, 5, 40, -30, 5, 40, -30])
    e = oracle.env(rf, ri)
    for _ in range(2):
      oracle.step(e, DOGGO, ctrl, noise=np.zeros(12))
    outs.append(oracle.record(e)[0])
  ra, rb = outs
  assert ra[ol.F_ROBOT] == pytest.approx(rb[ol.F_ROBOT], abs=1e-9)
  assert ra[ol.F_ROBOT + 1] == pytest.approx(-rb[ol.F_ROBOT + 1], abs=1e-9)
  assert ra[EXT] == pytest.approx(rb[EXT], abs=1e-9)
  assert abs(ra[ol.F_ROBOT + 1]) > 1e-6 or abs(ra[EXT + 9] - ra[EXT + 12]) > 1e-3   # not trivially symmetric
  ja, jb = ra[EXT + 9:EXT + 22], rb[EXT + 9:EXT + 22]
  # qpos order: leg1 (0-2), leg4 (3-5), waist (6), leg2 (7-9), leg3 (10-12)
  np.testing.assert_allclose(ja[0:3], jb[3:6], atol=1e-9)
  np.testing.assert_allclose(ja[7:10], jb[10:13], atol=1e-9)
  assert ja[6] == pytest.approx(-jb[6], abs=1e-9)


def test_joint_limits_hold(oracle):
  """Full torque against the hip_y upper limit (15 deg): the joint ends a little beyond it (soft
  constraint) and does not run away."""
  rf, ri = doggo_record(z=500.0)   # free flight for all 30 steps (it falls 92 m): only the limit stops the joint
  e = oracle.env(rf, ri)
  ctrl = np.zeros(12); ctrl[4] = 1.0  # hip_1_y
  for _ in range(30):
    oracle.step(e, DOGGO, ctrl, noise=np.zeros(12))
  r = oracle.record(e)[0]
  q = np.rad2deg(r[EXT + 9 + 1])
  assert 14.0 < q < 25.0, q


def test_observation_layout(oracle):
  """104 = lidar 48 + accelerometer/velocimeter/gyro/magnetometer 12 + touch 8 + joint rates 12 +
  12 x (sin, cos); at rest in the air the accelerometer reads 0 (free fall) and the magnetometer
  R^T (0, -.5, 0)."""
  yaw = 0.7
  rf, ri = doggo_record(yaw=yaw, z=5.0)
  rf[EXT + 9:EXT + 22] = np.deg2rad([0, -10, -20, 0, -10, -20, 0, 0, 0, -20, 0, 0, -20])
  out = oracle.observe(oracle.env(rf, ri), DOGGO)
  obs = np.array(out.obs)
  np.testing.assert_allclose(obs[48:51], 0, atol=1e-9)
  np.testing.assert_allclose(obs[57:60], [-0.5 * np.sin(yaw), -0.5 * np.cos(yaw), 0], atol=1e-12)
  # jointpos order hip_1..4_z, hip_1..4_y, ankle_1..4 (doggo.xml:92-107)
  ang = np.deg2rad([0, 0, 0, 0, -10, 0, 0, -10, -20, -20, -20, -20])
  np.testing.assert_allclose(obs[80:104:2], np.sin(ang), atol=1e-12)
  np.testing.assert_allclose(obs[81:104:2], np.cos(ang), atol=1e-12)
  # lidar: the goal at (5, 5) is out of range (> 5 m): zero; a goal 1 m ahead lights its bin
  rf[ol.F_GOAL:ol.F_GOAL + 2] = [np.cos(yaw), np.sin(yaw)]
  obs = np.array(oracle.observe(oracle.env(rf, ri), DOGGO).obs)
  assert obs[32] == pytest.approx(0.8, abs=1e-9)


# ---- rgb_observation specification (oracle/sag_oracle_render.inc) ---------------------------
def test_render_known_answers(oracle):
  """Camera of point.xml:14 (fovy 90, looking along +x, pitched 21.8 deg down): the top rows see
  the sky (blue gradient), the rows below the horizon the grey checker floor and the bottom of the image the
  robot's own red body (sphere r .1 right under the camera, point.xml:18-19); a pillar 1 m straight ahead
  fills the image centre with its colour (.5 .5 1) x shade; a hazard disc under the view tints the
  floor blue by alpha .25."""
  from golden_util import base_record
  rf, ri = base_record('go_to_goal', ['robot', 'goal'], {'robot': 0.4})
  rf[ol.F_GOAL:ol.F_GOAL + 2] = [-3.0, 0.0]   # behind the camera
  img = oracle.render_rgb(oracle.env(rf, ri), 0)
  top, floor_row, bottom = img[0].astype(int), img[34].astype(int), img[-1].astype(int)
  assert (top[:, 2] > top[:, 0] + 50).all() and (np.abs(floor_row[:, 0] - floor_row[:, 1]) <= 1).all()
  assert set(np.unique(floor_row[:, 0])) <= {int(0.7 * 255 + .5), int(0.8 * 255 + .5)}
  assert (bottom[:, 0] > 200).all() and (bottom[:, 1:] == 0).all()        # its own body
  # pillar ahead
  rf2, ri2 = base_record('go_to_goal', ['robot', 'goal', 'pillars0'], {'robot': 0.4})
  rf2[ol.F_GOAL:ol.F_GOAL + 2] = [-3.0, 0.0]
  rf2[ol.F_PILLARS:ol.F_PILLARS + 2] = [1.0, 0.0]
  img2 = oracle.render_rgb(oracle.env(rf2, ri2), 0)
  centre = img2[20, 32].astype(float) / 255
  assert centre[2] > 0.55 and abs(centre[0] - centre[1]) < 0.01 and abs(centre[0] / centre[2] - 0.5) < 0.02
  assert (img2[20, 2] == img[20, 2]).all()       # far left column unchanged
  # hazard disc on the floor ahead
  rf3, ri3 = base_record('go_to_goal', ['robot', 'goal', 'hazards0'], {'robot': 0.4})
  rf3[ol.F_GOAL:ol.F_GOAL + 2] = [-3.0, 0.0]
  rf3[ol.F_HAZARDS:ol.F_HAZARDS + 2] = [0.5, 0.0]
  img3 = oracle.render_rgb(oracle.env(rf3, ri3), 0)
  row = 36   # (between the horizon and the robot's own body)
  px = img3[row, 32].astype(int)
  assert px[2] > px[0] + 20 and (img3[row, 2] == img[row, 2]).all()
  # yaw rotates the view: a pillar at +y is centred after turning the robot by 90 degrees
  rf4 = rf2.copy(); rf4[ol.F_PILLARS:ol.F_PILLARS + 2] = [0.0, 1.0]; rf4[ol.F_ROBOT + 2] = np.pi / 2
  img4 = oracle.render_rgb(oracle.env(rf4, ri2), 0)
  assert np.abs(img4[:19].astype(int) - img2[:19].astype(int)).max() <= 1   # above the horizon: sky + pillar (the floor pattern is world-fixed)
  assert np.abs(img4[19:24, 31:33].astype(int) - img2[19:24, 31:33].astype(int)).max() <= 1      # the pillar's centre line


def test_doggo_lidar_uses_the_reference_pinned_routine(oracle):
  """The Doggo observation computes its lidar from the base quaternion; it must equal the general
  routine `sago_lidar(robot_pos, robot_mat, points)` that tests/golden/lidar.npz pins against the
  reference's `_lidar` (tilted-base cases included), for arbitrary base orientations."""
  rng = np.random.RandomState(4)
  names = ['robot', 'goal'] + [f'hazards{k}' for k in range(4)] + [f'vases{k}' for k in range(3)] + ['pillars0']
  for _ in range(20):
    rf, ri = base_record('go_to_goal', names, {'robot': 0.4}, robot='doggo')
    rf[EXT:EXT + 40] = 0
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    pos = np.r_[rng.uniform(-1, 1, 2), rng.uniform(0.1, 0.3)]
    rf[ol.F_ROBOT:ol.F_ROBOT + 2] = pos[:2]; rf[EXT] = pos[2]; rf[EXT + 1:EXT + 5] = q
    haz = rng.uniform(-2, 2, (4, 2)); vas = rng.uniform(-2, 2, (3, 2)); pil = rng.uniform(-2, 2, (1, 2))
    goal = rng.uniform(-1.5, 1.5, 2)
    rf[ol.F_HAZARDS:ol.F_HAZARDS + 8] = haz.ravel()
    for k in range(3):
      rf[ol.F_VASES + 6 * k:ol.F_VASES + 6 * k + 2] = vas[k]
    rf[ol.F_PILLARS:ol.F_PILLARS + 2] = pil[0]
    rf[ol.F_GOAL:ol.F_GOAL + 2] = goal
    obs = np.array(oracle.observe(oracle.env(rf, ri), DOGGO).obs)
    w, x, y, z = q
    R = np.array([[w*w+x*x-y*y-z*z, 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), w*w-x*x+y*y-z*z, 2*(y*z-w*x)],
                  [2*(x*z-w*y), 2*(y*z+w*x), w*w-x*x-y*y+z*z]])
    obst, _ = oracle.lidar(pos, R, np.vstack([haz, vas, pil]))
    gl, _ = oracle.lidar(pos, R, goal[None])
    np.testing.assert_allclose(obs[:16], obst, rtol=0, atol=1e-14)
    np.testing.assert_allclose(obs[16:32], 0, atol=0)
    np.testing.assert_allclose(obs[32:48], gl, rtol=0, atol=1e-14)


def test_first_doggo_call_may_come_from_many_threads():
  """The model tables are built on first use; the batch driver steps envs on OpenMP threads (sago_set_threads), so the
  first Doggo call of a process can come from all of them at once (it did in the GPU suite's free-running test, and the
  racing builds doubled the masses for the rest of the process).  Fresh processes whose FIRST Doggo call is an 8-thread
  batch (thread pool warm) must report the geoms' total mass afterwards."""
  import subprocess
  import sys
  code = '''
import numpy as np, sys
sys.path.insert(0, %r)
import oracle_lib as ol
from test_oracle_doggo import doggo_record
o = ol.Oracle()
o.lib.sago_set_threads(8)
from golden_util import base_record
prf, pri = base_record('go_to_goal', ['robot', 'goal'], {'robot': 0.4})
for _ in range(3):   # a Point batch first: the thread pool exists and all threads enter the Doggo batch together
  o.step_batch_full(o.make_batch(np.stack([prf] * 64), np.stack([pri] * 64)), 0, np.zeros((64, 2), np.float32), obs_dim=60)
recs = [doggo_record(x=0.1 * k) for k in range(64)]
arr = o.make_batch(np.stack([r[0] for r in recs]), np.stack([r[1] for r in recs]))
o.step_batch_full(arr, 2, np.zeros((64, 12), np.float32), obs_dim=104)
rf, ri = doggo_record()
print(repr(float(o.doggo_debug(o.env(rf, ri))[0][0, 0])))
''' % __import__('os').path.dirname(__import__('os').path.abspath(__file__))
  want = 2 * 0.5 * np.pi * 0.075**2 * 0.2
  for _ in range(6):
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = float(out.stdout.strip().splitlines()[-1])
    assert m > want and abs(m - 0.039679578751884034) < 1e-12, m


# ---- collision geometry (round 4): the XML's capsules and cylinders, not spheres ----------------------------------
def _decode(key):
  """(kind, floor point | (object, object geom, contact, robot geom)) of a contact row key (sag_oracle_doggo.inc)."""
  key = int(key)
  if key < 0x10000:
    return 'floor', (key - 0x1000) // 4
  q = (key - 0x10000) // 4
  return 'object', (q // 2048, q // 256 % 8, q // 32 % 8, q % 32)


VASE0, PILLAR0 = 2 + 6, 0   # object numbering of the keys: pillars 0.., buttons 2.., vases 8.., task object 18
ANKLE_1, HIP_1, TORSO_FRONT, TORSO_REAR = 4, 3, 0, 7   # indices into the 14 collision geoms (doggo.xml order)


def contact_scenarios():
  """Analytic contact cases shared by the oracle tests below and the device test (tests/test_gpu_parity.py:
  test_doggo_capsule_and_cylinder_contacts_on_device): name -> (record, expected cost flag)."""
  out = {}
  knee = np.array([.298, .1566, .17]); foot = knee + [-.1176, -.0679, -.1]
  spot = knee + 0.75 * (foot - knee)
  d = (foot - knee)[:2] / np.linalg.norm((foot - knee)[:2])
  nrm = np.array([d[1], -d[0]])
  for name, gap in [('vase corner 5 mm inside the shaft of ankle_1', 0.0), ('the same vase 1 cm further out', 0.01)]:
    rf, ri = doggo_record(names=('robot', 'vases0', 'goal'))
    corner = spot[:2] + (0.032 - 0.005 + gap) * nrm
    rf[ol.F_VASES:ol.F_VASES + 3] = [*(corner + 0.1 * np.sqrt(2) * nrm), np.arctan2(-nrm[1], -nrm[0]) - np.pi / 4]
    out[name] = (rf, ri, int(gap == 0))
  rf, ri = doggo_record(names=('robot', 'pillars0', 'goal'))
  a = np.array([.2, .1, .22])
  p = a + 0.8 * (knee - a)
  dh = (knee - a)[:2] / np.linalg.norm((knee - a)[:2])
  rf[ol.F_PILLARS:ol.F_PILLARS + 2] = p[:2] + (0.032 + 0.2 - 0.003) * np.array([-dh[1], dh[0]])
  out['pillar 3 mm inside the shafts of hip_1 and ankle_1'] = (rf, ri, 1)
  hw = np.sqrt(0.075**2 - 0.02**2)
  rf, ri = doggo_record(names=('robot', 'vases0', 'goal'))
  rf[ol.F_VASES:ol.F_VASES + 3] = [0.03, -(hw - 0.004) - 0.1, 0.0]
  out['vase face 4 mm inside the torso cylinders at the waist'] = (rf, ri, 1)
  rf, ri = doggo_record(names=('robot', 'vases0', 'goal'))
  rf[ol.F_VASES:ol.F_VASES + 3] = [0.2 + hw + 0.005 + 0.1, 0.0, 0.0]
  out['vase face 5 mm beyond where the old nose sphere ended, straight ahead'] = (rf, ri, 0)
  rf, ri = doggo_record(z=0.1566 + 0.032 - 0.005)
  rf[EXT + 1:EXT + 5] = [np.cos(-np.pi / 4), np.sin(-np.pi / 4), 0, 0]
  out['lying on its left side on the knees of legs 1 and 2'] = (rf, ri, 0)
  return out


def test_contact_scenarios_cost_flags(oracle):
  for name, (rf, ri, cost) in contact_scenarios().items():
    assert oracle.step(oracle.env(rf, ri), DOGGO, np.zeros(12), noise=np.zeros(12), nstep=0).cost == cost, name


def test_collision_geometry_is_the_xml(oracle):
  """Axis end points of the 14 geoms at the reset pose (doggo.xml:15-72) and the 16 floor points under them."""
  rf, ri = doggo_record(x=1.0, y=-2.0)
  geo = oracle.doggo_debug(oracle.env(rf, ri))[2]
  fpt, ax = geo[:48].reshape(16, 3), geo[48:].reshape(14, 6)
  o = np.array([1.0, -2.0, 0.22])
  knee1 = o + [.2, .1, 0] + [.098, .0566, -.05]
  np.testing.assert_allclose(ax[TORSO_FRONT], np.r_[o, o + [.2, 0, 0]], atol=1e-12)
  np.testing.assert_allclose(ax[TORSO_REAR], np.r_[o + [-.2, 0, 0], o], atol=1e-12)
  np.testing.assert_allclose(ax[1], np.r_[o + [.1, 0, 0], o + [.2, .1, 0]], atol=1e-12)            # aux_1
  np.testing.assert_allclose(ax[HIP_1], np.r_[o + [.2, .1, 0], knee1], atol=1e-12)
  np.testing.assert_allclose(ax[ANKLE_1], np.r_[knee1, knee1 + [-.1176, -.0679, -.1]], atol=1e-12)
  # floor points: rim of the front cylinder under x = .2 and x = 0, then hip / knee / foot of leg 1 (lowest points)
  np.testing.assert_allclose(fpt[0], o + [.2, 0, -.075], atol=1e-12)
  np.testing.assert_allclose(fpt[1], o + [0, 0, -.075], atol=1e-12)
  np.testing.assert_allclose(fpt[3], knee1 - [0, 0, .032], atol=1e-12)
  np.testing.assert_allclose(fpt[4], knee1 + [-.1176, -.0679, -.1 - .032], atol=1e-12)


def test_cylinder_rim_is_the_floor_point_of_a_pitched_torso(oracle):
  """A cylinder meets a plane with the lowest point of an end rim: nose down by th, that point is r cos th below the
  end-cap centre and r sin th BEHIND it along the heading - not a sphere's r straight below."""
  th = 0.4
  rf, ri = doggo_record(z=0.3)
  rf[EXT + 1:EXT + 5] = [np.cos(th / 2), 0, np.sin(th / 2), 0]      # rotation about +y: x axis dips
  fpt = oracle.doggo_debug(oracle.env(rf, ri))[2][:48].reshape(16, 3)
  end = np.array([.2 * np.cos(th), 0, .3 - .2 * np.sin(th)])
  np.testing.assert_allclose(fpt[0], end + .075 * np.array([-np.sin(th), 0, -np.cos(th)]), atol=1e-12)


def test_capsule_shaft_against_a_vase_corner_is_a_contact_and_a_cost(oracle):
  """a vase corner 5 mm inside ankle_1's SHAFT - three quarters of the way from the knee to the
  foot, 4.3 cm from the foot's end sphere (r 3.2 cm) and clear of every other geom (in plan view the ankle doubles back
  exactly under hip_1, so the spot is chosen beyond the part of the hip that is below the vase's top) - is a contact
  of that capsule and raises the cost (mujoco_bridge.py:177-191 counts any robot geom; world.py:144-155).  The
  17-sphere model of rounds 1-3 found nothing here."""
  rf, ri = doggo_record(names=('robot', 'vases0', 'goal'))
  knee = np.array([.298, .1566, .17]); foot = knee + [-.1176, -.0679, -.1]
  mid = knee + 0.75 * (foot - knee)
  d = (foot - knee)[:2] / np.linalg.norm((foot - knee)[:2])
  out = np.array([d[1], -d[0]])   # horizontal normal of the shaft on the +y side (away from the torso)
  assert out[1] > 0
  corner = mid[:2] + (0.032 - 0.005) * out
  rf[ol.F_VASES:ol.F_VASES + 3] = [*(corner + 0.1 * np.sqrt(2) * out), np.arctan2(-out[1], -out[0]) - np.pi / 4]
  e = oracle.env(rf, ri)
  rows, cc = oracle.doggo_contacts(e)
  obj = [(r, _decode(r[0])[1]) for r in rows if _decode(r[0])[0] == 'object']
  assert len(obj) == 1 and cc == 1
  r, (o, bg, k, g) = obj[0]
  assert (o, bg, k, g) == (VASE0, 0, 0, ANKLE_1)
  assert r[7] == pytest.approx(0.005, abs=1e-9)                      # depth
  np.testing.assert_allclose(r[4:7], [-out[0], -out[1], 0], atol=1e-9)   # the vase pushes the leg away from itself
  assert r[3] == pytest.approx(mid[2], abs=1e-9)                     # at the height of that axis point
  assert r[8] > 0                                                    # and the solver loads it
  assert oracle.step(oracle.env(rf, ri), DOGGO, np.zeros(12), noise=np.zeros(12), nstep=0).cost == 1
  # the same vase 1 cm further out touches nothing
  rf[ol.F_VASES:ol.F_VASES + 2] += 0.01 * out
  e = oracle.env(rf, ri)
  rows, cc = oracle.doggo_contacts(e)
  assert cc == 0 and oracle.step(e, DOGGO, np.zeros(12), noise=np.zeros(12), nstep=0).cost == 0


def test_capsule_shafts_against_a_pillar(oracle):
  """Circle footprint: the contact sits at the foot of the pillar's centre on the capsule's axis.  In plan view ankle_1
  doubles back exactly under hip_1 (doggo.xml:24,27: both run along (.866, .5)), so a pillar (1 m tall) beside the leg
  meets both shafts, each at its own height."""
  rf, ri = doggo_record(names=('robot', 'pillars0', 'goal'))
  a = np.array([.2, .1, .22]); b = np.array([.298, .1566, .17]); foot = b + [-.1176, -.0679, -.1]
  t = 0.8                                                           # 80 % down the hip: z = .18
  p = a + t * (b - a)
  d = (b - a)[:2] / np.linalg.norm((b - a)[:2])
  out = np.array([-d[1], d[0]])
  rf[ol.F_PILLARS:ol.F_PILLARS + 2] = p[:2] + (0.032 + 0.2 - 0.003) * out
  rows, cc = oracle.doggo_contacts(oracle.env(rf, ri))
  obj = [(r, _decode(r[0])[1]) for r in rows if _decode(r[0])[0] == 'object']
  assert cc == len(obj) == 2 and [o[1] for o in obj] == [(PILLAR0, 0, 0, HIP_1), (PILLAR0, 0, 0, ANKLE_1)]
  s_ankle = np.dot(p[:2] - b[:2], (foot - b)[:2]) / np.dot((foot - b)[:2], (foot - b)[:2])
  for (r, _), z in zip(obj, [p[2], b[2] + s_ankle * (foot[2] - b[2])]):
    assert r[7] == pytest.approx(0.003, abs=3e-6) and r[3] == pytest.approx(z, abs=1e-4)   # (the XML's .0566 / .0679 are rounded: not exactly parallel)
    np.testing.assert_allclose(r[4:6], -out, atol=2e-4)


def test_torso_cylinders_present_their_cross_section_at_a_vase_top(oracle):
  """The torso axis (z .22) is 2 cm above a vase (top .2): the cylinders present half width sqrt(.075^2 - .02^2) and
  flat ends.  A vase face 4 mm inside that width under the waist meets each cylinder's rectangle in two vertex
  contacts; the hemispherical caps the sphere model had in front of x = .2 are gone."""
  hw = np.sqrt(0.075**2 - 0.02**2)
  rf, ri = doggo_record(names=('robot', 'vases0', 'goal'))
  rf[ol.F_VASES:ol.F_VASES + 3] = [0.03, -(hw - 0.004) - 0.1, 0.0]   # (x = .03: clear of leg 3's knee and of aux_4)
  rows, cc = oracle.doggo_contacts(oracle.env(rf, ri))
  obj = [(r, _decode(r[0])[1]) for r in rows if _decode(r[0])[0] == 'object']
  assert cc == 4 and sorted(g for _, (_, _, _, g) in obj) == [TORSO_FRONT, TORSO_FRONT, TORSO_REAR, TORSO_REAR]
  for r, _ in obj:
    assert r[7] == pytest.approx(0.004, abs=1e-7) and r[3] == pytest.approx(0.22, abs=1e-9)
    np.testing.assert_allclose(r[4:7], [0, 1, 0], atol=1e-9)
  # a vase face 5.2 cm in front of the flat end cap (x = .2): the torso does not reach it (the sphere the round-3 model
  # had at the end of the axis did, by 2 cm); the legs on either side do
  rf[ol.F_VASES:ol.F_VASES + 3] = [0.2 + 0.032 + 0.02 + 0.1, 0.0, 0.0]
  rows, cc = oracle.doggo_contacts(oracle.env(rf, ri))
  geoms = [_decode(r[0])[1][3] for r in rows if _decode(r[0])[0] == 'object']
  assert cc == len(geoms) > 0 and TORSO_FRONT not in geoms


def test_merged_knee_row_and_touch_share(oracle):
  """Hip_k's end sphere and ankle_k's start sphere coincide at the knee: one row with half the regulariser, and the
  touch sensor ankle_ka (a site on the ANKLE body) reads half of its force; a foot's row is read in full.  Lying on
  its left side (base rolled by -90 degrees: body +y points down) the robot rests on the knees of legs 1 and 2."""
  rf, ri = doggo_record(z=0.1566 + 0.032 - 0.005)
  rf[EXT + 1:EXT + 5] = [np.cos(-np.pi / 4), np.sin(-np.pi / 4), 0, 0]
  e = oracle.env(rf, ri)
  rows, _ = oracle.doggo_contacts(e)
  force = {_decode(r[0])[1]: r[8] for r in rows if _decode(r[0])[0] == 'floor'}
  assert sorted(force) == [3, 11]            # floor points 3, 11 = knees of legs 1, 2
  touch = np.array(oracle.observe(e, DOGGO).obs[60:68])
  assert force[3] > 0 and force[11] > 0
  np.testing.assert_allclose(touch, [0.5 * force[3], 0.5 * force[11], 0, 0, 0, 0, 0, 0], rtol=1e-6)
  # standing (test_settles_standing_on_the_floor) the feet's rows are read in full: the sensors sum to m g there


def test_closest_axis_point_to_a_box_is_the_true_minimum(oracle):
  """dg_seg_box_t claims a closed form: the signed distance to a box is convex along a line, so its minimum over a
  segment is at an end, at the foot of a corner, or at a kink (the box's axes, the diagonals |x| - hx = |y| - hy) - twelve
  candidates.  Against a dense scan of 20 001 points on 3000 random segments (outside, grazing, crossing, inside,
  degenerate) the returned point is never worse than the best scanned one, and within the scan's resolution of it."""
  import ctypes as C
  f = oracle.lib.sago_seg_box_t
  f.restype = C.c_double
  f.argtypes = [C.c_double] * 8 + [C.POINTER(C.c_double)]
  rng = np.random.RandomState(12)
  ts = np.linspace(0, 1, 20001)

  def sd_box(x, y, hx, hy):
    qx, qy = np.abs(x) - hx, np.abs(y) - hy
    return np.hypot(np.maximum(qx, 0), np.maximum(qy, 0)) + np.minimum(np.maximum(qx, qy), 0)
  worst = 0.0
  for case in range(3000):
    hx, hy = rng.uniform(0.05, 0.4, 2)
    if case % 5 == 0:
      hy = hx                                   # squares: the diagonals' kinks coincide
    a = rng.uniform(-0.6, 0.6, 2)
    b = rng.uniform(-0.6, 0.6, 2) if case % 7 else a + rng.uniform(-1e-3, 1e-3, 2)   # (every 7th: a very short segment)
    if case % 11 == 0:
      b = np.array([b[0], a[1]])                # axis-parallel
    t0, t1 = (0.0, 1.0) if case % 3 else sorted(rng.uniform(0, 1, 2))
    d = b - a
    out = C.c_double()
    t = f(a[0], a[1], d[0], d[1], hx, hy, t0, t1, C.byref(out))
    assert t0 - 1e-12 <= t <= t1 + 1e-12
    tt = t0 + (t1 - t0) * ts
    scan = sd_box(a[0] + tt * d[0], a[1] + tt * d[1], hx, hy)
    assert out.value <= scan.min() + 1e-12, (case, out.value, scan.min())
    worst = max(worst, scan.min() - out.value)
  assert worst < 1e-4      # the scan's own resolution (segments up to 1.7 m long at 20 001 points, slope <= 1)
