"""ctypes loader for the CPU oracle (oracle/sag_oracle.c).  Test infrastructure:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')

# record schema (include/sag.h)
REC_FLOATS = 184
REC_INTS = 16
F_ROBOT, F_ROBOT0, F_GEAR, F_DAMP, F_ACTION_NOISE, F_CTRL_SCALE = 0, 6, 9, 10, 11, 12
F_HAZARD_SIZE, F_VASE_SIZE, F_PILLAR_SIZE, F_KEEPOUT = 24, 25, 26, 27
F_GOAL, F_CATCH, F_LAST, F_BOX = 32, 34, 38, 41
F_HAZARDS, F_PILLARS, F_BUTTONS, F_VASES = 47, 65, 69, 81
F_ROBOT_EXT = 144
F_BOUND = 141
(I_TASK, I_NH, I_NV, I_NP, I_NB, I_BOX_KIND, I_GOAL_BUTTON, I_BTN_STATE,
 I_BTN_TIMER, I_CATCH_TIMER, I_ACTIVE_MASK, I_STEP, I_ENV_ID, I_FLAGS, I_EPISODE, I_AWAKE) = range(16)


def build(force=False):
  out = os.path.join(ORACLE_DIR, '_build', 'libsag_oracle.so')
  src = os.path.join(ORACLE_DIR, 'sag_oracle.c')
  if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(
      src):
    subprocess.check_call(['make', '-s', '-C', ORACLE_DIR], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
  return out


class Oracle:

  def __init__(self, f32=False):
    path = build()
    if f32:
      path = path.replace('libsag_oracle.so', 'libsag_oracle_f32.so')
    self.lib = lib = C.CDLL(path)
    self.real = np.float32 if f32 else np.float64
    creal = C.c_float if f32 else C.c_double
    assert lib.sago_real_bytes() == np.dtype(self.real).itemsize

    class OEnv(C.Structure):
      _fields_ = [('f', creal * REC_FLOATS), ('i', C.c_int32 * REC_INTS)]

    class OOut(C.Structure):
      _fields_ = [('obs', creal * 104), ('reward', creal * 2), ('cost', C.c_int),
                  ('done', C.c_int), ('goal_met', C.c_int), ('tape_used', C.c_int),
                  ('qacc', creal * 3), ('cost_margin', creal),
                  ('btn_contact_mask', C.c_uint32), ('touch', creal * 8), ('comvel', creal * 4)]

    self.OEnv, self.OOut = OEnv, OOut
    assert lib.sago_sizeof_env() == C.sizeof(OEnv)
    assert lib.sago_sizeof_out() == C.sizeof(OOut)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    up = C.POINTER(C.c_uint32)
    bp = C.POINTER(C.c_uint8)
    lib.sago_load.argtypes = [C.POINTER(OEnv), fp, ip]
    lib.sago_load_f64.argtypes = [C.POINTER(OEnv), dp, ip]
    lib.sago_store.argtypes = [C.POINTER(OEnv), fp, ip]
    lib.sago_store_f64.argtypes = [C.POINTER(OEnv), dp, ip]
    lib.sago_step.argtypes = [
        C.POINTER(OEnv), C.c_int, fp, fp, up, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
        C.c_int, C.c_uint32, C.POINTER(OOut)
    ]
    lib.sago_observe.argtypes = [C.POINTER(OEnv), C.c_int, C.POINTER(OOut)]
    lib.sago_task_reset.argtypes = [C.POINTER(OEnv)]
    lib.sago_lidar.argtypes = [dp, dp, dp, C.c_int, dp, ip]
    lib.sago_lidar_cost.argtypes = [C.c_int, fp, fp, bp, C.c_float, fp, ip, bp]
    lib.sago_substeps.argtypes = [C.POINTER(OEnv), C.POINTER(creal), C.c_int, C.c_double]
    lib.sago_step_batch.argtypes = [
        C.POINTER(OEnv), C.c_int, C.c_int, fp, C.c_uint32, C.c_uint32, fp, fp, bp, bp, bp,
        C.c_int
    ]
    lib.sago_step_batch_full.argtypes = [
        C.POINTER(OEnv), C.c_int, C.c_int, fp, fp, up, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
        fp, fp, bp, bp, bp, ip, dp
    ]
    lib.sago_observe_batch.argtypes = [C.POINTER(OEnv), C.c_int, C.c_int, fp]
    lib.sago_noise.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, fp]
    lib.sago_actions.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, fp]
    lib.sago_philox.argtypes = [up, up, up]
    lib.sago_robot_info.argtypes = [C.c_int, ip, dp]
    lib.sago_render_rgb.argtypes = [C.POINTER(OEnv), C.c_int, bp]
    lib.sago_render.argtypes = [C.POINTER(OEnv), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, bp]
    lib.sago_doggo_substeps.argtypes = [C.POINTER(OEnv), dp, C.c_int, C.c_double]
    lib.sago_doggo_energy.argtypes = [C.POINTER(OEnv)]
    lib.sago_doggo_energy.restype = C.c_double
    lib.sago_doggo_debug.argtypes = [C.POINTER(OEnv), dp, dp, dp, dp, dp]

  def render_rgb(self, e, robot):
    img = np.zeros((64, 64, 3), np.uint8)
    self.lib.sago_render_rgb(C.byref(e), robot, img.ctypes.data_as(C.POINTER(C.c_uint8)))
    return img

  def render(self, e, robot, camera, width, height, overlays=False, lidar48=None, cost=0):
    img = np.zeros((height, width, 3), np.uint8)
    lid = None if lidar48 is None else np.ascontiguousarray(lidar48, np.float32)
    self.lib.sago_render(C.byref(e), robot, camera, width, height, 1 if overlays else 0,
                         None if lid is None else lid.ctypes.data_as(C.POINTER(C.c_float)), int(cost),
                         img.ctypes.data_as(C.POINTER(C.c_uint8)))
    return img

  # -- doggo diagnostics -----------------------------------------------------
  def doggo_substeps(self, e, ctrl12, nstep, h=0.012):
    c = np.ascontiguousarray(ctrl12, np.float64)
    self.lib.sago_doggo_substeps(C.byref(e), c.ctypes.data_as(C.POINTER(C.c_double)), nstep, h)

  def doggo_energy(self, e):
    return self.lib.sago_doggo_energy(C.byref(e))

  def doggo_debug(self, e):
    M, bias, sph = np.zeros((19, 19)), np.zeros(19), np.zeros(16 * 3 + 14 * 6)   # floor points [16x3], geom axes [14x6]
    mass, qacc = np.zeros(1), np.zeros(19)
    dp = C.POINTER(C.c_double)
    self.lib.sago_doggo_debug(C.byref(e), M.ctypes.data_as(dp), bias.ctypes.data_as(dp),
                              sph.ctypes.data_as(dp), mass.ctypes.data_as(dp), qacc.ctypes.data_as(dp))
    return M, bias, sph, float(mass[0]), qacc

  def doggo_contacts(self, e, max_contacts=64):
    """Contacts of one forward evaluation at zero control: rows [key, px, py, pz, nx, ny, nz, depth, force] and the
    count the cost rule sees."""
    out = np.zeros((max_contacts, 9))
    cc = C.c_int(0)
    n = self.lib.sago_doggo_contacts(C.byref(e), max_contacts, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cc))
    return out[:n], cc.value

  # -- single env ----------------------------------------------------------
  def env(self, rec_f, rec_i):
    e = self.OEnv()
    rf = np.ascontiguousarray(rec_f)
    ri = np.ascontiguousarray(rec_i, dtype=np.int32)
    if rf.dtype == np.float64:
      self.lib.sago_load_f64(C.byref(e), rf.ctypes.data_as(C.POINTER(C.c_double)),
                             ri.ctypes.data_as(C.POINTER(C.c_int32)))
    else:
      rf = rf.astype(np.float32)
      self.lib.sago_load(C.byref(e), rf.ctypes.data_as(C.POINTER(C.c_float)),
                         ri.ctypes.data_as(C.POINTER(C.c_int32)))
    return e

  def record(self, e, f64=True):
    rf = np.zeros(REC_FLOATS, np.float64 if f64 else np.float32)
    ri = np.zeros(REC_INTS, np.int32)
    if f64:
      self.lib.sago_store_f64(C.byref(e), rf.ctypes.data_as(C.POINTER(C.c_double)),
                              ri.ctypes.data_as(C.POINTER(C.c_int32)))
    else:
      self.lib.sago_store(C.byref(e), rf.ctypes.data_as(C.POINTER(C.c_float)),
                          ri.ctypes.data_as(C.POINTER(C.c_int32)))
    return rf, ri

  def step(self, e, robot, action, noise=None, tape=None, key=(0, 0), nstep=-1,
           ext_contacts=-1, ext_btn_mask=0):
    out = self.OOut()
    a = np.ascontiguousarray(action, np.float32)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
    tp = None if tape is None else np.ascontiguousarray(tape, np.uint32)
    self.lib.sago_step(
        C.byref(e), robot, a.ctypes.data_as(C.POINTER(C.c_float)),
        None if nz is None else nz.ctypes.data_as(C.POINTER(C.c_float)),
        None if tp is None else tp.ctypes.data_as(C.POINTER(C.c_uint32)),
        0 if tp is None else len(tp), key[0], key[1], nstep, ext_contacts, ext_btn_mask,
        C.byref(out))
    return out

  def observe(self, e, robot):
    out = self.OOut()
    self.lib.sago_observe(C.byref(e), robot, C.byref(out))
    return out

  def task_reset(self, e):
    self.lib.sago_task_reset(C.byref(e))

  def substeps(self, e, ctrl, nstep, h):
    c = (C.c_float if self.real == np.float32 else C.c_double) * 12
    cc = c(*[float(x) for x in ctrl] + [0.0] * (12 - len(ctrl)))
    self.lib.sago_substeps(C.byref(e), cc, nstep, h)

  # -- lidar ---------------------------------------------------------------
  def lidar(self, robot_pos, mat, pts):
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    rp = np.ascontiguousarray(robot_pos, np.float64)
    m = np.ascontiguousarray(mat, np.float64).reshape(9)
    obs = np.zeros(16)
    bins = np.zeros(max(len(pts), 1), np.int32)
    dp = C.POINTER(C.c_double)
    self.lib.sago_lidar(rp.ctypes.data_as(dp), m.ctypes.data_as(dp), pts.ctypes.data_as(dp),
                        len(pts), obs.ctypes.data_as(dp),
                        bins.ctypes.data_as(C.POINTER(C.c_int32)))
    return obs, bins[:len(pts)]

  def lidar_cost(self, robot, points, group, hazard_size=0.2):
    """Batched wrapper over the scalar sago_lidar_cost. robot [n,3], points [n,K,2], group [n,K]."""
    robot = np.ascontiguousarray(robot, np.float32)
    points = np.ascontiguousarray(points, np.float32)
    group = np.ascontiguousarray(group, np.uint8)
    n, K = group.shape
    lidar = np.zeros((n, 48), np.float32)
    bins = np.zeros((n, K), np.int32)
    cost = np.zeros(n, np.uint8)
    fp = C.POINTER(C.c_float)
    for i in range(n):
      self.lib.sago_lidar_cost(K, robot[i].ctypes.data_as(fp), points[i].ctypes.data_as(fp),
                               group[i].ctypes.data_as(C.POINTER(C.c_uint8)), hazard_size,
                               lidar[i].ctypes.data_as(fp),
                               bins[i].ctypes.data_as(C.POINTER(C.c_int32)),
                               cost[i:].ctypes.data_as(C.POINTER(C.c_uint8)))
    return lidar, bins, cost

  # -- batch ---------------------------------------------------------------
  def make_batch(self, rec_f, rec_i):
    n = len(rec_f)
    arr = (self.OEnv * n)()
    rec_f = np.ascontiguousarray(rec_f, np.float32)
    rec_i = np.ascontiguousarray(rec_i, np.int32)
    for k in range(n):
      self.lib.sago_load(C.byref(arr[k]), rec_f[k].ctypes.data_as(C.POINTER(C.c_float)),
                         rec_i[k].ctypes.data_as(C.POINTER(C.c_int32)))
    return arr

  def batch_records(self, arr):
    n = len(arr)
    rf = np.zeros((n, REC_FLOATS), np.float32)
    ri = np.zeros((n, REC_INTS), np.int32)
    for k in range(n):
      self.lib.sago_store(C.byref(arr[k]), rf[k].ctypes.data_as(C.POINTER(C.c_float)),
                          ri[k].ctypes.data_as(C.POINTER(C.c_int32)))
    return rf, ri

  def step_batch(self, arr, robot, actions, key=(0, 0), nthreads=1, obs_dim=60):
    n = len(arr)
    actions = np.ascontiguousarray(actions, np.float32)
    obs = np.zeros((n, obs_dim), np.float32)
    rew = np.zeros((n, 2), np.float32)
    cost = np.zeros(n, np.uint8)
    done = np.zeros(n, np.uint8)
    met = np.zeros(n, np.uint8)
    fp = C.POINTER(C.c_float)
    bp = C.POINTER(C.c_uint8)
    self.lib.sago_step_batch(arr, n, robot, actions.ctypes.data_as(fp), key[0], key[1],
                             obs.ctypes.data_as(fp), rew.ctypes.data_as(fp),
                             cost.ctypes.data_as(bp), done.ctypes.data_as(bp),
                             met.ctypes.data_as(bp), nthreads)
    return obs, rew, cost, done, met

  def step_batch_full(self, arr, robot, actions, noise=None, tape=None, key=(0, 0), nstep=-1,
                      obs_dim=60):
    n = len(arr)
    actions = np.ascontiguousarray(actions, np.float32)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
    tp = None if tape is None else np.ascontiguousarray(tape, np.uint32).reshape(n, -1)
    obs = np.zeros((n, obs_dim), np.float32)
    rew = np.zeros((n, 2), np.float32)
    cost, done, met = (np.zeros(n, np.uint8) for _ in range(3))
    used = np.zeros(n, np.int32)
    margin = np.zeros(n, np.float64)
    fp, bp = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    self.lib.sago_step_batch_full(
        arr, n, robot, actions.ctypes.data_as(fp),
        None if nz is None else nz.ctypes.data_as(fp),
        None if tp is None else tp.ctypes.data_as(C.POINTER(C.c_uint32)),
        0 if tp is None else tp.shape[1], key[0], key[1], nstep, obs.ctypes.data_as(fp),
        rew.ctypes.data_as(fp), cost.ctypes.data_as(bp), done.ctypes.data_as(bp),
        met.ctypes.data_as(bp), used.ctypes.data_as(C.POINTER(C.c_int32)),
        margin.ctypes.data_as(C.POINTER(C.c_double)))
    return obs, rew, cost, done, met, used, margin

  def observe_batch(self, arr, robot, obs_dim=60):
    obs = np.zeros((len(arr), obs_dim), np.float32)
    self.lib.sago_observe_batch(arr, len(arr), robot, obs.ctypes.data_as(C.POINTER(C.c_float)))
    return obs

  def noise(self, key, env_id, step, nu):
    out = np.zeros(nu + 1, np.float32)
    self.lib.sago_noise(key[0], key[1], env_id, step, nu,
                        out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:nu]

  def actions(self, key, env_id, step, nu):
    out = np.zeros(nu + 3, np.float32)
    self.lib.sago_actions(key[0], key[1], env_id, step, nu,
                          out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:nu]
