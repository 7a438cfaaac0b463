"""ctypes binding of libsag.so (include/sag.h).  NumPy only - no torch on the path.

There is no CPU fallback: if the HIP library is missing or no MI355X is visible,
every entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAG_LIB overrides the library file (A/B builds of the same ABI during kernel tuning)
LIB_PATH = os.environ.get('SAG_LIB') or os.path.join(_HERE, 'libsag.so')

ABI_VERSION = 5
MAX_HAZARDS, MAX_VASES, MAX_PILLARS, MAX_BUTTONS, MAX_NU = 9, 10, 2, 6, 12
REC_FLOATS, REC_INTS = 184, 16

# record field offsets (enum sag_rec_float / sag_rec_int)
F_ROBOT, F_ROBOT0, F_GEAR, F_DAMP, F_ACTION_NOISE, F_CTRL_SCALE = 0, 6, 9, 10, 11, 12
F_HAZARD_SIZE, F_VASE_SIZE, F_PILLAR_SIZE, F_KEEPOUT = 24, 25, 26, 27
F_GOAL, F_CATCH, F_LAST, F_BOX = 32, 34, 38, 41
F_HAZARDS, F_PILLARS, F_BUTTONS, F_VASES = 47, 65, 69, 81
F_ROBOT_EXT = 144
F_BOUND = 141
(I_TASK, I_NH, I_NV, I_NP, I_NB, I_BOX_KIND, I_GOAL_BUTTON, I_BTN_STATE, I_BTN_TIMER,
 I_CATCH_TIMER, I_ACTIVE_MASK, I_STEP, I_ENV_ID, I_FLAGS, I_EPISODE, I_AWAKE) = range(16)

ROBOT_IDS = {'point': 0, 'car': 1, 'doggo': 2}

EXPORTS = [
    'sag_robot_info', 'sag_create', 'sag_destroy', 'sag_last_error', 'sag_set_layout',
    'sag_reset', 'sag_get_state', 'sag_set_state', 'sag_step', 'sag_step_device', 'sag_wait',
    'sag_observe', 'sag_set_ext_contacts', 'sag_lidar_cost', 'sag_lidar_cost_device', 'sag_set_seed', 'sag_dev_alloc', 'sag_dev_free', 'sag_dev_upload',
    'sag_dev_download', 'sag_dev_fill_actions', 'sag_kernel_time_ms', 'sag_enable_timing', 'sag_busy_count', 'sag_debug_cycles', 'sag_render_rgb', 'sag_render_rgb_device', 'sag_render', 'sag_render_device', 'sag_debug_doggo_coop',
    'sag_device_count', 'sag_world_config_default', 'sag_sample_layouts', 'sag_sample_layouts_desc', 'sag_task_desc_default', 'sag_task_desc_check'
]


class SagError(RuntimeError):
  pass


class _Config(C.Structure):
  _fields_ = [('abi_version', C.c_int32), ('robot', C.c_int32), ('n_envs', C.c_int32),
              ('device', C.c_int32), ('max_hazards', C.c_int32), ('max_vases', C.c_int32),
              ('max_pillars', C.c_int32), ('max_buttons', C.c_int32), ('has_box', C.c_int32),
              ('reserved0', C.c_int32), ('seed', C.c_uint64)]


class WorldConfig(C.Structure):
  _fields_ = [(k, C.c_double) for k in ('placements_margin', 'robot_keepout', 'hazards_size',
                                         'vases_size', 'pillars_size', 'hazards_keepout',
                                         'vases_keepout', 'pillars_keepout',
                                         'robot_ctrl_range_scale', 'action_noise', 'max_bound')] + [
                                             ('random_bound', C.c_int32), ('reserved', C.c_int32)]


_lib = None


def load():
  """Load libsag.so; raises SagError (never falls back) if it is not there."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise SagError(
        f'{LIB_PATH} not found: build it with `python -m safe_adaptation_gym_amd.build` '
        '(hipcc, --offload-arch=gfx950). There is no CPU fallback.')
  lib = C.CDLL(LIB_PATH)
  vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
  up, bp = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
  lib.sag_last_error.restype = C.c_char_p
  lib.sag_last_error.argtypes = [vp]
  lib.sag_robot_info.argtypes = [C.c_int32, ip, C.POINTER(C.c_double)]
  lib.sag_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
  lib.sag_destroy.argtypes = [vp]
  lib.sag_set_layout.argtypes = [vp, ip, C.c_int32, fp, ip]
  lib.sag_set_state.argtypes = [vp, ip, C.c_int32, fp, ip]
  lib.sag_get_state.argtypes = [vp, ip, C.c_int32, fp, ip]
  lib.sag_reset.argtypes = [vp, ip, C.c_int32]
  lib.sag_step.argtypes = [vp, fp, fp, up, C.c_int32, C.c_int32, fp, fp, bp, bp, bp, ip]
  lib.sag_step_device.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
  lib.sag_wait.argtypes = [vp]
  lib.sag_observe.argtypes = [vp, fp]
  lib.sag_set_ext_contacts.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
  lib.sag_lidar_cost.argtypes = [vp, C.c_int32, C.c_int32, fp, fp, bp, C.c_float, fp, ip, bp]
  lib.sag_lidar_cost_device.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_float, vp, vp, vp]
  lib.sag_set_seed.argtypes = [vp, C.c_uint64]
  lib.sag_dev_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
  lib.sag_dev_free.argtypes = [vp, vp]
  lib.sag_dev_upload.argtypes = [vp, vp, vp, C.c_uint64]
  lib.sag_dev_download.argtypes = [vp, vp, vp, C.c_uint64]
  lib.sag_dev_fill_actions.argtypes = [vp, vp, C.c_uint32]
  lib.sag_kernel_time_ms.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
  lib.sag_enable_timing.argtypes = [vp, C.c_int32]
  lib.sag_busy_count.argtypes = [vp, C.POINTER(C.c_int32)]
  lib.sag_render_rgb.argtypes = [vp, C.POINTER(C.c_uint8)]
  lib.sag_render_rgb_device.argtypes = [vp, vp]
  lib.sag_render.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
  lib.sag_render_device.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
  lib.sag_debug_doggo_coop.argtypes = [vp, C.POINTER(C.c_double)]
  lib.sag_debug_cycles.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64), C.c_int32]
  lib.sag_world_config_default.argtypes = [C.POINTER(WorldConfig)]
  lib.sag_world_config_default.restype = None
  lib.sag_sample_layouts.argtypes = [C.c_int32, C.c_int32, up, ip, C.POINTER(WorldConfig), C.c_int32,
                                     C.c_int32, fp, ip, up, ip, ip, C.POINTER(C.c_double), ip, C.c_int32]
  _lib = lib
  return lib


def robot_info(robot):
  lib = load()
  out = (C.c_int32 * 5)()
  dt = C.c_double()
  rc = lib.sag_robot_info(ROBOT_IDS[robot] if isinstance(robot, str) else robot, out, C.byref(dt))
  if rc:
    raise SagError(f'sag_robot_info failed ({rc})')
  return dict(nu=out[0], obs_dim=out[1], nstep=out[2], nq=out[3], nv=out[4], dt=dt.value)


class TaskDesc(C.Structure):
  """sag_task_desc (include/sag.h)."""
  _fields_ = [(k, C.c_int32) for k in ('task_id', 'n_hazards', 'n_vases', 'n_pillars', 'has_goal', 'box_kind', 'box_yaw',
                                        'box_at_robot', 'n_buttons', 'button_reset', 'button_timer', 'reserved')] + [
      ('extents', C.c_double * 4), ('goal_keepout', C.c_double), ('box_keepout', C.c_double), ('button_keepout', C.c_double),
      ('box_offset', C.c_double), ('box_rect', C.c_double * 4), ('button_rect', C.c_double * 4), ('gear', C.c_double),
      ('damping', C.c_double)]

  @classmethod
  def from_dict(cls, d):
    t = cls()
    for k, v in d.items():
      if isinstance(v, (list, tuple)):
        setattr(t, k, (C.c_double * 4)(*v))
      else:
        setattr(t, k, v)
    return t

  def to_dict(self):
    return {k: (list(getattr(self, k)) if k in ('extents', 'box_rect', 'button_rect') else getattr(self, k))
            for k, _ in self._fields_ if k != 'reserved'}


def task_desc_default(task_id):
  """The descriptor the library holds for one of the reference's 14 tasks (what the Python Task classes are tested against)."""
  lib = load()
  lib.sag_task_desc_default.argtypes = [C.c_int32, C.POINTER(TaskDesc)]
  t = TaskDesc()
  if lib.sag_task_desc_default(int(task_id), C.byref(t)) != 0:
    raise SagError(f'no task {task_id}')
  return t.to_dict()


def task_desc_check(desc):
  """None, or what the library objects to in a descriptor dict (the field's name and the rule: include/sag.h)."""
  lib = load()
  lib.sag_task_desc_check.argtypes = [C.POINTER(TaskDesc)]
  lib.sag_task_desc_check.restype = C.c_char_p
  msg = lib.sag_task_desc_check(C.byref(TaskDesc.from_dict(desc)))
  return None if msg is None else msg.decode()


def sample_layouts(robot, seeds, task_ids, config=None, first_episode=True, env_id0=0,
                   want_rng=False, nthreads=None, descs=None):
  """Native reset path: records for len(seeds) envs, env j drawn with
  np.random.RandomState(seeds[j]) semantics.  config: dict over World.DEFAULT keys.
  task_ids: one of the 14 reference tasks per env - or, with `descs` (a list of Task.descriptor() dicts), the index of
  env j's descriptor in that list.
  Returns (rec_f, rec_i, status[, rng states as numpy RandomState set_state tuples])."""
  lib = load()
  seeds = np.ascontiguousarray(np.asarray(seeds, np.int64) % 2**32, np.uint32)
  n = len(seeds)
  tids = np.ascontiguousarray(np.broadcast_to(np.asarray(task_ids, np.int32), (n,)))
  cfg = WorldConfig()
  lib.sag_world_config_default(C.byref(cfg))
  for k, v in (config or {}).items():
    if k in ('gremlins_size', 'gremlins_keepout', 'gremlins_travel', 'obstacles_size_noise_scale'):
      continue  # accepted by the reference, without effect (no task spawns gremlins)
    if not hasattr(cfg, k):
      raise KeyError(f'unknown world config key {k!r}')
    setattr(cfg, k, int(v) if k == 'random_bound' else float(v))
  rf = np.zeros((n, REC_FLOATS), np.float32)
  ri = np.zeros((n, REC_INTS), np.int32)
  status = np.zeros(n, np.int32)
  key = np.zeros((n, 624), np.uint32) if want_rng else None
  pos = np.zeros(n, np.int32) if want_rng else None
  hg = np.zeros(n, np.int32) if want_rng else None
  g = np.zeros(n, np.float64) if want_rng else None
  if nthreads is None:
    nthreads = min(len(os.sched_getaffinity(0)), 16)
  rid = ROBOT_IDS[robot] if isinstance(robot, str) else robot
  tail = (C.byref(cfg), int(first_episode), env_id0, _ptr(rf, C.c_float), _ptr(ri, C.c_int32), _ptr(key, C.c_uint32),
          _ptr(pos, C.c_int32), _ptr(hg, C.c_int32), _ptr(g, C.c_double), _ptr(status, C.c_int32), nthreads)
  if descs is None:
    rc = lib.sag_sample_layouts(rid, n, _ptr(seeds, C.c_uint32), _ptr(tids, C.c_int32), *tail)
  else:
    for k, d in enumerate(descs):
      msg = task_desc_check(d)
      if msg:
        raise SagError(f'task descriptor {k}: {msg}')
    arr = (TaskDesc * len(descs))(*[TaskDesc.from_dict(d) for d in descs])
    rc = lib.sag_sample_layouts_desc(rid, n, _ptr(seeds, C.c_uint32), arr, len(descs), _ptr(tids, C.c_int32), *tail)
  if rc < 0:
    raise SagError(f'sag_sample_layouts failed ({rc}): a task descriptor exceeds the record capacity or is inconsistent')
  if want_rng:
    states = [('MT19937', key[j], int(pos[j]), int(hg[j]), float(g[j])) for j in range(n)]
    return rf, ri, status, states
  return rf, ri, status


def device_count():
  return load().sag_device_count()


def _ptr(a, ctype):
  return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


class DeviceArray:
  """A view of device memory owned by a Context (the outputs of a device-resident env.step, BatchedSafeAdaptationGym
  with device_buffers=True): pointer, shape, dtype, strides.  `__cuda_array_interface__` (version 2; HIP pointers are
  what PyTorch-ROCm and CuPy-ROCm expect there) lets a learner wrap it without a copy - torch.as_tensor(a, device='cuda')
  -; `numpy()` downloads.  The memory is written on the context's stream: read it after env.wait() (or step(sync=True))."""

  def __init__(self, ctx, ptr, shape, dtype, strides=None, base=None):
    self.ctx, self.ptr, self.shape, self.dtype, self.strides = ctx, int(ptr), tuple(shape), np.dtype(dtype), strides
    self._base = base   # (ptr, shape) of the contiguous buffer a strided view lives in

  @property
  def __cuda_array_interface__(self):
    return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (self.ptr, False), 'version': 2,
            'strides': self.strides}

  def numpy(self):
    if self.strides is None:
      return self.ctx.dev_download(C.c_void_p(self.ptr), self.shape, self.dtype)
    bptr, bshape = self._base
    full = self.ctx.dev_download(C.c_void_p(bptr), bshape, self.dtype)
    off = (self.ptr - bptr) // self.dtype.itemsize
    return np.lib.stride_tricks.as_strided(full.reshape(-1)[off:], self.shape, self.strides).copy()


def device_pointer(x):
  """Device pointer of a DeviceArray or of anything that exports __cuda_array_interface__ (a torch / cupy array on the GPU)."""
  if isinstance(x, DeviceArray):
    return x.ptr
  cai = getattr(x, '__cuda_array_interface__', None)
  if cai is None:
    raise TypeError(f'{type(x).__name__}: not a device array')
  if cai.get('strides') not in (None,) and tuple(cai['strides']) != tuple(np.zeros(cai['shape'], np.dtype(cai['typestr'])).strides):
    raise ValueError('device actions must be contiguous')
  return int(cai['data'][0])


class Context:
  """One GPU shard: owns the SoA world of n_envs environments on `device`."""

  def __init__(self, robot, n_envs, device=0, seed=0, max_hazards=MAX_HAZARDS,
               max_vases=MAX_VASES, max_pillars=MAX_PILLARS, max_buttons=MAX_BUTTONS,
               has_box=True):
    self.lib = load()
    self.robot = robot
    self.info = robot_info(robot)
    self.n_envs = int(n_envs)
    cfg = _Config(ABI_VERSION, ROBOT_IDS[robot], self.n_envs, device, max_hazards, max_vases,
                  max_pillars, max_buttons, int(has_box), 0, seed & (2**64 - 1))
    h = C.c_void_p()
    rc = self.lib.sag_create(C.byref(cfg), C.byref(h))
    if rc:
      raise SagError(f'sag_create failed ({rc}): {self.lib.sag_last_error(None).decode()}')
    self.h = h
    nu, od, n = self.info['nu'], self.info['obs_dim'], self.n_envs
    self._obs = np.zeros((n, od), np.float32)
    self._rew = np.zeros((n, 2), np.float32)
    self._cost = np.zeros(n, np.uint8)
    self._done = np.zeros(n, np.uint8)
    self._met = np.zeros(n, np.uint8)
    self._used = np.zeros(n, np.int32)

  def _check(self, rc, what):
    if rc:
      raise SagError(f'{what} failed ({rc}): {self.lib.sag_last_error(self.h).decode()}')

  def close(self):
    if getattr(self, 'h', None):
      self.lib.sag_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass

  # -- records ---------------------------------------------------------------
  def _recs(self, env_ids, rec_f, rec_i):
    rec_f = np.ascontiguousarray(rec_f, np.float32).reshape(-1, REC_FLOATS)
    rec_i = np.ascontiguousarray(rec_i, np.int32).reshape(-1, REC_INTS)
    n = len(rec_f)
    assert len(rec_i) == n
    ids = None
    if env_ids is not None:
      ids = np.ascontiguousarray(env_ids, np.int32)
      assert len(ids) == n
    return ids, n, rec_f, rec_i

  def set_layout(self, rec_f, rec_i, env_ids=None):
    ids, n, rf, ri = self._recs(env_ids, rec_f, rec_i)
    self._check(self.lib.sag_set_layout(self.h, _ptr(ids, C.c_int32), n, _ptr(rf, C.c_float),
                                        _ptr(ri, C.c_int32)), 'sag_set_layout')

  def set_state(self, rec_f, rec_i, env_ids=None):
    ids, n, rf, ri = self._recs(env_ids, rec_f, rec_i)
    self._check(self.lib.sag_set_state(self.h, _ptr(ids, C.c_int32), n, _ptr(rf, C.c_float),
                                       _ptr(ri, C.c_int32)), 'sag_set_state')

  def get_state(self, env_ids=None):
    ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
    n = self.n_envs if ids is None else len(ids)
    rf = np.zeros((n, REC_FLOATS), np.float32)
    ri = np.zeros((n, REC_INTS), np.int32)
    self._check(self.lib.sag_get_state(self.h, _ptr(ids, C.c_int32), n, _ptr(rf, C.c_float),
                                       _ptr(ri, C.c_int32)), 'sag_get_state')
    return rf, ri

  def reset(self, env_ids=None):
    ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
    n = self.n_envs if ids is None else len(ids)
    self._check(self.lib.sag_reset(self.h, _ptr(ids, C.c_int32), n), 'sag_reset')

  # -- stepping --------------------------------------------------------------
  def step(self, actions, noise=None, tape=None, nstep=-1, want_tape_used=False):
    n, nu = self.n_envs, self.info['nu']
    a = np.ascontiguousarray(actions, np.float32).reshape(n, nu)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float32).reshape(n, nu)
    tp, tl = None, 0
    if tape is not None:
      tp = np.ascontiguousarray(tape, np.uint32).reshape(n, -1)
      tl = tp.shape[1]
    self._check(
        self.lib.sag_step(self.h, _ptr(a, C.c_float), _ptr(nz, C.c_float), _ptr(tp, C.c_uint32), tl,
                          nstep, _ptr(self._obs, C.c_float), _ptr(self._rew, C.c_float),
                          _ptr(self._cost, C.c_uint8), _ptr(self._done, C.c_uint8),
                          _ptr(self._met, C.c_uint8),
                          _ptr(self._used, C.c_int32) if want_tape_used or tp is not None else None),
        'sag_step')
    return (self._obs.copy(), self._rew.copy(), self._cost.copy(), self._done.copy(),
            self._met.copy(), self._used.copy())

  def observe(self):
    self._check(self.lib.sag_observe(self.h, _ptr(self._obs, C.c_float)), 'sag_observe')
    return self._obs.copy()

  def set_ext_contacts(self, cost_contacts, btn_mask):
    """Contact results of the final state for the next step(nstep=0) (replayed episodes): per env the number of
    robot <-> obstacle contacts (-1: keep the device's own) and the mask of touched buttons
    (mujoco_bridge.py:177-191).  None, None clears a pending set."""
    if cost_contacts is None and btn_mask is None:
      self._check(self.lib.sag_set_ext_contacts(self.h, None, None), 'sag_set_ext_contacts')
      return
    cc = np.ascontiguousarray(cost_contacts, np.int32).reshape(self.n_envs)
    bm = np.ascontiguousarray(btn_mask, np.uint32).reshape(self.n_envs)
    self._check(self.lib.sag_set_ext_contacts(self.h, _ptr(cc, C.c_int32), _ptr(bm, C.c_uint32)), 'sag_set_ext_contacts')

  def lidar_cost(self, robot, points, group, hazard_size=0.2, want_bins=True):
    robot = np.ascontiguousarray(robot, np.float32).reshape(-1, 3)
    n = len(robot)
    group = np.ascontiguousarray(group, np.uint8).reshape(n, -1)
    K = group.shape[1]
    points = np.ascontiguousarray(points, np.float32).reshape(n, K, 2)
    lidar = np.zeros((n, 48), np.float32)
    bins = np.full((n, K), -1, np.int32) if want_bins else None
    cost = np.zeros(n, np.uint8)
    self._check(
        self.lib.sag_lidar_cost(self.h, n, K, _ptr(robot, C.c_float), _ptr(points, C.c_float),
                                _ptr(group, C.c_uint8), hazard_size, _ptr(lidar, C.c_float),
                                _ptr(bins, C.c_int32), _ptr(cost, C.c_uint8)), 'sag_lidar_cost')
    return lidar, bins, cost

  def lidar_cost_device(self, n, K, d_robot, d_points, d_group, d_lidar, d_bins, d_cost, hazard_size=0.2):
    """The same kernel on device buffers, asynchronous on the context stream (bench: kernel-only time)."""
    self._check(self.lib.sag_lidar_cost_device(self.h, n, K, d_robot, d_points, d_group, hazard_size, d_lidar, d_bins,
                                               d_cost), 'sag_lidar_cost_device')

  def set_seed(self, seed):
    """Key of the device-side generator of throughput mode (env.seed())."""
    self._check(self.lib.sag_set_seed(self.h, int(seed) & (2**64 - 1)), 'sag_set_seed')

  # -- device-resident stepping (bench harness / GPU learners) -----------------
  def dev_alloc(self, nbytes):
    p = C.c_void_p()
    self._check(self.lib.sag_dev_alloc(self.h, nbytes, C.byref(p)), 'sag_dev_alloc')
    return p

  def dev_free(self, p):
    self._check(self.lib.sag_dev_free(self.h, p), 'sag_dev_free')

  def dev_upload(self, dst, arr):
    arr = np.ascontiguousarray(arr)
    self._check(self.lib.sag_dev_upload(self.h, dst, arr.ctypes.data_as(C.c_void_p), arr.nbytes),
                'sag_dev_upload')

  def dev_download(self, src, shape, dtype):
    out = np.zeros(shape, dtype)
    self._check(self.lib.sag_dev_download(self.h, out.ctypes.data_as(C.c_void_p), src, out.nbytes),
                'sag_dev_download')
    return out

  def dev_fill_actions(self, d_actions, step_index):
    self._check(self.lib.sag_dev_fill_actions(self.h, d_actions, step_index), 'sag_dev_fill_actions')

  def step_device(self, d_actions, d_noise=None, nstep=-1, d_obs=None, d_reward=None, d_cost=None,
                  d_done=None, d_goal_met=None):
    self._check(
        self.lib.sag_step_device(self.h, d_actions, d_noise, nstep, d_obs, d_reward, d_cost, d_done,
                                 d_goal_met), 'sag_step_device')

  def wait(self):
    self._check(self.lib.sag_wait(self.h), 'sag_wait')

  def enable_timing(self, on=True):
    self._check(self.lib.sag_enable_timing(self.h, int(on)), 'sag_enable_timing')

  def debug_cycles(self, reset=False, worst=False):
    """[3 kernel forms, 15 sections + wavefront count] clock ticks summed over wavefronts (library built with -DSAG_CYCLES);
    worst=True: the sections of the slowest wavefront since the last reset (last column: its total) and its block index per form."""
    out = (C.c_uint64 * 291)()
    self._check(self.lib.sag_debug_cycles(self.h, int(reset), out, 291), 'sag_debug_cycles')
    a = np.array(out[:], np.uint64)
    self.cycles_hist = a[99:291].reshape(3, 64)   # wavefronts by total ticks: bucket b = [2^((b + 40) / 4), 2^((b + 41) / 4))
    return (a[48:96].reshape(3, 16), a[96:99]) if worst else a[:48].reshape(3, 16)

  def render_rgb(self):
    """[n_envs, 64, 64, 3] uint8: the robots' first-person camera images (rgb_observation)."""
    img = np.zeros((self.n_envs, 64, 64, 3), np.uint8)
    self._check(self.lib.sag_render_rgb(self.h, img.ctypes.data_as(C.POINTER(C.c_uint8))), 'sag_render_rgb')
    return img

  CAMERAS = {'vision': 0, 'fixednear': 1, 'fixedfar': 2, 'track': 3}

  def render(self, camera='fixedfar', width=256, height=256, overlays=True):
    """[n_envs, height, width, 3] uint8 from one of the scene's cameras (name or id), optionally with the lidar
    rings and the cost indicator of the last step."""
    cam = self.CAMERAS[camera] if isinstance(camera, str) else int(camera)
    img = np.zeros((self.n_envs, int(height), int(width), 3), np.uint8)
    self._check(self.lib.sag_render(self.h, cam, int(width), int(height), 1 if overlays else 0,
                                    img.ctypes.data_as(C.POINTER(C.c_uint8))), 'sag_render')
    return img

  def debug_doggo_coop(self):
    out = np.zeros((self.n_envs, 760), np.float64)
    self._check(self.lib.sag_debug_doggo_coop(self.h, out.ctypes.data_as(C.POINTER(C.c_double))), 'sag_debug_doggo_coop')
    return out[:, :361].reshape(-1, 19, 19), out[:, 361:380], out[:, 380:399], out[:, 399:].reshape(-1, 19, 19)

  def render_rgb_device(self, d_out):
    self._check(self.lib.sag_render_rgb_device(self.h, d_out), 'sag_render_rgb_device')

  def busy_count(self):
    n = C.c_int32()
    self._check(self.lib.sag_busy_count(self.h, C.byref(n)), 'sag_busy_count')
    return n.value

  def kernel_time_ms(self, reset=False):
    ms, n = C.c_double(), C.c_int64()
    self._check(self.lib.sag_kernel_time_ms(self.h, int(reset), C.byref(ms), C.byref(n)),
                'sag_kernel_time_ms')
    return ms.value, n.value
