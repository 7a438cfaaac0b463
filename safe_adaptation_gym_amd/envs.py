"""Batched SafeAdaptationGym: the reference's make / reset / step / set_task surface
(safe_adaptation_gym.py:21-257) over N independent environments stepped by the HIP
kernels of libsag.so.  NumPy only; no torch.

Environments never interact, so a batch is split into contiguous shards, one
native context (= one GPU, one stream) per shard, with no collective anywhere."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from safe_adaptation_gym_amd import _native as nat
from safe_adaptation_gym_amd.robot import Robot
from safe_adaptation_gym_amd.tasks.task import Task
from safe_adaptation_gym_amd.utils import ResamplingError
from safe_adaptation_gym_amd import consts

TAPE_WORDS = 1024  # raw generator words offered to the device per env per step (parity mode)


class Box:
  """Minimal stand-in for gym.spaces.Box (gym is not a dependency)."""

  def __init__(self, low, high, shape, dtype=np.float32, seed=None):
    self.low = np.broadcast_to(np.asarray(low, dtype), shape).copy()
    self.high = np.broadcast_to(np.asarray(high, dtype), shape).copy()
    self.shape, self.dtype = tuple(shape), dtype
    self._rs = np.random.RandomState(seed)

  def sample(self):
    return self._rs.uniform(self.low, self.high).astype(self.dtype)

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def shard_ranges(n, parts):
  """Contiguous [start, stop) ranges splitting n envs over `parts` devices."""
  base, rem = divmod(n, parts)
  out, s = [], 0
  for p in range(parts):
    e = s + base + (1 if p < rem else 0)
    out.append((s, e))
    s = e
  return out


class BatchedSafeAdaptationGym:
  NUM_LIDAR_BINS = 16
  LIDAR_MAX_DIST = 5.
  BASE_SENSORS = ['accelerometer', 'velocimeter', 'gyro', 'magnetometer']

  def __init__(self, robot_base, n_envs=1, rgb_observation=False, config=None, devices=None,
               parity_rng=False, device_seed=None, render_lidars_and_collision=False, render_options=None,
               device_buffers=False):
    # rgb_observation: the observation is the robot camera's 64 x 64 x 3 uint8 image
    # (safe_adaptation_gym.py:122-126,148-149), ray-cast on the device
    self._rgb_observation = bool(rgb_observation)
    # human view (safe_adaptation_gym.py:37-38,109-111): render(**render_options) of the scene's cameras, with the
    # lidar rings / cost sphere of render.py when render_lidars_and_collision
    self._render_lidars_and_collision = bool(render_lidars_and_collision)
    self._render_options = dict(render_options) if render_options else {}
    self.robot = Robot(robot_base)
    self.n_envs = int(n_envs)
    self.base_config = config
    self.parity_rng = bool(parity_rng)
    # device_buffers: step() / reset() leave their results in HBM and return _native.DeviceArray views of them (see
    # step()); actions may be device arrays too.  The reference's API, without the PCIe round trip per step.
    self.device_buffers = bool(device_buffers)
    if self.device_buffers and self.parity_rng:
      raise ValueError('device_buffers: the reference-order host generators of parity_rng draw on the host every step')
    self.devices = [0] if devices is None else list(devices)
    if self.n_envs < len(self.devices):
      self.devices = self.devices[:self.n_envs]
    self._ranges = shard_ranges(self.n_envs, len(self.devices))
    # throughput mode: the device-side generator is keyed by the env seed (seed() / reset(seed=...)) unless
    # device_seed pins it; counter = (global env id, step, draw, episode), so noise and in-step draws differ
    # between envs, steps, episodes and seeds
    self._device_seed = device_seed
    self._ctx = [
        nat.Context(self.robot.name, e - s, device=d, seed=0 if device_seed is None else device_seed)
        for (s, e), d in zip(self._ranges, self.devices)
    ]
    self._pool = ThreadPoolExecutor(len(self._ctx)) if len(self._ctx) > 1 else None
    self._dev = None   # device_buffers: per-shard output / action buffers, allocated on first use
    self._tasks = None
    self._episode = 0
    self._base_seed = int(np.random.randint(2**31))
    self._seeds = self._base_seed + np.arange(self.n_envs, dtype=np.int64)
    self._rs = None
    self._key_device()
    self.action_space = Box(-1, 1, (self.robot.nu,), np.float32)
    self._observation_space = None
    self._reward_dim = 1

  # -- reference surface ----------------------------------------------------------
  @property
  def rs(self):
    """Per-env host generators (safe_adaptation_gym.py:33,113-118).  Only parity mode draws from
    them, so they are built on first use (a RandomState costs ~0.1 ms: 100 s per million envs)."""
    if self._rs is None:
      self._rs = [np.random.RandomState(int(s) % 2**32) for s in self._seeds]
    return self._rs

  @property
  def observation_space(self):
    if self._observation_space is None and self._rgb_observation:
      self._observation_space = Box(0, 255, (64, 64, 3), np.uint8)
    if self._observation_space is None:
      d = self.robot.obs_dim
      lidar = 3 * self.NUM_LIDAR_BINS
      low = np.array([0.] * lidar + [-np.inf] * (d - lidar), np.float32)
      high = np.array([1.] * lidar + [np.inf] * (d - lidar), np.float32)
      self._observation_space = Box(low, high, (d,), np.float32)
    return self._observation_space

  def seed(self, seed=None):
    """Env i uses seed + i (the reference's single env uses `seed`,
    safe_adaptation_gym.py:113-118)."""
    self._base_seed = int(np.random.randint(2**31)) if seed is None else int(seed)
    self._seeds = self._base_seed + np.arange(self.n_envs, dtype=np.int64)
    self._rs = None
    self._key_device()

  def _key_device(self):
    if self._device_seed is None:
      for c in self._ctx:
        c.set_seed(self._base_seed)

  def set_task(self, task):
    """A Task instance / class (every env gets its own instance of that class) or a
    sequence of n_envs instances (heterogeneous batch)."""
    self._tasks = self._expand_tasks(task)
    unknown = set(self.base_config or {}) - set(consts.WORLD_DEFAULT)   # the reference accepts anything (world.py:43-44)
    if unknown:
      raise KeyError(f'unknown world config keys: {sorted(unknown)}')
    # the native sampler draws from what the Task objects say about themselves (tasks/task.py: obstacles,
    # placement_extents, setup_placements(), attributes), deduplicated: env j uses descriptor _desc_of_env[j]
    self._descs, self._desc_of_env, seen = [], np.zeros(self.n_envs, np.int32), {}
    for j, t in enumerate(self._tasks):
      d = t.descriptor()
      key = repr(sorted(d.items()))
      if key not in seen:
        seen[key] = len(self._descs)
        self._descs.append(d)
      self._desc_of_env[j] = seen[key]
    self._task_ids = np.array([t.TASK_ID for t in self._tasks], np.int32)
    self._reward_dim = max(t.REWARD_DIM for t in self._tasks)
    self._persist = None  # task attributes that outlive an episode (filled by _pull_task_state)
    self._build_world(first_episode=True)

  def reset(self, *, seed=None, return_info=False, options=None):
    assert self._tasks is not None or (options is not None and 'task' in options), (
        'A task should be first set before reset.')
    self._episode += 1
    if seed is not None:
      self._base_seed = int(seed)
      self._seeds = int(seed) + np.arange(self.n_envs, dtype=np.int64)
      self._key_device()
    else:
      # the reference's single env moves to seed + 1 (safe_adaptation_gym.py:97-101);
      # a batch moves every env past the whole batch so episodes never share a seed.
      self._seeds = self._seeds + self.n_envs
    self._rs = None
    if options is not None and 'task' in options:
      self.set_task(options['task'])
      return self._observe()
    self._pull_task_state()
    self._build_world(first_episode=False)
    return self._observe()

  def step(self, action, sync=True):
    """-> (obs [N, obs_dim] f32, reward [N] (or [N, 2]) f32, done [N] bool,
    info {'cost': [N] f32, 'bound': [N] f32, 'goal_met': [N] bool})

    With device_buffers=True the step is enqueued with sag_step_device and the same tuple comes back as
    _native.DeviceArray views of HBM (obs f32, reward f32, done / cost / goal_met uint8 flags; `bound` stays a host
    array): no copy in either direction.  `action` may be a host array (uploaded) or a device array - a DeviceArray or
    anything with __cuda_array_interface__, e.g. a torch tensor on the GPU - of shape [N, nu] float32 (one per shard, in
    a list, when the batch is sharded over several `devices`; the results are then lists, one entry per shard).
    sync=False returns as soon as the launches are enqueued on the contexts' streams: call env.wait() before the
    results are read on another stream.  The views are overwritten by the next step."""
    if self.device_buffers:
      return self._step_device(action, sync)
    a = np.asarray(action, np.float32).reshape(self.n_envs, self.robot.nu)
    noise = tapes = None
    if self.parity_rng:
      noise = np.stack([rs.normal(size=self.robot.nu) for rs in self.rs]).astype(np.float32)
      tapes = np.stack([_peek_words(rs, TAPE_WORDS) for rs in self.rs])
    outs = self._map(lambda c, s, e: c.step(a[s:e], None if noise is None else noise[s:e],
                                            None if tapes is None else tapes[s:e]))
    obs = np.concatenate([o[0] for o in outs])
    if self._rgb_observation:
      obs = self._render_rgb()
    rew = np.concatenate([o[1] for o in outs])
    cost = np.concatenate([o[2] for o in outs]).astype(np.float32)
    done = np.concatenate([o[3] for o in outs]).astype(bool)
    met = np.concatenate([o[4] for o in outs]).astype(bool)
    if self.parity_rng:
      used = np.concatenate([o[5] for o in outs])
      for rs, n in zip(self.rs, used):
        if n > TAPE_WORDS:
          raise RuntimeError('in-step random draws exceeded the tape; raise TAPE_WORDS')
        if n:
          rs.randint(0, 2**32, size=int(n), dtype=np.uint32)
    reward = rew if self._reward_dim == 2 else rew[:, 0]
    info = {'cost': cost, 'bound': self._bounds, 'goal_met': met}
    return obs, reward, done, info

  # -- device-resident path (device_buffers=True) ---------------------------------------------------
  def _dev_bufs(self):
    if self._dev is None:
      self._dev = []
      od, nu = self.robot.obs_dim, self.robot.nu
      for c, (s, e) in zip(self._ctx, self._ranges):
        n = e - s
        b = {'act': c.dev_alloc(n * nu * 4), 'obs': c.dev_alloc(n * od * 4), 'rew': c.dev_alloc(n * 2 * 4),
             'cost': c.dev_alloc(n), 'done': c.dev_alloc(n), 'met': c.dev_alloc(n)}
        if self._rgb_observation:
          b['img'] = c.dev_alloc(n * 64 * 64 * 3)
        self._dev.append(b)
    return self._dev

  def _views(self, k):
    c, (s, e), b = self._ctx[k], self._ranges[k], self._dev[k]
    n = e - s
    A = nat.DeviceArray
    if self._rgb_observation:
      obs = A(c, b['img'].value, (n, 64, 64, 3), np.uint8)
    else:
      obs = A(c, b['obs'].value, (n, self.robot.obs_dim), np.float32)
    rew = (A(c, b['rew'].value, (n, 2), np.float32) if self._reward_dim == 2 else
           A(c, b['rew'].value, (n,), np.float32, strides=(8,), base=(b['rew'].value, (n, 2))))
    return obs, rew, A(c, b['done'].value, (n,), np.uint8), A(c, b['cost'].value, (n,), np.uint8), A(c, b['met'].value, (n,), np.uint8)

  def _step_device(self, action, sync):
    bufs = self._dev_bufs()
    on_device = lambda x: isinstance(x, nat.DeviceArray) or hasattr(x, '__cuda_array_interface__')   # noqa: E731
    if isinstance(action, (list, tuple)) and len(action) == len(self._ctx) and all(on_device(x) for x in action):
      acts = list(action)          # one device array per shard
    elif on_device(action):
      if len(self._ctx) != 1:
        raise ValueError(f'{len(self._ctx)} shards: pass one device action array per shard')
      acts = [action]
    else:                          # host actions for the whole batch: split over the shards
      a = np.asarray(action, np.float32).reshape(self.n_envs, self.robot.nu)
      acts = [a[s:e] for s, e in self._ranges]
    for c, b, a in zip(self._ctx, bufs, acts):
      if on_device(a):
        d_act = nat.C.c_void_p(nat.device_pointer(a))
      else:
        c.dev_upload(b['act'], np.ascontiguousarray(a, np.float32))
        d_act = b['act']
      c.step_device(d_act, None, -1, b['obs'], b['rew'], b['cost'], b['done'], b['met'])
      if self._rgb_observation:
        c.render_rgb_device(b['img'])
    if sync:
      self.wait()
    outs = [self._views(k) for k in range(len(self._ctx))]
    one = len(outs) == 1
    pick = (lambda j: outs[0][j]) if one else (lambda j: [o[j] for o in outs])
    return pick(0), pick(1), pick(2), {'cost': pick(3), 'bound': self._bounds, 'goal_met': pick(4)}

  def wait(self):
    """Joins the contexts' streams (after step(sync=False))."""
    for c in self._ctx:
      c.wait()

  def _render_rgb(self):
    return np.concatenate(self._map(lambda c, s, e: c.render_rgb()))

  def render(self, mode='human', **options):
    """Images of every env from one of the scene's cameras, ray-cast on the device: [N, height, width, 3] uint8.
    Options as the reference passes to physics.render (render_options: camera_id 'vision' | 'fixednear' |
    'fixedfar' | 'track', height, width); defaults 'fixedfar', 256 x 256.  There is no window: mode 'human' and
    'rgb_array' both return the array (a viewer can show it)."""
    opt = dict(self._render_options)
    opt.update(options)
    cam, h, w = opt.get('camera_id', 'fixedfar'), int(opt.get('height', 256)), int(opt.get('width', 256))
    if isinstance(cam, str) and cam not in nat.Context.CAMERAS:
      raise KeyError(f'unknown camera {cam!r}: one of {sorted(nat.Context.CAMERAS)}')
    return np.concatenate(self._map(lambda c, s, e: c.render(cam, w, h, overlays=self._render_lidars_and_collision)))

  def close(self):
    for c, b in zip(self._ctx, self._dev or []):
      for p in b.values():
        c.dev_free(p)
    self._dev = None
    for c in self._ctx:
      c.close()
    if self._pool:
      self._pool.shutdown()

  # -- state access (checkpoint / tests) ---------------------------------------------
  def get_state(self):
    outs = self._map(lambda c, s, e: c.get_state())
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])

  def set_state(self, rec_f, rec_i):
    self._map(lambda c, s, e: c.set_state(rec_f[s:e], rec_i[s:e]))

  # -- internals ------------------------------------------------------------------------
  def _expand_tasks(self, task):
    if isinstance(task, type) and issubclass(task, Task):
      return [task() for _ in range(self.n_envs)]
    if isinstance(task, Task):
      return [task] + [type(task)() for _ in range(self.n_envs - 1)]
    tasks = list(task)
    assert len(tasks) == self.n_envs, 'one task per env'
    return tasks

  def _map(self, fn):
    jobs = list(zip(self._ctx, self._ranges))
    if self._pool is None:
      return [fn(c, s, e) for c, (s, e) in jobs]
    return list(self._pool.map(lambda j: fn(j[0], j[1][0], j[1][1]), jobs))

  def _build_world(self, first_episode):
    """World.sample_layout + World.reset for every env (safe_adaptation_gym.py:170-172) on the
    native sampler: env i draws from RandomState(seed_i) in the reference's order."""
    out = nat.sample_layouts(self.robot.name, self._seeds, self._desc_of_env, config=self.base_config,
                             first_episode=first_episode, want_rng=self.parity_rng, descs=self._descs)
    rf, ri, status = out[:3]
    states = out[3] if self.parity_rng else None
    if status.any():
      bad = np.flatnonzero(status)
      raise ResamplingError(f'Failed to generate layout for envs {bad[:8].tolist()} (seeds '
                            f'{self._seeds[bad[:8]].tolist()})')
    if self.parity_rng:  # continue the same streams on the host (noise, in-step draws)
      for rs, st in zip(self.rs, states):
        rs.set_state(st)
    cs = slice(nat.F_CTRL_SCALE, nat.F_CTRL_SCALE + nat.MAX_NU)
    if first_episode:  # drawn once per Task object (tasks/task.py:85-94), kept across resets
      self._ctrl_scale, self._bound0 = rf[:, cs].copy(), rf[:, nat.F_BOUND].copy()
    else:
      rf[:, cs], rf[:, nat.F_BOUND] = self._ctrl_scale, self._bound0
    if self._persist is not None:
      ri[:, nat.I_BTN_STATE] = self._persist['btn_state']
      ri[:, nat.I_CATCH_TIMER] = self._persist['catch_timer']
      rf[:, nat.F_CATCH + 2] = self._persist['catch_cur']
      rf[:, nat.F_CATCH + 3] = self._persist['catch_next']
    self._bounds = rf[:, nat.F_BOUND].copy()
    ri[:, nat.I_EPISODE] = self._episode & 0xffffff   # episode nonce of the device-side generator
    self._map(lambda c, s, e: c.set_layout(rf[s:e], ri[s:e]))

  def _pull_task_state(self):
    """Task attributes that outlive an episode in the reference because they live on
    the Task object, not in the simulator: PressButtons._state (press_buttons.py:23),
    CatchGoal radii and timer (catch_goal.py:14-18)."""
    rf, ri = self.get_state()
    self._persist = {
        'btn_state': ri[:, nat.I_BTN_STATE].copy(),
        'catch_timer': ri[:, nat.I_CATCH_TIMER].copy(),
        'catch_cur': rf[:, nat.F_CATCH + 2].copy(),
        'catch_next': rf[:, nat.F_CATCH + 3].copy(),
    }

  def _observe(self):
    if self.device_buffers:   # reset is the cold path: the observation is formed as usual and parked in the obs buffer
      bufs = self._dev_bufs()
      for c, b in zip(self._ctx, bufs):
        if self._rgb_observation:
          c.render_rgb_device(b['img'])
        else:
          c.dev_upload(b['obs'], c.observe())
      self.wait()
      outs = [self._views(k)[0] for k in range(len(self._ctx))]
      return outs[0] if len(outs) == 1 else outs
    if self._rgb_observation:
      return self._render_rgb()
    return np.concatenate(self._map(lambda c, s, e: c.observe()))


def _peek_words(rs, n):
  c = np.random.RandomState()
  c.set_state(rs.get_state())
  return c.randint(0, 2**32, size=n, dtype=np.uint32)
