#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched SafeAdaptationGym.step() on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1: run directly)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Environments are independent, so the batch is sharded by env
index with NO collective on the data path (weak scaling: --envs per GPU is fixed);
torch.distributed (gloo, CPU tensors) is used only for the barrier and the
max-over-ranks of the timed region.  The step itself never touches torch.

A "step" = one sag_step_device() over the whole shard: action noise + clip, nstep=5
physics substeps with contact, reward (+ goal resampling), cost, 3x16 lidar, sensors,
observation write.  Inputs (layouts, actions) are resident in HBM before timing.
Prints ONE JSON line on rank 0.

Every `roofline` block names its kernels and is checked against the committed rocprofv3 summary of the SAME device
sources (profiles/kernels.json, written by tools/prof_summary.py, keyed by `src_sha16`): if that summary is of these
sources and does not contain one of the block's kernels, the block is REFUSED (replaced by the reason) - a bench
figure nobody can recompute from profiles/ is not printed.  A summary of other sources is reported as such
(`profile: null`): re-run tools/gpu_final.sh.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

# algorithmic bytes of one Point/GoToGoal env-step (fp32 SoA state):
# action 8 + robot qpos,qvel r/w 48 + 10 vases x 6 floats r/w 480 + 9 hazards, pillar,
# goal xy 88 + task scalars r/w 16 + obs 240 + reward 4 + cost 4 + done 4
ALG_BYTES_PER_ENV_STEP = 892
# SURVEY 8(d) for the other configs: Car/PushBox full step; the lidar + cost kernel alone (robot 12 + 20 obstacle
# xy 160 + goal xy 8 + 48 lidar 192 + cost 4); rgb_observation = state read + one 64x64x3 uint8 image
ALG_BYTES = {'point': 892, 'car': 804, 'lidar_cost': 376, 'render': 64 * 64 * 3 + 736}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: fp64 vector (non-matrix) peak
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak
N_ACTION_BUFS = 8
# per robot: substeps of one env-step (safe_adaptation_gym.py:15-19), the kernels of one step() in the split form and
# in the single-launch form (below SAG_SPLIT_MIN_ENVS), algorithmic bytes per env-step (None: not HBM-bound)
ROBOT_LINE = {
    'point': {'substeps': 5, 'alg': 892, 'split_min': 262144, 'split': ['k_compact', 'k_step_quiet<0', 'k_step_busy<0'], 'single': ['k_step<0']},
    'car': {'substeps': 10, 'alg': 804, 'split_min': 393216, 'split': ['k_compact', 'k_step_quiet<1', 'k_step_busy<1'], 'single': ['k_step<1']},
    'doggo': {'substeps': 12, 'alg': None, 'split_min': 1 << 62, 'split': [], 'single': ['k_doggo_physics', 'k_step_doggo_post']},
}


def step_kernels(robot, envs):
  r = ROBOT_LINE[robot]
  return r['split'] if envs >= r['split_min'] else r['single']


def source_sha16():
  """Hash of the device sources: keys profiles/traffic.json to the code it was measured on."""
  import glob
  import hashlib
  h = hashlib.sha256()
  for f in sorted(glob.glob(os.path.join(ROOT, 'safe_adaptation_gym_amd', 'csrc', '*')) + [os.path.join(ROOT, 'include', 'sag.h')]):
    h.update(open(f, 'rb').read())
  return h.hexdigest()[:16]


def sampler_threads():
  """Host threads one rank gives the native layout sampler: the box's cores divided among the ranks that share it (on an
  8-GPU node every rank samples its own shard before the first barrier: 8 x 16 threads on one host was minutes of skew)."""
  world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
  return max(1, min(16, len(os.sched_getaffinity(0)) // max(world, 1)))


def build_records(task, envs_per_gpu, rank, seed=666, robot='point'):
  """Layouts with the reference sampler semantics (native sampler, exact numpy-legacy stream):
  global env g is drawn with RandomState(seed + g) exactly as make(robot, task, seed + g)
  would, and g is also its RNG stream id on the device."""
  from safe_adaptation_gym_amd import _native as nat
  from safe_adaptation_gym_amd import benchmark
  g0 = rank * envs_per_gpu
  if task == 'multitask':
    # BASELINE config 4: task of env g from the benchmark's TaskSampler order (task_sampler.py:15-19)
    names = [nm for nm, _ in benchmark.make('multitask', batch_size=envs_per_gpu, seed=seed + rank).train_tasks]
    tids = np.array([benchmark.TASKS[nm].TASK_ID for nm in names], np.int32)
  else:
    tids = benchmark.TASKS[task].TASK_ID
  rf, ri, status = nat.sample_layouts(robot, seed + g0 + np.arange(envs_per_gpu, dtype=np.int64), tids, env_id0=g0, nthreads=sampler_threads())
  assert not status.any(), 'layout sampling failed'
  return rf, ri


class DeviceRun:
  """Everything resident on the GPU: world, action buffers, output buffers."""

  def __init__(self, task, envs, device, rank, seed=666, robot='point'):
    from safe_adaptation_gym_amd import _native as nat
    self.nat = nat
    rf, ri = build_records(task, envs, rank, seed, robot)
    has_btn = int(ri[:, nat.I_NB].max()) > 0
    has_box = int(ri[:, nat.I_BOX_KIND].max()) > 0
    self.ctx = nat.Context(robot, envs, device=device, seed=seed,
                           max_buttons=nat.MAX_BUTTONS if has_btn else 0, has_box=has_box)
    self.ctx.set_layout(rf, ri)
    self.envs = envs
    od = self.ctx.info['obs_dim']
    self.d_act = [self.ctx.dev_alloc(envs * self.ctx.info['nu'] * 4) for _ in range(N_ACTION_BUFS)]
    for k, b in enumerate(self.d_act):
      self.ctx.dev_fill_actions(b, k)
    self.d_obs = self.ctx.dev_alloc(envs * od * 4)
    self.d_rew = self.ctx.dev_alloc(envs * 2 * 4)
    self.d_cost = self.ctx.dev_alloc(envs)
    self.d_done = self.ctx.dev_alloc(envs)
    self.d_met = self.ctx.dev_alloc(envs)
    self.ctx.wait()
    self.t = 0

  def burn_in(self, steps):
    """Untimed ageing: the cost of a step depends on the state's age (how many vases the robots have set in motion:
    the busy fraction keeps drifting - 0.11 at step 225, 0.08 at step 420 of the headline run), so the line states
    the burn-in and the busy fraction at its first and last timed step instead of calling the load stationary."""
    self.run(steps)
    self.ctx.wait()

  def busy_fraction(self):
    self.ctx.wait()
    return self.ctx.busy_count() / self.envs

  def step(self):
    c = self.ctx
    c.step_device(self.d_act[self.t % N_ACTION_BUFS], None, -1, self.d_obs, self.d_rew, self.d_cost,
                  self.d_done, self.d_met)
    self.t += 1

  def run(self, steps):
    for _ in range(steps):
      self.step()

  def wait(self):
    self.ctx.wait()

  def timing(self, on):
    self.ctx.enable_timing(on)
    self.ctx.kernel_time_ms(reset=True)

  def kernel_time_ms(self):
    return self.ctx.kernel_time_ms(reset=True)

  def stats(self):
    cost = self.ctx.dev_download(self.d_cost, (self.envs,), np.uint8)
    done = self.ctx.dev_download(self.d_done, (self.envs,), np.uint8)
    obs = self.ctx.dev_download(self.d_obs, (min(self.envs, 4096) * self.ctx.info['obs_dim'],), np.float32)
    self.met_rate = float(self.ctx.dev_download(self.d_met, (self.envs,), np.uint8).mean())
    self.busy_frac = self.ctx.busy_count() / self.envs
    return float(cost.mean()), int(done.sum()), bool(np.isfinite(obs).all())

  def close(self):
    self.ctx.close()


def timed(run, steps, warmup, barrier):
  """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides.  Returns THIS rank's time
  from the opening barrier to the end of its own K steps (device synchronised): the clock stops BEFORE the closing
  barrier, so that what the caller max-reduces over the ranks is the slowest rank's work, not that plus the latency of
  a gloo barrier - 1 ms of it would read as a 5 % scaling loss on the driver's 18-ms timed window."""
  run.run(warmup)
  run.wait()
  if hasattr(run, 'busy_fraction'):
    run.busy_first = run.busy_fraction()   # load at the first timed step (untimed read-back)
  barrier()
  t0 = time.perf_counter()
  run.run(steps)
  run.wait()
  t1 = time.perf_counter()
  barrier()
  return t1 - t0


_PROFILE = None


def profile_kernels():
  """profiles/kernels.json: {src_sha16, summary, kernels: {name: {calls, avg_us}}, doggo_flops_per_env_step: {...}}."""
  global _PROFILE
  if _PROFILE is None:
    try:
      _PROFILE = json.load(open(os.path.join(ROOT, 'profiles', 'kernels.json')))
    except (OSError, ValueError):
      _PROFILE = {}
  return _PROFILE


def checked(block, kernels):
  """Attach the committed profile's figures for the block's kernels; refuse the block if the profile is of these
  sources and lacks one of them (see the module docstring)."""
  prof = profile_kernels()
  if prof.get('src_sha16') != source_sha16():
    block['profile'] = None
    block['profile_note'] = 'profiles/kernels.json was measured on other device sources (or is absent): nothing to check this block against'
    return block
  found = {}
  for k in kernels:
    hit = [n for n in prof.get('kernels', {}) if k in n]
    if not hit:
      return {'refused': f'kernel {k!r} is not in {prof.get("summary")} (src_sha16 {prof.get("src_sha16")}): this block cannot be recomputed from profiles/',
              'kernel': block.get('kernel')}
    found[k] = {n: prof['kernels'][n] for n in hit}
  block['profile'] = {'summary': prof.get('summary'), 'src_sha16': prof.get('src_sha16'), 'kernels': found}
  return block


def doggo_roofline(ms, envs):
  """Doggo: not an HBM kernel.  Vector-ALU utilisation from the issued lane-operations per env-step that rocprofv3 counted
  on these sources (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 / _F32 x 64 lanes, FMA = 2; profiles/kernels.json):
  time the fp64 operations need at the 78.6 TFLOP/s fp64 vector peak + time the fp32 ones need at 157.3, over the step time."""
  fl = profile_kernels().get('doggo_flops_per_env_step') if profile_kernels().get('src_sha16') == source_sha16() else None
  blk = {'bound': 'vector-alu (fp64 + fp32)', 'kernel': 'sag::k_doggo_physics + k_step_doggo_post', 'kernel_ms': ms, 'traffic': None,
         'units_per_launch': envs,
         'note': 'chains of dependent fp64 (kinematics, mass matrix, Cholesky, rows) and fp32 (projected Gauss-Seidel) operations on 2 envs per '
                 'wavefront, one wavefront per SIMD: latency-bound, far from any roofline'}
  if fl:
    t64, t32 = fl['fp64'] * envs / (FP64_VECTOR_PEAK_TFLOPS * 1e12), fl['fp32'] * envs / (FP32_VECTOR_PEAK_TFLOPS * 1e12)
    blk.update({'achieved': (fl['fp64'] + fl['fp32']) * envs / (ms * 1e-3) / 1e12, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': (t64 + t32) / (ms * 1e-3), 'flops_per_env_step': fl})
  else:
    blk.update({'achieved': None, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': None,
                'flops_per_env_step': None, 'profile_note': 'no operation counts for these sources in profiles/kernels.json'})
  return checked(blk, ['k_doggo_physics', 'k_step_doggo_post'])


def traffic_per_launch(envs, key='point'):
  """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json:
  FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc runs of this same command), scaled to
  this launch's env count.  None if no profile has been committed FOR THESE SOURCES: the file carries the
  hash of the device sources it was measured on (tools/prof_steady.py) and a stale one is not quoted."""
  try:
    t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    if t.get('src_sha16') != source_sha16():
      return None
    return t['configs'][key]['bytes_per_env_step'] * envs
  except (OSError, KeyError, ValueError, TypeError):
    return None


def roofline_block(alg_bytes, envs, kernel_ms, kernels, key=None, launches=None):
  """kernels: the names (substrings of the profile's kernel names) whose launches make up one timed unit."""
  achieved = alg_bytes * envs / (kernel_ms * 1e-3) / 1e9
  return checked({'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                  'traffic': traffic_per_launch(envs, key) if key else None, 'kernel': ' + '.join('sag::' + k for k in kernels),
                  'kernel_ms': kernel_ms, 'launches_timed': launches, 'alg_bytes_per_unit': alg_bytes, 'units_per_launch': envs}, kernels)


def cpu_baseline(task, seconds=12.0, robot='point', n=4096, single_core=True):
  """The CPU oracle (oracle/sag_oracle.c, fp64, OpenMP over envs) on this host: same
  layouts, same counter-based actions/noise.  A reported baseline, not the target."""
  from oracle_lib import Oracle
  o = Oracle()
  rid, nu, od = {'point': (0, 2, 60), 'car': (1, 2, 72), 'doggo': (2, 12, 104)}[robot]
  rf, ri = build_records(task, n, 0, robot=robot)
  arr = o.make_batch(rf, ri)
  acts = np.stack([[o.actions((666, 0), int(ri[i, 12]), s, nu) for i in range(n)] for s in range(4)])
  cores = min(len(os.sched_getaffinity(0)), 16)  # a 1-GPU box is given 16 host cores
  out = {}
  for label, nt in ((('1', 1),) if single_core else ()) + (('all', cores),):
    o.step_batch(arr, rid, acts[0], key=(666, 0), nthreads=nt, obs_dim=od)  # warm
    t0 = time.perf_counter()
    done_steps = 0
    budget = seconds * ((0.35 if nt == 1 else 0.65) if single_core else 1.0)
    while time.perf_counter() - t0 < budget:
      o.step_batch(arr, rid, acts[done_steps % 4], key=(666, 0), nthreads=nt, obs_dim=od)
      done_steps += 1
    dt = time.perf_counter() - t0
    out[label] = (n * done_steps / dt, done_steps, dt)
  v, k, dt = out['all']
  res = {
      'value': v,
      'unit': 'env-steps/s',
      'cores': cores,
      'kind': 'port',
      'sample': f'{robot}/{task}, {n} envs x {k} steps in {dt:.1f}s, oracle/sag_oracle.c fp64 -O2 OpenMP {cores} threads'
                + (f' (1 thread: {out["1"][0]:.0f} env-steps/s)' if single_core else '') + '; CPU restatement, not MuJoCo',
  }
  if single_core:
    res['single_core_value'] = out['1'][0]
  return res


def main(argv=None, run_factory=None, emit=print):
  """run_factory(task, envs, device, rank) -> run object; tests inject a CPU stand-in to
  exercise the sharding / barrier / max-over-ranks harness under gloo without a GPU."""
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  # 4 M envs per GPU (4 GB of device state): above ~2 M the busy and the quiet kernel both keep the chip
  # full and throughput is ~12 % above the 1 M-env figure (profiles/r01_v16_batch_size_sweep.txt)
  ap.add_argument('--envs', type=int, default=1 << 22, help='environments per GPU (weak scaling)')
  ap.add_argument('--task', default='go_to_goal')
  ap.add_argument('--robot', default='point', help='point (headline) | car | doggo: profile another config as the main line')
  ap.add_argument('--burn-in', type=int, default=200, help='untimed steps before warmup (the contact load depends on the state age: see checks.state_age_steps)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0, help='CPU time spent on the cpu_baseline sample')
  ap.add_argument('--no-c2', action='store_true', help='skip the 4096-env (BASELINE config 2 size) line and the other single-GPU config lines')
  ap.add_argument('--no-c4', action='store_true', help='skip the sharded Doggo / multitask line (BASELINE config 4)')
  ap.add_argument('--c4-envs', type=int, default=4096, help='Doggo envs per GPU of the config-4 line')
  ap.add_argument('--c4-steps', type=int, default=30)
  args = ap.parse_args(argv)

  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      sys.exit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N '
               '--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...')
    sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
  dist = None
  if world > 1:
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # the data path has no collective; gloo on CPU tensors carries the barrier and the max
    dist.init_process_group('gloo', rank=rank, world_size=world)

  def barrier():
    if dist is not None:
      dist.barrier()

  def max_over_ranks(x):
    if dist is None:
      return x
    import torch
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  def gather_ranks(x):
    if dist is None:
      return [x]
    import torch
    t = torch.zeros(world, dtype=torch.float64)
    t[rank] = x
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t]

  if run_factory is None:
    from safe_adaptation_gym_amd import _native as nat
    ndev = nat.device_count()
    if ndev < 1:
      sys.exit('no HIP device visible (there is no CPU fallback)')
    device = local % ndev
    run_factory = DeviceRun
  else:
    device = local

  run = run_factory(args.task, args.envs, device, rank) if args.robot == 'point' else run_factory(
      args.task, args.envs, device, rank, robot=args.robot)
  run.burn_in(args.burn_in)
  # timed region with per-launch HIP events on the context stream (kernel time for the roofline)
  run.timing(True)
  elapsed = max_over_ranks(timed(run, args.steps, args.warmup, barrier))
  k_ms_rank, k_n = run.kernel_time_ms()
  # kernel_time covers warmup + timed launches; both are the same kernel on the same data
  run.timing(False)
  k_ms = max_over_ranks(k_ms_rank)            # the roofline is quoted on the slowest rank's kernel time
  k_ms_per_rank = gather_ranks(k_ms_rank)
  cost_rate, n_done, finite = run.stats()
  total_env_steps = world * args.envs * args.steps
  value = total_env_steps / elapsed
  line = ROBOT_LINE[args.robot]
  kernels = step_kernels(args.robot, args.envs)
  if line['alg'] is None:   # Doggo: no HBM fraction is quoted for a kernel that is not memory-bound
    roof = doggo_roofline(k_ms, args.envs)
  else:
    roof = roofline_block(line['alg'], args.envs, k_ms, kernels, args.robot if args.envs >= line['split_min'] else None, k_n)
    if 'refused' not in roof:
      roof['kernel'] += ' (one step() = these launches)'
      roof['alg_bytes_per_env_step'] = line['alg']
  roof['kernel_ms_per_rank'] = k_ms_per_rank
  roof['src_sha16'] = source_sha16()
  res = {
      'metric': 'env-steps/sec (batched) Point/GoToGoal' if args.robot == 'point' else f'env-steps/sec (batched) {args.robot}/{args.task}',
      'value': value,
      'unit': 'env-steps/s',
      'n_gpus': world,
      'steps': args.steps,
      'warmup': args.warmup,
      'ms_per_step': elapsed / args.steps * 1e3,
      'higher_is_better': True,
      'scaling': 'weak',
      'vs_baseline': None,
      'dtype': 'f32',
      'data': 'synthetic',
      'config': {
          'workload': f'{args.robot}/{args.task} full step() ({line["substeps"]} substeps + contact + reward + cost + lidar + obs), '
                      f'one layout per env from the reference sampler semantics (env i <- RandomState(666 + i)), '
                      f'{args.envs} envs per GPU, counter-based actions U(-1,1) and noise 0.01 on device; '
                      f'{args.burn_in} untimed burn-in steps + {args.warmup} warmup steps before the timed region',
          'burn_in': args.burn_in,
          'envs_per_gpu': args.envs,
          'global_envs': world * args.envs,
          'env_id_ranges_per_rank': [[r * args.envs, (r + 1) * args.envs] for r in range(world)],
          'parallelism': f'env-sharded x{world}, no collective',
      },
      'roofline': roof,
      'checks': {'cost_rate_last_step': cost_rate, 'done_envs': n_done, 'obs_finite': finite,
                 'goal_met_rate_last_step': getattr(run, 'met_rate', None),
                 'busy_env_fraction_first_timed_step': getattr(run, 'busy_first', None),
                 'busy_env_fraction_last_step': getattr(run, 'busy_frac', None),
                 'state_age_steps': [args.burn_in + args.warmup, args.burn_in + args.warmup + args.steps],
                 'spread_note': ('the same library measures this step between 0.81 and 0.99 ms on MI355X boxes of one pool: the value comes with the allocation and is '
                                 'constant within a context (DESIGN.md 9, tools/alloc_spread.py, tools/clock_probe.py); compare builds interleaved in one call (tools/ab.sh run -r N)')
                                if args.robot == 'point' and args.envs == 1 << 22 else None},
  }
  run.close()

  # BASELINE config 4 as it is stated: Doggo, multitask sampler, 4096 envs PER GPU, sharded by env index over the
  # ranks (8 x 4096 = 32768 on a node) - emitted at every world size; max over ranks like the headline
  if not args.no_c4:
    n4, k4, w4 = args.c4_envs, args.c4_steps, 5
    r4 = run_factory('multitask', n4, device, rank, robot='doggo')
    r4.burn_in(20)
    r4.timing(True)
    t4 = max_over_ranks(timed(r4, k4, w4, barrier))
    ms4_rank, _ = r4.kernel_time_ms()
    ms4 = max_over_ranks(ms4_rank)
    res['c4_doggo_multitask'] = {
        'value': world * n4 * k4 / t4, 'unit': 'env-steps/s', 'n_gpus': world, 'envs_per_gpu': n4, 'global_envs': world * n4,
        'env_id_ranges_per_rank': [[r * n4, (r + 1) * n4] for r in range(world)], 'steps': k4, 'warmup': w4, 'burn_in': 20,
        'ms_per_step': t4 / k4 * 1e3, 'kernel_ms': ms4, 'kernel_ms_per_rank': gather_ranks(ms4_rank), 'scaling': 'weak',
        'note': 'BASELINE config 4: task of global env g from the benchmark TaskSampler order; whole-job env-steps over the max-over-ranks '
                'time; 12 substeps, warm-started PGS 24 sweeps (48 cold)',
        'roofline': doggo_roofline(ms4, n4)}
    r4.close()

  if rank == 0 and world == 1 and run_factory is DeviceRun:
    if not args.no_c2:
      c2 = DeviceRun(args.task, 4096, device, 0)
      c2.burn_in(args.burn_in)
      c2.timing(True)
      t = timed(c2, max(args.steps, 200), args.warmup, lambda: None)
      ms, _ = c2.kernel_time_ms()
      res['c2_4096_envs'] = {
          'value': 4096 * max(args.steps, 200) / t,
          'unit': 'env-steps/s',
          'ms_per_step': t / max(args.steps, 200) * 1e3,
          'kernel_ms': ms,
          'roofline': roofline_block(ALG_BYTES['point'], 4096, ms, step_kernels('point', 4096)),
          'note': 'BASELINE config-2 batch size (4096 envs on one GPU): latency bound, 16 envs per wavefront x 256 wavefronts'
      }
      # the lidar + hazard-cost kernel alone on explicit poses (BASELINE config 2 "lidar + cost only"), device
      # buffers, kernel-only HIP-event time: at the config's 4096 poses (launch-bound) and at 4 M (roofline)
      lc = {}
      nK = 21
      for n_lc in (4096, 1 << 22):
        rs = np.random.RandomState(0)
        rob = np.concatenate([rs.uniform(-2, 2, (n_lc, 2)), rs.uniform(-np.pi, np.pi, (n_lc, 1))], 1).astype(np.float32)
        pts = rs.uniform(-2.5, 2.5, (n_lc, nK, 2)).astype(np.float32)
        grp = np.tile(np.array([1 + 128] * 8 + [1] * 12 + [2], np.uint8), (n_lc, 1))   # 8 hazards (+128: cost-tested), 12 vases / pillars, the goal
        cx = nat.Context('point', 64, device=device)
        d = [cx.dev_alloc(x.nbytes) for x in (rob, pts, grp)]
        for dp, x in zip(d, (rob, pts, grp)):
          cx.dev_upload(dp, x)
        d_lid, d_cost = cx.dev_alloc(n_lc * 48 * 4), cx.dev_alloc(n_lc)
        for _ in range(5):
          cx.lidar_cost_device(n_lc, nK, d[0], d[1], d[2], d_lid, None, d_cost)
        cx.wait()
        cx.enable_timing(True)
        cx.kernel_time_ms(reset=True)
        reps = 200 if n_lc == 4096 else 30
        t0 = time.perf_counter()
        for _ in range(reps):
          cx.lidar_cost_device(n_lc, nK, d[0], d[1], d[2], d_lid, None, d_cost)
        cx.wait()
        t_lc = (time.perf_counter() - t0) / reps
        ms, cnt = cx.kernel_time_ms(reset=True)
        lc[str(n_lc)] = {'value': n_lc / t_lc, 'unit': 'env-evaluations/s', 'ms_per_call': t_lc * 1e3,
                         'roofline': roofline_block(ALG_BYTES['lidar_cost'], n_lc, ms, ['k_lidar_cost_team<16>' if n_lc <= 16384 else 'k_lidar_cost_team<4>'], 'lidar_cost', cnt)}
        cx.close()
      lc['note'] = ('sag_lidar_cost_device on n poses x 21 points resident in HBM, kernel-only time; bit-exact bins and cost flags vs the reference fixtures. '
                    '4096 poses: k_lidar_cost_team<16> (16 lanes share a pose; [bin][pose] ds_max tile); 4 M poses: k_lidar_cost_team<4> (4 lanes share a pose; 3-KB tile, eight wavefronts per SIMD)')
      res['c2_lidar_cost_only'] = lc
      c2.close()
    if not args.no_c2:
      # BASELINE config 3 (Car / push_box), 4096 envs and a loaded batch; 804 algorithmic B/env-step
      c3 = {}
      for n_c3 in (4096, 1 << 22):   # the config's own batch (latency-bound) and a chip-filling one (split form)
        r3 = DeviceRun('push_box', n_c3, device, 0, robot='car')
        r3.burn_in(60)
        r3.timing(True)
        k3 = 100 if n_c3 == 4096 else 30
        t = timed(r3, k3, 10, lambda: None)
        ms, cnt = r3.kernel_time_ms()
        c3[str(n_c3)] = {'value': n_c3 * k3 / t, 'unit': 'env-steps/s', 'ms_per_step': t / k3 * 1e3,
                         'roofline': roofline_block(ALG_BYTES['car'], n_c3, ms, step_kernels('car', n_c3),
                                                    'car' if n_c3 >= ROBOT_LINE['car']['split_min'] else None, cnt)}
        r3.close()
      if not args.no_cpu_baseline:   # the oracle on the same batch: a few seconds each
        c3['cpu_baseline'] = cpu_baseline('push_box', 3.0, robot='car', single_core=False)
        if 'c4_doggo_multitask' in res:
          res['c4_doggo_multitask']['cpu_baseline'] = cpu_baseline('multitask', 5.0, robot='doggo', n=args.c4_envs, single_core=False)
      res['c3_car_push_box'] = c3
      # BASELINE config 5 (stretch): Doggo / haul_box with rgb_observation: step + 64x64x3 render per env
      r5 = DeviceRun('haul_box', 4096, device, 0, robot='doggo')
      d_img = r5.ctx.dev_alloc(4096 * 64 * 64 * 3)
      r5.burn_in(10)
      import time as _t
      r5.ctx.render_rgb_device(d_img); r5.wait()
      t0 = _t.perf_counter()
      for _ in range(20):
        r5.ctx.render_rgb_device(d_img)
      r5.wait()
      t_render = (_t.perf_counter() - t0) / 20
      t = timed(r5, 20, 2, lambda: None)
      res['c5_doggo_haul_box_rgb_4096'] = {'render_ms': t_render * 1e3, 'step_ms': t / 20 * 1e3,
                                           'value': 4096 / (t / 20 + t_render), 'unit': 'env-steps/s',
                                           'roofline_render': roofline_block(ALG_BYTES['render'], 4096, t_render * 1e3, ['k_render_rgb']),
                                           'note': 'per GPU: one step + one 64x64x3 uint8 first-person image per env; the ray caster '
                                                   'is fp64 compute per pixel (a hard decision like a lidar bin), not memory-bound'}
      r5.close()
    if not args.no_cpu_baseline:
      res['cpu_baseline'] = cpu_baseline(args.task, args.cpu_baseline_seconds)
  if rank == 0:
    emit(json.dumps(res))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
