"""safe_adaptation_gym_amd: batched SafeAdaptationGym.step() on MI355X.

`make()` mirrors the reference factory (safe_adaptation_gym/__init__.py:6-24) with
an extra `n_envs` (batch size), `devices` (GPU ordinals to shard over) and `device_buffers` (step() / reset() return
views of HBM instead of NumPy copies: envs.BatchedSafeAdaptationGym.step)."""
from typing import Dict, Optional


def make(robot_name: str,
         task_name: Optional[str] = None,
         seed: int = 666,
         config: Optional[Dict] = None,
         rgb_observation: bool = False,
         render_options: Optional[Dict] = None,
         render_lidar_and_collision=True,
         n_envs: int = 1,
         devices=None,
         parity_rng: bool = False,
         device_buffers: bool = False):
  from safe_adaptation_gym_amd.benchmark import ROBOTS_BASENAMES, TASKS
  from safe_adaptation_gym_amd.envs import BatchedSafeAdaptationGym
  env = BatchedSafeAdaptationGym(
      ROBOTS_BASENAMES[robot_name.lower()],
      n_envs=n_envs,
      config=config,
      rgb_observation=rgb_observation,
      devices=devices,
      parity_rng=parity_rng,
      device_buffers=device_buffers,
      render_lidars_and_collision=render_lidar_and_collision,
      render_options=render_options)
  env.seed(seed)
  if task_name is not None:
    env.set_task(TASKS[task_name.lower()])
  return env
