"""Builds of the library's OWN device sources for the host (tests/hostemu: a launch runs its workgroups one after the
other, lanes are fibers) compared with each other - checks that need no GPU and that the `-m gpu` suite cannot make,
because they compare two BUILDS:
  * the separating-axis pair mask of the narrowphase (collide_list_nb) against the plain geom-by-geom loop
    (-DSAG_COLLIDE_REF): bit-identical Car / push_box trajectories (an edit to the mask's margins could drop a
    real contact and only a tolerance-based lockstep case that happens to hit the pair would notice);
  * the 64-lane projected Gauss-Seidel path of the Doggo kernel (envs with more than 32 constraint rows - rare), forced
    onto ordinary states by -DSAG_DC_FAST_ROWS=8, against the 32-lane path, one step at a time from identical state.
The builds are unsanitized (-O1, ~30 s each, in parallel); the sanitizer builds stay a manual tool (tests/hostemu/run.sh)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'hostemu'))
VARIANTS = {'base': [], 'cref': ['-DSAG_COLLIDE_REF'], 'wide': ['-DSAG_DC_FAST_ROWS=8']}


@pytest.fixture(scope='module')
def libs():
  import build as hb   # tests/hostemu/build.py
  if not os.path.exists(hb.CLANG):
    pytest.skip('no clang for the host build')
  with ThreadPoolExecutor(3) as ex:
    futs = {k: ex.submit(hb.build, 'clang', False, False, v, False, False, 'var_' + k) for k, v in VARIANTS.items()}
    return {k: f.result() for k, f in futs.items()}


def _traj(lib, out, *args):
  env = dict(os.environ, SAG_LIB=lib, SAG_HOSTEMU='1', PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'))
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'diag_traj.py'), out, *args], env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
  return np.load(out)


def test_pair_mask_equals_plain_loop_bitwise(libs, tmp_path):
  a = _traj(libs['base'], str(tmp_path / 'a.npz'), 'car', 'push_box', '96', '40')
  b = _traj(libs['cref'], str(tmp_path / 'b.npz'), 'car', 'push_box', '96', '40')
  assert a['outs'][:, :, -3].sum() > 10, 'the rollout should contain contacts with obstacles'
  for key in ('states', 'ints', 'outs'):
    np.testing.assert_array_equal(a[key], b[key], err_msg=key)


def test_wide_pgs_path_matches_the_32_lane_path(libs, tmp_path):
  base = str(tmp_path / 'a.npz')
  a = _traj(libs['base'], base, 'doggo', 'haul_box', '24', '8')
  b = _traj(libs['wide'], str(tmp_path / 'b.npz'), 'doggo', 'haul_box', '24', '8', '--replay', base)
  assert np.isfinite(a['states']).all() and np.isfinite(b['states']).all()
  E = 144
  pos = [0, 1, 2, E] + list(range(E + 1, E + 5)) + list(range(E + 9, E + 22))
  tol = np.full(a['states'].shape[-1], 2e-3)
  tol[pos] = 2e-5
  d = np.abs(a['states'] - b['states'])
  bad = (d > tol + tol * np.abs(a['states'])).any(-1)
  # (the two paths sum the same products in another order: rounding only; a threshold event may flip a contact)
  assert bad.mean() <= 0.02, f'{bad.sum()} of {bad.size} env-steps differ beyond the lockstep tolerance'
  np.testing.assert_array_equal(a['ints'][~bad], b['ints'][~bad])


def test_front_end_api_on_the_emulated_device(libs):
  """The NumPy front end (make / reset / step / set_task / seed / get_state, envs.py and _native.py) and the C ABI's error
  and edge behaviour, exercised WITHOUT a GPU: the API cases of the `-m gpu` suite run in a subprocess against the host build
  of the device sources.  (A checker, not a fallback: the product loader opens libsag.so only and reports no device here;
  the host build is selected by the test through SAG_LIB + SAG_HOSTEMU.)"""
  env = dict(os.environ, SAG_LIB=libs['base'], SAG_HOSTEMU='1', PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'))
  pick = ('test_env_api_shapes or test_car_env_api or state_roundtrip or physics_error_is_data or parity_rng_mode or '
          'maximum_capacity or lidar_cost_empty or lidar_cost_kernel_golden')
  r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-m', 'gpu', '-q', '-x',
                      '-p', 'no:cacheprovider', '-k', pick], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  tail = r.stdout[-1500:] + r.stderr[-1500:]
  assert r.returncode == 0, tail
  import re
  m = re.search(r'(\d+) passed', r.stdout)
  assert m and int(m.group(1)) >= 8 and 'skipped' not in r.stdout.splitlines()[-1], tail
