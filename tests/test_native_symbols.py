"""The C-ABI library loads and exports every symbol include/sag.h declares (no GPU needed,
no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  src = open(os.path.join(ROOT, 'include', 'sag.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(sag_[a-z_0-9]+)\s*\(', src)))


def test_header_and_binding_agree():
  from safe_adaptation_gym_amd import _native
  assert sorted(_native.EXPORTS) == declared_symbols()


def test_library_exports_every_declared_symbol():
  from safe_adaptation_gym_amd import _native
  lib = _native.load()
  for name in declared_symbols():
    assert hasattr(lib, name), f'libsag.so does not export {name}'


def test_robot_table_without_gpu():
  from safe_adaptation_gym_amd import _native
  assert _native.robot_info('point') == dict(nu=2, obs_dim=60, nstep=5, nq=3, nv=3, dt=0.004)
  assert _native.robot_info('car') == dict(nu=2, obs_dim=72, nstep=10, nq=13, nv=11, dt=0.008)
  assert _native.robot_info('doggo') == dict(nu=12, obs_dim=104, nstep=12, nq=20, nv=19, dt=0.012)


def test_record_offsets_match_header():
  """Python-side offsets (package and test helpers) are the header's enum values."""
  src = open(os.path.join(ROOT, 'include', 'sag.h')).read()
  vals = dict(re.findall(r'\b(SAG_[FI]_[A-Z0-9_]+|SAG_REC_[A-Z]+)\s*=\s*(\d+)', src))
  from safe_adaptation_gym_amd import _native as n
  import oracle_lib as o
  for mod in (n, o):
    for py, c in [('F_ROBOT', 'SAG_F_ROBOT'), ('F_ROBOT0', 'SAG_F_ROBOT0'), ('F_GEAR', 'SAG_F_GEAR'),
                  ('F_DAMP', 'SAG_F_DAMP'), ('F_ACTION_NOISE', 'SAG_F_ACTION_NOISE'),
                  ('F_CTRL_SCALE', 'SAG_F_CTRL_SCALE'), ('F_HAZARD_SIZE', 'SAG_F_HAZARD_SIZE'),
                  ('F_VASE_SIZE', 'SAG_F_VASE_SIZE'), ('F_PILLAR_SIZE', 'SAG_F_PILLAR_SIZE'),
                  ('F_KEEPOUT', 'SAG_F_KEEPOUT'), ('F_GOAL', 'SAG_F_GOAL'), ('F_CATCH', 'SAG_F_CATCH'),
                  ('F_LAST', 'SAG_F_LAST'), ('F_BOX', 'SAG_F_BOX'), ('F_HAZARDS', 'SAG_F_HAZARDS'),
                  ('F_PILLARS', 'SAG_F_PILLARS'), ('F_BUTTONS', 'SAG_F_BUTTONS'), ('F_VASES', 'SAG_F_VASES'),
                  ('F_ROBOT_EXT', 'SAG_F_ROBOT_EXT'),
                  ('REC_FLOATS', 'SAG_REC_FLOATS'), ('REC_INTS', 'SAG_REC_INTS'),
                  ('I_TASK', 'SAG_I_TASK'), ('I_NB', 'SAG_I_NB'), ('I_BOX_KIND', 'SAG_I_BOX_KIND'),
                  ('I_GOAL_BUTTON', 'SAG_I_GOAL_BUTTON'), ('I_ACTIVE_MASK', 'SAG_I_ACTIVE_MASK'),
                  ('I_STEP', 'SAG_I_STEP'), ('I_ENV_ID', 'SAG_I_ENV_ID'), ('I_FLAGS', 'SAG_I_FLAGS'),
                  ('I_EPISODE', 'SAG_I_EPISODE')]:
      assert getattr(mod, py) == int(vals[c]), (mod.__name__, py)


def test_product_never_imports_the_oracle():
  """The oracle is test infrastructure: nothing under the package may reference it."""
  pkg = os.path.join(ROOT, 'safe_adaptation_gym_amd')
  for dp, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
        txt = open(os.path.join(dp, f)).read()
        if not f.endswith('.py'):   # device / host C++: comments may cite the specification's file, code may not
          txt = re.sub(r'//[^\n]*', '', re.sub(r'/\*.*?\*/', '', txt, flags=re.S))
        assert 'oracle' not in txt.lower(), f'{f} references the oracle outside a comment'


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
  import importlib
  from safe_adaptation_gym_amd import _native
  monkeypatch.setattr(_native, '_lib', None)
  monkeypatch.setattr(_native, 'LIB_PATH', str(tmp_path / 'nope.so'))
  with pytest.raises(_native.SagError, match='no CPU fallback'):
    _native.load()
  monkeypatch.undo()
  importlib.reload(_native)
