"""Static robot descriptors.  The reference's Robot (robot.py:8-62) loads the MJCF
into MuJoCo to introspect it; the facts it records are constants of the three XMLs
(assets/xmls/{point,car,doggo}.xml), restated here, with nu/obs_dim/nstep/dt taken
from the native library's table so host and device agree."""
from safe_adaptation_gym_amd import _native

_BASE = 'xmls/'
ROBOTS_BASENAMES = {name: _BASE + name + '.xml' for name in ['point', 'car', 'doggo']}

_GEOMS = {
    'point': {'robot', 'pointarrow'},
    'car': {'robot', 'back_bumper', 'back_connector', 'front_bumper', 'front_connector', 'left',
            'right', 'rear'},
    'doggo': {'robot', 'robot2'} | {f'{p}_{i}' for p in ('aux', 'hip', 'ankle') for i in range(1, 5)},
}
_Z = {'point': 0.1, 'car': 0.1, 'doggo': 0.22}
_DOGGO_JOINTS = ([f'hip_{i}_z' for i in range(1, 5)] + [f'hip_{i}_y' for i in range(1, 5)] +
                 [f'ankle_{i}' for i in range(1, 5)])


def _get_robot_name(path):
  import os
  return os.path.splitext(os.path.basename(path))[0]


class Robot:
  """Same attributes as the reference's Robot: name, z_height, geom_names, nq, nv, nu,
  hinge_pos_names, hinge_vel_names, ballquat_names, ballangvel_names."""

  def __init__(self, path):
    self.base_path = path
    self.name = _get_robot_name(path)
    if self.name not in _Z:
      raise ValueError(f'unknown robot {path!r}')
    info = _native.robot_info(self.name)
    self.z_height = _Z[self.name]
    self.geom_names = set(_GEOMS[self.name])
    self.nq, self.nv, self.nu = info['nq'], info['nv'], info['nu']
    self.obs_dim, self.nstep, self.dt = info['obs_dim'], info['nstep'], info['dt']
    self.hinge_pos_names, self.hinge_vel_names = [], []
    self.ballquat_names, self.ballangvel_names = [], []
    if self.name == 'car':
      self.ballquat_names, self.ballangvel_names = ['ballquat_rear'], ['ballangvel_rear']
    if self.name == 'doggo':
      self.hinge_pos_names = ['jointpos_' + j for j in _DOGGO_JOINTS]
      self.hinge_vel_names = ['jointvel_' + j for j in _DOGGO_JOINTS]
