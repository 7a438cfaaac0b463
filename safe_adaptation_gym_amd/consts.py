"""Constants of the world (reference safe_adaptation_gym/consts.py:9-36,
primitive_objects.py, world.py:17-34)."""
PLACEMENT_EXTENTS = (-2, -2, 2, 2)
OBSTACLES = ['hazards', 'vases', 'gremlins', 'pillars']

GROUP_INACTIVE = 0
GROUP_OBSTACLES = 1
GROUP_GOAL = 2
GROUP_OBJECTS = 3

NUM_LIDAR_BINS = 16
LIDAR_MAX_DIST = 5.

# World.DEFAULT (world.py:17-34)
WORLD_DEFAULT = {
    'placements_margin': 0.0,
    'robot_keepout': 0.4,
    'hazards_size': 0.2,
    'vases_size': 0.1,
    'pillars_size': 0.2,
    'gremlins_size': 0.1,
    'hazards_keepout': 0.18,
    'gremlins_keepout': 0.4,
    'vases_keepout': 0.15,
    'pillars_keepout': 0.3,
    'gremlins_travel': 0.35,
    'obstacles_size_noise_scale': 0.0,
    'robot_ctrl_range_scale': 0.0,
    'action_noise': 0.01,
    'max_bound': 25,
    'random_bound': False,
}
