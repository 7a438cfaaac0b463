/*
 * sag.h - C ABI of libsag.so: batched SafeAdaptationGym.step() on MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path.  The reference has no FFI; its
 * seam is the Python object MujocoBridge (reference safe_adaptation_gym/
 * mujoco_bridge.py:15-280) as consumed by SafeAdaptationGym.step/reset/observation
 * (safe_adaptation_gym.py:56-139), World (world.py:139-165,219-231) and the tasks'
 * compute_reward/reset/set_mocaps (tasks/<task>.py).  Each entry point below names the
 * reference calls it replaces.  Plain pointers and sizes only; no exceptions cross
 * the ABI; every function returns 0 on success and a negative sag_status on error,
 * with a message available from sag_last_error().
 *
 * Ownership: the caller owns every host buffer it passes; the library owns all
 * device memory.  One context per GPU, one HIP stream per context.  A context is
 * not thread-safe; distinct contexts are independent (one host thread each).
 *
 * Per-env physics failure (non-finite state, the reference's PhysicsError branch,
 * safe_adaptation_gym.py:73-75) is reported as DATA (done=1, reward=-10, cost=0),
 * not as an error code.
 */
#ifndef SAG_H_
#define SAG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAG_ABI_VERSION 5

/* ---- capacities (maxima over the reference's task set: Task.obstacles) ---- */
#define SAG_MAX_HAZARDS 9  /* tasks/go_to_goal.py:83-84  [9,10,0,1]          */
#define SAG_MAX_VASES 10   /* tasks/go_to_goal.py:83-84                     */
#define SAG_MAX_PILLARS 2  /* every task has <= 1; one spare                */
#define SAG_MAX_BUTTONS 6  /* tasks/collect.py:11 NUM_BUTTONS = 6           */
#define SAG_MAX_NU 12      /* doggo.xml:134-145                             */
#define SAG_LIDAR_BINS 16  /* safe_adaptation_gym.py:22                     */

/* ---- robots (safe_adaptation_gym.py:15-19 control substeps) ---- */
enum sag_robot { SAG_ROBOT_POINT = 0, SAG_ROBOT_CAR = 1, SAG_ROBOT_DOGGO = 2 };

/* ---- task ids = index into sorted benchmark.TASKS keys
 *      (benchmark/__init__.py:14-20; pinned by tests/golden/sampler.json) ---- */
enum sag_task {
  SAG_TASK_CATCH_GOAL = 0,
  SAG_TASK_COLLECT = 1,
  SAG_TASK_DRIBBLE_BALL = 2,
  SAG_TASK_GO_TO_GOAL = 3,
  SAG_TASK_GO_TO_GOAL_DAMPING = 4,
  SAG_TASK_GO_TO_GOAL_MOTOR = 5,
  SAG_TASK_GO_TO_GOAL_SCARCE = 6,
  SAG_TASK_HAUL_BOX = 7,
  SAG_TASK_PRESS_BUTTONS = 8,
  SAG_TASK_PRESS_BUTTONS_SCARCE = 9,
  SAG_TASK_PUSH_BOX = 10,
  SAG_TASK_PUSH_BOX_SCARCE = 11,
  SAG_TASK_ROLL_ROD = 12,
  SAG_TASK_UNSUPERVISED = 13,
  SAG_NUM_TASKS = 14
};

enum sag_box_kind { SAG_BOX_NONE = 0, SAG_BOX_BOX = 1, SAG_BOX_ROD = 2, SAG_BOX_BALL = 3 };

enum sag_status {
  SAG_OK = 0,
  SAG_ERR_ARG = -1,      /* bad argument                                  */
  SAG_ERR_HIP = -2,      /* HIP runtime error (message has the HIP text)  */
  SAG_ERR_NODEVICE = -3, /* no gfx950 device / device ordinal out of range */
  SAG_ERR_STATE = -4,    /* call out of order (step before set_layout)    */
  SAG_ERR_UNSUPPORTED = -5
};

/*
 * Per-env record: what rebuild() + World.reset() install in the reference
 * (world.py:108-137 world_config, mujoco_bridge.py:59-63,154-166, tasks' reset)
 * and, read back, the full simulator + task state (checkpoint / parity tests).
 * Host side it is an array-of-records [n_envs][SAG_REC_FLOATS] float32 plus
 * [n_envs][SAG_REC_INTS] int32; the library transposes to its SoA device layout.
 * World-frame planar coordinates; angles in radians; SI units.
 */
enum sag_rec_float {
  SAG_F_ROBOT = 0,        /* x, y, yaw, vx, vy, w  (world frame)                      */
  SAG_F_ROBOT0 = 6,       /* x0, y0, rot0: pose at rebuild (qpos frame, mujoco_bridge.py:59-63) */
  SAG_F_GEAR = 9,         /* point motor-x gear: 0.3, Motor variant 3.0 (go_to_goal_motor.py:12-16) */
  SAG_F_DAMP = 10,        /* slide damping: 0.01, Damping variant 0.001 (go_to_goal_damping.py:12-17) */
  SAG_F_ACTION_NOISE = 11, /* world.py:31                                             */
  SAG_F_CTRL_SCALE = 12,  /* [SAG_MAX_NU] ctrlrange scale (world.py:72-73)            */
  SAG_F_HAZARD_SIZE = 24, /* world.py:20                                              */
  SAG_F_VASE_SIZE = 25,   /* world.py:21  (half extent)                               */
  SAG_F_PILLAR_SIZE = 26, /* world.py:22  (radius)                                    */
  SAG_F_KEEPOUT = 27,     /* robot, hazards, vases, pillars, box  (world.py:60-70, push_box.py:13) */
  SAG_F_GOAL = 32,        /* x, y                                                     */
  SAG_F_CATCH = 34,       /* origin x, y, current_radius, next_radius (catch_goal.py:14-18) */
  SAG_F_LAST = 38,        /* last goal dist, last box dist, last box-goal dist        */
  SAG_F_BOX = 41,         /* x, y, yaw, vx, vy, w                                     */
  SAG_F_HAZARDS = 47,     /* (x, y) * SAG_MAX_HAZARDS                                 */
  SAG_F_PILLARS = 65,     /* (x, y) * SAG_MAX_PILLARS                                 */
  SAG_F_BUTTONS = 69,     /* (x, y) * SAG_MAX_BUTTONS                                 */
  SAG_F_VASES = 81,       /* (x, y, yaw, vx, vy, w) * SAG_MAX_VASES                   */
  SAG_F_BOUND = 141,      /* info['bound'] (world.py:75-78); carried, not used on the device */
  SAG_F_ROBOT_EXT = 144,  /* robot DoF beyond the planar base. car (car.xml:21-32): wheel rates left,
                           * right (rad/s about the axle); rear ball: angular velocity x,y,z
                           * (relative to the base, base frame); ball quaternion w,x,y,z.
                           * doggo (doggo.xml): [0] base z, [1..4] base quaternion w,x,y,z, [5] vz,
                           * [6..8] base angular velocity in the base frame (MuJoCo free-joint qvel),
                           * [9..21] the 13 hinge angles in qpos order (hip_1_z, hip_1_y, ankle_1,
                           * hip_4_z, hip_4_y, ankle_4, waist_x, hip_2_z, hip_2_y, ankle_2, hip_3_z,
                           * hip_3_y, ankle_3), [22..34] their rates.  ROBOT x, y, vx, vy are the base
                           * position / world velocity; ROBOT yaw and w are derived (heading, world
                           * yaw rate) and written back every step.  A zero quaternion at install
                           * time means "upright at ROBOT yaw, z = 0.22, joints at 0".            */
  SAG_ROBOT_EXT_FLOATS = 40,
  SAG_REC_FLOATS = 184
};

enum sag_rec_int {
  SAG_I_TASK = 0,
  SAG_I_NH = 1,
  SAG_I_NV = 2,
  SAG_I_NP = 3,
  SAG_I_NB = 4,
  SAG_I_BOX_KIND = 5,
  SAG_I_GOAL_BUTTON = 6,  /* press_buttons.py:72                                   */
  SAG_I_BTN_STATE = 7,    /* 0 = BUTTON_CHANGE, 1 = NORMAL (press_buttons.py:101-104) */
  SAG_I_BTN_TIMER = 8,    /* press_buttons.py:107-121                              */
  SAG_I_CATCH_TIMER = 9,  /* catch_goal.py:17                                      */
  SAG_I_ACTIVE_MASK = 10, /* collect.py:15-16: bit b = buttons{b} still active     */
  SAG_I_STEP = 11,        /* env steps since rebuild (time = step * nstep * dt)    */
  SAG_I_ENV_ID = 12,      /* global env index (counter-based RNG stream id)        */
  SAG_I_FLAGS = 13,       /* bit0: resample failed (ResamplingError), bit1: tape exhausted,
                           * bit2: a Doggo constraint did not fit the row budget and was dropped */
  SAG_I_EPISODE = 14,     /* episode nonce of the counter-based generator (24 bits): with SAG_I_ENV_ID
                           * and SAG_I_STEP it forms the counter, so that the device-side draws of
                           * throughput mode (action noise, goal resampling, CatchGoal radii, button
                           * choice) differ from episode to episode as the reference's reseeded
                           * RandomState does (safe_adaptation_gym.py:97-101).  sag_set_layout takes it
                           * from the record, sag_reset adds one */
  SAG_I_AWAKE = 15,       /* bit k: free body k (vases 0 .., the task object = bit SAG_MAX_VASES) takes part in the NEXT forward
                           * evaluation although it is at rest.  sag_set_layout sets it for bodies whose bounding circle
                           * overlaps another free body's, a pillar's or a button's (HaulBox spawns its box at robot + .6
                           * without a keep-out check, haul_box.py:17-18: MuJoCo separates such a pair in the first steps,
                           * and so does this); the first substep clears it (a body in motion is awake by its velocity);
                           * sag_set_state / sag_get_state carry it like any other field (ABI v5) */
  SAG_REC_INTS = 16
};

typedef struct sag_config {
  int32_t abi_version;  /* SAG_ABI_VERSION                                     */
  int32_t robot;        /* enum sag_robot                                      */
  int32_t n_envs;       /* envs owned by this context (one shard)              */
  int32_t device;       /* HIP device ordinal                                  */
  int32_t max_hazards;  /* SoA capacities, <= SAG_MAX_*; 0 drops the arrays    */
  int32_t max_vases;
  int32_t max_pillars;
  int32_t max_buttons;
  int32_t has_box;      /* allocate box state                                  */
  int32_t reserved0;
  uint64_t seed;        /* key of the counter-based generator (throughput mode); sag_set_seed changes it */
} sag_config;

typedef struct sag_ctx sag_ctx;

/* Static per-robot facts: replaces Robot (robot.py:11-59) + the substep table
 * (safe_adaptation_gym.py:15-19).  out[0]=nu out[1]=obs_dim out[2]=nstep
 * out[3]=nq out[4]=nv; dt in *dt. */
int sag_robot_info(int32_t robot, int32_t out[5], double* dt);

/* Replaces MujocoBridge.__init__ (mujoco_bridge.py:26-36): allocates the SoA
 * world for n_envs on one device and creates the context's stream. */
int sag_create(const sag_config* cfg, sag_ctx** out);
int sag_destroy(sag_ctx* ctx);
const char* sag_last_error(const sag_ctx* ctx); /* ctx may be NULL: last create error */

/* Replaces MujocoBridge.rebuild(world_config) + World.reset()
 * (mujoco_bridge.py:170-175, world.py:167-170, safe_adaptation_gym.py:170-172)
 * for the listed envs, without compiling anything.  env_ids == NULL means
 * 0..n-1.  Velocities in the record are honoured (normally 0). */
int sag_set_layout(sag_ctx* ctx, const int32_t* env_ids, int32_t n,
                   const float* rec_f, const int32_t* rec_i);

/* Re-install the records last given to sag_set_layout for these envs (start
 * of a fixed-layout episode); their episode nonce (SAG_I_EPISODE) advances by one. */
int sag_reset(sag_ctx* ctx, const int32_t* env_ids, int32_t n);

/* env.seed(s) (safe_adaptation_gym.py:113-118) for the device-side generator of throughput
 * mode: the key of the counter-based stream from the next step on. */
int sag_set_seed(sag_ctx* ctx, uint64_t seed);

/* Checkpoint / parity access to the complete per-env state (same records). */
int sag_get_state(sag_ctx* ctx, const int32_t* env_ids, int32_t n, float* rec_f,
                  int32_t* rec_i);
int sag_set_state(sag_ctx* ctx, const int32_t* env_ids, int32_t n,
                  const float* rec_f, const int32_t* rec_i);

/* Host-buffer step = SafeAdaptationGym.step (safe_adaptation_gym.py:56-83) for
 * every env of the context.
 *   actions  [n_envs][nu] f32, before noise/clip (:58-67)
 *   noise    [n_envs][nu] f32 standard normals (parity mode: the reference's
 *            rs.normal draws) or NULL: counter-based generator on device
 *   tape     [n_envs][tape_len] u32 raw generator words consumed by in-step
 *            draws in the reference's order (App. B.6: CatchGoal radius,
 *            goal resample uniforms, button choice) or NULL: counter-based
 *   nstep    physics substeps; <0 = the robot's table value (5/10/12);
 *            0 = no physics: evaluate reward/cost/observation at the current state
 *   obs      [n_envs][obs_dim] f32   reward [n_envs][2] f32 (col 1 only used by
 *            Unsupervised, tasks/unsupervised.py:66)   cost/done/goal_met
 *            [n_envs] u8   tape_used [n_envs] i32 (may be NULL)
 * Any output pointer may be NULL (skipped).  Synchronous. */
int sag_step(sag_ctx* ctx, const float* actions, const float* noise,
             const uint32_t* tape, int32_t tape_len, int32_t nstep, float* obs,
             float* reward, uint8_t* cost, uint8_t* done, uint8_t* goal_met,
             int32_t* tape_used);

/* External contact results for the NEXT step with nstep == 0 (one-shot; replay of recorded episodes: the caller sets
 * the poses with sag_set_state and supplies what the physics would have reported).  Replaces, per env, the outcome of
 * MujocoBridge.robot_contacts on the final state (mujoco_bridge.py:177-191): cost_contacts[i] = number of contacts
 * robot geom <-> obstacle geom (world.py:146; -1 = keep the device's own geometric result for this env),
 * btn_mask[i] = buttons the robot touches (tasks/press_buttons.py:51, tasks/collect.py:32).  Host arrays [n_envs];
 * NULL, NULL clears a pending set.  A step with nstep != 0 ignores and clears it. */
int sag_set_ext_contacts(sag_ctx* ctx, const int32_t* cost_contacts, const uint32_t* btn_mask);

/* Device-buffer step for learners that live on the GPU: same semantics, all
 * pointers are device pointers on ctx's device (or NULL as above), enqueued on
 * the context stream; returns without waiting.  sag_wait() joins. */
int sag_step_device(sag_ctx* ctx, const float* d_actions, const float* d_noise,
                    int32_t nstep, float* d_obs, float* d_reward, uint8_t* d_cost,
                    uint8_t* d_done, uint8_t* d_goal_met);
int sag_wait(sag_ctx* ctx);

/* First observation after reset: `self.observation` at safe_adaptation_gym.py:104,107
 * (mj_forward + sensors + 3 lidars, no reward/cost). */
int sag_observe(sag_ctx* ctx, float* obs);

/* The lidar + hazard-cost kernel on its own (BASELINE config 2), on explicit
 * poses; replaces lidar_observations/_lidar (safe_adaptation_gym.py:133-139,
 * 174-223) and the hazard loop of World.compute_cost (world.py:147-153).
 *   robot   [n][3] f32 x, y, yaw          points [n][K][2] f32
 *   group   [n][K] u8: 0 inactive, 1 obstacle, 2 goal, 3 object (consts.py:13-16),
 *           +128 flags a hazard (tested for cost with hazard_size)
 *   lidar   [n][48] f32 = [obstacles, objects, goal]    bins [n][K] i32 (-1 =
 *           inactive)      cost [n] u8
 * Host pointers; synchronous. */
int sag_lidar_cost(sag_ctx* ctx, int32_t n, int32_t K, const float* robot,
                   const float* points, const uint8_t* group, float hazard_size,
                   float* lidar, int32_t* bins, uint8_t* cost);

/* The same kernel on DEVICE buffers of the context's device, enqueued on the context stream
 * without host copies or synchronisation (sag_wait joins); bracketed by the HIP events of
 * sag_enable_timing / sag_kernel_time_ms like a step launch.  d_bins may be NULL. */
int sag_lidar_cost_device(sag_ctx* ctx, int32_t n, int32_t K, const float* d_robot,
                          const float* d_points, const uint8_t* d_group, float hazard_size,
                          float* d_lidar, int32_t* d_bins, uint8_t* d_cost);

/* Device allocation helpers so NumPy-only hosts can keep action/observation
 * buffers resident (bench harness, GPU learners without torch). */
int sag_dev_alloc(sag_ctx* ctx, uint64_t bytes, void** dptr);
int sag_dev_free(sag_ctx* ctx, void* dptr);
int sag_dev_upload(sag_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int sag_dev_download(sag_ctx* ctx, void* dst, const void* src, uint64_t bytes);
/* Fill d_actions [n_envs][nu] with U(-1,1) from the counter-based generator
 * (stream = step index): the synthetic policy of the bench. */
int sag_dev_fill_actions(sag_ctx* ctx, float* d_actions, uint32_t step_index);

/* Timing of the step kernel alone, measured with HIP events on the context
 * stream: mean milliseconds per launch over the launches since the last call
 * with reset != 0.  Feeds bench.py's roofline.achieved. */
int sag_kernel_time_ms(sag_ctx* ctx, int32_t reset, double* mean_ms, int64_t* launches);
/* Diagnostic (tests): Doggo mass matrix [19x19], bias [19], contact-free qacc [19] and M^-1 [19x19]
 * per env from the wave-cooperative routines; out[n_envs][760] doubles. */
int sag_debug_doggo_coop(sag_ctx* ctx, double* out);

/* rgb_observation (safe_adaptation_gym.py:122-126 `physics.render(height=64, width=64,
 * camera_id='vision')`): the first-person image of every env at its current state,
 * out[n_envs][64][64][3] uint8 (row 0 = top).  Host buffer / device buffer variants. */
int sag_render_rgb(sag_ctx* ctx, uint8_t* out);
int sag_render_rgb_device(sag_ctx* ctx, void* d_out);

/* Human / debugging view (safe_adaptation_gym.py:109-111 render() with render_options, render.py,
 * mujoco_bridge.py:126-153): the same device ray caster with any of the scene's cameras -
 *   camera 0 the robot's `vision` camera, 1 `fixednear` (pos 0 -2 2, zaxis 0 -1 1), 2 `fixedfar` (0 -5 5),
 *          3 `track` (fixednear's view, following the robot body)
 * any image size, and with flags & 1 the overlays of render_lidars_and_collision: three rings of 16 spheres above
 * the robot (obstacles red, goal green, objects blue; alpha = min(1, lidar value + .1), safe_adaptation_gym.py:239-257)
 * and the red cost sphere.  out[n_envs][height][width][3] uint8, row 0 = top.  sag_render shows the observation / cost
 * of the context's last host-buffer step; the device variant takes them as (device) pointers or NULL. */
int sag_render(sag_ctx* ctx, int32_t camera, int32_t width, int32_t height, int32_t flags, uint8_t* out);
int sag_render_device(sag_ctx* ctx, int32_t camera, int32_t width, int32_t height, int32_t flags,
                      const float* d_obs, const uint8_t* d_cost, void* d_out);

/* Diagnostic: how many envs the last split step handed to the busy kernel (0 for the
 * single-kernel form).  Synchronises the context stream. */
int sag_busy_count(sag_ctx* ctx, int32_t* count);
/* Diagnostic, only in libraries built with -DSAG_CYCLES (SAG_ERR_UNSUPPORTED otherwise):
 * wavefront clock ticks per code section of the step kernel, [3 launch forms][16]. */
int sag_debug_cycles(sag_ctx* ctx, int32_t reset, uint64_t* out, int32_t n);
/* Per-launch HIP-event bracketing of the step kernel is off by default. */
int sag_enable_timing(sag_ctx* ctx, int32_t on);

int sag_device_count(void);

/* ---- native reset path ------------------------------------------------
 * World.DEFAULT (world.py:17-34), the keys that influence sampling or the installed record. */
typedef struct sag_world_config {
  double placements_margin, robot_keepout, hazards_size, vases_size, pillars_size;
  double hazards_keepout, vases_keepout, pillars_keepout;
  double robot_ctrl_range_scale, action_noise, max_bound;
  int32_t random_bound, reserved;
} sag_world_config;
void sag_world_config_default(sag_world_config* cfg);

/* What the sampler needs to know about a Task object (tasks/task.py:14-97): the host mirror builds one from the
 * Task's own `obstacles`, `placement_extents`, `setup_placements()` and attributes (safe_adaptation_gym_amd/tasks), so a
 * subclass that overrides them changes the layouts; `task_id` selects the device-side per-step logic (compute_reward /
 * set_mocaps / on-goal reset), which cannot be overridden from Python.  Rectangles are (xmin, ymin, xmax, ymax). */
typedef struct sag_task_desc {
  int32_t task_id;                       /* enum sag_task */
  int32_t n_hazards, n_vases, n_pillars; /* Task.obstacles = [hazards, vases, gremlins (always 0), pillars] */
  int32_t has_goal;                      /* a 'goal' placement, re-drawn by GoToGoal.reset (go_to_goal.py:50-80) */
  int32_t box_kind;                      /* enum sag_box_kind; SAG_BOX_NONE: no task object */
  int32_t box_yaw;                       /* build_world_config draws a yaw for it (push_box.py; not roll_rod / dribble_ball) */
  int32_t box_at_robot;                  /* haul_box.py:17-18: after sampling the object sits at robot + (box_offset, 0) */
  int32_t n_buttons;
  int32_t button_reset;                  /* task.reset: 0 none, 1 rs.choice(n_buttons) + timer (press_buttons.py:71-77), 2 all active (collect.py) */
  int32_t button_timer;                  /* BUTTON_TICKING_DELAY */
  int32_t reserved;
  double extents[4];                     /* Task.placement_extents */
  double goal_keepout, box_keepout, button_keepout, box_offset;
  double box_rect[4];                    /* all zero: the extents (placement None) */
  double button_rect[4];
  double gear, damping;                  /* point.xml variants (go_to_goal_motor.py, go_to_goal_damping.py) */
} sag_task_desc;
/* the descriptor of one of the reference's 14 tasks; returns SAG_ERR_ARG for an unknown id */
int sag_task_desc_default(int32_t task_id, sag_task_desc* out);
/* NULL if the descriptor is one the sampler and the device can serve, else a static string naming the offending field
 * and the rule (capacities, button_timer 0..5 and 5 with button_reset 1, finite non-negative keep-outs, rectangles with
 * xmin < xmax and ymin < ymax or all zero, no goal together with buttons).  sag_sample_layouts_desc returns SAG_ERR_ARG
 * when this is not NULL for one of its descriptors. */
const char* sag_task_desc_check(const sag_task_desc* desc);

/* Replaces World.__init__ + sample_layout + _build_world_config + World.reset's host draws
 * (world.py:36-137,172-217; tasks' setup_placements/build_world_config/reset) for n envs on the
 * host cores.  Env j owns np.random.RandomState(seeds[j]) (exact legacy MT19937 stream, the
 * reference's draw order) and yields record j of include/sag.h, env id env_id0 + j.
 * first_episode != 0: the Task object is new (Cauchy ctrl-scale draw, world.py:72-73), else the
 * World persists (safe_adaptation_gym.py:105-106).  mt_* (each may be NULL) return the generator
 * after the draws (key [n][624], pos, has_gauss, gauss) so a parity harness can continue the
 * same stream in numpy.  status[j] (may be NULL): 0 ok, <0 ResamplingError.  Returns the number
 * of envs that failed, or a negative sag_status.  No GPU involved. */
int sag_sample_layouts(int32_t robot, int32_t n, const uint32_t* seeds, const int32_t* task_ids,
                       const sag_world_config* cfg, int32_t first_episode, int32_t env_id0,
                       float* rec_f, int32_t* rec_i, uint32_t* mt_key, int32_t* mt_pos,
                       int32_t* mt_has_gauss, double* mt_gauss, int32_t* status, int32_t nthreads);
/* The same with explicit task descriptors: env j uses descs[desc_of_env[j]] (n_descs of them).  sag_sample_layouts is
 * this call with the 14 default descriptors and desc_of_env = task_ids.  SAG_ERR_ARG if sag_task_desc_check rejects a
 * descriptor. */
int sag_sample_layouts_desc(int32_t robot, int32_t n, const uint32_t* seeds, const sag_task_desc* descs, int32_t n_descs,
                            const int32_t* desc_of_env, const sag_world_config* cfg, int32_t first_episode, int32_t env_id0,
                            float* rec_f, int32_t* rec_i, uint32_t* mt_key, int32_t* mt_pos,
                            int32_t* mt_has_gauss, double* mt_gauss, int32_t* status, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif /* SAG_H_ */
