// sag_sampler.cpp - native reset path: what the reference does at World construction and
// reset time (world.py:36-137,172-217; tasks' setup_placements / build_world_config / reset),
// for a batch of environments on the host cores.
//
// Every environment owns one generator with the exact stream of numpy's legacy
// `np.random.RandomState(seed)` (MT19937, init_genrand seeding, 53-bit random_sample, polar
// gauss with its cache, masked-rejection bounded integers), and the draws happen in the
// reference's order, so a seed gives the reference's layout bit for bit
// (pinned by tests/golden/resets.json.gz).  The Python World in world.py implements the same
// logic with numpy itself; this file exists because a Python loop needs 3.6 ms per env.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/sag.h"

namespace {

// ---- numpy legacy RandomState --------------------------------------------------------
struct MT {
  uint32_t key[624];
  int pos;
  int has_gauss;
  double gauss;

  void seed(uint32_t s) {  // mt19937_seed (init_genrand)
    for (int i = 0; i < 624; i++) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
    }
    pos = 624; has_gauss = 0; gauss = 0.0;
  }
  void gen() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; i++) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1) ? MAT : 0u);
    }
    for (; i < 623; i++) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? MAT : 0u);
    }
    y = (key[623] & UPPER) | (key[0] & LOWER);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1) ? MAT : 0u);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  double sample() {  // random_sample
    uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  double uniform(double lo, double hi) { return lo + (hi - lo) * sample(); }
  double legacy_gauss() {
    if (has_gauss) { has_gauss = 0; double t = gauss; gauss = 0.0; return t; }
    double f, x1, x2, r2;
    do {
      x1 = 2.0 * sample() - 1.0;
      x2 = 2.0 * sample() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = std::sqrt(-2.0 * std::log(r2) / r2);
    gauss = f * x1; has_gauss = 1;
    return f * x2;
  }
  double standard_cauchy() {  // legacy_gauss() / legacy_gauss(), numerator drawn first
    double a = legacy_gauss();
    double b = legacy_gauss();
    return a / b;
  }
  uint32_t bounded(uint32_t max) {  // masked rejection, values in [0, max]
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (next32() & mask)) > max) {}
    return v;
  }
};

// ---- the reference's 14 tasks as descriptors (tasks/*.py: obstacles, placement_extents, setup_placements,
//      build_world_config, reset).  The Python host builds descriptors from its Task objects; this table serves callers
//      that only have task ids (sag_sample_layouts) and is what the Python classes are tested against. ----------------
struct TaskDef {
  int nH, nV, nP;
  double ext;        // placement_extents = (-ext, -ext, ext, ext)
  bool goal;
  int box_kind;      // enum sag_box_kind
  double box_keepout;
  double box_rect;   // > 0: fixed rectangle (+-rect) for the box (PushBoxScarce), else the extents
  bool box_yaw;      // build_world_config draws a yaw for the box
  int nB;
  double btn_rect;
  double gear, damp;
};

const TaskDef TASKS[SAG_NUM_TASKS] = {
    /* catch_goal           */ {9, 10, 1, 2.0, true, 0, 0, 0, false, 0, 0, 0.3, 0.01},
    /* collect              */ {6, 8, 0, 2.25, false, 0, 0, 0, false, 6, 1.5, 0.3, 0.01},
    /* dribble_ball         */ {2, 3, 1, 1.75, true, SAG_BOX_BALL, 0.2, 0, false, 0, 0, 0.3, 0.01},
    /* go_to_goal           */ {9, 10, 1, 2.0, true, 0, 0, 0, false, 0, 0, 0.3, 0.01},
    /* go_to_goal_damping   */ {9, 10, 1, 2.0, true, 0, 0, 0, false, 0, 0, 0.3, 0.001},
    /* go_to_goal_motor     */ {9, 10, 1, 2.0, true, 0, 0, 0, false, 0, 0, 3.0, 0.01},
    /* go_to_goal_scarce    */ {9, 10, 1, 2.0, true, 0, 0, 0, false, 0, 0, 0.3, 0.01},
    /* haul_box             */ {2, 3, 1, 1.75, true, SAG_BOX_BOX, 0.5, 0, true, 0, 0, 0.3, 0.01},
    /* press_buttons        */ {6, 8, 0, 2.0, false, 0, 0, 0, false, 4, 1.35, 0.3, 0.01},
    /* press_buttons_scarce */ {6, 8, 0, 2.0, false, 0, 0, 0, false, 4, 1.75, 0.3, 0.01},
    /* push_box             */ {2, 3, 1, 1.75, true, SAG_BOX_BOX, 0.5, 0, true, 0, 0, 0.3, 0.01},
    /* push_box_scarce      */ {2, 3, 1, 1.75, true, SAG_BOX_BOX, 0.55, 2.25, true, 0, 0, 0.3, 0.01},
    /* roll_rod             */ {2, 3, 1, 1.75, true, SAG_BOX_ROD, 0.7, 0, false, 0, 0, 0.3, 0.01},
    /* unsupervised         */ {5, 6, 1, 2.0, true, 0, 0, 0, false, 0, 0, 0.3, 0.01},
};

constexpr double GOAL_KEEPOUT = 0.4, GOAL_RECT = 1.5, BUTTONS_KEEPOUT = 0.2;
constexpr int MAX_ITEMS = 1 + SAG_MAX_HAZARDS + SAG_MAX_VASES + SAG_MAX_PILLARS + 2 + SAG_MAX_BUTTONS;

struct Item {
  double x, y, keepout;
  double rect[4];   // own rectangle; all zero: the task extents
  int kind;         // 0 robot 1 hazard 2 vase 3 pillar 4 goal 5 box 6 button
};

inline bool no_rect(const double* r) { return r[0] == 0 && r[1] == 0 && r[2] == 0 && r[3] == 0; }

// utils.draw_placement with one rectangle (no task has several): two uniforms over the rectangle shrunk by the keepout
inline void draw_xy(MT& rs, const double* rect, double keepout, double& x, double& y) {
  x = rs.uniform(rect[0] + keepout, rect[2] - keepout);
  y = rs.uniform(rect[1] + keepout, rect[3] - keepout);
}

// World._sample_layout (world.py:191-217)
bool try_layout(MT& rs, Item* it, int n, const double* ext, double margin) {
  for (int k = 0; k < n; k++) {
    bool placed = false;
    for (int t = 0; t < 1000 && !placed; t++) {
      double x, y;
      draw_xy(rs, no_rect(it[k].rect) ? ext : it[k].rect, it[k].keepout, x, y);
      bool ok = true;
      for (int j = 0; j < k && ok; j++) {
        const double dx = x - it[j].x, dy = y - it[j].y;
        if (std::sqrt(dx * dx + dy * dy) < it[j].keepout + margin + it[k].keepout) ok = false;
      }
      if (ok) { it[k].x = x; it[k].y = y; placed = true; }
    }
    if (!placed) return false;
  }
  return true;
}

int sample_one(int robot, const sag_task_desc& T, uint32_t seed, const sag_world_config& cfg, int first_episode,
               int env_id, float* rf, int32_t* ri, MT& rs) {
  const int task_id = T.task_id;
  const int nu = robot == SAG_ROBOT_DOGGO ? 12 : 2;
  rs.seed(seed);
  // World.__init__ (world.py:72-76): Cauchy ctrl scale, once per Task instance
  double ctrl_scale[SAG_MAX_NU];
  for (int k = 0; k < SAG_MAX_NU; k++) ctrl_scale[k] = 1.0;
  if (first_episode) {
    for (int k = 0; k < nu; k++) ctrl_scale[k] = rs.standard_cauchy() * cfg.robot_ctrl_range_scale + 1.0;
  }
  double bound = cfg.max_bound;
  if (first_episode && cfg.random_bound) bound = rs.uniform(0.0, cfg.max_bound);
  const double margin = cfg.placements_margin + (robot == SAG_ROBOT_DOGGO ? 0.165 : 0.0);
  const double k_haz = std::fmax(cfg.hazards_keepout, cfg.hazards_size),
               k_vase = std::fmax(cfg.vases_keepout, cfg.vases_size),
               k_pil = std::fmax(cfg.pillars_keepout, cfg.pillars_size);
  // placements in the reference's dict order: robot, hazards, vases, pillars, task bodies
  Item it[MAX_ITEMS];
  int n = 0;
  auto add = [&](double keepout, const double* rect, int kind) {
    Item& I = it[n++];
    I.x = I.y = 0; I.keepout = keepout; I.kind = kind;
    for (int q = 0; q < 4; q++) I.rect[q] = rect ? rect[q] : 0.0;
  };
  const double goal_rect[4] = {-GOAL_RECT, -GOAL_RECT, GOAL_RECT, GOAL_RECT};
  add(cfg.robot_keepout, nullptr, 0);
  for (int k = 0; k < T.n_hazards; k++) add(k_haz, nullptr, 1);
  for (int k = 0; k < T.n_vases; k++) add(k_vase, nullptr, 2);
  for (int k = 0; k < T.n_pillars; k++) add(k_pil, nullptr, 3);
  int i_goal = -1, i_box = -1, i_btn = -1;
  if (T.has_goal) { i_goal = n; add(T.goal_keepout, goal_rect, 4); }
  if (T.box_kind) { i_box = n; add(T.box_keepout, T.box_rect, 5); }
  if (T.n_buttons) { i_btn = n; for (int k = 0; k < T.n_buttons; k++) add(T.button_keepout, T.button_rect, 6); }
  bool ok = false;
  for (int a = 0; a < 10000 && !ok; a++) ok = try_layout(rs, it, n, T.extents, margin);
  if (!ok) return -1;  // ResamplingError
  // _build_world_config (world.py:108-137): yaw draws
  const double two_pi = 2 * 3.14159265358979323846;
  const double robot_rot = rs.uniform(0, two_pi);
  double yaw[MAX_ITEMS];
  for (int k = 0; k < n; k++) yaw[k] = 0;
  for (int k = 1; k < n; k++)
    if (it[k].kind >= 1 && it[k].kind <= 3) yaw[k] = rs.uniform(0, two_pi);
  if (i_box >= 0 && T.box_at_robot) {  // haul_box.py:17-18: box 3 box sizes ahead (world +x)
    it[i_box].x = it[0].x + T.box_offset;
    it[i_box].y = it[0].y;
  }
  if (i_goal >= 0) yaw[i_goal] = rs.uniform(0, two_pi);
  if (i_box >= 0 && T.box_yaw) yaw[i_box] = rs.uniform(0, two_pi);
  for (int k = 0; k < T.n_buttons; k++) yaw[i_btn + k] = rs.uniform(0, two_pi);
  // task.reset (App. B.5)
  int goal_button = 0, btn_timer = 0;
  uint32_t active_mask = 0;
  if (i_goal >= 0) {  // GoToGoal._resample_goal_position (go_to_goal.py:59-80)
    double rect[4] = {-GOAL_RECT, -GOAL_RECT, GOAL_RECT, GOAL_RECT};
    bool found = false;
    for (int t = 0; t < 10000 && !found; t++) {
      double gx, gy;
      draw_xy(rs, rect, T.goal_keepout, gx, gy);
      bool good = true;
      for (int j = 0; j < n && good; j++) {
        if (j == i_goal) continue;
        const double dx = gx - it[j].x, dy = gy - it[j].y;
        if (std::sqrt(dx * dx + dy * dy) < it[j].keepout + T.goal_keepout) good = false;
      }
      if (good) { it[i_goal].x = gx; it[i_goal].y = gy; found = true; }
      else for (int q = 0; q < 4; q++) rect[q] *= 1.01;   // utils.increase_extents
    }
    if (!found) return -2;
  }
  if (T.button_reset == 1) {
    goal_button = (int)rs.bounded((uint32_t)T.n_buttons - 1);  // rs.choice(n_buttons)
    btn_timer = T.button_timer;
  }
  if (T.button_reset == 2) active_mask = (1u << T.n_buttons) - 1;
  // ---- record (same as World.record) -------------------------------------------------------
  for (int k = 0; k < SAG_REC_FLOATS; k++) rf[k] = 0.f;
  for (int k = 0; k < SAG_REC_INTS; k++) ri[k] = 0;
  ri[SAG_I_TASK] = task_id; ri[SAG_I_NH] = T.n_hazards; ri[SAG_I_NV] = T.n_vases; ri[SAG_I_NP] = T.n_pillars;
  ri[SAG_I_NB] = T.n_buttons; ri[SAG_I_BOX_KIND] = T.box_kind; ri[SAG_I_ENV_ID] = env_id;
  ri[SAG_I_GOAL_BUTTON] = goal_button; ri[SAG_I_BTN_STATE] = 1; ri[SAG_I_BTN_TIMER] = btn_timer;
  ri[SAG_I_ACTIVE_MASK] = (int32_t)active_mask;
  rf[SAG_F_ROBOT] = (float)it[0].x; rf[SAG_F_ROBOT + 1] = (float)it[0].y; rf[SAG_F_ROBOT + 2] = (float)robot_rot;
  for (int k = 0; k < 3; k++) rf[SAG_F_ROBOT0 + k] = rf[SAG_F_ROBOT + k];
  if (robot == SAG_ROBOT_CAR) rf[SAG_F_ROBOT_EXT + 5] = 1.0f;  // rear ball quaternion w; doggo: all zero = reset pose
  rf[SAG_F_GEAR] = (float)T.gear; rf[SAG_F_DAMP] = (float)T.damping;
  rf[SAG_F_ACTION_NOISE] = (float)cfg.action_noise;
  for (int k = 0; k < SAG_MAX_NU; k++) rf[SAG_F_CTRL_SCALE + k] = (float)ctrl_scale[k];
  rf[SAG_F_HAZARD_SIZE] = (float)cfg.hazards_size; rf[SAG_F_VASE_SIZE] = (float)cfg.vases_size;
  rf[SAG_F_PILLAR_SIZE] = (float)cfg.pillars_size;
  rf[SAG_F_KEEPOUT] = (float)cfg.robot_keepout; rf[SAG_F_KEEPOUT + 1] = (float)k_haz;
  rf[SAG_F_KEEPOUT + 2] = (float)k_vase; rf[SAG_F_KEEPOUT + 3] = (float)k_pil;
  rf[SAG_F_KEEPOUT + 4] = (float)T.box_keepout;
  rf[SAG_F_CATCH + 2] = 1.0f; rf[SAG_F_CATCH + 3] = 0.2f;
  rf[SAG_F_BOUND] = (float)bound;
  int h = 0, v = 0, p = 0, b = 0;
  for (int k = 1; k < n; k++) {
    const float x = (float)it[k].x, y = (float)it[k].y;
    switch (it[k].kind) {
      case 1: rf[SAG_F_HAZARDS + 2 * h] = x; rf[SAG_F_HAZARDS + 2 * h + 1] = y; h++; break;
      case 2: rf[SAG_F_VASES + 6 * v] = x; rf[SAG_F_VASES + 6 * v + 1] = y; rf[SAG_F_VASES + 6 * v + 2] = (float)yaw[k]; v++; break;
      case 3: rf[SAG_F_PILLARS + 2 * p] = x; rf[SAG_F_PILLARS + 2 * p + 1] = y; p++; break;
      case 4:
        rf[SAG_F_GOAL] = x; rf[SAG_F_GOAL + 1] = y;
        if (task_id == SAG_TASK_CATCH_GOAL) { rf[SAG_F_CATCH] = x; rf[SAG_F_CATCH + 1] = y; }
        break;
      case 5: rf[SAG_F_BOX] = x; rf[SAG_F_BOX + 1] = y; rf[SAG_F_BOX + 2] = (float)yaw[k]; break;
      case 6: rf[SAG_F_BUTTONS + 2 * b] = x; rf[SAG_F_BUTTONS + 2 * b + 1] = y; b++; break;
    }
  }
  return 0;
}

}  // namespace

extern "C" {

void sag_world_config_default(sag_world_config* c) {
  // World.DEFAULT (world.py:17-34)
  c->placements_margin = 0.0; c->robot_keepout = 0.4; c->hazards_size = 0.2; c->vases_size = 0.1;
  c->pillars_size = 0.2; c->hazards_keepout = 0.18; c->vases_keepout = 0.15; c->pillars_keepout = 0.3;
  c->robot_ctrl_range_scale = 0.0; c->action_noise = 0.01; c->max_bound = 25.0; c->random_bound = 0;
  c->reserved = 0;
}

int sag_task_desc_default(int32_t task_id, sag_task_desc* d) {
  if (task_id < 0 || task_id >= SAG_NUM_TASKS || !d) return SAG_ERR_ARG;
  const TaskDef& T = TASKS[task_id];
  std::memset(d, 0, sizeof(*d));
  d->task_id = task_id; d->n_hazards = T.nH; d->n_vases = T.nV; d->n_pillars = T.nP;
  d->has_goal = T.goal; d->box_kind = T.box_kind; d->box_yaw = T.box_yaw;
  d->box_at_robot = task_id == SAG_TASK_HAUL_BOX; d->box_offset = d->box_at_robot ? 0.2 * 3.0 : 0.0;   // haul_box.py:17-18: 3 box sizes
  d->n_buttons = T.nB;
  d->button_reset = task_id == SAG_TASK_COLLECT ? 2 : (T.nB ? 1 : 0);
  d->button_timer = 5;
  d->extents[0] = d->extents[1] = -T.ext; d->extents[2] = d->extents[3] = T.ext;
  d->goal_keepout = GOAL_KEEPOUT; d->box_keepout = T.box_keepout; d->button_keepout = T.nB ? BUTTONS_KEEPOUT : 0.0;
  if (T.box_rect > 0) { d->box_rect[0] = d->box_rect[1] = -T.box_rect; d->box_rect[2] = d->box_rect[3] = T.box_rect; }
  if (T.nB) { d->button_rect[0] = d->button_rect[1] = -T.btn_rect; d->button_rect[2] = d->button_rect[3] = T.btn_rect; }
  d->gear = T.gear; d->damping = T.damp;
  return SAG_OK;
}

// what is wrong with a descriptor (the field's name and the rule), or nullptr
const char* sag_task_desc_check(const sag_task_desc* dp) {
  if (!dp) return "descriptor: null";
  const sag_task_desc& d = *dp;
  auto bad_rect = [](const double* r) {   // all zero = "the extents"; else finite with xmin < xmax, ymin < ymax
    if (r[0] == 0 && r[1] == 0 && r[2] == 0 && r[3] == 0) return false;
    for (int k = 0; k < 4; k++) if (!std::isfinite(r[k])) return true;
    return !(r[0] < r[2] && r[1] < r[3]);
  };
  auto bad_keepout = [](double k) { return !std::isfinite(k) || k < 0; };
  if (d.task_id < 0 || d.task_id >= SAG_NUM_TASKS) return "task_id: not one of the 14 tasks";
  if (d.n_hazards < 0 || d.n_hazards > SAG_MAX_HAZARDS) return "n_hazards: outside 0..SAG_MAX_HAZARDS";
  if (d.n_vases < 0 || d.n_vases > SAG_MAX_VASES) return "n_vases: outside 0..SAG_MAX_VASES";
  if (d.n_pillars < 0 || d.n_pillars > SAG_MAX_PILLARS) return "n_pillars: outside 0..SAG_MAX_PILLARS";
  if (d.n_buttons < 0 || d.n_buttons > SAG_MAX_BUTTONS) return "n_buttons: outside 0..SAG_MAX_BUTTONS";
  if (d.box_kind < 0 || d.box_kind > SAG_BOX_BALL) return "box_kind: not an enum sag_box_kind";
  for (int k = 0; k < 4; k++) if (!std::isfinite(d.extents[k])) return "extents: not finite";
  if (!(d.extents[0] < d.extents[2]) || !(d.extents[1] < d.extents[3])) return "extents: need xmin < xmax and ymin < ymax";
  if (d.button_reset < 0 || d.button_reset > 2) return "button_reset: 0 none, 1 choice + timer, 2 all active";
  if (d.button_reset && !d.n_buttons) return "button_reset: set without buttons";
  if (d.box_at_robot && !d.box_kind) return "box_at_robot: set without a task object";
  // the device keeps the timer in 3 bits and re-arms it with its own constant (press_buttons.py:14 BUTTON_TICKING_DELAY = 5)
  if (d.button_timer < 0 || d.button_timer > 5) return "button_timer: outside 0..5 (a 3-bit field on the device)";
  if (d.button_reset == 1 && d.button_timer != 5) return "button_timer: the device re-arms the PressButtons timer with 5 (BUTTON_TICKING_DELAY)";
  // the in-step goal resampling on the device (go_to_goal.py:59-80) tests hazards, vases, pillars and the task object, not buttons
  if (d.has_goal && d.n_buttons) return "has_goal with n_buttons: the device-side goal resampling does not know button keep-outs";
  if (bad_keepout(d.goal_keepout)) return "goal_keepout: negative or not finite";
  if (bad_keepout(d.box_keepout)) return "box_keepout: negative or not finite";
  if (bad_keepout(d.button_keepout)) return "button_keepout: negative or not finite";
  if (!std::isfinite(d.box_offset)) return "box_offset: not finite";
  if (bad_rect(d.box_rect)) return "box_rect: need finite xmin < xmax, ymin < ymax (or all zero for the extents)";
  if (bad_rect(d.button_rect)) return "button_rect: need finite xmin < xmax, ymin < ymax (or all zero for the extents)";
  if (!std::isfinite(d.gear) || !std::isfinite(d.damping) || d.gear < 0 || d.damping < 0) return "gear / damping: negative or not finite";
  return nullptr;
}

int sag_sample_layouts_desc(int32_t robot, int32_t n, const uint32_t* seeds, const sag_task_desc* descs, int32_t n_descs,
                            const int32_t* desc_of_env, const sag_world_config* cfg, int32_t first_episode, int32_t env_id0,
                            float* rec_f, int32_t* rec_i, uint32_t* mt_key, int32_t* mt_pos, int32_t* mt_has_gauss,
                            double* mt_gauss, int32_t* status, int32_t nthreads) {
  if (robot < 0 || robot > 2 || n <= 0 || !seeds || !descs || n_descs <= 0 || !desc_of_env || !rec_f || !rec_i) return SAG_ERR_ARG;
  for (int k = 0; k < n_descs; k++)
    if (sag_task_desc_check(&descs[k])) return SAG_ERR_ARG;   // (sag_task_desc_check names the field)
  for (int i = 0; i < n; i++)
    if (desc_of_env[i] < 0 || desc_of_env[i] >= n_descs) return SAG_ERR_ARG;
  sag_world_config c;
  if (cfg) c = *cfg; else sag_world_config_default(&c);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > n) nthreads = n;
  std::vector<int> fails(nthreads, 0);
  auto work = [&](int t) {
    MT rs;
    for (int i = t; i < n; i += nthreads) {
      int rc = sample_one(robot, descs[desc_of_env[i]], seeds[i], c, first_episode, env_id0 + i,
                          rec_f + (size_t)i * SAG_REC_FLOATS, rec_i + (size_t)i * SAG_REC_INTS, rs);
      if (status) status[i] = rc;
      if (rc) fails[t]++;
      if (mt_key) std::memcpy(mt_key + (size_t)i * 624, rs.key, sizeof(rs.key));
      if (mt_pos) mt_pos[i] = rs.pos;
      if (mt_has_gauss) mt_has_gauss[i] = rs.has_gauss;
      if (mt_gauss) mt_gauss[i] = rs.gauss;
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int nf = 0;
  for (int f : fails) nf += f;
  return nf;  // number of envs whose sampling failed (ResamplingError), 0 = all good
}

int sag_sample_layouts(int32_t robot, int32_t n, const uint32_t* seeds, const int32_t* task_ids,
                       const sag_world_config* cfg, int32_t first_episode, int32_t env_id0, float* rec_f,
                       int32_t* rec_i, uint32_t* mt_key, int32_t* mt_pos, int32_t* mt_has_gauss,
                       double* mt_gauss, int32_t* status, int32_t nthreads) {
  if (!task_ids) return SAG_ERR_ARG;
  sag_task_desc descs[SAG_NUM_TASKS];
  for (int k = 0; k < SAG_NUM_TASKS; k++) sag_task_desc_default(k, &descs[k]);
  return sag_sample_layouts_desc(robot, n, seeds, descs, SAG_NUM_TASKS, task_ids, cfg, first_episode, env_id0, rec_f, rec_i, mt_key,
                                 mt_pos, mt_has_gauss, mt_gauss, status, nthreads);
}

}  // extern "C"
