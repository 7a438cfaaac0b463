// sag_api.hip - host side of the C ABI declared in include/sag.h.
//
// One context = one GPU = one HIP stream.  Device memory: the SoA world
// (S: [DEV_GROUPS][N] float4, I: tstate [N] + int4 [N]; sag_device.hpp didx / iaddr), AoS staging for records,
// pinned host + device staging for the host-pointer step.  No allocation, no
// synchronisation and no host<->device copy happens inside sag_step_device()
// (every buffer it needs exists after sag_create; timing events, off by default,
// are the exception), so callers may capture it into a hipGraph.  Every entry point
// selects the context's device first: contexts on different GPUs may share a host thread.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sag_device.hpp"

#ifndef SAG_SPLIT_MIN_ENVS
#define SAG_EARLY_FORK_MIN_ENVS 2097152  // tools/ab.sh run sweep: crossover between 1.5 M and 2 M envs
#define SAG_SPLIT_MIN_ENVS 262144  // measured crossover (tools/split_sweep.py, overlapped launches): Point ~260k, Car ~390k
#define SAG_SPLIT_MIN_ENVS_CAR 393216
#endif

using namespace sag;

namespace {

struct RobotInfo { int nu, obs_dim, nstep, nq, nv; double dt; };
const RobotInfo ROBOTS[3] = {
    {2, 60, 5, 3, 3, 0.004},      // point.xml:3, safe_adaptation_gym.py:15-19
    {2, 72, 10, 13, 11, 0.008},   // car.xml:3
    {12, 104, 12, 20, 19, 0.012}  // doggo.xml:2
};

thread_local std::string g_create_error;

}  // namespace

struct sag_ctx {
  sag_config cfg;
  RobotInfo rb;
  int N;
  hipStream_t stream = nullptr;
  // split form: the busy kernel (few, long, VALU-bound wavefronts) runs on a second stream beside
  // the quiet kernel (many short memory-bound ones); fork/join events order them against the main stream
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // Side by side with the busy kernel the quiet kernel requests this much unused LDS on top of its own
  // 10 KB: a quiet wavefront then frees at least what a busy one needs (13 KB), so a waiting busy
  // workgroup (higher stream priority) fits into any hole a quiet one leaves.  With 10-KB holes the busy
  // kernel, which sets the length of a step, is starved of LDS by the far more numerous quiet
  // workgroups once the batch exceeds what is resident at once (measured at 4 M envs: +8 % throughput,
  // tools/ab.sh run; SAG_QUIET_LDS_EXTRA overrides, bytes; < 0 = busy kernel's footprint minus the quiet one's).
  int quiet_lds_extra = -1;
  int early_fork = -1;  // quiet stream forks before the compaction: 1 / 0, -1 = by batch size (SAG_EARLY_FORK)
  bool overlap = true;  // SAG_OVERLAP=0: both kernels on the main stream, one after the other
  float* S = nullptr;
  int32_t* I = nullptr;
  float* G = nullptr;  // [3][NBODY][N] spill of body accelerations beyond the LDS pool
  // busy lists (split launches): [2][N] rows and two counters, used alternately by `phase`.  A step
  // consumes list[phase] and its classification appends list[phase ^ 1]; k_compact rebuilds
  // list[phase] from the busy bits whenever state was installed from outside (list_valid = false)
  int32_t* d_rows = nullptr; int32_t* d_count = nullptr;
  uint8_t* d_kind = nullptr;   // [N] kind of every busy env (BUSY_CLASSES)
  bool list_valid = false;
  int count_flip = 0;          // which of the counter sets 2 and 3 of d_count this step's compaction fills
  int32_t* last_count = nullptr;
  float* d_hot = nullptr;     // [N][HOT_FLOATS] hot records (split form), see sag_device.hpp
  bool hot_valid = false, use_hot = true;   // SAG_HOT=0 disables
  // SAG_INKERNEL_LIST=1: the step kernels append the next busy list themselves and k_compact only runs
  // after installs.  Measured no faster (0.37 vs 0.35 ms at 1M envs): the 15 us saved are lost to the
  // worse row locality of chunks in arrival order (k_compact emits rows sorted per 4096-env block).
  bool inkernel_list = false;
  // last installed layout (sag_reset)
  float* L_f = nullptr;   // [N][SAG_REC_FLOATS] AoS, device
  int32_t* L_i = nullptr; // [N][SAG_REC_INTS]
  // staging
  float* st_f = nullptr; int32_t* st_i = nullptr; int32_t* st_ids = nullptr;  // device, N records
  void* pin = nullptr; size_t pin_bytes = 0;  // pinned host
  // step staging (device)
  float* d_act = nullptr; float* d_noise = nullptr; uint32_t* d_tape = nullptr; size_t tape_cap = 0;
  float* d_obs = nullptr; float* d_rew = nullptr; uint8_t* d_cost = nullptr; uint8_t* d_done = nullptr;
  uint8_t* d_met = nullptr; int32_t* d_used = nullptr;
  int32_t* d_ext_cc = nullptr; uint32_t* d_ext_btn = nullptr; bool ext_pending = false;   // sag_set_ext_contacts
  // generic buffers for sag_lidar_cost
  void* scratch = nullptr; size_t scratch_bytes = 0;
  bool have_layout = false;
  // kernel timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double ev_ms = 0; int64_t ev_n = 0;
  bool timing = false;
  int phase = 0;       // busy-bit copy read by the next step launch
  int phase_used = 0;  // (the phase of the launch being built)
  int n_cu = 256;      // compute units of the device
  uint8_t* d_rgb = nullptr; size_t rgb_bytes = 0;  // [N][H][W][3] staging of sag_render, grown on demand
  double* d_dr = nullptr;    // Doggo: per-env result block of the physics kernel (k_doggo_physics: 2 envs per wavefront)
  int32_t* d_dg_sched = nullptr; int dg_phase = 0;   // Doggo: longest-first launch order (sag_doggo_coop.hpp); SAG_DOGGO_SCHED=0 turns it off
  bool dg_sched_on = true;
  int epw_override = 0;  // SAG_EPW (read once at create): envs per wavefront of the single-launch form
  int busy_e = 64;       // SAG_BUSY_E: envs per busy wavefront (0 = balanced over busy_slots: busy_wave_envs - measured slower, see there)
  int busy_slots = 0;    // SAG_BUSY_SLOTS: busy wavefronts resident at once (0 = 8 per CU: two per SIMD)
  int busy_kinds = -1;   // SAG_BUSY_KINDS: 1 = the busy list by kind of contact, 0 = one list; default: the Car (the Point's step is its quiet kernel: no gain)
  int kinds_min = -1;    // SAG_BUSY_KINDS_MIN: busy envs of the step before above which the kinds are used (default: 64 per resident slot; tests: 0)
  bool split = true;   // QUIET + BUSY launches; SAG_SPLIT=0/1 in the environment forces the form
  std::string err;
};

namespace {

int fail(sag_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(ctx, SAG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                 \
  } while (0)

int ensure_pin(sag_ctx* c, size_t bytes) {
  if (c->pin_bytes >= bytes) return 0;
  if (c->pin) (void)hipHostFree(c->pin);
  c->pin = nullptr; c->pin_bytes = 0;
  HIPCHK(c, hipHostMalloc(&c->pin, bytes, hipHostMallocDefault));
  c->pin_bytes = bytes;
  return 0;
}

int ensure_scratch(sag_ctx* c, size_t bytes) {
  if (c->scratch_bytes >= bytes) return 0;
  if (c->scratch) (void)hipFree(c->scratch);
  c->scratch = nullptr; c->scratch_bytes = 0;
  HIPCHK(c, hipMalloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return 0;
}

// ---- Doggo model tables (doggo.xml), built in fp64 and copied to constant memory ----
// Geometry numbers are the XML's: body offsets (:18-79), joint axes / ranges / springref (:21-71),
// geoms as (from, to, radius, density) with capsules unless noted (:5-6,15,48), touch sites (:27-28 ...).
void dg_build_model(DgModel& M) {
  static const int parent[DG_NB] = {-1, 0, 1, 0, 3, 0, 5, 6, 5, 8};
  static const double bpos[DG_NB][3] = {{0, 0, 0}, {.2, .1, 0}, {.098, .0566, -.05}, {.2, -.1, 0}, {.098, -.0566, -.05},
                                        {0, 0, 0}, {-.2, .1, 0}, {.098, .0566, -.05}, {-.2, -.1, 0}, {.098, -.0566, -.05}};
  static const int dof_body[DG_NV] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 5, 6, 6, 7, 8, 8, 9};
  static const double axis[DG_NJ][3] = {{0, 0, 1}, {0, 1, 0}, {-.5, .866, 0}, {0, 0, -1}, {0, 1, 0}, {.5, .866, 0}, {1, 0, 0},
                                        {0, 0, 1}, {0, 1, 0}, {-.5, .866, 0}, {0, 0, -1}, {0, 1, 0}, {.5, .866, 0}};
  static const double range_deg[DG_NJ][2] = {{-10, 30}, {-75, 15}, {-75, 0}, {-10, 30}, {-75, 15}, {-75, 0}, {-30, 30},
                                             {-10, 30}, {0, 135}, {-75, 0}, {-10, 30}, {0, 135}, {-75, 0}};
  static const double springref_deg[DG_NJ] = {0, -10, -20, 0, -10, -20, 0, 0, 0, -20, 0, 0, -20};
  static const int act_joint[12] = {0, 7, 10, 3, 1, 8, 11, 4, 2, 9, 12, 5};
  struct G { int body; double a[3], b[3], r, dens; bool capsule; };
  static const G geoms[14] = {
      {0, {0, 0, 0}, {.2, 0, 0}, .075, .5, false}, {0, {.1, 0, 0}, {.2, .1, 0}, .032, 5, true},
      {0, {.1, 0, 0}, {.2, -.1, 0}, .032, 5, true}, {1, {0, 0, 0}, {.098, .0566, -.05}, .032, 5, true},
      {2, {0, 0, 0}, {-.1176, -.0679, -.1}, .032, 5, true}, {3, {0, 0, 0}, {.098, -.0566, -.05}, .032, 5, true},
      {4, {0, 0, 0}, {-.1176, .0679, -.1}, .032, 5, true}, {5, {-.2, 0, 0}, {0, 0, 0}, .075, .5, false},
      {5, {-.1, 0, 0}, {-.2, .1, 0}, .032, 5, true}, {5, {-.1, 0, 0}, {-.2, -.1, 0}, .032, 5, true},
      {6, {0, 0, 0}, {.098, .0566, -.05}, .032, 5, true}, {7, {0, 0, 0}, {-.1176, -.0679, -.1}, .032, 5, true},
      {8, {0, 0, 0}, {.098, -.0566, -.05}, .032, 5, true}, {9, {0, 0, 0}, {-.1176, .0679, -.1}, .032, 5, true}};
  // floor contact points (oracle DG_FLOORPTS): geom, end, rim, merged, touch slot, half share
  struct Fp { int geom, end, rim, dbl, touch, half; };
  static const Fp floorpts[DG_NFP] = {
      {0, 1, 1, 0, -1, 0}, {0, 0, 1, 0, -1, 0},
      {3, 0, 0, 1, -1, 0}, {4, 0, 0, 1, 0, 1}, {4, 1, 0, 0, 4, 0},
      {5, 0, 0, 1, -1, 0}, {6, 0, 0, 1, 3, 1}, {6, 1, 0, 0, 7, 0},
      {7, 1, 1, 0, -1, 0}, {7, 0, 1, 0, -1, 0},
      {10, 0, 0, 1, -1, 0}, {11, 0, 0, 1, 1, 1}, {11, 1, 0, 0, 5, 0},
      {12, 0, 0, 1, -1, 0}, {13, 0, 0, 1, 2, 1}, {13, 1, 0, 0, 6, 0}};
  static const int geom_touch[14] = {-1, -1, -1, -1, 0, -1, 3, -1, -1, -1, -1, 1, -1, 2};
  const double pi = 3.14159265358979323846;
  memset(&M, 0, sizeof(M));
  for (int b = 0; b < DG_NB; b++) {
    M.parent[b] = parent[b];
    for (int k = 0; k < 3; k++) M.bpos[b][k] = bpos[b][k];
    unsigned anc = 0;
    for (int a = b; a >= 0; a = parent[a]) anc |= 1u << a;
    M.anc[b] = anc;
  }
  for (int i = 0; i < DG_NV; i++) M.dof_body[i] = dof_body[i];
  for (int j = 0; j < DG_NJ; j++) {
    const double n = std::sqrt(axis[j][0] * axis[j][0] + axis[j][1] * axis[j][1] + axis[j][2] * axis[j][2]);
    for (int k = 0; k < 3; k++) M.axis[j][k] = axis[j][k] / n;
    M.lo[j] = range_deg[j][0] * pi / 180; M.hi[j] = range_deg[j][1] * pi / 180;
    M.springref[j] = springref_deg[j] * pi / 180;
  }
  for (int k = 0; k < 12; k++) M.act_joint[k] = act_joint[k];
  double gm[14], gc[14][3], gI[14][9];
  for (int g = 0; g < 14; g++) {
    const G& q = geoms[g];
    const double d[3] = {q.b[0] - q.a[0], q.b[1] - q.a[1], q.b[2] - q.a[2]};
    const double L = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), r = q.r;
    const double u[3] = {d[0] / L, d[1] / L, d[2] / L};
    const double mc = q.dens * pi * r * r * L, ms = q.capsule ? q.dens * 4.0 / 3.0 * pi * r * r * r : 0.0;
    const double Ia = mc * r * r / 2 + ms * 2 * r * r / 5;
    const double It = mc * (r * r / 4 + L * L / 12) + ms * (2 * r * r / 5 + L * L / 4 + 3 * L * r / 8);
    gm[g] = mc + ms;
    for (int k = 0; k < 3; k++) gc[g][k] = 0.5 * (q.a[k] + q.b[k]);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) gI[g][3 * a + b] = It * ((a == b ? 1.0 : 0.0) - u[a] * u[b]) + Ia * u[a] * u[b];
    M.m[q.body] += gm[g];
    for (int k = 0; k < 3; k++) M.com[q.body][k] += gm[g] * gc[g][k];
  }
  for (int b = 0; b < DG_NB; b++)
    for (int k = 0; k < 3; k++) M.com[b][k] /= M.m[b];
  for (int g = 0; g < 14; g++) {
    const int b = geoms[g].body;
    const double d[3] = {gc[g][0] - M.com[b][0], gc[g][1] - M.com[b][1], gc[g][2] - M.com[b][2]};
    const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 3; c++) M.I[b][3 * a + c] += gI[g][3 * a + c] + gm[g] * ((a == c ? d2 : 0.0) - d[a] * d[c]);
  }
  for (int s = 0; s < DG_NFP; s++) {
    const Fp& f = floorpts[s];
    M.fp_code[s] = f.geom | f.end << 4 | f.rim << 5 | f.dbl << 6 | (f.touch + 1) << 7 | f.half << 11;
  }
  static_assert(DG_NGEOM == 14, "geom table");
  for (int g = 0; g < DG_NGEOM; g++) {
    M.geom_body[g] = geoms[g].body; M.geom_capsule[g] = geoms[g].capsule ? 1 : 0; M.geom_r[g] = geoms[g].r;
    M.geom_touch[g] = geom_touch[g];
    // ankle geoms = the shin capsules on bodies 2, 4 (front legs: blue) and 7, 9 (rear legs: green)
    M.geom_ankle[g] = (geoms[g].body == 2 || geoms[g].body == 4) ? 1 : ((geoms[g].body == 7 || geoms[g].body == 9) ? 2 : 0);
    for (int k = 0; k < 3; k++) { M.geom_a[g][k] = geoms[g].a[k]; M.geom_b[g][k] = geoms[g].b[k]; }
  }
}

// drain finished timing events into the running mean
void drain_events(sag_ctx* c) {
  for (size_t k = 0; k < c->ev_used; k++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev_pool[k].first, c->ev_pool[k].second) == hipSuccess) {
      c->ev_ms += ms; c->ev_n += 1;
    }
  }
  c->ev_used = 0;
}

// a pair of timing events for one launch (nullptr, nullptr when timing is off); bounded pool
int timing_events(sag_ctx* c, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!c->timing) return 0;
  if (c->ev_used == c->ev_pool.size()) {
    if (c->ev_pool.size() >= 4096) {  // fold what has finished
      HIPCHK(c, hipStreamSynchronize(c->stream));
      drain_events(c);
    } else {
      hipEvent_t a0, a1;
      HIPCHK(c, hipEventCreate(&a0));
      HIPCHK(c, hipEventCreate(&a1));
      c->ev_pool.emplace_back(a0, a1);
    }
  }
  *e0 = c->ev_pool[c->ev_used].first; *e1 = c->ev_pool[c->ev_used].second;
  c->ev_used++;
  return 0;
}

int launch_step(sag_ctx* c, const float* d_act, const float* d_noise, const uint32_t* d_tape,
                int tape_len, int nstep, float* d_obs, float* d_rew, uint8_t* d_cost,
                uint8_t* d_done, uint8_t* d_met, int32_t* d_used, int observe_only) {
  HIPCHK(c, hipSetDevice(c->cfg.device));
  StepArgs a;
  a.S = c->S; a.I = c->I; a.N = c->N;
  a.actions = d_act; a.noise = d_noise; a.tape = d_tape; a.tape_len = tape_len;
  a.nstep = nstep < 0 ? c->rb.nstep : nstep;
  a.nstep_table = c->rb.nstep;
  a.h = (float)c->rb.dt;
  a.car = car_fric_constants(a.h);
  a.key0 = (uint32_t)(c->cfg.seed & 0xffffffffu); a.key1 = (uint32_t)(c->cfg.seed >> 32);  // sag_set_seed
  a.obs = d_obs; a.reward = d_rew; a.cost = d_cost; a.done = d_done; a.goal_met = d_met;
  a.tape_used = d_used; a.max_vases = c->cfg.max_vases; a.max_hazards = c->cfg.max_hazards;
  a.max_pillars = c->cfg.max_pillars; a.max_buttons = c->cfg.max_buttons; a.observe_only = observe_only;
  a.has_box = c->cfg.has_box; a.G = c->G;
  // busy lists: [phase][BUSY_CLASSES][N] rows; counters [4][BUSY_CLASSES]: two for the in-kernel lists (by phase), two used alternately by k_compact
  a.phase = c->phase; a.rows = c->d_rows + (size_t)c->phase * BUSY_CLASSES * c->N; a.count = c->d_count + c->phase * BUSY_CLASSES;
  a.rows_next = nullptr; a.count_next = nullptr; a.DR = nullptr; a.dg_sched = nullptr; a.dg_phase = -1;
  a.hot = nullptr; a.hot_haz = nullptr;
  // external contact results: for the one step with nstep == 0 that follows sag_set_ext_contacts
  const bool use_ext = c->ext_pending && !observe_only && a.nstep == 0;
  a.ext_cc = use_ext ? c->d_ext_cc : nullptr; a.ext_btn = use_ext ? c->d_ext_btn : nullptr;
  if (!observe_only) c->ext_pending = false;
  {
    // single-launch form: aim for four wavefronts per CU (Doggo: one - only one fits its LDS working
    // set), down to 16 (Doggo 8) envs per wavefront (tools/epw_sweep.py: Car 4096 envs 0.58 -> 0.48 ms)
    const bool dg = c->cfg.robot == SAG_ROBOT_DOGGO;
    int epw = 64;
    while (epw > (dg ? 8 : 16) && (c->N + epw - 1) / epw < (dg ? 1 : 4) * c->n_cu) epw >>= 1;
    if (c->epw_override > 0) epw = c->epw_override;
    a.envs_per_wave = epw < 1 ? 1 : (epw > 64 ? 64 : epw);
    a.busy_envs = c->busy_e; a.busy_kinds = c->busy_kinds < 0 ? c->cfg.robot == SAG_ROBOT_CAR : c->busy_kinds; a.kind = a.busy_kinds ? c->d_kind : nullptr; a.busy_total = c->d_count + 4 * BUSY_CLASSES;
    a.busy_slots = c->busy_slots > 0 ? c->busy_slots : 8 * c->n_cu;
  }
  c->phase_used = c->phase;
  if (!observe_only) c->phase ^= 1;
  const int blocks = (c->N + WAVE - 1) / WAVE;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!observe_only) {
    int rc = timing_events(c, &e0, &e1);
    if (rc) return rc;
    if (e0) HIPCHK(c, hipEventRecord(e0, c->stream));
  }
  const bool btn = c->cfg.max_buttons > 0, tbox = c->cfg.has_box != 0;
  // split form: QUIET kernel over every env whose busy bit is clear, then BUSY kernel over the
  // rest (compacted per 256-env neighbourhood).  observe() and SAG_SPLIT=0 use the single form.
  const bool split = c->split && !observe_only && c->cfg.robot != SAG_ROBOT_DOGGO;
  hipStream_t quiet_stream = c->stream;
  if (split) {
    a.rows_next = c->d_rows + (size_t)(c->phase_used ^ 1) * BUSY_CLASSES * c->N; a.count_next = c->d_count + (c->phase_used ^ 1) * BUSY_CLASSES;
    if (c->use_hot && c->d_hot) {
      a.hot = c->d_hot; a.hot_haz = c->d_hot + (size_t)c->N * HOT_FLOATS;
      if (!c->hot_valid) {
        hipLaunchKernelGGL(k_hot_refresh, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N, c->d_hot, c->d_hot + (size_t)c->N * HOT_FLOATS);
        c->hot_valid = true;
      }
    }
    if (!c->inkernel_list) { a.rows_next = nullptr; a.count_next = nullptr; c->list_valid = false; }
    else HIPCHK(c, hipMemsetAsync(a.count_next, 0, BUSY_CLASSES * sizeof(int32_t), c->stream));
    // The quiet kernel needs no list (it reads the busy bits), so its stream forks off BEFORE the
    // compaction: k_compact only looks at this step's copy of the bit, which the quiet kernel never
    // changes (it rewrites tstate words of quiet envs with that bit still clear), so the two may overlap.
    // Only when the batch is large enough to keep the chip full (>= SAG_EARLY_FORK_MIN_ENVS): below that a
    // step is as long as its busy wavefronts, which must then be resident first (1 M envs: 15 % slower
    // with the early fork, 4 M envs: 10 % faster; profiles/r01_v16_early_fork_ab.txt).
    const bool early_fork = c->overlap && !c->inkernel_list &&
                            (c->early_fork < 0 ? (c->N >= SAG_EARLY_FORK_MIN_ENVS && c->cfg.robot != SAG_ROBOT_CAR) : c->early_fork != 0);
    // (Car: its busy kernel is 4x longer than the quiet one and sets the step; with the early fork the quiet grid
    // occupies the chip first and the busy wavefronts start late: 3.49 vs 3.15 ms at 4 M envs)
    if (early_fork) {
      HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      quiet_stream = c->stream2;
    }
    if (!c->inkernel_list) {
      // two counters used alternately: this step's compaction zeroes the one the next step will use
      a.count = c->d_count + (2 + c->count_flip) * BUSY_CLASSES;
      if (a.busy_kinds)
        hipLaunchKernelGGL(k_compact<true>, dim3((c->N + COMPACT_ENVS - 1) / COMPACT_ENVS), dim3(256), 0, c->stream, c->I, c->d_kind, c->N, a.phase,
                           a.rows, a.count, c->d_count + (2 + (c->count_flip ^ 1)) * BUSY_CLASSES, a.busy_total, c->kinds_min >= 0 ? c->kinds_min : 64 * a.busy_slots);
      else
        hipLaunchKernelGGL(k_compact<false>, dim3((c->N + COMPACT_ENVS - 1) / COMPACT_ENVS), dim3(256), 0, c->stream, c->I, c->d_kind, c->N, a.phase,
                           a.rows, a.count, c->d_count + (2 + (c->count_flip ^ 1)) * BUSY_CLASSES, a.busy_total, 0);
      c->count_flip ^= 1;
    } else if (!c->list_valid) {
      HIPCHK(c, hipMemsetAsync(a.count, 0, BUSY_CLASSES * sizeof(int32_t), c->stream));
      if (a.busy_kinds)
        hipLaunchKernelGGL(k_compact<true>, dim3((c->N + COMPACT_ENVS - 1) / COMPACT_ENVS), dim3(256), 0, c->stream, c->I, c->d_kind, c->N, a.phase,
                           a.rows, a.count, (int32_t*)nullptr, a.busy_total, c->kinds_min >= 0 ? c->kinds_min : 64 * a.busy_slots);
      else
        hipLaunchKernelGGL(k_compact<false>, dim3((c->N + COMPACT_ENVS - 1) / COMPACT_ENVS), dim3(256), 0, c->stream, c->I, c->d_kind, c->N, a.phase,
                           a.rows, a.count, (int32_t*)nullptr, a.busy_total, 0);
    }
    c->last_count = a.count;
    c->list_valid = true;
    if (c->overlap && !early_fork) {
      HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      quiet_stream = c->stream2;
    }
  }
  // LDS the quiet kernel asks for beyond what it uses (see sag_ctx::quiet_lds_extra)
  size_t quiet_lds_extra = 0;
  if (split && c->overlap) {
    const int quiet_own = (LS_YAW + (c->cfg.robot == SAG_ROBOT_CAR ? 25 : 17)) * WAVE * (int)sizeof(float);
    const int busy_own = LDS_FLOATS * (int)sizeof(float) + WAVE * (int)sizeof(int);
    const int want = c->quiet_lds_extra >= 0 ? c->quiet_lds_extra : (busy_own > quiet_own ? (busy_own - quiet_own + 511) / 512 * 512 : 0);
    quiet_lds_extra = (size_t)want;
  }
#define SAG_LAUNCH3(ROB, B_, X_)                                                                         \
  do {                                                                                                  \
    if (split) {                                                                                        \
      hipLaunchKernelGGL((k_step_busy<ROB, B_, X_>), dim3(busy_grid(c->N, a.busy_envs, a.busy_slots)), dim3(WAVE), 0, c->stream, a); \
      hipLaunchKernelGGL((k_step_quiet<ROB, B_, X_>), dim3(blocks), dim3(WAVE), quiet_lds_extra, quiet_stream, a); \
    } else {                                                                                            \
      hipLaunchKernelGGL((k_step<ROB, B_, X_>), dim3((c->N + a.envs_per_wave - 1) / a.envs_per_wave), dim3(WAVE), 0, c->stream, a); \
    }                                                                                                   \
  } while (0)
#define SAG_LAUNCH(ROB)                            \
  do {                                             \
    if (!btn && !tbox) SAG_LAUNCH3(ROB, false, false); \
    else if (btn && !tbox) SAG_LAUNCH3(ROB, true, false); \
    else if (!btn && tbox) SAG_LAUNCH3(ROB, false, true); \
    else SAG_LAUNCH3(ROB, true, true);             \
  } while (0)
  // (-DSAG_ONLY_ROBOT=<id>: kernel-tuning builds that instantiate one robot's step kernels only - a third of the
  // compile time; the shipped library has them all)
#ifndef SAG_ONLY_ROBOT
#define SAG_ONLY_ROBOT -1
#endif
  if (false) {}
#if SAG_ONLY_ROBOT < 0 || SAG_ONLY_ROBOT == 0
  else if (c->cfg.robot == SAG_ROBOT_POINT) SAG_LAUNCH(SAG_ROBOT_POINT);
#endif
#if SAG_ONLY_ROBOT < 0 || SAG_ONLY_ROBOT == 1
  else if (c->cfg.robot == SAG_ROBOT_CAR) SAG_LAUNCH(SAG_ROBOT_CAR);
#endif
  else if (c->cfg.robot != SAG_ROBOT_DOGGO) return fail(c, SAG_ERR_UNSUPPORTED, "this build holds the step kernels of robot %d only", SAG_ONLY_ROBOT);
  else {
    // Doggo: wave-cooperative physics (32 lanes per env) + the generic step without physics
    a.DR = c->d_dr;
    const bool sched = c->dg_sched_on && c->d_dg_sched && !observe_only;
    a.dg_sched = sched ? c->d_dg_sched : nullptr;
    a.dg_phase = sched ? c->dg_phase : -1;
    if (sched) c->dg_phase = (c->dg_phase + 1) % 6;   // (the kernel uses it mod 2 and mod 3)
    hipLaunchKernelGGL(k_doggo_physics, dim3(sched ? dc_sched_blocks(c->N) : (c->N + DC_EPW - 1) / DC_EPW), dim3(32 * DC_EPW), 0, c->stream, a, c->d_dr);
    hipLaunchKernelGGL((k_step_doggo_post<true, true>), dim3(blocks), dim3(WAVE), 0, c->stream, a);
  }
#undef SAG_LAUNCH
#undef SAG_LAUNCH3
  if (split && c->overlap) {
    HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  }
  if (e1) HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipGetLastError());
  return 0;
}

// A team of lanes per pose (k_lidar_cost_team): lane s of a team takes the points s, s + LPP, ...  Few poses (the reference's own
// 4096): 16 lanes, so that the chip is full and a pose's points are not one serial chain; more poses, smaller teams - 8 up to 262 144
// poses, 4 beyond (4 M poses x 21 points, kernel only: teams of 4 0.474 ms, of 8 0.502, of 2 0.641, of 16 0.684; the lane-per-pose
// kernel with its points in registers 0.537, LDS-staged 0.792 - its 12.8 KB of staging / tile per wavefront hold it at three
// wavefronts per SIMD, a team's 3-KB tile does not; profiles/r04_lidar_forms.txt).  SAG_LIDAR_TEAM = 0 / 2 / 4 / 8 / 16 forces the
// form (0: a lane per pose - K <= 21 the register-resident kernel, more points per env or SAG_LIDAR_REG=0 the LDS-staged one;
// the tests run every form on the same inputs).
void launch_lidar_cost(sag_ctx* c, int n, int K, const float* d_robot, const float* d_pts, const uint8_t* d_grp, float hazard_size,
                       float* d_lid, int32_t* d_bins, uint8_t* d_cost) {
  static const bool use_reg = [] { const char* e = getenv("SAG_LIDAR_REG"); return !e || atoi(e) != 0; }();
  int team = n <= 16384 ? 16 : (n <= 262144 ? 8 : 4);
  if (const char* e = getenv("SAG_LIDAR_TEAM")) team = atoi(e);
  if (team == 2)
    hipLaunchKernelGGL(k_lidar_cost_team<2>, dim3((n + 31) / 32), dim3(WAVE), 0, c->stream, n, K, d_robot, d_pts, d_grp, hazard_size,
                       d_lid, d_bins, d_cost);
  else if (team == 8)
    hipLaunchKernelGGL(k_lidar_cost_team<8>, dim3((n + 7) / 8), dim3(WAVE), 0, c->stream, n, K, d_robot, d_pts, d_grp, hazard_size,
                       d_lid, d_bins, d_cost);
  else if (team >= 16)
    hipLaunchKernelGGL(k_lidar_cost_team<16>, dim3((n + 3) / 4), dim3(WAVE), 0, c->stream, n, K, d_robot, d_pts, d_grp, hazard_size,
                       d_lid, d_bins, d_cost);
  else if (team >= 4)
    hipLaunchKernelGGL(k_lidar_cost_team<4>, dim3((n + 15) / 16), dim3(WAVE), 0, c->stream, n, K, d_robot, d_pts, d_grp, hazard_size,
                       d_lid, d_bins, d_cost);
  else if (K <= LC_KREG && use_reg)
    hipLaunchKernelGGL(k_lidar_cost_reg, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), 0, c->stream, n, K, d_robot, d_pts, d_grp,
                       hazard_size, d_lid, d_bins, d_cost);
  else
    hipLaunchKernelGGL(k_lidar_cost, dim3((n + WAVE - 1) / WAVE), dim3(WAVE), lidar_cost_lds_bytes(K), c->stream, n, K, d_robot, d_pts,
                       d_grp, hazard_size, d_lid, d_bins, d_cost);
}

}  // namespace

extern "C" {

int sag_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sag_robot_info(int32_t robot, int32_t out[5], double* dt) {
  if (robot < 0 || robot > 2 || !out || !dt) return SAG_ERR_ARG;
  const RobotInfo& r = ROBOTS[robot];
  out[0] = r.nu; out[1] = r.obs_dim; out[2] = r.nstep; out[3] = r.nq; out[4] = r.nv;
  *dt = r.dt;
  return SAG_OK;
}

const char* sag_last_error(const sag_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sag_create(const sag_config* cfg, sag_ctx** out) {
  if (!cfg || !out) return fail(nullptr, SAG_ERR_ARG, "null argument");
  *out = nullptr;
  if (cfg->abi_version != SAG_ABI_VERSION)
    return fail(nullptr, SAG_ERR_ARG, "ABI version %d, library is %d", cfg->abi_version, SAG_ABI_VERSION);
  if (cfg->robot < 0 || cfg->robot > 2) return fail(nullptr, SAG_ERR_ARG, "bad robot %d", cfg->robot);
  if (cfg->n_envs <= 0) return fail(nullptr, SAG_ERR_ARG, "n_envs must be positive");
  if (cfg->max_hazards < 0 || cfg->max_hazards > SAG_MAX_HAZARDS || cfg->max_vases < 0 ||
      cfg->max_vases > SAG_MAX_VASES || cfg->max_pillars < 0 || cfg->max_pillars > SAG_MAX_PILLARS ||
      cfg->max_buttons < 0 || cfg->max_buttons > SAG_MAX_BUTTONS)
    return fail(nullptr, SAG_ERR_ARG, "capacity out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, SAG_ERR_NODEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, SAG_ERR_NODEVICE, "device %d out of range (%d visible)", cfg->device, ndev);
  sag_ctx* c = new sag_ctx();
  c->cfg = *cfg; c->rb = ROBOTS[cfg->robot]; c->N = cfg->n_envs;
  // below ~one resident round of wavefronts a step is latency-bound and two dependent launches
  // cost more than the divergence they remove
  c->split = c->N >= (cfg->robot == SAG_ROBOT_CAR ? SAG_SPLIT_MIN_ENVS_CAR : SAG_SPLIT_MIN_ENVS);
  if (const char* e = getenv("SAG_SPLIT")) c->split = atoi(e) != 0;
  const size_t N = (size_t)c->N;
#define CREATE_CHK(call)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      int rc_ = fail(nullptr, SAG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));       \
      sag_destroy(c);                                                                            \
      return rc_;                                                                                \
    }                                                                                            \
  } while (0)
  CREATE_CHK(hipSetDevice(cfg->device));
  CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  {
    int lo = 0, hi = 0;  // (numerically lower = higher priority)
    CREATE_CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CREATE_CHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
  }
  CREATE_CHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  CREATE_CHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  if (const char* e = getenv("SAG_OVERLAP")) c->overlap = atoi(e) != 0;
  if (const char* e = getenv("SAG_EARLY_FORK")) c->early_fork = atoi(e);
  if (const char* e = getenv("SAG_QUIET_LDS_EXTRA")) c->quiet_lds_extra = atoi(e) > 65536 ? 65536 : atoi(e);
  if (const char* e = getenv("SAG_INKERNEL_LIST")) c->inkernel_list = atoi(e) != 0;
  if (const char* e = getenv("SAG_HOT")) c->use_hot = atoi(e) != 0;
  if (const char* e = getenv("SAG_EPW")) c->epw_override = atoi(e);
  if (const char* e = getenv("SAG_BUSY_E")) c->busy_e = atoi(e) > 64 ? 64 : atoi(e);
  if (const char* e = getenv("SAG_BUSY_SLOTS")) c->busy_slots = atoi(e);
  if (const char* e = getenv("SAG_BUSY_KINDS")) c->busy_kinds = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("SAG_BUSY_KINDS_MIN")) c->kinds_min = atoi(e);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  }
  CREATE_CHK(hipMalloc(&c->S, N * DEV_FLOATS * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->I, icount(N) * sizeof(int32_t)));
  CREATE_CHK(hipMalloc(&c->G, N * 3 * NBODY * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->d_rows, 2 * BUSY_CLASSES * N * sizeof(int32_t)));
  CREATE_CHK(hipMalloc(&c->d_kind, N));
  CREATE_CHK(hipMemset(c->d_kind, 0, N));
  CREATE_CHK(hipMalloc(&c->d_count, (4 * BUSY_CLASSES + 1) * sizeof(int32_t)));   // (+ the env count of the last busy launch)
  CREATE_CHK(hipMemset(c->d_count, 0, (4 * BUSY_CLASSES + 1) * sizeof(int32_t)));
  CREATE_CHK(hipMalloc(&c->L_f, N * SAG_REC_FLOATS * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->L_i, N * SAG_REC_INTS * sizeof(int32_t)));
  CREATE_CHK(hipMalloc(&c->st_f, N * SAG_REC_FLOATS * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->st_i, N * SAG_REC_INTS * sizeof(int32_t)));
  CREATE_CHK(hipMalloc(&c->st_ids, N * sizeof(int32_t)));
  CREATE_CHK(hipMalloc(&c->d_act, N * c->rb.nu * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->d_noise, N * c->rb.nu * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->d_obs, N * c->rb.obs_dim * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->d_rew, N * 2 * sizeof(float)));
  CREATE_CHK(hipMalloc(&c->d_cost, N));
  CREATE_CHK(hipMalloc(&c->d_done, N));
  CREATE_CHK(hipMalloc(&c->d_met, N));
  CREATE_CHK(hipMalloc(&c->d_used, N * sizeof(int32_t)));
  if (c->split && c->use_hot && cfg->robot != SAG_ROBOT_DOGGO)
    CREATE_CHK(hipMalloc(&c->d_hot, N * (HOT_FLOATS + 20) * sizeof(float)));
  if (cfg->robot == SAG_ROBOT_DOGGO) {
    CREATE_CHK(hipMalloc(&c->d_dr, N * DR_STRIDE * sizeof(double)));
    CREATE_CHK(hipMalloc(&c->d_dg_sched, dc_sched_ints(N) * sizeof(int32_t)));
    CREATE_CHK(hipMemset(c->d_dg_sched, 0, dc_sched_ints(N) * sizeof(int32_t)));
    if (const char* e = getenv("SAG_DOGGO_SCHED")) c->dg_sched_on = atoi(e) != 0;
  }
  if (cfg->robot == SAG_ROBOT_DOGGO) {
    DgModel model;
    dg_build_model(model);
    CREATE_CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dg), &model, sizeof(model)));
  }
  CREATE_CHK(hipMemsetAsync(c->d_cost, 0, N, c->stream));
  CREATE_CHK(hipMemsetAsync(c->d_obs, 0, N * c->rb.obs_dim * sizeof(float), c->stream));
  CREATE_CHK(hipMemsetAsync(c->S, 0, N * DEV_FLOATS * sizeof(float), c->stream));
  CREATE_CHK(hipMemsetAsync(c->I, 0, icount(N) * sizeof(int32_t), c->stream));
  CREATE_CHK(hipStreamSynchronize(c->stream));
#undef CREATE_CHK
  *out = c;
  return SAG_OK;
}

int sag_destroy(sag_ctx* c) {
  if (!c) return SAG_OK;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  void* bufs[] = {c->S, c->I, c->G, c->d_rows, c->d_count, c->d_kind, c->L_f, c->L_i, c->st_f, c->st_i, c->st_ids, c->d_act, c->d_noise,
                  c->d_tape, c->d_obs, c->d_rew, c->d_cost, c->d_done, c->d_met, c->d_used, c->scratch, c->d_rgb, c->d_dr, c->d_dg_sched, c->d_hot,
                  c->d_ext_cc, c->d_ext_btn};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SAG_OK;
}

// a new episode starts without a cost flag: sag_render's cost overlay reads d_cost of the last host-buffer step
__global__ void k_clear_cost(uint8_t* cost, const int32_t* ids, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) cost[ids ? ids[j] : j] = 0;
}

static int upload_records(sag_ctx* c, const int32_t* env_ids, int32_t n, const float* rec_f,
                          const int32_t* rec_i) {
  if (n <= 0 || n > c->N || !rec_f || !rec_i) return fail(c, SAG_ERR_ARG, "bad record batch (n=%d)", n);
  if (env_ids)
    for (int k = 0; k < n; k++)
      if (env_ids[k] < 0 || env_ids[k] >= c->N) return fail(c, SAG_ERR_ARG, "env id %d out of range", env_ids[k]);
  for (int k = 0; k < n; k++) {
    const int32_t* ri = rec_i + (size_t)k * SAG_REC_INTS;
    if (ri[SAG_I_TASK] < 0 || ri[SAG_I_TASK] >= SAG_NUM_TASKS || ri[SAG_I_NH] < 0 ||
        ri[SAG_I_NH] > c->cfg.max_hazards || ri[SAG_I_NV] < 0 || ri[SAG_I_NV] > c->cfg.max_vases ||
        ri[SAG_I_NP] < 0 || ri[SAG_I_NP] > c->cfg.max_pillars || ri[SAG_I_NB] < 0 ||
        ri[SAG_I_NB] > c->cfg.max_buttons || ri[SAG_I_BOX_KIND] < 0 || ri[SAG_I_BOX_KIND] > 3 ||
        (ri[SAG_I_BOX_KIND] != 0 && !c->cfg.has_box) || ri[SAG_I_GOAL_BUTTON] < 0 ||
        ri[SAG_I_GOAL_BUTTON] >= SAG_MAX_BUTTONS || ri[SAG_I_BTN_TIMER] < 0 || ri[SAG_I_BTN_TIMER] > 5 ||
        ri[SAG_I_CATCH_TIMER] < 0 || ri[SAG_I_CATCH_TIMER] > 10 ||
        ri[SAG_I_EPISODE] < 0 || ri[SAG_I_EPISODE] > 0xffffff)   // the nonce is 24 bits of the flags word: no silent truncation
      return fail(c, SAG_ERR_ARG, "record %d exceeds the context's capacities or has a bad field", k);
  }
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipMemcpyAsync(c->st_f, rec_f, (size_t)n * SAG_REC_FLOATS * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_i, rec_i, (size_t)n * SAG_REC_INTS * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (env_ids)
    HIPCHK(c, hipMemcpyAsync(c->st_ids, env_ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  return 0;
}

int sag_set_layout(sag_ctx* c, const int32_t* env_ids, int32_t n, const float* rec_f, const int32_t* rec_i) {
  if (!c) return SAG_ERR_ARG;
  int rc = upload_records(c, env_ids, n, rec_f, rec_i);
  if (rc) return rc;
  hipLaunchKernelGGL(k_install, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N,
                     env_ids ? c->st_ids : nullptr, n, c->st_f, c->st_i, 1);
  c->list_valid = false; c->hot_valid = false;  // busy bits / state changed outside a step
  c->ext_pending = false;   // contact results supplied for another state do not carry over (sag_set_ext_contacts: "for the NEXT step")
  hipLaunchKernelGGL(k_clear_cost, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_cost, env_ids ? c->st_ids : nullptr, n);
  // keep a copy for sag_reset: read the installed state back into the AoS layout store
  hipLaunchKernelGGL(k_extract, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N,
                     env_ids ? c->st_ids : nullptr, n, c->st_f, c->st_i);
  HIPCHK(c, hipGetLastError());
  // the layout store keeps the records as installed, by env id (one scatter launch, not 2 n copies)
  if (!env_ids) {
    HIPCHK(c, hipMemcpyAsync(c->L_f, c->st_f, (size_t)n * SAG_REC_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->L_i, c->st_i, (size_t)n * SAG_REC_INTS * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
  } else {
    const size_t pieces = (size_t)n * (SAG_REC_FLOATS / 4 + SAG_REC_INTS / 4);
    hipLaunchKernelGGL(k_move_rows, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, c->stream, c->L_f, c->L_i, c->st_f,
                       c->st_i, c->st_ids, n, 0, 0);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_layout = true;
  return SAG_OK;
}

int sag_set_state(sag_ctx* c, const int32_t* env_ids, int32_t n, const float* rec_f, const int32_t* rec_i) {
  if (!c) return SAG_ERR_ARG;
  int rc = upload_records(c, env_ids, n, rec_f, rec_i);
  if (rc) return rc;
  hipLaunchKernelGGL(k_install, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N,
                     env_ids ? c->st_ids : nullptr, n, c->st_f, c->st_i, 0);
  c->list_valid = false; c->hot_valid = false;  // busy bits / state changed outside a step
  c->ext_pending = false;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_layout = true;
  return SAG_OK;
}

int sag_get_state(sag_ctx* c, const int32_t* env_ids, int32_t n, float* rec_f, int32_t* rec_i) {
  if (!c) return SAG_ERR_ARG;
  if (n <= 0 || n > c->N || !rec_f || !rec_i) return fail(c, SAG_ERR_ARG, "bad record batch (n=%d)", n);
  if (env_ids)
    for (int k = 0; k < n; k++)
      if (env_ids[k] < 0 || env_ids[k] >= c->N) return fail(c, SAG_ERR_ARG, "env id %d out of range", env_ids[k]);
  HIPCHK(c, hipSetDevice(c->cfg.device));
  if (env_ids)
    HIPCHK(c, hipMemcpyAsync(c->st_ids, env_ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_extract, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N,
                     env_ids ? c->st_ids : nullptr, n, c->st_f, c->st_i);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(rec_f, c->st_f, (size_t)n * SAG_REC_FLOATS * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(rec_i, c->st_i, (size_t)n * SAG_REC_INTS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}

int sag_reset(sag_ctx* c, const int32_t* env_ids, int32_t n) {
  if (!c) return SAG_ERR_ARG;
  if (!c->have_layout) return fail(c, SAG_ERR_STATE, "sag_reset before sag_set_layout");
  if (n <= 0 || n > c->N) return fail(c, SAG_ERR_ARG, "bad n=%d", n);
  HIPCHK(c, hipSetDevice(c->cfg.device));
  // a new episode of the same layout: the episode nonce of the counter-based generator advances
  if (!env_ids) {
    hipLaunchKernelGGL(k_bump_episode, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->L_i, n);
    hipLaunchKernelGGL(k_install, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N,
                       (const int32_t*)nullptr, n, c->L_f, c->L_i, 0);
  } else {
    for (int k = 0; k < n; k++)
      if (env_ids[k] < 0 || env_ids[k] >= c->N) return fail(c, SAG_ERR_ARG, "env id %d out of range", env_ids[k]);
    // gather the chosen layout rows into staging (one launch), then install
    HIPCHK(c, hipMemcpyAsync(c->st_ids, env_ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    const size_t pieces = (size_t)n * (SAG_REC_FLOATS / 4 + SAG_REC_INTS / 4);
    hipLaunchKernelGGL(k_move_rows, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, c->stream, c->st_f, c->st_i, c->L_f,
                       c->L_i, c->st_ids, n, 1, 1);
    hipLaunchKernelGGL(k_install, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->I, c->N,
                       c->st_ids, n, c->st_f, c->st_i, 0);
  }
  hipLaunchKernelGGL(k_clear_cost, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_cost, env_ids ? c->st_ids : nullptr, n);
  c->list_valid = false; c->hot_valid = false;  // busy bits / state changed outside a step
  c->ext_pending = false;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}

int sag_step_device(sag_ctx* c, const float* d_actions, const float* d_noise, int32_t nstep, float* d_obs,
                    float* d_reward, uint8_t* d_cost, uint8_t* d_done, uint8_t* d_goal_met) {
  if (!c) return SAG_ERR_ARG;
  if (!c->have_layout) return fail(c, SAG_ERR_STATE, "sag_step before sag_set_layout");
  if (!d_actions) return fail(c, SAG_ERR_ARG, "actions is NULL");
  return launch_step(c, d_actions, d_noise, nullptr, 0, nstep, d_obs, d_reward, d_cost, d_done, d_goal_met,
                     nullptr, 0);
}

int sag_wait(sag_ctx* c) {
  if (!c) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}

int sag_set_seed(sag_ctx* c, uint64_t seed) {
  if (!c) return SAG_ERR_ARG;
  c->cfg.seed = seed;   // read by the next launch (StepArgs::key0/1)
  return SAG_OK;
}

int sag_step(sag_ctx* c, const float* actions, const float* noise, const uint32_t* tape, int32_t tape_len,
             int32_t nstep, float* obs, float* reward, uint8_t* cost, uint8_t* done, uint8_t* goal_met,
             int32_t* tape_used) {
  if (!c) return SAG_ERR_ARG;
  if (!c->have_layout) return fail(c, SAG_ERR_STATE, "sag_step before sag_set_layout");
  if (!actions) return fail(c, SAG_ERR_ARG, "actions is NULL");
  if (tape && tape_len <= 0) return fail(c, SAG_ERR_ARG, "tape without tape_len");
  const size_t N = (size_t)c->N, nu = (size_t)c->rb.nu, od = (size_t)c->rb.obs_dim;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipMemcpyAsync(c->d_act, actions, N * nu * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (noise) HIPCHK(c, hipMemcpyAsync(c->d_noise, noise, N * nu * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (tape) {
    size_t need = N * (size_t)tape_len * sizeof(uint32_t);
    if (c->tape_cap < need) {
      if (c->d_tape) (void)hipFree(c->d_tape);
      c->d_tape = nullptr; c->tape_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_tape, need));
      c->tape_cap = need;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_tape, tape, need, hipMemcpyHostToDevice, c->stream));
  }
  int rc = launch_step(c, c->d_act, noise ? c->d_noise : nullptr, tape ? c->d_tape : nullptr, tape_len, nstep,
                       obs ? c->d_obs : nullptr, reward ? c->d_rew : nullptr, cost ? c->d_cost : nullptr,
                       done ? c->d_done : nullptr, goal_met ? c->d_met : nullptr,
                       tape_used ? c->d_used : nullptr, 0);
  if (rc) return rc;
  if (obs) HIPCHK(c, hipMemcpyAsync(obs, c->d_obs, N * od * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (reward) HIPCHK(c, hipMemcpyAsync(reward, c->d_rew, N * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (cost) HIPCHK(c, hipMemcpyAsync(cost, c->d_cost, N, hipMemcpyDeviceToHost, c->stream));
  if (done) HIPCHK(c, hipMemcpyAsync(done, c->d_done, N, hipMemcpyDeviceToHost, c->stream));
  if (goal_met) HIPCHK(c, hipMemcpyAsync(goal_met, c->d_met, N, hipMemcpyDeviceToHost, c->stream));
  if (tape_used) HIPCHK(c, hipMemcpyAsync(tape_used, c->d_used, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}

int sag_set_ext_contacts(sag_ctx* c, const int32_t* cost_contacts, const uint32_t* btn_mask) {
  if (!c) return SAG_ERR_ARG;
  if (!cost_contacts && !btn_mask) { c->ext_pending = false; return SAG_OK; }
  if (!cost_contacts || !btn_mask) return fail(c, SAG_ERR_ARG, "sag_set_ext_contacts: both arrays or neither");
  const size_t N = (size_t)c->N;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  if (!c->d_ext_cc) {
    HIPCHK(c, hipMalloc(&c->d_ext_cc, N * sizeof(int32_t)));
    HIPCHK(c, hipMalloc(&c->d_ext_btn, N * sizeof(uint32_t)));
  }
  HIPCHK(c, hipMemcpyAsync(c->d_ext_cc, cost_contacts, N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_ext_btn, btn_mask, N * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the host arrays are the caller's: do not keep reading them
  c->ext_pending = true;
  return SAG_OK;
}

int sag_observe(sag_ctx* c, float* obs) {
  if (!c || !obs) return SAG_ERR_ARG;
  if (!c->have_layout) return fail(c, SAG_ERR_STATE, "sag_observe before sag_set_layout");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  int rc = launch_step(c, nullptr, nullptr, nullptr, 0, 0, c->d_obs, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(obs, c->d_obs, (size_t)c->N * c->rb.obs_dim * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}

int sag_lidar_cost(sag_ctx* c, int32_t n, int32_t K, const float* robot, const float* points,
                   const uint8_t* group, float hazard_size, float* lidar, int32_t* bins, uint8_t* cost) {
  if (!c) return SAG_ERR_ARG;
  if (n <= 0 || K < 0 || !robot || (K > 0 && (!points || !group)) || !lidar || !cost)
    return fail(c, SAG_ERR_ARG, "bad sag_lidar_cost arguments");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };   // every buffer 256-byte aligned (vector loads)
  const size_t b_robot = up((size_t)n * 3 * 4), b_pts = up((size_t)n * K * 2 * 4), b_grp = up((size_t)n * K),
               b_lid = up((size_t)n * 48 * 4), b_bins = up((size_t)n * K * 4), b_cost = up((size_t)n);
  int rc = ensure_scratch(c, b_robot + b_pts + b_grp + b_lid + b_bins + b_cost + 256);
  if (rc) return rc;
  char* base = (char*)c->scratch;
  float* d_robot = (float*)base; base += b_robot;
  float* d_pts = (float*)base; base += b_pts;
  float* d_lid = (float*)base; base += b_lid;
  int32_t* d_bins = (int32_t*)base; base += b_bins;
  uint8_t* d_grp = (uint8_t*)base; base += b_grp;
  uint8_t* d_cost = (uint8_t*)base;
  if (lidar_cost_lds_bytes(K) > 64 * 1024) return fail(c, SAG_ERR_ARG, "K = %d points per env exceed the kernel's LDS staging (K <= 100)", K);
  HIPCHK(c, hipMemcpyAsync(d_robot, robot, (size_t)n * 3 * 4, hipMemcpyHostToDevice, c->stream));
  if (K > 0) {
    HIPCHK(c, hipMemcpyAsync(d_pts, points, (size_t)n * K * 2 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_grp, group, (size_t)n * K, hipMemcpyHostToDevice, c->stream));
  }
  launch_lidar_cost(c, n, K, d_robot, d_pts, d_grp, hazard_size, d_lid, bins ? d_bins : nullptr, d_cost);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(lidar, d_lid, (size_t)n * 48 * 4, hipMemcpyDeviceToHost, c->stream));
  if (bins && K > 0) HIPCHK(c, hipMemcpyAsync(bins, d_bins, (size_t)n * K * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cost, d_cost, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}

int sag_lidar_cost_device(sag_ctx* c, int32_t n, int32_t K, const float* d_robot, const float* d_points,
                          const uint8_t* d_group, float hazard_size, float* d_lidar, int32_t* d_bins, uint8_t* d_cost) {
  if (!c) return SAG_ERR_ARG;
  if (n <= 0 || K < 0 || !d_robot || (K > 0 && (!d_points || !d_group)) || !d_lidar || !d_cost)
    return fail(c, SAG_ERR_ARG, "bad sag_lidar_cost_device arguments");
  // (every argument is checked BEFORE a timing-event pair is reserved: a pair that is counted but never recorded would
  //  add an earlier launch's timestamps to the kernel-time mean)
  if (lidar_cost_lds_bytes(K) > 64 * 1024) return fail(c, SAG_ERR_ARG, "K = %d points per env exceed the kernel's LDS staging (K <= 100)", K);
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = timing_events(c, &e0, &e1);
  if (rc) return rc;
  if (e0) HIPCHK(c, hipEventRecord(e0, c->stream));
  launch_lidar_cost(c, n, K, d_robot, d_points, d_group, hazard_size, d_lidar, d_bins, d_cost);
  if (e1) HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipGetLastError());
  return SAG_OK;
}

int sag_dev_alloc(sag_ctx* c, uint64_t bytes, void** dptr) {
  if (!c || !dptr) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipMalloc(dptr, bytes));
  return SAG_OK;
}
int sag_dev_free(sag_ctx* c, void* dptr) {
  if (!c) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(dptr));
  return SAG_OK;
}
int sag_dev_upload(sag_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}
int sag_dev_download(sag_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}
int sag_dev_fill_actions(sag_ctx* c, float* d_actions, uint32_t step_index) {
  if (!c || !d_actions) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(k_fill_actions, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, d_actions, c->I, c->N,
                     c->rb.nu, (uint32_t)(c->cfg.seed & 0xffffffffu), (uint32_t)(c->cfg.seed >> 32), step_index);
  HIPCHK(c, hipGetLastError());
  return SAG_OK;
}

int sag_enable_timing(sag_ctx* c, int32_t on) {
  if (!c) return SAG_ERR_ARG;
  c->timing = on != 0;
  return SAG_OK;
}

int sag_kernel_time_ms(sag_ctx* c, int32_t reset, double* mean_ms, int64_t* launches) {
  if (!c) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  if (mean_ms) *mean_ms = c->ev_n ? c->ev_ms / (double)c->ev_n : 0.0;
  if (launches) *launches = c->ev_n;
  if (reset) { c->ev_ms = 0; c->ev_n = 0; }
  return SAG_OK;
}

int sag_debug_cycles(sag_ctx* c, int32_t reset, uint64_t* out, int32_t n) {
  if (!c || !out || n < 0) return SAG_ERR_ARG;
#ifdef SAG_WAVE_TIMES   // (start, end) per busy wavefront of the last launch; reset: cleared
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
  {
    // out: [WT_MAX][2] times, then [WT_MAX][32] env ids (two per word), then [WT_MAX][64] work bytes (eight per word)
    std::vector<uint64_t> host((size_t)(2 + 32 + 64 + 2) * WT_MAX);   // ... then [WT_MAX][8] wavefront passes (four per word)
    HIPCHK(c, hipMemcpyFromSymbol(host.data() + (size_t)98 * WT_MAX, HIP_SYMBOL(g_wt_trips), (size_t)2 * WT_MAX * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(g_wt), (size_t)2 * WT_MAX * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpyFromSymbol(host.data() + (size_t)2 * WT_MAX, HIP_SYMBOL(g_wt_env), (size_t)32 * WT_MAX * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpyFromSymbol(host.data() + (size_t)34 * WT_MAX, HIP_SYMBOL(g_wt_work), (size_t)64 * WT_MAX * sizeof(uint64_t)));
    for (int k = 0; k < n; k++) out[k] = (size_t)k < host.size() ? host[k] : 0;
    if (reset) {
      std::fill(host.begin(), host.end(), 0);
      HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_wt), host.data(), (size_t)2 * WT_MAX * sizeof(uint64_t)));
    }
  }
  return SAG_OK;
#elif defined(SAG_CYCLES)
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // out[0 .. 48): sums per kernel form and section (+ wavefront count); out[48 .. 96): the sections of the slowest wavefront;
  // out[96 .. 99): its block index; out[99 .. 291): wavefronts by total ticks, 3 x 64 buckets of a quarter octave from 2^10
  constexpr int M = 3 * (CY_N + 1);
  uint64_t host[2 * M + 3 + 192];
  HIPCHK(c, hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cyc), M * sizeof(uint64_t)));
  HIPCHK(c, hipMemcpyFromSymbol(host + M, HIP_SYMBOL(g_cyc_worst), M * sizeof(uint64_t)));
  HIPCHK(c, hipMemcpyFromSymbol(host + 2 * M, HIP_SYMBOL(g_cyc_worst_block), 3 * sizeof(uint64_t)));
  HIPCHK(c, hipMemcpyFromSymbol(host + 2 * M + 3, HIP_SYMBOL(g_cyc_hist), 192 * sizeof(uint64_t)));
  for (int k = 0; k < n; k++) out[k] = k < 2 * M + 3 + 192 ? host[k] : 0;
  if (reset) {
    memset(host, 0, sizeof(host));
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), host, M * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_cyc_worst), host, M * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_cyc_hist), host, 192 * sizeof(uint64_t)));
  }
  return SAG_OK;
#else
  (void)reset;
  for (int k = 0; k < n; k++) out[k] = 0;
  return SAG_ERR_UNSUPPORTED;
#endif
}

// diagnostic (tests): mass matrix, bias, contact-free qacc and M^-1 of every Doggo env from the
// wave-cooperative routines; out[n_envs][2*361 + 38] doubles (host)
int sag_debug_doggo_coop(sag_ctx* c, double* out) {
  if (!c || !out) return SAG_ERR_ARG;
  if (c->cfg.robot != SAG_ROBOT_DOGGO) return fail(c, SAG_ERR_ARG, "not a doggo context");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  const size_t bytes = (size_t)c->N * (2 * DG_NV * DG_NV + 2 * DG_NV) * sizeof(double);
  double* d = nullptr;
  HIPCHK(c, hipMalloc(&d, bytes));
  hipLaunchKernelGGL(k_doggo_coop_debug, dim3((c->N + DC_EPW - 1) / DC_EPW), dim3(32 * DC_EPW), 0, c->stream, c->S, c->N, d);
  hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  HIPCHK(c, e);
  return SAG_OK;
}

// images of every env at its current state, ray-cast on the device (sag_render.hpp): [n_envs][height][width][3] uint8.
// d_obs / d_cost: the observation and cost flags the overlays show (device pointers or NULL)
int sag_render_device(sag_ctx* c, int32_t camera, int32_t width, int32_t height, int32_t flags, const float* d_obs,
                      const uint8_t* d_cost, void* d_out) {
  if (!c || !d_out) return c ? fail(c, SAG_ERR_ARG, "null argument") : SAG_ERR_ARG;
  if (camera < 0 || camera > SAG_CAM_TRACK || width <= 0 || height <= 0 || width > 4096 || height > 4096)
    return fail(c, SAG_ERR_ARG, "bad camera %d or image size %d x %d", camera, width, height);
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(k_render_rgb, dim3(c->N), dim3(256), 0, c->stream, c->S, c->I, c->N, c->cfg.robot, camera, width, height,
                     flags, d_obs, c->rb.obs_dim, d_cost, (uint8_t*)d_out);
  HIPCHK(c, hipGetLastError());
  return SAG_OK;
}
int sag_render(sag_ctx* c, int32_t camera, int32_t width, int32_t height, int32_t flags, uint8_t* out) {
  if (!c || !out) return c ? fail(c, SAG_ERR_ARG, "null argument") : SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  const size_t bytes = (size_t)c->N * (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0) * 3;
  if (c->rgb_bytes < bytes) {
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    c->d_rgb = nullptr; c->rgb_bytes = 0;
    HIPCHK(c, hipMalloc(&c->d_rgb, bytes ? bytes : 1));
    c->rgb_bytes = bytes;
  }
  // the overlays show what the last host-buffer step / observe left in the context's output buffers
  int rc = sag_render_device(c, camera, width, height, flags, c->d_obs, c->d_cost, c->d_rgb);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(out, c->d_rgb, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SAG_OK;
}
// rgb_observation: the 64 x 64 image of the robot's own camera, no overlays
int sag_render_rgb_device(sag_ctx* c, void* d_out) { return sag_render_device(c, SAG_CAM_VISION, R_W, R_H, 0, nullptr, nullptr, d_out); }
int sag_render_rgb(sag_ctx* c, uint8_t* out) { return sag_render(c, SAG_CAM_VISION, R_W, R_H, 0, out); }

int sag_busy_count(sag_ctx* c, int32_t* count) {
  if (!c || !count) return SAG_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!c->last_count) { *count = 0; return SAG_OK; }
  int32_t per_kind[BUSY_CLASSES];   // the lists the last step consumed
  HIPCHK(c, hipMemcpy(per_kind, c->last_count, sizeof(per_kind), hipMemcpyDeviceToHost));
  *count = 0;
  for (int k = 0; k < BUSY_CLASSES; k++) *count += per_kind[k];
  return SAG_OK;
}

}  // extern "C"
