// sag_device.hpp - gfx950 device code of the batched SafeAdaptationGym.step().
//
// One environment per lane, 64 environments per wavefront, one wavefront per
// workgroup.  World state lives in HBM as a structure of arrays S[field][env]
// (fields = the sag.h record, transposed) so every load/store below is a
// coalesced dword-per-lane access.  A step touches HBM once in and once out:
// the robot, the task scalars and the static bodies stay in registers and the
// free bodies (vases) in LDS, laid out [component][lane] (bank-conflict free),
// across all `nstep` physics substeps.
//
// Precision: dynamics in fp32; everything whose result is a discrete decision
// the reference takes in fp64 (lidar bin index, hazard test, goal-met test,
// goal resampling) is evaluated in fp64 from the fp32 state, with the
// reference's operation order (safe_adaptation_gym.py:174-223, world.py:144-155,
// tasks/go_to_goal.py:31-80).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sag.h"

namespace sag {

constexpr int WAVE = 64;
constexpr int VCOMP = 9;  // x y yaw vx vy w ax ay aw per vase in LDS

// device int fields
enum { DI_META = 0, DI_TSTATE = 1, DI_STEP = 2, DI_ENVID = 3, DI_FLAGS = 4, DI_COUNT = 5 };

// ---- packing of the int record into two device words -----------------------
__host__ __device__ inline uint32_t pack_meta(const int32_t* ri) {
  return (uint32_t)ri[SAG_I_TASK] | (uint32_t)ri[SAG_I_NH] << 4 | (uint32_t)ri[SAG_I_NV] << 8 |
         (uint32_t)ri[SAG_I_NP] << 12 | (uint32_t)ri[SAG_I_NB] << 14 |
         (uint32_t)ri[SAG_I_BOX_KIND] << 17;
}
__host__ __device__ inline void unpack_meta(uint32_t m, int32_t* ri) {
  ri[SAG_I_TASK] = m & 15; ri[SAG_I_NH] = m >> 4 & 15; ri[SAG_I_NV] = m >> 8 & 15;
  ri[SAG_I_NP] = m >> 12 & 3; ri[SAG_I_NB] = m >> 14 & 7; ri[SAG_I_BOX_KIND] = m >> 17 & 3;
}
__host__ __device__ inline uint32_t pack_tstate(const int32_t* ri) {
  return (uint32_t)ri[SAG_I_GOAL_BUTTON] | (uint32_t)ri[SAG_I_BTN_STATE] << 3 |
         (uint32_t)ri[SAG_I_BTN_TIMER] << 4 | (uint32_t)ri[SAG_I_CATCH_TIMER] << 7 |
         (uint32_t)ri[SAG_I_ACTIVE_MASK] << 11;
}
__host__ __device__ inline void unpack_tstate(uint32_t t, int32_t* ri) {
  ri[SAG_I_GOAL_BUTTON] = t & 7; ri[SAG_I_BTN_STATE] = t >> 3 & 1; ri[SAG_I_BTN_TIMER] = t >> 4 & 7;
  ri[SAG_I_CATCH_TIMER] = t >> 7 & 15; ri[SAG_I_ACTIVE_MASK] = t >> 11 & 63;
}

struct StepArgs {
  float* S;          // [SAG_REC_FLOATS][N]
  int32_t* I;        // [DI_COUNT][N]
  int32_t N;
  const float* actions;   // [N][nu]
  const float* noise;     // [N][nu] or nullptr
  const uint32_t* tape;   // [N][tape_len] or nullptr
  int32_t tape_len;
  int32_t nstep;          // substeps (0 = none)
  int32_t nstep_table;    // robot's table value (time base)
  float h;
  uint32_t key0, key1;
  float* obs;             // [N][60] or nullptr
  float* reward;          // [N][2]
  uint8_t* cost;
  uint8_t* done;
  uint8_t* goal_met;
  int32_t* tape_used;
  int32_t max_vases, max_hazards, max_pillars, max_buttons;  // context capacities (load bounds)
  int32_t observe_only;   // 1: sag_observe (no noise/physics/reward/cost)
};

// ---- constants of the Point robot (assets/xmls/point.xml) -------------------
constexpr float PT_M_SPHERE = 4.0f / 3.0f * 3.14159265358979323846f * 0.001f;
constexpr float PT_M_BOX = 0.001f;
constexpr float PT_MASS = PT_M_SPHERE + PT_M_BOX;
constexpr float PT_MC = PT_M_BOX * 0.1f;
constexpr float PT_IO = 0.4f * PT_M_SPHERE * 0.01f + PT_M_BOX * 0.02f / 12.0f + PT_M_BOX * 0.01f;
constexpr float PT_DAMP_Z = 0.005f, PT_FLIM = 0.05f, PT_GEAR_Z = 0.3f;
constexpr double PT_Z = 0.1, GOAL_Z = 0.3 / 2.0 + 1e-2, GOAL_SIZE = 0.3, GOAL_KEEPOUT = 0.4;
constexpr double PI_D = 3.14159265358979323846;
constexpr float SOL_D0 = 0.9f, SOL_D1 = 0.95f, SOL_WIDTH = 0.001f, MU = 1.0f, GRAV = 9.81f;
constexpr float BUTTON_R = 0.1f, VASE_DENSITY = 0.001f;
constexpr float REST_V = 1e-5f, REST_W = 1e-4f;
constexpr float ROBOT_BOUND = 0.15811388300841897f;  // arrow corner (0.15, 0.05)

// ---- counter-based generator ------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct Rng {
  const uint32_t* tape;  // this env's row or nullptr
  int len, pos, exhausted;
  uint32_t k0, k1, env, step;
  __device__ uint32_t word() {
    if (tape) {
      if (pos >= len) { exhausted = 1; pos++; return 0u; }
      return tape[pos++];
    }
    uint32_t c[4] = {env, step, (uint32_t)(pos >> 2), 0u};
    philox4x32_10(c, k0, k1);
    uint32_t w = c[pos & 3];
    pos++;
    return w;
  }
  // numpy legacy random_sample(): 53 bits from two words
  __device__ double sample() {
    uint32_t a = word() >> 5, b = word() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  __device__ double uniform(double lo, double hi) { return lo + (hi - lo) * sample(); }
};

// ---- lidar (fp64, the reference's arithmetic) -------------------------------
__device__ inline int lidar_accum(double ex, double ey, double* obs) {
  const double two_pi = PI_D * 2;
  double dist = hypot(ex, ey);
  double ang = atan2(ey, ex);
  if (ang < 0) ang += two_pi;  // python float %: fmod leaves |ang| < 2pi untouched, then shifts
  const double bin_size = two_pi / SAG_LIDAR_BINS;
  int bin = (int)(ang / bin_size);
  if (bin >= SAG_LIDAR_BINS) bin -= SAG_LIDAR_BINS;
  double bin_angle = bin_size * bin;
  double sensor = (5.0 - dist > 0 ? 5.0 - dist : 0.0) / 5.0;
  double alias = (ang - bin_angle) / bin_size;
  int bp = (bin + 1) & 15, bm = (bin + 15) & 15;
  // obs is indexed dynamically: callers keep it in LDS or accept scratch
  obs[bin] = fmax(obs[bin], sensor);
  obs[bp] = fmax(obs[bp], alias * sensor);
  obs[bm] = fmax(obs[bm], (1 - alias) * sensor);
  return bin;
}

// ---- planar bodies ----------------------------------------------------------
struct BV {  // body view used by the contact solver
  float x, y, vx, vy, w, ax, ay, aw;
  float m0, m1, m2, m3, m4, m5;  // symmetric inverse inertia xx xy xw yy yw ww
  int dyn;
};

__device__ inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ inline float minv_apply(const BV& b, float dx, float dy, float rxd, float u[3]) {
  u[0] = b.m0 * dx + b.m1 * dy + b.m2 * rxd;
  u[1] = b.m1 * dx + b.m3 * dy + b.m4 * rxd;
  u[2] = b.m2 * dx + b.m4 * dy + b.m5 * rxd;
  return dx * u[0] + dy * u[1] + rxd * u[2];
}

__device__ inline void rel_at(const BV& A, const BV& B, float rax, float ray, float rbx, float rby,
                              float& vx, float& vy, float& ax, float& ay) {
  vx = (B.vx - B.w * rby) - (A.vx - A.w * ray);
  vy = (B.vy + B.w * rbx) - (A.vy + A.w * rax);
  ax = (B.ax - B.aw * rby) - (A.ax - A.aw * ray);
  ay = (B.ay + B.aw * rbx) - (A.ay + A.aw * rax);
}

__device__ inline float impedance(float depth) {
  float x = fminf(depth * (1.0f / SOL_WIDTH), 1.0f);
  float y = x < 0.5f ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);
  return SOL_D0 + (SOL_D1 - SOL_D0) * y;
}

// soft contact, normal then friction; n points from A to B
__device__ inline void solve_contact(BV& A, BV& B, float nx, float ny, float px, float py,
                                     float depth, float bcoef, float kcoef) {
  float rax = px - A.x, ray = py - A.y, rbx = px - B.x, rby = py - B.y;
  float vx, vy, ax, ay, ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
  rel_at(A, B, rax, ray, rbx, rby, vx, vy, ax, ay);
  float vn = vx * nx + vy * ny, an = ax * nx + ay * ny, An = 0;
  if (A.dyn) An += minv_apply(A, nx, ny, rax * ny - ray * nx, ua);
  if (B.dyn) An += minv_apply(B, nx, ny, rbx * ny - rby * nx, ub);
  if (!(An > 0)) return;
  float d = impedance(depth);
  float fn = d * ((-bcoef * vn + kcoef * depth) - an) / An;
  if (!(fn > 0)) return;
  if (A.dyn) { A.ax -= ua[0] * fn; A.ay -= ua[1] * fn; A.aw -= ua[2] * fn; }
  if (B.dyn) { B.ax += ub[0] * fn; B.ay += ub[1] * fn; B.aw += ub[2] * fn; }
  float tx = -ny, ty = nx;
  rel_at(A, B, rax, ray, rbx, rby, vx, vy, ax, ay);
  float vt = vx * tx + vy * ty, at = ax * tx + ay * ty, At = 0;
  if (A.dyn) At += minv_apply(A, tx, ty, rax * ty - ray * tx, ua);
  if (B.dyn) At += minv_apply(B, tx, ty, rbx * ty - rby * tx, ub);
  if (!(At > 0)) return;
  float ft = clampf(d * (-bcoef * vt - at) / At, -MU * fn, MU * fn);
  if (A.dyn) { A.ax -= ua[0] * ft; A.ay -= ua[1] * ft; A.aw -= ua[2] * ft; }
  if (B.dyn) { B.ax += ub[0] * ft; B.ay += ub[1] * ft; B.aw += ub[2] * ft; }
}

// circle (A, centre ax,ay radius ra) vs circle (B)
__device__ inline int cc_contact(BV& A, BV& B, float ax, float ay, float ra, float bx, float by,
                                 float rb, float bcoef, float kcoef) {
  float dx = bx - ax, dy = by - ay, d2 = dx * dx + dy * dy, rs = ra + rb;
  if (d2 >= rs * rs) return 0;
  float d = sqrtf(d2), nx = 1, ny = 0;
  if (d > 1e-12f) { nx = dx / d; ny = dy / d; }
  float depth = rs - d;
  solve_contact(A, B, nx, ny, ax + nx * (ra - 0.5f * depth), ay + ny * (ra - 0.5f * depth), depth,
                bcoef, kcoef);
  return 1;
}

// circle vs oriented box; `circle_is_A` selects the normal direction (A -> B)
__device__ inline int cb_contact(BV& Circ, BV& Box, float cx, float cy, float r, float bx, float by,
                                 float cb, float sb, float hx, float hy, bool circle_is_A,
                                 float bcoef, float kcoef) {
  float wx = cx - bx, wy = cy - by;
  float lx = cb * wx + sb * wy, ly = -sb * wx + cb * wy;
  float qx = clampf(lx, -hx, hx), qy = clampf(ly, -hy, hy);
  float ddx = lx - qx, ddy = ly - qy, onx, ony, depth;
  if (ddx == 0 && ddy == 0) {
    float px = hx - fabsf(lx), py = hy - fabsf(ly);
    if (px < py) { onx = lx >= 0 ? 1.f : -1.f; ony = 0; depth = r + px; qx = onx * hx; }
    else { onx = 0; ony = ly >= 0 ? 1.f : -1.f; depth = r + py; qy = ony * hy; }
  } else {
    float d2 = ddx * ddx + ddy * ddy;
    if (d2 >= r * r) return 0;
    float d = sqrtf(d2);
    onx = ddx / d; ony = ddy / d; depth = r - d;
  }
  // outward normal of the box, world frame; points from box to circle
  float wnx = cb * onx - sb * ony, wny = sb * onx + cb * ony;
  float px = bx + cb * qx - sb * qy, py = by + sb * qx + cb * qy;
  if (circle_is_A) solve_contact(Circ, Box, -wnx, -wny, px, py, depth, bcoef, kcoef);
  else solve_contact(Box, Circ, wnx, wny, px, py, depth, bcoef, kcoef);
  return 1;
}

// vertices of box P strictly inside box Q. If q_is_A the normal A->B is Q's outward
// normal, else its negative.  A/B are passed in solver order.
__device__ inline int verts_in_box(BV& A, BV& B, float pxc, float pyc, float cp, float sp, float phx,
                                   float phy, float qxc, float qyc, float cq, float sq, float qhx,
                                   float qhy, bool q_is_A, float bcoef, float kcoef) {
  int n = 0;
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    float sx = (k == 0 || k == 3) ? phx : -phx, sy = (k < 2) ? phy : -phy;
    float vx = pxc + cp * sx - sp * sy, vy = pyc + sp * sx + cp * sy;
    float wx = vx - qxc, wy = vy - qyc;
    float lx = cq * wx + sq * wy, ly = -sq * wx + cq * wy;
    float dx = qhx - fabsf(lx), dy = qhy - fabsf(ly);
    if (dx <= 0 || dy <= 0) continue;
    float onx, ony, depth;
    if (dx < dy) { onx = lx >= 0 ? 1.f : -1.f; ony = 0; depth = dx; }
    else { onx = 0; ony = ly >= 0 ? 1.f : -1.f; depth = dy; }
    float sgn = q_is_A ? 1.f : -1.f;
    solve_contact(A, B, sgn * (cq * onx - sq * ony), sgn * (sq * onx + cq * ony), vx, vy, depth,
                  bcoef, kcoef);
    n++;
  }
  return n;
}

__device__ inline int bb_contact(BV& A, BV& B, float ax, float ay, float ca, float sa, float ahx,
                                 float ahy, float bx, float by, float cb, float sb, float bhx,
                                 float bhy, float bcoef, float kcoef) {
  int n = verts_in_box(A, B, ax, ay, ca, sa, ahx, ahy, bx, by, cb, sb, bhx, bhy, false, bcoef, kcoef);
  n += verts_in_box(A, B, bx, by, cb, sb, bhx, bhy, ax, ay, ca, sa, ahx, ahy, true, bcoef, kcoef);
  return n;
}

__device__ inline double dist2d(double ax, double ay, double bx, double by) {
  double dx = ax - bx, dy = ay - by;
  return sqrt(dx * dx + dy * dy);
}

// ---------------------------------------------------------------------------
// the fused step kernel (Point robot)
// ---------------------------------------------------------------------------
// LDS map (floats), one wavefront per workgroup, [slot][lane]:
//   slots  0.. 9 vase x      10..19 vase y     20..29 vase yaw
//         30..39 vase vx     40..49 vase vy    50..59 vase w
//         60..69 vase ax     70..79 vase ay    80..89 vase aw
// After the physics the region from slot 30 on is reused as the observation staging
// area [lane][61] (48 lidar bins + 12 sensors, row stride 61 = conflict-free both for
// the per-lane writes and for the transposed, fully coalesced read-out).
#ifdef SAG_LDS_ALIAS  // timing probe only (wrong results): fold the 90 slots onto 45
constexpr int LDS_FLOATS = 3968;
constexpr int STG_BASE = 0, STG_STRIDE = 61;
#define SLOT_MOD(x) ((x) % 45)
#else
constexpr int LDS_FLOATS = 6016;  // 90 slots * 64 + tail of the staging area
constexpr int STG_BASE = 30 * WAVE, STG_STRIDE = 61;
#define SLOT_MOD(x) (x)
#endif
enum { VS_X = 0, VS_Y = 10, VS_YAW = 20, VS_VX = 30, VS_VY = 40, VS_W = 50, VS_AX = 60, VS_AY = 70, VS_AW = 80 };

// timing-only ablations (tools/ablate.py): -DSAG_ABLATE=<mask>; results are wrong by design
#ifndef SAG_ABLATE
#define SAG_ABLATE 0
#endif
enum { ABL_NO_LIDAR = 1, ABL_NO_OBS_STORE = 2, ABL_NO_VV = 4, ABL_NO_VS = 8, ABL_NO_RV = 16, ABL_NO_RS = 32,
       ABL_NSUB1 = 64, ABL_NO_ACTIVE = 128 };
#define ABL(f) ((SAG_ABLATE & (f)) != 0)
#define SF(k) S[(size_t)(k) * N + i]
#define LV(comp, k) lds[SLOT_MOD((comp) + (k)) * WAVE + lane]
#define STG(j) lds[STG_BASE + lane * STG_STRIDE + (j)]

// tstate bits 17..26: vase k has non-zero velocity (derived; lets the kernel skip the
// velocity loads and the write-back of sleeping vases).  meta bit 19: some bodies overlap
// at rest, so no vase may be treated as asleep (never the case for sampled layouts).
constexpr uint32_t TS_AWAKE_SHIFT = 17, TS_AWAKE_MASK = 0x3ffu << 17;
constexpr uint32_t META_FULL_PAIRS = 1u << 19;

__device__ inline void load_vase(const float* lds, int lane, int k, float inv_m, float inv_I, BV& V) {
  V.x = LV(VS_X, k); V.y = LV(VS_Y, k);
  V.vx = LV(VS_VX, k); V.vy = LV(VS_VY, k); V.w = LV(VS_W, k);
  V.ax = LV(VS_AX, k); V.ay = LV(VS_AY, k); V.aw = LV(VS_AW, k);
  V.m0 = inv_m; V.m1 = 0; V.m2 = 0; V.m3 = inv_m; V.m4 = 0; V.m5 = inv_I; V.dyn = 1;
}
__device__ inline void store_vase_acc(float* lds, int lane, int k, const BV& V) {
  LV(VS_AX, k) = V.ax; LV(VS_AY, k) = V.ay; LV(VS_AW, k) = V.aw;
}

// lidar: fp32 estimate of (bin, alias); the fp64 evaluation (the reference's arithmetic,
// safe_adaptation_gym.py:208-216) is redone whenever the estimate is within 2e-5 bins of a
// bin boundary, so the bin index is always the fp64 one.
struct LidarHit { int bin; float alias, sensor; };

__device__ __attribute__((noinline)) LidarHit lidar_exact(double rx, double ry, double cd, double sd,
                                                           float px, float py) {
  const double W0 = (double)px - rx, W1 = (double)py - ry;
  const double EX = W0 * cd + W1 * sd, EY = W0 * -sd + W1 * cd;
  const double two_pi = PI_D * 2, bin_size = two_pi / SAG_LIDAR_BINS;
  double a = atan2(EY, EX);
  if (a < 0) a += two_pi;
  LidarHit h;
  h.bin = (int)(a / bin_size);
  if (h.bin >= SAG_LIDAR_BINS) h.bin -= SAG_LIDAR_BINS;
  h.alias = (float)((a - bin_size * h.bin) / bin_size);
  const double D = hypot(EX, EY);
  h.sensor = (float)((5.0 - D > 0 ? 5.0 - D : 0.0) / 5.0);
  return h;
}

__device__ inline void lidar_point(float* lds, int lane, int group_off, double rx, double ry,
                                   double cd, double sd, float rxf, float ryf, float cf, float sf,
                                   float px, float py) {
  const float w0 = px - rxf, w1 = py - ryf;
  const float ex = w0 * cf + w1 * sf, ey = w1 * cf - w0 * sf;
  const float dist = sqrtf(ex * ex + ey * ey);
  float ang = atan2f(ey, ex);
  if (ang < 0) ang += 6.28318530717958647692f;
  float t = ang * (16.0f / 6.28318530717958647692f);
  int bin = (int)t;
  float alias = t - (float)bin;
  float sensor = fmaxf(5.0f - dist, 0.0f) * 0.2f;
  if (alias < 2e-5f || alias > 1.0f - 2e-5f || bin > 15) {
    const LidarHit h = lidar_exact(rx, ry, cd, sd, px, py);
    bin = h.bin; alias = h.alias; sensor = h.sensor;
  }
  const int bp = (bin + 1) & 15, bm = (bin + 15) & 15;
  float* o = &STG(group_off);
  o[bin] = fmaxf(o[bin], sensor);
  o[bp] = fmaxf(o[bp], alias * sensor);
  o[bm] = fmaxf(o[bm], (1.0f - alias) * sensor);
}

#ifndef SAG_STEP_MIN_WAVES
#define SAG_STEP_MIN_WAVES 1
#endif
__global__ __launch_bounds__(WAVE, SAG_STEP_MIN_WAVES) void k_step_point(StepArgs p) {
#ifndef SAG_LDS_PAD
#define SAG_LDS_PAD 0
#endif
  __shared__ float lds[LDS_FLOATS + SAG_LDS_PAD];  // PAD: occupancy probe (tools/ablate.py)
  const int lane = threadIdx.x;
  const int N = p.N;
  const int gi = blockIdx.x * WAVE + lane;
  const bool live = gi < N;
  const int i = live ? gi : N - 1;  // idle lanes shadow the last env; they never store
  float* __restrict__ S = p.S;
  int32_t* __restrict__ I = p.I;
  const int capV = p.max_vases, capH = p.max_hazards, capP = p.max_pillars, capB = p.max_buttons;

  // ---- issue every load up front (bounds are context capacities, not per-env counts, so
  //      nothing waits on the meta word) ------------------------------------------------------
  const uint32_t meta = (uint32_t)I[(size_t)DI_META * N + i];
  uint32_t tstate = (uint32_t)I[(size_t)DI_TSTATE * N + i];
  int step = I[(size_t)DI_STEP * N + i];
  const uint32_t env_id = (uint32_t)I[(size_t)DI_ENVID * N + i];
  BV R;
  R.x = SF(SAG_F_ROBOT); R.y = SF(SAG_F_ROBOT + 1);
  float yaw = SF(SAG_F_ROBOT + 2);
  R.vx = SF(SAG_F_ROBOT + 3); R.vy = SF(SAG_F_ROBOT + 4); R.w = SF(SAG_F_ROBOT + 5);
  R.ax = R.ay = R.aw = 0; R.dyn = 1;
  const float gear = SF(SAG_F_GEAR), damp = SF(SAG_F_DAMP);
  const float vsz = SF(SAG_F_VASE_SIZE), psz = SF(SAG_F_PILLAR_SIZE);
  float goalx = SF(SAG_F_GOAL), goaly = SF(SAG_F_GOAL + 1);
  float last0 = SF(SAG_F_LAST);
  float a0 = 0, a1 = 0, n0 = 0, n1 = 0;
  if (!p.observe_only) {
    const float2 a = reinterpret_cast<const float2*>(p.actions)[i];
    a0 = a.x; a1 = a.y;
    if (p.noise) { const float2 z = reinterpret_cast<const float2*>(p.noise)[i]; n0 = z.x; n1 = z.y; }
  }
  uint32_t awake = (tstate >> TS_AWAKE_SHIFT) & 0x3ffu;
  if (meta & META_FULL_PAIRS) awake = 0x3ffu;
  {
    float vpos[SAG_MAX_VASES * 3];
#pragma unroll
    for (int k = 0; k < SAG_MAX_VASES; k++) {
      if (k < capV) {
        vpos[3 * k] = SF(SAG_F_VASES + 6 * k); vpos[3 * k + 1] = SF(SAG_F_VASES + 6 * k + 1);
        vpos[3 * k + 2] = SF(SAG_F_VASES + 6 * k + 2);
      } else { vpos[3 * k] = vpos[3 * k + 1] = vpos[3 * k + 2] = 0; }
    }
#pragma unroll
    for (int k = 0; k < SAG_MAX_VASES; k++) {
      LV(VS_X, k) = vpos[3 * k]; LV(VS_Y, k) = vpos[3 * k + 1]; LV(VS_YAW, k) = vpos[3 * k + 2];
      float vx = 0, vy = 0, w = 0;
      if (k < capV && (awake >> k & 1)) {
        vx = SF(SAG_F_VASES + 6 * k + 3); vy = SF(SAG_F_VASES + 6 * k + 4); w = SF(SAG_F_VASES + 6 * k + 5);
      }
      LV(VS_VX, k) = vx; LV(VS_VY, k) = vy; LV(VS_W, k) = w;
    }
  }
  // static colliders: pillars then buttons, in registers
  float stx[SAG_MAX_PILLARS + SAG_MAX_BUTTONS], sty[SAG_MAX_PILLARS + SAG_MAX_BUTTONS];
#pragma unroll
  for (int k = 0; k < SAG_MAX_PILLARS; k++) {
    stx[k] = k < capP ? SF(SAG_F_PILLARS + 2 * k) : 0.f;
    sty[k] = k < capP ? SF(SAG_F_PILLARS + 2 * k + 1) : 0.f;
  }
#pragma unroll
  for (int k = 0; k < SAG_MAX_BUTTONS; k++) {
    stx[SAG_MAX_PILLARS + k] = k < capB ? SF(SAG_F_BUTTONS + 2 * k) : 0.f;
    sty[SAG_MAX_PILLARS + k] = k < capB ? SF(SAG_F_BUTTONS + 2 * k + 1) : 0.f;
  }
  float hzx[SAG_MAX_HAZARDS], hzy[SAG_MAX_HAZARDS];
#pragma unroll
  for (int k = 0; k < SAG_MAX_HAZARDS; k++) {
    hzx[k] = k < capH ? SF(SAG_F_HAZARDS + 2 * k) : 0.f;
    hzy[k] = k < capH ? SF(SAG_F_HAZARDS + 2 * k + 1) : 0.f;
  }
  const float hsz = SF(SAG_F_HAZARD_SIZE);

  const int task = meta & 15, nH = meta >> 4 & 15, nV = meta >> 8 & 15, nP = meta >> 12 & 3,
            nB = meta >> 14 & 7, box_kind = meta >> 17 & 3;
  int flags = 0;

  Rng rng;
  rng.tape = p.tape ? p.tape + (size_t)i * p.tape_len : nullptr;
  rng.len = p.tape_len; rng.pos = 0; rng.exhausted = 0;
  rng.k0 = p.key0; rng.k1 = p.key1; rng.env = env_id; rng.step = (uint32_t)step;

  // ---- action noise + clip (safe_adaptation_gym.py:58-67) ---------------------
  float ctrl0 = 0, ctrl1 = 0;
  if (!p.observe_only) {
    if (!p.noise) {
      uint32_t c[4] = {rng.env, rng.step, 0u, 1u};
      philox4x32_10(c, p.key0, p.key1);
      float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      float r = sqrtf(-2.0f * logf(u1)), a = 6.28318530717958647692f * u2;
      n0 = r * cosf(a); n1 = r * sinf(a);
    }
    const float an = SF(SAG_F_ACTION_NOISE);
    const float cs0 = SF(SAG_F_CTRL_SCALE), cs1 = SF(SAG_F_CTRL_SCALE + 1);
    ctrl0 = clampf(a0 + an * n0, -cs0, cs0);
    ctrl1 = clampf(a1 + an * n1, -cs1, cs1);
    // ---- CatchGoal.set_mocaps (tasks/catch_goal.py:20-31), time before the step ---
    if (task == SAG_TASK_CATCH_GOAL) {
      int t = tstate >> 7 & 15;
      t = t - 1 > 0 ? t - 1 : 0;
      float cur = SF(SAG_F_CATCH + 2), nxt = SF(SAG_F_CATCH + 3);
      if (t == 0) {
        cur = nxt;
        nxt = (float)rng.uniform(0.2, 1.0);
        if (rng.exhausted) flags |= 2;
        if (live) { SF(SAG_F_CATCH + 2) = cur; SF(SAG_F_CATCH + 3) = nxt; }
        t = 10;
      }
      tstate = (tstate & ~(15u << 7)) | (uint32_t)t << 7;
      double time = (double)step * p.nstep_table * (double)p.h;
      double progress = (10 - t) / 10.0;
      double radius = progress * ((double)nxt - (double)cur) + (double)cur;
      goalx = (float)((double)SF(SAG_F_CATCH) + sin(time) * radius);
      goaly = (float)((double)SF(SAG_F_CATCH + 1) + cos(time) * radius);
    }
  }

  // ---- physics: nstep x (forward, integrate) + one forward at the final state ---
  const float h = p.h;
  const float tc = fmaxf(0.02f, 2.0f * h);
  const float bcoef = 2.0f / (SOL_D1 * tc), kcoef = 1.0f / (SOL_D1 * SOL_D1 * tc * tc);
  const float vase_m = VASE_DENSITY * 8 * vsz * vsz * vsz;
  const float vase_I = vase_m * (8 * vsz * vsz) / 12;
  const float vase_r = vsz * 1.41421356237309504880f;
  const float inv_vm = 1.0f / vase_m, inv_vI = 1.0f / vase_I;
  const int nsub = p.observe_only ? 0 : (ABL(ABL_NSUB1) ? 1 : p.nstep);
  // statics occupy [0, capP) and [SAG_MAX_PILLARS, SAG_MAX_PILLARS + capB) of stx/sty
  const int n_static = capB ? SAG_MAX_PILLARS + capB : capP;
  const uint32_t vmask = (1u << nV) - 1;
  awake &= vmask;
  uint32_t dirty = 0;  // vases whose state changed during this step
  int cost_contacts = 0;
  uint32_t btn_mask = 0;
  float cy = 1, sy = 0;

#pragma unroll 1
  for (int sub = 0; sub <= nsub; sub++) {
    sincosf(yaw, &sy, &cy);
    // robot smooth dynamics (point.xml)
    {
      float f0 = gear * clampf(ctrl0, -PT_FLIM, PT_FLIM);
      float Fx = f0 * cy - damp * R.vx + PT_MC * cy * R.w * R.w;
      float Fy = f0 * sy - damp * R.vy + PT_MC * sy * R.w * R.w;
      float Tz0 = -PT_DAMP_Z * R.w;
      float m = PT_MASS + h * damp, Iz = PT_IO + h * PT_DAMP_Z;
      float a = -PT_MC * sy, b = PT_MC * cy;
      float id = 1.0f / (m * (m * Iz - a * a - b * b));
      R.m0 = (m * Iz - b * b) * id; R.m1 = (a * b) * id; R.m2 = (-a * m) * id;
      R.m3 = (m * Iz - a * a) * id; R.m4 = (-b * m) * id; R.m5 = (m * m) * id;
      // yaw servo evaluated at the end-of-substep rate (implicit, exact for the clipped-linear
      // law; see DESIGN.md "servo"): explicit feedback would have gain 12.7 per substep
      float A = R.w + h * (R.m2 * Fx + R.m4 * Fy + R.m5 * Tz0);
      float g = h * R.m5;
      float w_lin = (A + g * PT_GEAR_Z * ctrl1) / (1 + g * (PT_GEAR_Z * PT_GEAR_Z));
      float Tz = PT_GEAR_Z * clampf(ctrl1 - PT_GEAR_Z * w_lin, -PT_FLIM, PT_FLIM) + Tz0;
      R.ax = R.m0 * Fx + R.m1 * Fy + R.m2 * Tz;
      R.ay = R.m1 * Fx + R.m3 * Fy + R.m4 * Tz;
      R.aw = R.m2 * Fx + R.m4 * Fy + R.m5 * Tz;
    }
    const float arx = R.x + cy * 0.1f, ary = R.y + sy * 0.1f;  // arrow box centre
    cost_contacts = 0; btn_mask = 0;
    // `active` = vases that can have a non-zero acceleration or velocity this substep.
    // A sleeping vase (v = 0, untouched) overlaps nothing (invariant of sampled layouts, else
    // META_FULL_PAIRS), so every pair test it would take part in is a no-op and is skipped;
    // the pair ORDER of the specification is kept for the ones that run.
    uint32_t active = awake;
    for (uint32_t m = active; m; m &= m - 1) {
      const int k = __ffs(m) - 1;
      LV(VS_AX, k) = 0; LV(VS_AY, k) = 0; LV(VS_AW, k) = 0;
    }
    // robot vs static circles (pillars, then buttons)
#pragma unroll 1
    for (int k = 0; k < n_static; k++) {
      if (k == capP) k = SAG_MAX_PILLARS;  // jump over unused pillar slots to the buttons
      const bool is_p = k < SAG_MAX_PILLARS;
      const bool on = (is_p ? (k < nP) : (k - SAG_MAX_PILLARS < nB)) && !ABL(ABL_NO_RS);
      const float sx = stx[k], syy = sty[k], sr = is_p ? psz : BUTTON_R;
      const float dx = sx - R.x, dy = syy - R.y, rs = ROBOT_BOUND + sr;
      if (on && dx * dx + dy * dy <= rs * rs) {
        BV St; St.x = sx; St.y = syy; St.vx = St.vy = St.w = St.ax = St.ay = St.aw = 0;
        St.m0 = St.m1 = St.m2 = St.m3 = St.m4 = St.m5 = 0; St.dyn = 0;
        int n = cc_contact(R, St, R.x, R.y, 0.1f, sx, syy, sr, bcoef, kcoef);
        n += cb_contact(St, R, sx, syy, sr, arx, ary, cy, sy, 0.05f, 0.05f, false, bcoef, kcoef);
        if (is_p) cost_contacts += n;
        else if (n) btn_mask |= 1u << (k - SAG_MAX_PILLARS);
      }
    }
    // robot vs vases: cheap broadphase for all, then each lane walks ITS OWN hit list (ascending
    // index = the specification's order), so a wavefront runs the narrowphase max-hits times,
    // not once per vase index that any lane happens to touch
    uint32_t hits = 0;
#pragma unroll 1
    for (int k = 0; k < nV; k++) {
      const float dx = LV(VS_X, k) - R.x, dy = LV(VS_Y, k) - R.y, rs = ROBOT_BOUND + vase_r;
      if (dx * dx + dy * dy <= rs * rs) hits |= 1u << k;
    }
    if (ABL(ABL_NO_RV)) hits = 0;
    for (uint32_t m = hits; m; m &= m - 1) {
      const int k = __ffs(m) - 1;
      const float vx_ = LV(VS_X, k), vy_ = LV(VS_Y, k);
      if (!(active >> k & 1)) { LV(VS_AX, k) = 0; LV(VS_AY, k) = 0; LV(VS_AW, k) = 0; }
      BV V; load_vase(lds, lane, k, inv_vm, inv_vI, V);
      float cv, sv; sincosf(LV(VS_YAW, k), &sv, &cv);
      // per-geom bounding circles first: sphere (r .1) and arrow box (half diagonal .0707)
      int n = 0;
      {
        const float dx = vx_ - R.x, dy = vy_ - R.y, rs = 0.1f + vase_r;
        if (dx * dx + dy * dy <= rs * rs)
          n = cb_contact(R, V, R.x, R.y, 0.1f, vx_, vy_, cv, sv, vsz, vsz, true, bcoef, kcoef);
      }
      {
        const float dx = vx_ - arx, dy = vy_ - ary, rs = 0.0707106781186548f + vase_r;
        if (dx * dx + dy * dy <= rs * rs)
          n += bb_contact(R, V, arx, ary, cy, sy, 0.05f, 0.05f, vx_, vy_, cv, sv, vsz, vsz, bcoef, kcoef);
      }
      cost_contacts += n;
      if (n) { store_vase_acc(lds, lane, k, V); active |= 1u << k; }
    }
    if (sub == nsub) break;  // final forward: robot acceleration + contact flags only
    if (active && !ABL(ABL_NO_ACTIVE)) {
      // vases vs static circles (pillars then buttons), per active vase
      for (uint32_t m = active; m; m &= m - 1) {
        const int k = __ffs(m) - 1;
        const float vx_ = LV(VS_X, k), vy_ = LV(VS_Y, k);
        uint32_t shit = 0;
#pragma unroll 1
        for (int q = 0; q < n_static; q++) {
          if (q == capP) q = SAG_MAX_PILLARS;
          const bool is_p = q < SAG_MAX_PILLARS;
          const bool on = (is_p ? (q < nP) : (q - SAG_MAX_PILLARS < nB)) && !ABL(ABL_NO_VS);
          const float dx = stx[q] - vx_, dy = sty[q] - vy_, rs = vase_r + (is_p ? psz : BUTTON_R);
          if (on && dx * dx + dy * dy <= rs * rs) shit |= 1u << q;
        }
        for (uint32_t mq = shit; mq; mq &= mq - 1) {
          const int q = __ffs(mq) - 1;
          // per-lane q: select from the register arrays without dynamic indexing
          float sx = 0, syy = 0;
#pragma unroll
          for (int z = 0; z < SAG_MAX_PILLARS + SAG_MAX_BUTTONS; z++) if (z == q) { sx = stx[z]; syy = sty[z]; }
          const float sr = q < SAG_MAX_PILLARS ? psz : BUTTON_R;
          BV V; load_vase(lds, lane, k, inv_vm, inv_vI, V);
          BV St; St.x = sx; St.y = syy; St.vx = St.vy = St.w = St.ax = St.ay = St.aw = 0;
          St.m0 = St.m1 = St.m2 = St.m3 = St.m4 = St.m5 = 0; St.dyn = 0;
          float cv, sv; sincosf(LV(VS_YAW, k), &sv, &cv);
          cb_contact(St, V, sx, syy, sr, vx_, vy_, cv, sv, vsz, vsz, false, bcoef, kcoef);
          store_vase_acc(lds, lane, k, V);
        }
      }
      // vase vs vase, pairs (a < b) in lexicographic order, at least one of them active;
      // per a: broadphase mask over b, then the lane's own hit list.  (A per-lane walk over a
      // 45-bit pair list was measured slower: 64-bit ffs + index decode per pair.)
#pragma unroll 1
      for (int a = 0; a < nV - 1; a++) {
        const uint32_t above = vmask & ~((2u << a) - 1);          // b > a
        uint32_t cand = (active >> a & 1) ? above : (active & above);
        if (ABL(ABL_NO_VV)) cand = 0;
        if (!cand) continue;
        const float axp = LV(VS_X, a), ayp = LV(VS_Y, a);
        uint32_t hit = 0;
        for (uint32_t m = cand; m; m &= m - 1) {
          const int b = __ffs(m) - 1;
          const float dx = LV(VS_X, b) - axp, dy = LV(VS_Y, b) - ayp, rs = 2 * vase_r;
          if (dx * dx + dy * dy <= rs * rs) hit |= 1u << b;
        }
        for (uint32_t m = hit; m; m &= m - 1) {
          const int b = __ffs(m) - 1;
          const float bxp = LV(VS_X, b), byp = LV(VS_Y, b);
          if (!(active >> a & 1)) { LV(VS_AX, a) = 0; LV(VS_AY, a) = 0; LV(VS_AW, a) = 0; }
          if (!(active >> b & 1)) { LV(VS_AX, b) = 0; LV(VS_AY, b) = 0; LV(VS_AW, b) = 0; }
          BV A, B;
          load_vase(lds, lane, a, inv_vm, inv_vI, A);
          load_vase(lds, lane, b, inv_vm, inv_vI, B);
          float ca, sa, cb, sb;
          sincosf(LV(VS_YAW, a), &sa, &ca); sincosf(LV(VS_YAW, b), &sb, &cb);
          int n = bb_contact(A, B, axp, ayp, ca, sa, vsz, vsz, bxp, byp, cb, sb, vsz, vsz, bcoef, kcoef);
          if (n) {
            store_vase_acc(lds, lane, a, A); store_vase_acc(lds, lane, b, B);
            active |= 1u << a | 1u << b;
          }
        }
      }
      // floor friction + semi-implicit Euler for the active vases
      const float fmax_ = MU * GRAV * vase_m;
      for (uint32_t m = active; m; m &= m - 1) {
        const int k = __ffs(m) - 1;
        float vx_ = LV(VS_VX, k), vy_ = LV(VS_VY, k), w_ = LV(VS_W, k);
        float ax_ = LV(VS_AX, k), ay_ = LV(VS_AY, k), aw_ = LV(VS_AW, k);
        float fx = -SOL_D0 * vase_m * (bcoef * vx_ + ax_), fy = -SOL_D0 * vase_m * (bcoef * vy_ + ay_);
        float f2 = fx * fx + fy * fy;
        if (f2 > fmax_ * fmax_) { float sc = fmax_ / sqrtf(f2); fx *= sc; fy *= sc; }
        ax_ += fx / vase_m; ay_ += fy / vase_m;
        float t = clampf(-SOL_D0 * vase_I * (bcoef * w_ + aw_), -fmax_ * vase_r, fmax_ * vase_r);
        aw_ += t / vase_I;
        vx_ += h * ax_; vy_ += h * ay_; w_ += h * aw_;
        // static friction capture (specification, oracle integrate_free)
        if (fabsf(vx_) < REST_V && fabsf(vy_) < REST_V && fabsf(w_) < REST_W && fabsf(h * ax_) < REST_V &&
            fabsf(h * ay_) < REST_V && fabsf(h * aw_) < REST_W) { vx_ = 0; vy_ = 0; w_ = 0; }
        LV(VS_VX, k) = vx_; LV(VS_VY, k) = vy_; LV(VS_W, k) = w_;
        LV(VS_X, k) += h * vx_; LV(VS_Y, k) += h * vy_; LV(VS_YAW, k) += h * w_;
        if (vx_ == 0 && vy_ == 0 && w_ == 0) awake &= ~(1u << k); else awake |= 1u << k;
      }
      dirty |= active;
    }
    if (meta & META_FULL_PAIRS) awake = vmask;
    R.vx += h * R.ax; R.vy += h * R.ay; R.w += h * R.aw;
    R.x += h * R.vx; R.y += h * R.vy; yaw += h * R.w;
  }

  // ---- write back dynamic state -------------------------------------------------
  if (!p.observe_only && live) {
    SF(SAG_F_ROBOT) = R.x; SF(SAG_F_ROBOT + 1) = R.y; SF(SAG_F_ROBOT + 2) = yaw;
    SF(SAG_F_ROBOT + 3) = R.vx; SF(SAG_F_ROBOT + 4) = R.vy; SF(SAG_F_ROBOT + 5) = R.w;
    for (uint32_t m = dirty; m; m &= m - 1) {
      const int k = __ffs(m) - 1;
      SF(SAG_F_VASES + 6 * k) = LV(VS_X, k); SF(SAG_F_VASES + 6 * k + 1) = LV(VS_Y, k);
      SF(SAG_F_VASES + 6 * k + 2) = LV(VS_YAW, k); SF(SAG_F_VASES + 6 * k + 3) = LV(VS_VX, k);
      SF(SAG_F_VASES + 6 * k + 4) = LV(VS_VY, k); SF(SAG_F_VASES + 6 * k + 5) = LV(VS_W, k);
    }
    step += 1;
    I[(size_t)DI_STEP * N + i] = step;
  }
  tstate = (tstate & ~TS_AWAKE_MASK) | (awake & 0x3ffu) << TS_AWAKE_SHIFT;

  // ---- PhysicsError branch (safe_adaptation_gym.py:73-75) -------------------------
  bool bad = false;
  {
    float v[6] = {R.x, R.y, yaw, R.vx, R.vy, R.w};
#pragma unroll
    for (int k = 0; k < 6; k++) bad |= !(fabsf(v[k]) <= 1e10f);
  }

  // ---- reward (tasks/<task>.py compute_reward), fp64 from the fp32 state -----------
  double rew0 = 0, rew1 = 0;
  int met = 0;
  const double rx = R.x, ry = R.y;
  if (!p.observe_only && !bad) {
    if (task == SAG_TASK_PRESS_BUTTONS || task == SAG_TASK_PRESS_BUTTONS_SCARCE) {
      int gb = tstate & 7, bstate = tstate >> 3 & 1, timer = tstate >> 4 & 7;
      double gd = dist2d(rx, ry, SF(SAG_F_BUTTONS + 2 * gb), SF(SAG_F_BUTTONS + 2 * gb + 1));
      double r = task == SAG_TASK_PRESS_BUTTONS ? (double)last0 - gd : 0.0;
      float last = (float)gd;
      if (btn_mask >> gb & 1) {
        r += 1.0; met = 1;
        gb = (int)(rng.word() & 3u);
        if (rng.exhausted) flags |= 2;
        timer = 5;
        last = (float)dist2d(rx, ry, SF(SAG_F_BUTTONS + 2 * gb), SF(SAG_F_BUTTONS + 2 * gb + 1));
        bstate = 0;
      }
      if (bstate == 0) {
        if (timer != 0) timer -= 1;
        else { bstate = 1; timer = 5; }
      }
      last0 = last;
      tstate = (tstate & ~127u) | (uint32_t)gb | (uint32_t)bstate << 3 | (uint32_t)timer << 4;
      rew0 = r;
    } else if (task == SAG_TASK_COLLECT) {
      uint32_t act = tstate >> 11 & 63;
      if (act == 0) act = (1u << nB) - 1;
      uint32_t hit = act & btn_mask;
      if (hit) {
        rew0 = 1.0; met = 1;
        act &= ~(hit & (0u - hit));  // lowest touched active button
      }
      tstate = (tstate & ~(63u << 11)) | act << 11;
    } else if (box_kind == SAG_BOX_NONE) {
      // GoToGoal family (tasks/go_to_goal.py:31-45): 3-D distance incl. dz
      double gx = goalx, gy = goaly;
      double dx = rx - gx, dy = ry - gy, dz = PT_Z - GOAL_Z;
      double dist = sqrt(dx * dx + dy * dy + dz * dz);
      double r = (double)last0 - dist;
      if (task == SAG_TASK_GO_TO_GOAL_SCARCE) r *= (dist <= GOAL_SIZE * 1.5) ? 1.0 : 0.0;
      float last = (float)dist;
      if (dist <= GOAL_SIZE) {
        met = 1;
        // GoToGoal._resample_goal_position (:59-80) + utils.draw_placement (utils.py:28-70)
        double xmin = -1.5, ymin = -1.5, xmax = 1.5, ymax = 1.5;
        const double k_robot = SF(SAG_F_KEEPOUT), k_haz = SF(SAG_F_KEEPOUT + 1),
                     k_vase = SF(SAG_F_KEEPOUT + 2), k_pil = SF(SAG_F_KEEPOUT + 3);
        bool found = false;
#pragma unroll 1
        for (int t = 0; t < 10000 && !found; t++) {
          double ngx = rng.uniform(xmin + GOAL_KEEPOUT, xmax - GOAL_KEEPOUT);
          double ngy = rng.uniform(ymin + GOAL_KEEPOUT, ymax - GOAL_KEEPOUT);
          if (rng.exhausted) break;
          bool ok = !(dist2d(ngx, ngy, rx, ry) < k_robot + GOAL_KEEPOUT);
          for (int k = 0; k < nH && ok; k++)
            ok = !(dist2d(ngx, ngy, SF(SAG_F_HAZARDS + 2 * k), SF(SAG_F_HAZARDS + 2 * k + 1)) < k_haz + GOAL_KEEPOUT);
          for (int k = 0; k < nV && ok; k++)
            ok = !(dist2d(ngx, ngy, LV(VS_X, k), LV(VS_Y, k)) < k_vase + GOAL_KEEPOUT);
          for (int k = 0; k < nP && ok; k++)
            ok = !(dist2d(ngx, ngy, SF(SAG_F_PILLARS + 2 * k), SF(SAG_F_PILLARS + 2 * k + 1)) < k_pil + GOAL_KEEPOUT);
          if (ok) { goalx = (float)ngx; goaly = (float)ngy; found = true; }
          else { xmin *= 1.01; ymin *= 1.01; xmax *= 1.01; ymax *= 1.01; }
        }
        if (!found) flags |= rng.exhausted ? 2 : 1;
        last = (float)dist2d(rx, ry, goalx, goaly);  // GoToGoal.reset: 2-D (:54-55)
        if (task == SAG_TASK_CATCH_GOAL && live) { SF(SAG_F_CATCH) = goalx; SF(SAG_F_CATCH + 1) = goaly; }
        r += 1.0;
      }
      last0 = last;
      if (task == SAG_TASK_UNSUPERVISED) {  // tasks/unsupervised.py:48-67
        double c = cos((double)yaw), s = sin((double)yaw), off = (double)PT_MC / (double)PT_MASS;
        double x = rx + c * off, y = ry + s * off, w = R.w;
        double u = (double)R.vx - w * (s * off), v = (double)R.vy + w * (c * off);
        double radius = sqrt(x * x + y * y);
        rew0 = (((-u * y + v * x) / radius) / (1 + fabs(radius - 1.5))) * 1e-1;
        rew1 = r;
      } else rew0 = r;
    }
    if (live) {
      SF(SAG_F_LAST) = last0;
      SF(SAG_F_GOAL) = goalx; SF(SAG_F_GOAL + 1) = goaly;
    }
  }
  if (!p.observe_only && live) I[(size_t)DI_TSTATE * N + i] = (int32_t)tstate;

  // ---- cost (world.py:144-155): hazard test in fp32 unless within 1e-5 of the threshold,
  //      where the reference's fp64 expression decides -----------------------------------
  int cost = 0;
  if (!p.observe_only && !bad) {
    int c = cost_contacts;
#pragma unroll 1
    for (int k = 0; k < SAG_MAX_HAZARDS; k++) {
      if (k < nH) {
        const float dx = R.x - hzx[k], dy = R.y - hzy[k];
        const float d2 = dx * dx + dy * dy, t2 = hsz * hsz;
        bool in = d2 <= t2;
        if (fabsf(d2 - t2) < 1e-5f) in = dist2d(rx, ry, hzx[k], hzy[k]) <= (double)hsz;
        c += in;
      }
    }
    cost = c > 0;
  }
  if (flags && live) I[(size_t)DI_FLAGS * N + i] |= flags;

  // ---- observation (safe_adaptation_gym.py:120-139, 225-237) -> LDS staging -----------
  if (p.obs) {
#pragma unroll
    for (int k = 0; k < 48; k++) STG(k) = 0.0f;
    const double cd = cos((double)yaw), sd = sin((double)yaw);
    const float cf = (float)cd, sf = (float)sd;
#pragma unroll 1
    for (int k = 0; k < SAG_MAX_HAZARDS; k++)
      if (k < nH && !ABL(ABL_NO_LIDAR)) lidar_point(lds, lane, 0, rx, ry, cd, sd, R.x, R.y, cf, sf, hzx[k], hzy[k]);
#pragma unroll 1
    for (int k = 0; k < nV; k++)
      if (!ABL(ABL_NO_LIDAR)) lidar_point(lds, lane, 0, rx, ry, cd, sd, R.x, R.y, cf, sf, LV(VS_X, k), LV(VS_Y, k));
#pragma unroll 1
    for (int k = 0; k < SAG_MAX_PILLARS; k++)
      if (k < nP) lidar_point(lds, lane, 0, rx, ry, cd, sd, R.x, R.y, cf, sf, stx[k], sty[k]);
    if (nB) {
      const int gb = tstate & 7, bstate = tstate >> 3 & 1;
      const uint32_t act = tstate >> 11 & 63;
#pragma unroll 1
      for (int b = 0; b < SAG_MAX_BUTTONS; b++) {
        if (b < nB) {
          int g;
          if (task == SAG_TASK_COLLECT) g = (act >> b & 1) ? 2 : 0;
          else g = bstate == 0 ? 0 : (b == gb ? 2 : 3);
          if (g) lidar_point(lds, lane, g == 3 ? 16 : 32, rx, ry, cd, sd, R.x, R.y, cf, sf,
                             stx[SAG_MAX_PILLARS + b], sty[SAG_MAX_PILLARS + b]);
        }
      }
    } else {
      lidar_point(lds, lane, 32, rx, ry, cd, sd, R.x, R.y, cf, sf, goalx, goaly);
    }
    const float qax = bad ? 0.0f : R.ax, qay = bad ? 0.0f : R.ay;
    STG(48) = cf * qax + sf * qay;
    STG(49) = cf * qay - sf * qax;
    STG(50) = GRAV;
    STG(51) = cf * R.vx + sf * R.vy;
    STG(52) = cf * R.vy - sf * R.vx;
    STG(53) = 0; STG(54) = 0; STG(55) = 0; STG(56) = R.w;
    STG(57) = -0.5f * sf; STG(58) = -0.5f * cf; STG(59) = 0;
    __syncthreads();
    // transposed read-out: element e of this wavefront's contiguous [64][60] block
    const int base_env = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - base_env);
    float* __restrict__ o = p.obs + (size_t)base_env * 60;
#pragma unroll 4
    for (int j = 0; j < 60; j++) {
      const int e = j * WAVE + lane;
      const int env = (int)(((uint32_t)e * 34953u) >> 21);  // e / 60, exact for e < 3840
      const float v = lds[STG_BASE + e + env];  // env*61 + (e - env*60)
      if (env < nvalid && !ABL(ABL_NO_OBS_STORE)) o[e] = v;
    }
  }
  if (live) {
    if (p.reward) {
      float2 r;
      r.x = bad ? -10.0f : (float)rew0; r.y = bad ? 0.0f : (float)rew1;
      reinterpret_cast<float2*>(p.reward)[i] = r;
    }
    if (p.cost) p.cost[i] = (uint8_t)cost;
    if (p.done) p.done[i] = bad ? 1 : 0;
    if (p.goal_met) p.goal_met[i] = (uint8_t)met;
    if (p.tape_used) p.tape_used[i] = rng.pos;
  }
}

// ---------------------------------------------------------------------------
// record <-> SoA
// ---------------------------------------------------------------------------
// rec_f [n][SAG_REC_FLOATS], rec_i [n][SAG_REC_INTS] in device memory (AoS staging).
__global__ void k_install(float* S, int32_t* I, int N, const int32_t* env_ids, int n,
                          const float* rec_f, const int32_t* rec_i, int init_task) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int i = env_ids ? env_ids[j] : j;
  const float* rf = rec_f + (size_t)j * SAG_REC_FLOATS;
  const int32_t* ri = rec_i + (size_t)j * SAG_REC_INTS;
  for (int k = 0; k < SAG_REC_FLOATS; k++) S[(size_t)k * N + i] = rf[k];
  // derived words: which vases move (velocity loads / write-backs of the others are skipped)
  // and whether any bodies overlap at rest (then nothing may be treated as asleep)
  uint32_t awake = 0, overlap = 0;
  {
    const int nV = ri[SAG_I_NV], nP = ri[SAG_I_NP], nB = ri[SAG_I_NB];
    const float vr = rf[SAG_F_VASE_SIZE] * 1.41421356237309504880f;
    for (int a = 0; a < nV; a++) {
      const float* va = rf + SAG_F_VASES + 6 * a;
      if (va[3] != 0 || va[4] != 0 || va[5] != 0) awake |= 1u << a;
      for (int b = a + 1; b < nV; b++) {
        const float* vb = rf + SAG_F_VASES + 6 * b;
        float dx = va[0] - vb[0], dy = va[1] - vb[1];
        if (dx * dx + dy * dy <= 4 * vr * vr) overlap = 1;
      }
      for (int q = 0; q < nP; q++) {
        float dx = va[0] - rf[SAG_F_PILLARS + 2 * q], dy = va[1] - rf[SAG_F_PILLARS + 2 * q + 1];
        float rs = vr + rf[SAG_F_PILLAR_SIZE];
        if (dx * dx + dy * dy <= rs * rs) overlap = 1;
      }
      for (int q = 0; q < nB; q++) {
        float dx = va[0] - rf[SAG_F_BUTTONS + 2 * q], dy = va[1] - rf[SAG_F_BUTTONS + 2 * q + 1];
        float rs = vr + BUTTON_R;
        if (dx * dx + dy * dy <= rs * rs) overlap = 1;
      }
    }
  }
  I[(size_t)DI_META * N + i] = (int32_t)(pack_meta(ri) | (overlap ? META_FULL_PAIRS : 0u));
  I[(size_t)DI_TSTATE * N + i] = (int32_t)(pack_tstate(ri) | awake << TS_AWAKE_SHIFT);
  I[(size_t)DI_STEP * N + i] = ri[SAG_I_STEP];
  I[(size_t)DI_ENVID * N + i] = ri[SAG_I_ENV_ID];
  I[(size_t)DI_FLAGS * N + i] = ri[SAG_I_FLAGS];
  if (init_task) {
    // task.reset() as run by World.reset right after rebuild (world.py:167-170):
    // `last` distances from the installed positions (go_to_goal.py:54-55,
    // press_buttons.py:75-77).
    int task = ri[SAG_I_TASK];
    double rx = rf[SAG_F_ROBOT], ry = rf[SAG_F_ROBOT + 1];
    bool buttons = task == SAG_TASK_PRESS_BUTTONS || task == SAG_TASK_PRESS_BUTTONS_SCARCE;
    if (buttons) {
      int b = ri[SAG_I_GOAL_BUTTON];
      S[(size_t)SAG_F_LAST * N + i] = (float)dist2d(rx, ry, rf[SAG_F_BUTTONS + 2 * b], rf[SAG_F_BUTTONS + 2 * b + 1]);
    } else if (task != SAG_TASK_COLLECT) {
      S[(size_t)SAG_F_LAST * N + i] = (float)dist2d(rx, ry, rf[SAG_F_GOAL], rf[SAG_F_GOAL + 1]);
    }
  }
}

__global__ void k_extract(const float* S, const int32_t* I, int N, const int32_t* env_ids, int n,
                          float* rec_f, int32_t* rec_i) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int i = env_ids ? env_ids[j] : j;
  float* rf = rec_f + (size_t)j * SAG_REC_FLOATS;
  int32_t* ri = rec_i + (size_t)j * SAG_REC_INTS;
  for (int k = 0; k < SAG_REC_FLOATS; k++) rf[k] = S[(size_t)k * N + i];
  for (int k = 0; k < SAG_REC_INTS; k++) ri[k] = 0;
  unpack_meta((uint32_t)I[(size_t)DI_META * N + i], ri);
  unpack_tstate((uint32_t)I[(size_t)DI_TSTATE * N + i], ri);
  ri[SAG_I_STEP] = I[(size_t)DI_STEP * N + i];
  ri[SAG_I_ENV_ID] = I[(size_t)DI_ENVID * N + i];
  ri[SAG_I_FLAGS] = I[(size_t)DI_FLAGS * N + i];
}

// ---------------------------------------------------------------------------
// stand-alone lidar + hazard cost (BASELINE config 2): one env per lane
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_lidar_cost(int n, int K, const float* robot,
                                                     const float* points, const uint8_t* group,
                                                     float hazard_size, float* lidar,
                                                     int32_t* bins, uint8_t* cost) {
  int i = blockIdx.x * WAVE + threadIdx.x;
  if (i >= n) return;
  double obs[48];
#pragma unroll
  for (int k = 0; k < 48; k++) obs[k] = 0;
  const double rx = robot[(size_t)i * 3], ry = robot[(size_t)i * 3 + 1], yaw = robot[(size_t)i * 3 + 2];
  const double c = cos(yaw), s = sin(yaw);
  int cst = 0;
  for (int j = 0; j < K; j++) {
    uint8_t gg = group[(size_t)i * K + j];
    int g = gg & 127;
    double px = points[((size_t)i * K + j) * 2], py = points[((size_t)i * K + j) * 2 + 1];
    if (gg & 128) {
      if (dist2d(rx, ry, px, py) <= (double)hazard_size) cst = 1;
    }
    int b = -1;
    double w0 = px - rx, w1 = py - ry;
    double ex = w0 * c + w1 * s, ey = w0 * -s + w1 * c;
    if (g == 1) b = lidar_accum(ex, ey, obs);
    else if (g == 3) b = lidar_accum(ex, ey, obs + 16);
    else if (g == 2) b = lidar_accum(ex, ey, obs + 32);
    if (bins) bins[(size_t)i * K + j] = b;
  }
#pragma unroll
  for (int k = 0; k < 48; k++) lidar[(size_t)i * 48 + k] = (float)obs[k];
  cost[i] = (uint8_t)cst;
}

// synthetic policy: U(-1,1)^nu from Philox stream 2
__global__ void k_fill_actions(float* actions, const int32_t* I, int N, int nu, uint32_t k0,
                               uint32_t k1, uint32_t step_index) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint32_t env = (uint32_t)I[(size_t)DI_ENVID * N + i];
  for (int j = 0; j < nu; j += 4) {
    uint32_t c[4] = {env, step_index, (uint32_t)(j >> 2), 2u};
    philox4x32_10(c, k0, k1);
    for (int k = 0; k < 4 && j + k < nu; k++)
      actions[(size_t)i * nu + j + k] = ((float)(c[k] >> 8) + 0.5f) * (2.0f / 16777216.0f) - 1.0f;
  }
}

}  // namespace sag
