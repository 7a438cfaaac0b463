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

#include <type_traits>

#include "../../include/sag.h"

namespace sag {

constexpr int WAVE = 64;

// device int fields.  TSTATE is a plain [N] column (k_compact and the quiet kernel's busy test read
// only it); the other four words of an env sit together as one int4 behind it.
enum { DI_META = 0, DI_TSTATE = 1, DI_STEP = 2, DI_ENVID = 3, DI_FLAGS = 4, DI_COUNT = 5 };
// the FLAGS word: record flags in the low byte, the episode nonce (SAG_I_EPISODE) above it
constexpr int FLAG_EPISODE_SHIFT = 8;
// (the int4 block starts on a 16-byte boundary for any N: the TSTATE column is padded to a multiple of four words)
__host__ __device__ constexpr size_t ipad(size_t N) { return (N + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t icount(size_t N) { return ipad(N) + 4 * N; }   // words of the whole block
__host__ __device__ constexpr size_t iaddr(int w, size_t N, size_t i) {
  return w == DI_TSTATE ? i : ipad(N) + i * 4 + (w == DI_META ? 0 : (w == DI_STEP ? 1 : (w == DI_ENVID ? 2 : 3)));
}

// ---- device layout of the float state ---------------------------------------------------------
// Groups of four floats per env, group-major: float4 number g of env i is ((float4*)S)[g * N + i].
// A wavefront reading group g of 64 consecutive envs moves one contiguous KB (as coalesced as the
// plain field-major layout), and a GATHER - the busy kernel reading scattered envs - moves a quarter
// of the cache lines per useful float.  The order inside the device record follows what the step
// prologue reads together, not the order of the ABI record (k_install / k_extract translate):
//   0 robot x y yaw vx | 4 vy w goal_x goal_y | 8 last0 gear damp action_noise | 12 ctrl_scale0,1
//   vase_size pillar_size | 16 pillars | 20..37 hazards, 38 hazard_size, 39 bound |
//   40.. positions (x y yaw) of the 11 free bodies | 76.. their velocities (vx vy w) |
//   112 buttons | 124 catch | 128 last1,2 | 130 keepouts | 136 rebuild pose | 140 ctrl_scale2.. |
//   152 robot extension
constexpr int DEV_FLOATS = 192, DEV_GROUPS = DEV_FLOATS / 4;
constexpr int DV_POS = 40, DV_VEL = 76, DG_HAZ = 5, DG_POS = 10, DG_PILLARS = 4, DG_BUTTONS = 28;
__host__ __device__ constexpr int didx(int k) {
  return k < 6 ? k
       : k < 9 ? 136 + (k - 6)
       : k < 12 ? k
       : k < 14 ? k
       : k < 24 ? 140 + (k - 14)
       : k == 24 ? 38
       : k < 27 ? 14 + (k - 25)
       : k < 32 ? 130 + (k - 27)
       : k < 34 ? 6 + (k - 32)
       : k < 38 ? 124 + (k - 34)
       : k == 38 ? 8
       : k < 41 ? 128 + (k - 39)
       : k < 44 ? DV_POS + 30 + (k - 41)
       : k < 47 ? DV_VEL + 30 + (k - 44)
       : k < 65 ? 20 + (k - 47)
       : k < 69 ? 16 + (k - 65)
       : k < 81 ? 112 + (k - 69)
       : k < 141 ? ((k - 81) % 6 < 3 ? DV_POS + 3 * ((k - 81) / 6) + (k - 81) % 6
                                     : DV_VEL + 3 * ((k - 81) / 6) + (k - 81) % 6 - 3)
       : k == 141 ? 39
       : k < 144 ? 150 + (k - 142)
       : 152 + (k - 144);
}
static_assert(didx(SAG_F_BUTTONS) == 112 && didx(SAG_F_VASES + 6 * 3 + 4) == DV_VEL + 10 && didx(SAG_F_BOX + 1) == DV_POS + 31 &&
              didx(SAG_F_HAZARDS) == 4 * DG_HAZ && didx(SAG_F_PILLARS) == 4 * DG_PILLARS, "device layout constants");
__host__ __device__ constexpr size_t saddr(int k, size_t N, size_t i) {
  return ((size_t)(didx(k) >> 2) * N + i) * 4 + (didx(k) & 3);
}

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

// Car floor friction: what depends on the substep length only (oracle car_floor_friction).  Computed on the host
// (launch_step) and handed over as kernel arguments: wave-uniform values that live in SGPRs instead of ~19 VGPRs of
// kernels that are short of them (the busy Car kernel spilled its friction solve to scratch: 50x slower there).
struct CarFricK {
  float iIw, iIb;                    // 1 / (inertia + h damping) of a wheel / the rear ball
  float iA_L, iA_T, iA_X, iA_Y;      // 1 / (A + R) per element kind: L wheels rolling, T lateral (merged), X / Y caster
  float R_L, R_T, R_X, R_Y;          // regulariser R = A (1 - d0) / d0 (T: half, two equal elements merged)
  float w_L, w_T, w_X, w_Y;          // 1 / D = 1 / (r_w^2 / I_spin + R): weight of the eliminated element in the direct solve
  float K00, K02, K22, idk, iK11;    // friction-augmented inertia K = M + sum J^T J / D: {x, yaw} block, its 1 / det, 1 / K_yy
};

struct StepArgs {
  CarFricK car;
  int envs_per_wave;  // single-launch form (k_step): 8..64 envs per wavefront, by batch size
  int busy_envs;      // split form (k_step_busy): envs per busy wavefront, or 0 = balanced over `busy_slots` (busy_wave_envs)
  int busy_slots;     // busy wavefronts the chip holds at once
  int busy_kinds;     // 1: the busy list is kept by kind (BUSY_CLASSES); 0: one list (SAG_BUSY_KINDS=0, A/B)
  uint8_t* kind;      // [N] kind of every busy env, for the next step's compaction
  int32_t* busy_total;  // the busy launch leaves its env count here
  float* S;          // [DEV_GROUPS][N] float4 (see didx)
  int32_t* I;        // tstate [N], then (meta, step, envid, flags) [N] int4 (see iaddr)
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
  int32_t has_box;
  int32_t phase;          // which copy of the busy bit this launch reads (0 / 1)
  int32_t* rows;          // [BUSY_CLASSES][N] env ids of the busy envs by kind, compacted (k_compact)
  int32_t* count;         // [BUSY_CLASSES] entries in each list
  int32_t* rows_next;     // split form: the busy list of the NEXT step, appended to by the classification
  int32_t* count_next;    //   (nullptr: not built; the host then runs k_compact)
  float* G;               // [3][NBODY][N] spill of body accelerations beyond the LDS pool
  int32_t observe_only;   // 1: sag_observe (no noise/physics/reward/cost)
  const int32_t* ext_cc;  // sag_set_ext_contacts: [N] contact counts (>= 0: replaces the geometric result) or nullptr
  const uint32_t* ext_btn;  // [N] touched-button masks
  float* hot;             // split form: [N][HOT_FLOATS] env-major copy of what the busy prologue reads, or nullptr
  const float* hot_haz;   // [N][20]: the five hazard groups env-major (static: written by k_hot_refresh only)
  double* DR;             // Doggo, cooperative form: per-env result block of k_doggo_physics [N][DR_STRIDE]
                          // (k_step<DOGGO> with DR set skips the physics and reads it); else nullptr
  int32_t* dg_sched;      // Doggo: longest-first launch order of k_doggo_physics (sag_doggo_coop.hpp), or nullptr
  int32_t dg_phase;       //   step counter of that rotation; < 0: this launch neither uses nor builds the lists (observe)
};
// "Hot record": the 16 float4 a busy env's prologue needs (groups 0-4, the nine position groups, the int4
// words and tstate), contiguous per env = two cache lines.  The busy kernel reads compacted, scattered
// envs: from the group-major state that is one cache line per group (~16 per env); the hot record is
// written by whichever kernel classifies an env busy for the next step (~10 % of the envs) and by
// k_hot_refresh after an install.
constexpr int HOT_GROUPS = 16, HOT_FLOATS = 4 * HOT_GROUPS;
constexpr int DR_STRIDE = 20;  // qacc_lin 3, touch 8, comvel 4, cost_contacts, btn_mask, row overflow, pad

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

// ---- constants of the Car robot (assets/xmls/car.xml, density 5) -------------------
// masses: boxes 8 hx hy hz rho; wheels pi r^2 L rho; rear ball 4/3 pi r^3 rho
constexpr float PI_F = 3.14159265358979323846f;
constexpr float CAR_MB0 = 5 * 8 * 0.1f * 0.1f * 0.05f, CAR_MB1 = 5 * 8 * 0.1f * 0.01f * 0.05f,
                CAR_MB2 = 5 * 8 * 0.01f * 0.025f * 0.03f, CAR_MB3 = 5 * 8 * 0.05f * 0.01f * 0.05f,
                CAR_MB4 = 5 * 8 * 0.05f * 0.03f * 0.01f, CAR_MW = 5 * PI_F * 0.05f * 0.05f * 0.05f,
                CAR_MBALL = 5 * 4.0f / 3.0f * PI_F * 0.05f * 0.05f * 0.05f;
constexpr float CAR_M = CAR_MB0 + CAR_MB1 + CAR_MB2 + CAR_MB3 + CAR_MB4 + 2 * CAR_MW + CAR_MBALL;
constexpr float CAR_OY = (CAR_MB1 * 0.15f + CAR_MB2 * 0.125f - CAR_MB3 * 0.165f - CAR_MB4 * 0.13f +
                          2 * CAR_MW * 0.1f - CAR_MBALL * 0.1f) / CAR_M;
// yaw inertia about the base origin: own (box m (a^2+b^2)/3, cylinder-on-its-side m (3r^2+L^2)/12,
// sphere 2/5 m r^2) + m d^2
constexpr float CAR_IO =
    CAR_MB0 * (0.01f + 0.01f) / 3 + CAR_MB1 * ((0.01f + 0.0001f) / 3 + 0.15f * 0.15f) +
    CAR_MB2 * ((0.0001f + 0.000625f) / 3 + 0.125f * 0.125f) + CAR_MB3 * ((0.0025f + 0.0001f) / 3 + 0.165f * 0.165f) +
    CAR_MB4 * ((0.0025f + 0.0009f) / 3 + 0.13f * 0.13f) +
    2 * CAR_MW * ((3 * 0.0025f + 0.0025f) / 12 + 0.13f * 0.13f + 0.01f) + CAR_MBALL * (0.4f * 0.0025f + 0.01f);
constexpr float CAR_IW = 0.5f * CAR_MW * 0.0025f + 0.00025f;  // axle inertia + armature (car.xml:22,26)
constexpr float CAR_IB = 0.4f * CAR_MBALL * 0.0025f;
// projected Gauss-Seidel sweeps over the car's five floor-friction elements: after the direct solve (the first solve of an
// env-step) / in a solve that starts from the previous solve's forces (oracle car_floor_friction)
constexpr int CAR_FRICTION_SWEEPS = 4, CAR_FRICTION_SWEEPS_WARM = 2;

// ---- counter-based generator ------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    // (one 64-bit product per multiplier - v_mad_u64_u32 - instead of a mul_hi + mul_lo pair: integer multiplies
    // issue at a quarter of the rate, and this runs for every env-step)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct Rng {
  const uint32_t* tape;  // this env's row or nullptr
  int len, pos, exhausted;
  uint32_t k0, k1, env, step;
  uint32_t ep4;   // episode nonce << 2 (counter word 3 = nonce | stream: 0 in-step draws, 1 action noise, 2 bench policy)
  __device__ uint32_t word() {
    if (tape) {
      if (pos >= len) { exhausted = 1; pos++; return 0u; }
      return tape[pos++];
    }
    uint32_t c[4] = {env, step, (uint32_t)(pos >> 2), ep4};
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

// two standard normals from two random words (throughput mode's action noise; the oracle's
// sago_noise is the same transform in libm arithmetic).  Hardware log2 / sin / cos (1 ulp-class, the
// trigonometric ones take revolutions): the draws are noise, nothing discrete depends on their last bits.
__device__ inline void box_muller(uint32_t w0, uint32_t w1, float& z0, float& z1) {
  const float u1 = ((float)(w0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));  // -2 ln u1
  z0 = r * __builtin_amdgcn_cosf(u2); z1 = r * __builtin_amdgcn_sinf(u2);
}

// ---- lidar (fp64, the reference's arithmetic) -------------------------------
__device__ inline int lidar_accum(double ex, double ey, double* obs) {
  const double two_pi = PI_D * 2;
  double dist = hypot(ex, ey);
  double ang = atan2(ey, ex);
  if (ang < 0) ang += two_pi;  // python float %: fmod leaves |ang| < 2pi untouched, then shifts
  const double bin_size = two_pi / SAG_LIDAR_BINS;
  int bin = ang >= 0 ? (int)(ang / bin_size) : 0;   // (NaN pose: bin 0, no conversion of a NaN)
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
struct Sol { float bcoef, kcoef, mu; };  // soft-contact reference acceleration + friction of a pair

__device__ inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
// a[k] of a small register array for a run-time k, as a select chain: an indexed access would put the array into
// scratch memory (the 32 B of scratch the quiet kernel had: the two pillar coordinates, read back in two loops)
template <int N>
__device__ inline float pick(const float (&a)[N], int k) {
  float v = a[0];
#pragma unroll
  for (int z = 1; z < N; z++) v = k == z ? a[z] : v;
  return v;
}

// The planar contact code below contracts a * b + c within an expression into one fused multiply-add
// (`#pragma clang fp contract(on)`; the translation unit is built with -ffp-contract=off for the fp64 paths that
// follow the reference's arithmetic operation by operation).  The busy kernels are chains of dependent fp32
// operations at two wavefronts per SIMD: a fused pair is one link instead of two.
__device__ inline float minv_apply(const BV& b, float dx, float dy, float rxd, float u[3]) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  u[0] = b.m0 * dx + b.m1 * dy + b.m2 * rxd;
  u[1] = b.m1 * dx + b.m3 * dy + b.m4 * rxd;
  u[2] = b.m2 * dx + b.m4 * dy + b.m5 * rxd;
  return dx * u[0] + dy * u[1] + rxd * u[2];
}

__device__ inline void rel_at(const BV& A, const BV& B, float rax, float ray, float rbx, float rby,
                              float& vx, float& vy, float& ax, float& ay) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  vx = (B.vx - B.w * rby) - (A.vx - A.w * ray);
  vy = (B.vy + B.w * rbx) - (A.vy + A.w * rax);
  ax = (B.ax - B.aw * rby) - (A.ax - A.aw * ray);
  ay = (B.ay + B.aw * rbx) - (A.ay + A.aw * rax);
}

__device__ inline float impedance(float depth) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  float x = fminf(depth * (1.0f / SOL_WIDTH), 1.0f);
  float y = x < 0.5f ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);
  return SOL_D0 + (SOL_D1 - SOL_D0) * y;
}

// soft contact, normal then friction; n points from A to B
__device__ inline void solve_contact(BV& A, BV& B, float nx, float ny, float px, float py,
                                     float depth, const Sol& sol) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
#if defined(SAG_ABL_CL) && SAG_ABL_CL == 4   // timing-only: pair and vertex tests without the contact solve
  return;
#endif
  float rax = px - A.x, ray = py - A.y, rbx = px - B.x, rby = py - B.y;
  float vx, vy, ax, ay, ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
  rel_at(A, B, rax, ray, rbx, rby, vx, vy, ax, ay);
  float vn = vx * nx + vy * ny, an = ax * nx + ay * ny, An = 0;
  if (A.dyn) An += minv_apply(A, nx, ny, rax * ny - ray * nx, ua);
  if (B.dyn) An += minv_apply(B, nx, ny, rbx * ny - rby * nx, ub);
  if (!(An > 0)) return;
  float d = impedance(depth);
  float fn = d * ((-sol.bcoef * vn + sol.kcoef * depth) - an) / An;
  if (!(fn > 0)) return;
  if (A.dyn) { A.ax -= ua[0] * fn; A.ay -= ua[1] * fn; A.aw -= ua[2] * fn; }
  if (B.dyn) { B.ax += ub[0] * fn; B.ay += ub[1] * fn; B.aw += ub[2] * fn; }
  float tx = -ny, ty = nx;
  rel_at(A, B, rax, ray, rbx, rby, vx, vy, ax, ay);
  float vt = vx * tx + vy * ty, at = ax * tx + ay * ty, At = 0;
  if (A.dyn) At += minv_apply(A, tx, ty, rax * ty - ray * tx, ua);
  if (B.dyn) At += minv_apply(B, tx, ty, rbx * ty - rby * tx, ub);
  if (!(At > 0)) return;
  float ft = clampf(d * (-sol.bcoef * vt - at) / At, -sol.mu * fn, sol.mu * fn);
  if (A.dyn) { A.ax -= ua[0] * ft; A.ay -= ua[1] * ft; A.aw -= ua[2] * ft; }
  if (B.dyn) { B.ax += ub[0] * ft; B.ay += ub[1] * ft; B.aw += ub[2] * ft; }
}

// circle (A, centre ax,ay radius ra) vs circle (B)
__device__ inline int cc_contact(BV& A, BV& B, float ax, float ay, float ra, float bx, float by,
                                 float rb, const Sol& sol) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  float dx = bx - ax, dy = by - ay, d2 = dx * dx + dy * dy, rs = ra + rb;
  if (d2 >= rs * rs) return 0;
  float d = sqrtf(d2), nx = 1, ny = 0;
  if (d > 1e-12f) { nx = dx / d; ny = dy / d; }
  float depth = rs - d;
  solve_contact(A, B, nx, ny, ax + nx * (ra - 0.5f * depth), ay + ny * (ra - 0.5f * depth), depth, sol);
  return 1;
}

// circle vs oriented box; `circle_is_A` selects the normal direction (A -> B)
__device__ inline int cb_contact(BV& Circ, BV& Box, float cx, float cy, float r, float bx, float by,
                                 float cb, float sb, float hx, float hy, bool circle_is_A,
                                 const Sol& sol) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
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
  if (circle_is_A) solve_contact(Circ, Box, -wnx, -wny, px, py, depth, sol);
  else solve_contact(Box, Circ, wnx, wny, px, py, depth, sol);
  return 1;
}

// vertices of box P strictly inside box Q. If q_is_A the normal A->B is Q's outward
// normal, else its negative.  A/B are passed in solver order.
// All four vertices are tested in one straight-line pass (a 4-bit mask per lane); each lane then solves ITS inside
// vertices in ascending order, so a wavefront runs the contact solve max-inside-count times (1 - 2), not once per
// vertex index that any lane has inside.
__device__ inline uint32_t verts_inside_mask(float pxc, float pyc, float cp, float sp, float phx, float phy, float qxc,
                                             float qyc, float cq, float sq, float qhx, float qhy) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  uint32_t inside = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float sx = (k == 0 || k == 3) ? phx : -phx, sy = (k < 2) ? phy : -phy;
    const float vx = pxc + cp * sx - sp * sy, vy = pyc + sp * sx + cp * sy;
    const float wx = vx - qxc, wy = vy - qyc;
    const float lx = cq * wx + sq * wy, ly = -sq * wx + cq * wy;
    const float dx = qhx - fabsf(lx), dy = qhy - fabsf(ly);
    inside |= (uint32_t)!(dx <= 0 || dy <= 0) << k;
  }
  return inside;
}
// the contact of vertex k of P (known to be inside Q): the same expressions as the test above
__device__ inline void vert_contact(BV& A, BV& B, int k, float pxc, float pyc, float cp, float sp, float phx, float phy,
                                    float qxc, float qyc, float cq, float sq, float qhx, float qhy, bool q_is_A,
                                    const Sol& sol) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  const float sx = (k == 0 || k == 3) ? phx : -phx, sy = (k < 2) ? phy : -phy;
  const float vx = pxc + cp * sx - sp * sy, vy = pyc + sp * sx + cp * sy;
  const float wx = vx - qxc, wy = vy - qyc;
  const float lx = cq * wx + sq * wy, ly = -sq * wx + cq * wy;
  const float dx = qhx - fabsf(lx), dy = qhy - fabsf(ly);
  float onx, ony, depth;
  if (dx < dy) { onx = lx >= 0 ? 1.f : -1.f; ony = 0; depth = dx; }
  else { onx = 0; ony = ly >= 0 ? 1.f : -1.f; depth = dy; }
  const float sgn = q_is_A ? 1.f : -1.f;
  solve_contact(A, B, sgn * (cq * onx - sq * ony), sgn * (sq * onx + cq * ony), vx, vy, depth, sol);
}
__device__ inline int verts_in_box(BV& A, BV& B, float pxc, float pyc, float cp, float sp, float phx,
                                   float phy, float qxc, float qyc, float cq, float sq, float qhx,
                                   float qhy, bool q_is_A, const Sol& sol) {
  int n = 0;
  for (uint32_t inside = verts_inside_mask(pxc, pyc, cp, sp, phx, phy, qxc, qyc, cq, sq, qhx, qhy); inside; inside &= inside - 1) {
    vert_contact(A, B, __ffs(inside) - 1, pxc, pyc, cp, sp, phx, phy, qxc, qyc, cq, sq, qhx, qhy, q_is_A, sol);
    n++;
  }
  return n;
}

__device__ inline int bb_contact(BV& A, BV& B, float ax, float ay, float ca, float sa, float ahx,
                                 float ahy, float bx, float by, float cb, float sb, float bhx,
                                 float bhy, const Sol& sol) {
  int n = verts_in_box(A, B, ax, ay, ca, sa, ahx, ahy, bx, by, cb, sb, bhx, bhy, false, sol);
  n += verts_in_box(A, B, bx, by, cb, sb, bhx, bhy, ax, ay, ca, sa, ahx, ahy, true, sol);
  return n;
}

// Separating-axis test of two oriented boxes (centre, cos / sin, half extents): true = some axis of one of them
// separates the two, so no vertex of either lies inside the other and bb_contact would find nothing.  A cull the
// specification does not have; SAT_EPS makes it conservative against its own rounding (a pair is only skipped when
// the gap along an axis exceeds 1e-5 m; bb_contact's vertices must be strictly inside).  Most pairs that pass the
// bounding circles are thin bumpers beside a box: 40 instructions here instead of 2 x 4 vertex transforms there.
constexpr float SAT_EPS = 1e-5f;
#ifndef SAG_NO_SAT
__device__ inline bool boxes_separated(float ax, float ay, float ca, float sa, float ahx, float ahy, float bx, float by,
                                       float cb, float sb, float bhx, float bhy) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  const float c = fabsf(ca * cb + sa * sb), s = fabsf(sa * cb - ca * sb);   // |axis . axis| of the two frames
  const float tx = bx - ax, ty = by - ay;
  const float ta_x = ca * tx + sa * ty, ta_y = ca * ty - sa * tx;   // centre offset in A's frame
  const float tb_x = cb * tx + sb * ty, tb_y = cb * ty - sb * tx;   // ... in B's
  return fabsf(ta_x) > ahx + bhx * c + bhy * s + SAT_EPS || fabsf(ta_y) > ahy + bhx * s + bhy * c + SAT_EPS ||
         fabsf(tb_x) > bhx + ahx * c + ahy * s + SAT_EPS || fabsf(tb_y) > bhy + ahx * s + ahy * c + SAT_EPS;
}
#else   // (diagnosis: results must be bit-identical with and without the cull - tests/diag_traj.py)
__device__ inline bool boxes_separated(float, float, float, float, float, float, float, float, float, float, float, float) { return false; }
#endif

__device__ inline double dist2d(double ax, double ay, double bx, double by) {
  double dx = ax - bx, dy = ay - by;
  return sqrt(dx * dx + dy * dy);
}

// ---- shapes -------------------------------------------------------------------
// A body's footprint is a list of geoms (circle or axis-aligned box in the body frame).
// Shape codes: the robot, a vase, the PushBox-family task object (box / rod / ball), a static
// circle.  Values follow point.xml:18-19, primitive_objects.py, push_box.py:28-72,
// roll_rod.py:19-43, dribble_ball.py:18-41.
enum { SH_ROBOT = 0, SH_VASE = 1, SH_BOX = 2, SH_ROD = 3, SH_BALL = 4, SH_STATIC = 5, SH_CAR = 6 };
struct Geom { int box; float ox, oy, a, b, r; };  // box ? half extents (a, b) : radius a; r: bounding radius, rounded up

__device__ inline int shape_ngeom(int sh) { return sh == SH_ROBOT ? 2 : (sh == SH_BOX ? 5 : (sh == SH_CAR ? 8 : 1)); }

__device__ inline Geom shape_geom(int sh, int g, float vsz, float rstatic) {
  Geom q; q.ox = 0; q.oy = 0; q.b = 0; q.r = 0;
  switch (sh) {
    case SH_ROBOT: q.box = g; q.ox = g ? 0.1f : 0.f; q.a = g ? 0.05f : 0.1f; q.b = 0.05f; q.r = g ? 0.0707108f : 0.1f; break;
    case SH_VASE: q.box = 1; q.a = vsz; q.b = vsz; q.r = vsz * 1.4142137f; break;
    case SH_BOX:
      q.box = 1; q.a = q.b = g ? 0.1f : 0.2f; q.r = g ? 0.1414215f : 0.2828429f;
      q.ox = g == 0 ? 0.f : ((g & 1) ? 0.2f : -0.2f);      // g: 1 (+,+) 2 (-,+) 3 (+,-) 4 (-,-)
      q.oy = g == 0 ? 0.f : (g <= 2 ? 0.2f : -0.2f);
      break;
    case SH_ROD: q.box = 1; q.a = 0.08f; q.b = 0.3f; q.r = 0.3104836f; break;
    case SH_BALL: q.box = 0; q.a = 0.14f; q.r = 0.14f; break;
    case SH_CAR:  // car.xml:16-32 footprints: base, back bumper/connector, front bumper/connector,
                  // left/right wheel (cylinders along x), rear ball
      q.box = g < 7;
      q.ox = g == 5 ? -0.13f : (g == 6 ? 0.13f : 0.f);
      q.oy = g == 1 ? 0.15f : (g == 2 ? 0.125f : (g == 3 ? -0.165f : (g == 4 ? -0.13f : (g == 7 ? -0.1f : (g >= 5 ? 0.1f : 0.f)))));
      q.a = g == 0 || g == 1 ? 0.1f : (g == 2 ? 0.01f : (g == 5 || g == 6 ? 0.025f : 0.05f));
      q.b = g == 0 ? 0.1f : (g == 1 || g == 3 ? 0.01f : (g == 2 ? 0.025f : (g == 4 ? 0.03f : 0.05f)));
      // sqrt(a^2 + b^2) of the seven boxes (rounded up), the ball's radius
      q.r = g == 0 ? 0.1414215f : (g == 1 ? 0.1004989f : (g == 2 ? 0.0269260f : (g == 3 ? 0.0509903f : (g == 4 ? 0.0583097f : (g < 7 ? 0.0559018f : 0.05f)))));
      break;
    default: q.box = 0; q.a = rstatic; q.r = rstatic; break;
  }
  return q;
}

__device__ inline float shape_bound(int sh, float vsz, float rstatic) {
  switch (sh) {
    case SH_ROBOT: return ROBOT_BOUND;
    case SH_VASE: return vsz * 1.41421356237309504880f;
    case SH_BOX: return 0.42426406871192851f;   // column corner (0.3, 0.3)
    case SH_ROD: return 0.31048349392520047f;   // sqrt(.08^2 + .3^2)
    case SH_BALL: return 0.14f;
    case SH_CAR: return 0.21569654610114636f;  // wheel corner (.155, .15)
    default: return rstatic;
  }
}

// every geom pair of two bodies, geoms of A outer, of B inner (the specification's order).
// CONST_R: cull with the constant bounding radii of shape_geom; the Doggo's planar world (sag_doggo_coop.hpp)
// keeps the run-time sqrt form it was validated with
template <bool CONST_R = false>
__device__ inline int collide_shapes(BV& A, int shA, float ca, float sa, BV& B, int shB, float cb,
                                     float sb, float vsz, float rstatic, const Sol& sol) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  int n = 0;
  const int na = shape_ngeom(shA), nb = shape_ngeom(shB);
#pragma unroll 1
  for (int ga = 0; ga < na; ga++) {
    const Geom a = shape_geom(shA, ga, vsz, rstatic);
    const float ax = A.x + ca * a.ox - sa * a.oy, ay = A.y + sa * a.ox + ca * a.oy;
    if constexpr (CONST_R) {
      // a geom of A that cannot reach B's bounding circle skips B's geoms altogether (the car's rear
      // geoms against the 5-geom box it pushes with its front)
      const float ex = B.x - ax, ey = B.y - ay, rb_all = a.r + shape_bound(shB, vsz, rstatic) * 1.000001f;
      if (nb > 1 && ex * ex + ey * ey > rb_all * rb_all) continue;
    }
#pragma unroll 1
    for (int gb = 0; gb < nb; gb++) {
      Geom b = shape_geom(shB, gb, vsz, rstatic);
      // the ball's centre is .04 above the robot sphere's: it presents sqrt(.24^2-.04^2)-.1
      if (shA == SH_ROBOT && ga == 0 && shB == SH_BALL) b.a = b.r = 0.13664319132398464f;
      const float bx = B.x + cb * b.ox - sb * b.oy, by = B.y + sb * b.ox + cb * b.oy;
      // per-geom bounding circles first (a cull the specification does not have: the radii only need
      // to contain the geoms, so they are constants rounded up, not a run-time sqrt per pair)
      float ra = a.r, rb = b.r;
      if constexpr (!CONST_R) { ra = a.box ? sqrtf(a.a * a.a + a.b * a.b) : a.a; rb = b.box ? sqrtf(b.a * b.a + b.b * b.b) : b.a; }
      const float dx = bx - ax, dy = by - ay, rs = ra + rb;
      if (dx * dx + dy * dy > rs * rs) continue;
      if (!a.box && !b.box) n += cc_contact(A, B, ax, ay, a.a, bx, by, b.a, sol);
      else if (!a.box) n += cb_contact(A, B, ax, ay, a.a, bx, by, cb, sb, b.a, b.b, true, sol);
      else if (!b.box) n += cb_contact(B, A, bx, by, b.a, ax, ay, ca, sa, a.a, a.b, false, sol);
      else if (!boxes_separated(ax, ay, ca, sa, a.a, a.b, bx, by, cb, sb, b.a, b.b))
        n += bb_contact(A, B, ax, ay, ca, sa, a.a, a.b, bx, by, cb, sb, b.a, b.b, sol);
    }
  }
  return n;
}

// The same pair tests for a compile-time shape A (the robot's footprint, a vase), organised for the wavefront: one
// branch-free pass tests every geom pair for a separating axis into a per-lane bit mask, then each lane walks ITS OWN
// hits in ascending (geom of A, geom of B) order - the specification's order.  collide_shapes runs its 8 x 5 loop for
// the union of what 64 lanes need; here a wavefront pays 40 straight-line tests + the maximum number of pairs of one
// lane that really overlap (2 - 3).
// The test: every geom is axis-aligned in its body's frame, so ONE relative rotation (|cos|, |sin| of the angle between
// the two bodies) serves all pairs; a pair costs four compares of a centre offset against half extents + projected
// half extents (circles count as their bounding squares).  With bounding circles alone (the first version) a thin
// bumper 'hit' everything within 0.1 m of its centre and the contact loop ran ~8 times a substep for a car at the
// box, finding nothing in most of them: 70 % of the busy kernel's instructions (PMC, contact solve switched off).
// Conservative by SAT_EPS; the contacts found are the same (bit-identical trajectories, tests/diag_traj.py).
template <int SHA, int NB>
__device__ inline int collide_list_nb(BV& A, float ca, float sa, BV& B, int shB, float cb, float sb, float vsz,
                                      float rstatic, const Sol& sol) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
  constexpr int NA = SHA == SH_ROBOT ? 2 : (SHA == SH_CAR ? 8 : 1);
  static_assert(SHA == SH_ROBOT || SHA == SH_CAR || SHA == SH_VASE, "shapes with a compile-time geom list");
#if defined(SAG_ABL_CL) && SAG_ABL_CL == 2   // timing-only: no pair test at all
  return 0;
#endif
  const float rc = ca * cb + sa * sb, rs = sa * cb - ca * sb;   // x axis of A . x axis of B, x axis of A . y axis of B
  const float ac = fabsf(rc), as = fabsf(rs);
  // B's geoms: centre in A's frame (relative to A's origin), half extents projected on A's axes
  float lbx[NB], lby[NB], pax[NB], pay[NB];
#pragma unroll
  for (int gb = 0; gb < NB; gb++) {
    const Geom b = shape_geom(shB, gb, vsz, rstatic);
    const float wx = B.x + cb * b.ox - sb * b.oy - A.x, wy = B.y + sb * b.ox + cb * b.oy - A.y;
    lbx[gb] = ca * wx + sa * wy; lby[gb] = ca * wy - sa * wx;
    const float hb = b.box ? b.b : b.a;
    pax[gb] = b.a * ac + hb * as + SAT_EPS; pay[gb] = b.a * as + hb * ac + SAT_EPS;
  }
  // A's origin in B's frame  (tried: a wave-uniform skip of A's geoms that no lane has within B's bounding circle -
  // no gain, some lane always needs each geom)
  const float odx = A.x - B.x, ody = A.y - B.y;
  const float obx = cb * odx + sb * ody, oby = cb * ody - sb * odx;
  typedef typename std::conditional<(NA > 4), uint64_t, uint32_t>::type mask_t;
  mask_t mask = 0;   // bit 8 ga + gb
#ifndef SAG_CL_UNROLL_A
#define SAG_CL_UNROLL_A 8
#endif
  // (the car's 8 x 5 pairs fully unrolled keep 15 + temporaries alive and spill: its geom loop stays rolled)
  constexpr int UNROLL_A = NA * NB > SAG_CL_UNROLL_A ? 1 : NA;
#pragma unroll UNROLL_A
  for (int ga = 0; ga < NA; ga++) {
    const Geom a = shape_geom(SHA, ga, vsz, rstatic);
    const float ha = a.box ? a.b : a.a;
    // this geom's centre in B's frame, its half extents projected on B's axes
    const float lax = obx + a.ox * rc - a.oy * rs, lay = oby + a.ox * rs + a.oy * rc;
    const float pbx = a.a * ac + ha * as + SAT_EPS, pby = a.a * as + ha * ac + SAT_EPS;
#pragma unroll
    for (int gb = 0; gb < NB; gb++) {
      const Geom b = shape_geom(shB, gb, vsz, rstatic);
      const float hb = b.box ? b.b : b.a;
      const bool hit = !(fabsf(lbx[gb] - a.ox) > a.a + pax[gb]) && !(fabsf(lby[gb] - a.oy) > ha + pay[gb]) &&
                       !(fabsf(lax - b.ox) > b.a + pbx) && !(fabsf(lay - b.oy) > hb + pby);
      mask |= (mask_t)hit << (8 * ga + gb);
    }
  }
  int n = 0;
#if defined(SAG_ABL_CL) && SAG_ABL_CL == 1   // timing-only: the bounding-circle pass alone
  mask = 0;
#endif
  while (mask) {
    const int pbit = (NA > 4 ? __ffsll((unsigned long long)mask) : __ffs((unsigned int)mask)) - 1;
    mask &= mask - 1;
    const int ga = NA > 1 ? pbit >> 3 : 0, gb = NB > 1 ? pbit & 7 : 0;
    const Geom a = shape_geom(SHA, ga, vsz, rstatic);
    Geom b = shape_geom(shB, gb, vsz, rstatic);
    const float ax = A.x + ca * a.ox - sa * a.oy, ay = A.y + sa * a.ox + ca * a.oy;
    const float qx = B.x + cb * b.ox - sb * b.oy, qy = B.y + sb * b.ox + cb * b.oy;
    if (!a.box && !b.box) n += cc_contact(A, B, ax, ay, a.a, qx, qy, b.a, sol);
    else if (!a.box) n += cb_contact(A, B, ax, ay, a.a, qx, qy, cb, sb, b.a, b.b, true, sol);
    else if (!b.box) n += cb_contact(B, A, qx, qy, b.a, ax, ay, ca, sa, a.a, a.b, false, sol);
    else n += bb_contact(A, B, ax, ay, ca, sa, a.a, a.b, qx, qy, cb, sb, b.a, b.b, sol);
  }
  return n;
}
// MULTI: the other body may be the 5-geom PushBox box (instances that hold a task object); else it has one geom
template <int SHA, bool MULTI>
__device__ inline int collide_list(BV& A, float ca, float sa, BV& B, int shB, float cb, float sb, float vsz,
                                   float rstatic, const Sol& sol) {
#ifdef SAG_COLLIDE_REF   // (diagnosis: the plain loop form; results must be bit-identical - tests/diag_traj.py)
  return collide_shapes<true>(A, SHA, ca, sa, B, shB, cb, sb, vsz, rstatic, sol);
#endif
  if constexpr (MULTI) {
    if (shape_ngeom(shB) > 1) return collide_list_nb<SHA, 5>(A, ca, sa, B, SH_BOX, cb, sb, vsz, rstatic, sol);   // (the only 5-geom shape)
  }
  // one or two pairs (Point robot or a vase against a single-geom body): nothing to gain from a list, the plain loop
  // is the shorter code (measured: the list form costs the Point step 5 %)
  if constexpr (SHA != SH_CAR) return collide_shapes<true>(A, SHA, ca, sa, B, shB, cb, sb, vsz, rstatic, sol);
  return collide_list_nb<SHA, 1>(A, ca, sa, B, shB, cb, sb, vsz, rstatic, sol);
}

// ---------------------------------------------------------------------------
// the fused step kernel (Point robot)
// ---------------------------------------------------------------------------
// Free bodies: index 0..9 vases, 10 the task object (PushBox family).
// LDS map (floats), one wavefront per workgroup, [slot][lane]:
//   slots  0..10 x      11..21 y      22..32 yaw            (every free body)
//         33..50 a pool of DPOOL dynamic entries (vx vy w ax ay aw), handed to the bodies that
//                move or get touched during the step; further ones spill to global memory
// = 13 KB per wavefront -> 12 wavefronts per CU.  For the observation the region from slot 22
// on is reused as a [lane][17] staging tile per 16-column chunk, read back transposed so the
// row-major [N][60] output is written in 64-B runs instead of scattered dwords.
constexpr int NBODY = SAG_MAX_VASES + 1, BOX_ID = SAG_MAX_VASES;
constexpr int DPOOL = 3;
enum { LS_X = 0, LS_Y = NBODY, LS_YAW = 2 * NBODY, LS_POOL = 3 * NBODY };
constexpr int LDS_SLOTS = LS_POOL + 6 * DPOOL;
constexpr int LDS_FLOATS = LDS_SLOTS * WAVE;
// Car, kernels with contact code (BUSY / ALL): values that live across the contact phases of a substep but are touched
// once or twice in it - the friction solve's warm-start forces and its share of the base acceleration, the spin
// accelerations, the rear ball's quaternion - are parked in [slot][lane] LDS slots behind the body slots instead of
// holding registers the pair walk needs (that instance spilled 73 VGPRs = 256 B of scratch per lane, and 2048 resident
// wavefronts x 64 x 256 B is more than the chip's L2: the spills went to HBM).  SAG_CAR_PARK=0: in registers (A/B).
#ifndef SAG_CAR_PARK
#define SAG_CAR_PARK 1
#endif
constexpr int CAR_PARK_SLOTS = SAG_CAR_PARK ? 23 : 0;
enum { CP_FL = 0, CP_F0 = 5, CP_PX = 8, CP_EACC = 11, CP_Q = 16, CP_TAIL = 20 };
#define CPK(j) lds[(LDS_SLOTS + (j)) * WAVE + lane]
constexpr int STG_BASE = LS_YAW * WAVE;

// timing-only ablations (tools/ablate.py): -DSAG_ABLATE=<mask>; results are wrong by design
#ifndef SAG_ABLATE
#define SAG_ABLATE 0
#endif
enum { ABL_NO_LIDAR = 1, ABL_NO_OBS_STORE = 2, ABL_NO_VV = 4, ABL_NO_VS = 8, ABL_NO_RV = 16, ABL_NO_RS = 32,
       ABL_NSUB1 = 64, ABL_NO_ACTIVE = 128, ABL_NO_RESAMPLE = 8192 };
#define ABL(f) ((SAG_ABLATE & (f)) != 0)
// section profile (tools/cycles.py): -DSAG_CYCLES accumulates wavefront clock ticks per code
// section of step_body into g_cyc[mode][section]; off in the shipped library
enum { CY_LOAD = 0, CY_ROBOT, CY_RS, CY_RV, CY_VS, CY_VV_BROAD, CY_VV_NARROW, CY_INTEG, CY_WRITEBACK,
       CY_REWARD, CY_RESAMPLE, CY_COST, CY_LIDAR, CY_OBS_STORE, CY_TAIL, CY_N };
#ifdef SAG_WAVE_TIMES   // (tools/busy_timeline.py) per busy wavefront of the LAST launch: start / end (100-MHz wall clock), its env ids and, per
// lane, how often each contact loop ran over the step - the raw material of a cost model of a busy wavefront
constexpr int WT_MAX = 16384;
static __device__ unsigned long long g_wt[WT_MAX][2];
static __device__ int g_wt_env[WT_MAX][64];
static __device__ unsigned char g_wt_work[WT_MAX][64][8];
static __device__ unsigned short g_wt_trips[WT_MAX][8];   // loop passes the WAVEFRONT ran (sum over substeps of the max over its lanes)
#define WTW(k) do { if (wt_work[k] < 255) wt_work[k]++; } while (0)
__device__ inline int wt_wave_max(int v) { for (int off = 32; off; off >>= 1) v = max(v, __shfl_xor(v, off)); return v; }
#define WTT(k, v) do { wt_trips[k] += (unsigned)wt_wave_max((int)(v)); } while (0)
#else
#define WTW(k) do {} while (0)
#define WTT(k, v) do {} while (0)
#endif
#ifdef SAG_CYCLES
static __device__ unsigned long long g_cyc[3][CY_N + 1];
static __device__ unsigned long long g_cyc_worst[3][CY_N + 1];   // the sections of the wavefront with the largest total (last column)
static __device__ unsigned long long g_cyc_worst_block[3];        // ... and its block index
static __device__ unsigned long long g_cyc_hist[3][64];           // wavefronts by total ticks, four buckets per octave (bucket = floor(4 log2 ticks) - 40)
#define CYC_DECL unsigned long long cyc_acc[CY_N] = {}; unsigned long long cyc_t = __builtin_readcyclecounter();
#define CYC(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); cyc_acc[k] += t_ - cyc_t; cyc_t = t_; } while (0)
// the sections inside the active-body block run under a divergent branch: a lane outside it books the wavefront's time there on its next
// marker (`robot`).  Per section the wavefront's figure is therefore the MAX over lanes, except `robot`: the MIN over the lanes that carry an env.
__device__ inline unsigned long long cyc_wave_red(unsigned long long v, bool take_max) {
  for (int off = 32; off; off >>= 1) {
    const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), off) << 32) | (unsigned)__shfl_xor((int)(v & 0xffffffffu), off);
    v = take_max ? (o > v ? o : v) : (o < v ? o : v);
  }
  return v;
}
#define CYC_FLUSH(mode) do { unsigned long long tot_ = 0, red_[CY_N]; \
    for (int k_ = 0; k_ < CY_N; k_++) { red_[k_] = cyc_wave_red(k_ == CY_ROBOT && !live ? ~0ull : cyc_acc[k_], k_ != CY_ROBOT); tot_ += red_[k_]; } \
    if (lane == 0) { for (int k_ = 0; k_ < CY_N; k_++) atomicAdd(&g_cyc[mode][k_], red_[k_]); \
    atomicAdd(&g_cyc[mode][CY_N], 1ull); \
    { const int lg_ = 63 - __clzll((long long)(tot_ | 1ull)); const int q_ = lg_ >= 2 ? (int)(tot_ >> (lg_ - 2) & 3ull) : 0; \
      const int b_ = 4 * lg_ + q_ - 40; atomicAdd(&g_cyc_hist[mode][b_ < 0 ? 0 : (b_ > 63 ? 63 : b_)], 1ull); } \
    if (atomicMax(&g_cyc_worst[mode][CY_N], tot_) < tot_) { for (int k_ = 0; k_ < CY_N; k_++) g_cyc_worst[mode][k_] = red_[k_]; g_cyc_worst_block[mode] = blockIdx.x; } } } while (0)
#else
#define CYC_DECL
#define CYC(k) do {} while (0)
#define CYC_FLUSH(mode) do {} while (0)
#endif
}  // namespace sag
#include "sag_doggo.hpp"
namespace sag {
#define SF(k) S[saddr((k), (size_t)N, (size_t)i)]
// by DEVICE index (free-body positions DV_POS + 3 k + c, velocities DV_VEL + 3 k + c with k = BOX_ID for
// the task object): no record-to-device translation for run-time body indices
#define SD(d) S[((size_t)((d) >> 2) * N + i) * 4 + ((d) & 3)]
#define LP(base, k) lds[((base) + (k)) * WAVE + lane]
#define POOL(d, c) lds[(LS_POOL + (d) * 6 + (c)) * WAVE + lane]
#define STG(j) lds[STG_BASE + lane * STG_STRIDE + (j)]  // STG_STRIDE: constexpr of the enclosing kernel
// The 16 lidar bins of a chunk accumulate in a [bin][lane] tile over the same region (the bank is the lane: the
// data-dependent bin of an atomic never conflicts; the [lane][17] rows did - SQ_LDS_BANK_CONFLICT was 1.4 x
// SQ_ACTIVE_INST_LDS in k_step_quiet<0>, round 3).  SAG_LIDAR_TILE=0: the row layout (A/B).
#ifndef SAG_LIDAR_TILE
#define SAG_LIDAR_TILE 1
#endif
#define TILE(b) lds[STG_BASE + (b) * WAVE + lane]

// tstate bits 17..27: body k has non-zero velocity or may overlap something (derived at
// install); lets the kernel skip the loads, pair tests and write-back of sleeping bodies.
constexpr uint32_t TS_AWAKE_SHIFT = 17, TS_AWAKE_BITS = (1u << NBODY) - 1;

// per-lane view of the dynamic part of the free bodies
struct Dyn {
  float* lds; float* S; float* G; int lane, N, i;
  uint32_t dmap;  // 2 bits per body: 0 = no pool entry, 1..DPOOL
  uint32_t ovf;   // bodies whose velocity/acceleration live in global memory
  int nd;
  __device__ int vel_field(int k) const { return DV_VEL + 3 * k; }  // device index
  __device__ int slot(int k) const { return (int)(dmap >> (2 * k) & 3u) - 1; }
  __device__ bool has(int k) const { return (dmap >> (2 * k) & 3u) || (ovf >> k & 1u); }
  // give body k a dynamic entry (zero velocity unless `from_state`, zero acceleration)
  __device__ void ensure(int k, bool from_state) {
    if (has(k)) return;
    float vx = 0, vy = 0, w = 0;
    if (nd < DPOOL) {
      if (from_state) { const int f = vel_field(k); vx = SD(f); vy = SD(f + 1); w = SD(f + 2); }
      const int d = nd++;
      dmap |= (uint32_t)(d + 1) << (2 * k);
      POOL(d, 0) = vx; POOL(d, 1) = vy; POOL(d, 2) = w; POOL(d, 3) = 0; POOL(d, 4) = 0; POOL(d, 5) = 0;
    } else {
      ovf |= 1u << k;  // velocity stays in its SoA home, acceleration in the spill array
      if (!from_state) { const int f = vel_field(k); SD(f) = 0; SD(f + 1) = 0; SD(f + 2) = 0; }
      for (int c = 0; c < 3; c++) G[((size_t)c * NBODY + k) * N + i] = 0;
    }
  }
  __device__ void get(int k, float& vx, float& vy, float& w, float& ax, float& ay, float& aw) const {
    const int d = slot(k);
    if (d >= 0) { vx = POOL(d, 0); vy = POOL(d, 1); w = POOL(d, 2); ax = POOL(d, 3); ay = POOL(d, 4); aw = POOL(d, 5); }
    else if (ovf >> k & 1u) {
      const int f = vel_field(k);
      vx = SD(f); vy = SD(f + 1); w = SD(f + 2);
      ax = G[((size_t)0 * NBODY + k) * N + i]; ay = G[((size_t)1 * NBODY + k) * N + i]; aw = G[((size_t)2 * NBODY + k) * N + i];
    } else { vx = vy = w = ax = ay = aw = 0; }
  }
  __device__ void set_acc(int k, float ax, float ay, float aw) {
    const int d = slot(k);
    if (d >= 0) { POOL(d, 3) = ax; POOL(d, 4) = ay; POOL(d, 5) = aw; }
    else { G[((size_t)0 * NBODY + k) * N + i] = ax; G[((size_t)1 * NBODY + k) * N + i] = ay; G[((size_t)2 * NBODY + k) * N + i] = aw; }
  }
  __device__ void set_vel(int k, float vx, float vy, float w) {
    const int d = slot(k);
    if (d >= 0) { POOL(d, 0) = vx; POOL(d, 1) = vy; POOL(d, 2) = w; }
    else { const int f = vel_field(k); SD(f) = vx; SD(f + 1) = vy; SD(f + 2) = w; }
  }
};

// constants of a free body: shape code, inverse inertia (world frame), friction model
struct BodyK { int sh; float m, I, reff; };

// lidar: fp32 estimate of (bin, alias); the fp64 evaluation (the reference's arithmetic,
// safe_adaptation_gym.py:208-216) is redone whenever the estimate is within 2e-5 bins of a
// bin boundary, so the bin index is always the fp64 one.
struct LidarHit { int bin; float alias, sensor; };

__device__ __attribute__((noinline)) LidarHit lidar_exact(float rxf, float ryf, float yawf, float px, float py) {
  const double rx = rxf, ry = ryf, cd = cos((double)yawf), sd = sin((double)yawf);
  const double W0 = (double)px - rx, W1 = (double)py - ry;
  const double EX = W0 * cd + W1 * sd, EY = W0 * -sd + W1 * cd;
  const double two_pi = PI_D * 2, bin_size = two_pi / SAG_LIDAR_BINS;
  double a = atan2(EY, EX);
  if (a < 0) a += two_pi;
  LidarHit h;
  h.bin = a >= 0 ? (int)(a / bin_size) : 0;   // (NaN state, PhysicsError branch: bin 0, no conversion of a NaN)
  if (h.bin >= SAG_LIDAR_BINS) h.bin -= SAG_LIDAR_BINS;
  h.alias = (float)((a - bin_size * h.bin) / bin_size);
  const double D = hypot(EX, EY);
  h.sensor = (float)((5.0 - D > 0 ? 5.0 - D : 0.0) / 5.0);
  return h;
}

// angle of (ex, ey) in lidar bins, [0, 16]: octant reduction + odd minimax polynomial for
// atan(q) 8/pi on [0, 1] (|error| < 1e-6 bins in fp32 arithmetic, tools/fit_atan_bins.py); a quarter
// of the instructions of atan2f, whose special cases (inf, nan, signed zeros) cannot occur or
// are sent to the fp64 evaluation by the caller (0/0 -> NaN fails the caller's band test)
__device__ inline float angle_bins(float ex, float ey) {
  const float ax = fabsf(ex), ay = fabsf(ey);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float q = mn * __builtin_amdgcn_rcpf(mx), z = q * q;
  // (explicit fused multiply-adds: the translation unit is built with -ffp-contract=off, and this estimate - the
  // step's most executed arithmetic, ~20 points per env - only has to stay inside the band the caller re-checks in fp64)
  float t = -1.017929272e-02f;
  t = __builtin_fmaf(t, z, 5.515013699e-02f);
  t = __builtin_fmaf(t, z, -1.416326720e-01f);
  t = __builtin_fmaf(t, z, 2.449992122e-01f);
  t = __builtin_fmaf(t, z, -3.539759922e-01f);
  t = __builtin_fmaf(t, z, 5.078958901e-01f);
  t = __builtin_fmaf(t, z, -8.487346847e-01f);
  t = __builtin_fmaf(t, z, 2.546477322e+00f);
  t *= q;
  if (ay > ax) t = 4.0f - t;
  if (ex < 0) t = 8.0f - t;
  if (ey < 0) t = 16.0f - t;
  return t;
}

template <int STG_STRIDE>
__device__ inline void lidar_point(float* lds, int lane, float rxf, float ryf, float yawf, float cf, float sf,
                                   float px, float py) {
  const float w0 = px - rxf, w1 = py - ryf;
  const float ex = __builtin_fmaf(w0, cf, w1 * sf), ey = __builtin_fmaf(w1, cf, -(w0 * sf));
  const float dist = __builtin_amdgcn_sqrtf(__builtin_fmaf(ex, ex, ey * ey));   // 1 ulp: feeds the closeness value only
  const float t = angle_bins(ex, ey);
  // NaN (the point is the robot's own position: 0 * rcp(0)), or a non-finite state: no float -> int
  // conversion of it (undefined in C++ and poison in LLVM, whatever v_cvt_i32_f32 does with it); bin 16
  // sends the point to the fp64 evaluation below
  const bool t_ok = t >= 0.0f && t <= 16.0f;
  int bin = t_ok ? (int)t : 16;
  float alias = t - (float)bin;
  float sensor = fmaxf(5.0f - dist, 0.0f) * 0.2f;
  // the fp32 estimate is off by < 4e-6 bins + 1e-6 / dist (rounding of the relative position);
  // inside that band of a bin boundary fp64 decides
  const float edge = fminf(alias, 1.0f - alias);
  if (!(edge * dist >= __builtin_fmaf(2e-5f, dist, 1.5e-6f)) || bin > 15) {
    const LidarHit h = lidar_exact(rxf, ryf, yawf, px, py);
    bin = h.bin; alias = h.alias; sensor = h.sensor;
  }
  const int bp = (bin + 1) & 15, bm = (bin + 15) & 15;
  // closeness values are >= +0, so their bit patterns order like the floats: LDS integer
  // atomic max (ds_max_i32, no return value) replaces read-max-write and its round trips
#if SAG_LIDAR_TILE
  int* o = reinterpret_cast<int*>(&TILE(0));
  atomicMax(o + bin * WAVE, __float_as_int(sensor));
  atomicMax(o + bp * WAVE, __float_as_int(alias * sensor));
  atomicMax(o + bm * WAVE, __float_as_int((1.0f - alias) * sensor));
#else
  int* o = reinterpret_cast<int*>(&STG(0));
  atomicMax(o + bin, __float_as_int(sensor));
  atomicMax(o + bp, __float_as_int(alias * sensor));
  atomicMax(o + bm, __float_as_int((1.0f - alias) * sensor));
#endif
}

#ifndef SAG_STEP_MIN_WAVES
#define SAG_STEP_MIN_WAVES 2  // 245 VGPRs: 8 waves per CU; forcing 3 per SIMD spills (measured slower)
#endif
#ifndef SAG_LDS_PAD
#define SAG_LDS_PAD 0
#endif
// HAS_BTN / HAS_TBOX: compile-time knowledge that the context holds no buttons / no task object
// (capacities of sag_create); the specialised instances drop those arrays, loops and the box
// shapes altogether.  <true, true> serves mixed (multitask) batches.
// MODE: how a launch covers the batch.
//  ALL    every env of the wavefront's 64-env slice, full physics (single-launch form);
//  QUIET  the envs whose `busy` bit is clear: nothing can touch the robot or move during this
//         step (classified conservatively at the end of the previous step), so all contact code
//         is compiled out: small, high-occupancy, close to the memory roofline;
//  BUSY   the envs whose bit is set, compacted (k_compact) into full wavefronts, full physics.  QUIET + BUSY together do exactly what ALL does (tests compare them).
// Doggo lidar: the base is tilted, so e = (d @ R)[:2] with d = [p_xy, 0] - robot_xpos
// (safe_adaptation_gym.py:197-216) has a -z R[2,:2] term; evaluated in fp64 as the reference does
template <int STG_STRIDE>
__device__ inline void lidar_point_tilted(float* lds, int lane, const double* pos, const double* Rm, float px, float py) {
  const double dx = (double)px - pos[0], dy = (double)py - pos[1], dz = -pos[2];
  const double EX = Rm[0] * dx + Rm[3] * dy + Rm[6] * dz;
  const double EY = Rm[1] * dx + Rm[4] * dy + Rm[7] * dz;
  const double two_pi = PI_D * 2, bin_size = two_pi / SAG_LIDAR_BINS;
  double a = atan2(EY, EX);
  if (a < 0) a += two_pi;
  int bin = a >= 0 ? (int)(a / bin_size) : 0;
  if (bin >= SAG_LIDAR_BINS) bin -= SAG_LIDAR_BINS;
  const float alias = (float)((a - bin_size * bin) / bin_size);
  const double Dd = hypot(EX, EY);
  const float sensor = (float)((5.0 - Dd > 0 ? 5.0 - Dd : 0.0) / 5.0);
  const int bp = (bin + 1) & 15, bm = (bin + 15) & 15;
#if SAG_LIDAR_TILE
  int* o = reinterpret_cast<int*>(&TILE(0));
  atomicMax(o + bin * WAVE, __float_as_int(sensor));
  atomicMax(o + bp * WAVE, __float_as_int(alias * sensor));
  atomicMax(o + bm * WAVE, __float_as_int((1.0f - alias) * sensor));
#else
  int* o = reinterpret_cast<int*>(&STG(0));
  atomicMax(o + bin, __float_as_int(sensor));
  atomicMax(o + bp, __float_as_int(alias * sensor));
  atomicMax(o + bm, __float_as_int((1.0f - alias) * sensor));
#endif
}

// ---------------------------------------------------------------------------
// Doggo: the physics is k_doggo_physics (sag_doggo_coop.hpp: specification oracle/sag_oracle_doggo.inc +
// world_forward order); the generic step post-processes its per-env result block (DgResult).
// ---------------------------------------------------------------------------
struct DgResult {
  double qacc_lin[3], touch[8], comvel[4];
  int cost_contacts;
  uint32_t btn_mask;
  int overflow;   // a constraint row did not fit into DG_MAXROWS during this step
};

// floor friction + semi-implicit Euler + rest capture of one planar free body (the block of
// step_body, on the [x y yaw vx vy w ax ay aw] layout)
__device__ __attribute__((noinline)) void dg_free_body_finish(float* B, bool is_task_obj, const BodyK& vk,
                                                              const BodyK& bk, float bc, float h) {
  float vx_ = B[3], vy_ = B[4], w_ = B[5], ax_ = B[6], ay_ = B[7], aw_ = B[8];
  if (!is_task_obj || bk.sh == SH_BOX) {
    const float mm = !is_task_obj ? vk.m : bk.m, II = !is_task_obj ? vk.I : bk.I, rr = !is_task_obj ? vk.reff : bk.reff;
    const float fmax_ = MU * GRAV * mm;
    float fx = -SOL_D0 * mm * (bc * vx_ + ax_), fy = -SOL_D0 * mm * (bc * vy_ + ay_);
    const float f2 = fx * fx + fy * fy;
    if (f2 > fmax_ * fmax_) { const float sc = fmax_ / sqrtf(f2); fx *= sc; fy *= sc; }
    ax_ += fx / mm; ay_ += fy / mm;
    const float t = clampf(-SOL_D0 * II * (bc * w_ + aw_), -fmax_ * rr, fmax_ * rr);
    aw_ += t / II;
  } else {
    float c, s; sincosf(B[2], &s, &c);
    float bvx = c * vx_ + s * vy_, bvy = -s * vx_ + c * vy_, bax = c * ax_ + s * ay_, bay = -s * ax_ + c * ay_;
    const float mg = bk.m * GRAV;
    float tlim;
    if (bk.sh == SH_ROD) {
      const float mx = 1.5f * bk.m, lx = 0.05f * mg / 0.08f, ly = 1.2f * mg;
      const float fx = clampf(-SOL_D0 * mx * (bc * bvx + bax), -lx, lx);
      const float fy = clampf(-SOL_D0 * bk.m * (bc * bvy + bay), -ly, ly);
      bax += fx / mx; bay += fy / bk.m;
      tlim = 1.2f * mg * 0.15f;
    } else {
      const float me = 1.4f * bk.m, lim = 0.05f * mg / 0.14f;
      float fx = -SOL_D0 * me * (bc * bvx + bax), fy = -SOL_D0 * me * (bc * bvy + bay);
      const float f2 = fx * fx + fy * fy;
      if (f2 > lim * lim) { const float sc = lim / sqrtf(f2); fx *= sc; fy *= sc; }
      bax += fx / me; bay += fy / me;
      tlim = 0.003f * mg;
    }
    ax_ = c * bax - s * bay; ay_ = s * bax + c * bay;
    aw_ += clampf(-SOL_D0 * bk.I * (bc * w_ + aw_), -tlim, tlim) / bk.I;
  }
  vx_ += h * ax_; vy_ += h * ay_; w_ += h * aw_;
  if (fabsf(vx_) < REST_V && fabsf(vy_) < REST_V && fabsf(w_) < REST_W && fabsf(h * ax_) < REST_V &&
      fabsf(h * ay_) < REST_V && fabsf(h * aw_) < REST_W) { vx_ = 0; vy_ = 0; w_ = 0; }
  B[3] = vx_; B[4] = vy_; B[5] = w_;
  B[0] += h * vx_; B[1] += h * vy_; B[2] += h * w_;
}


// MODE_POST: Doggo only - the generic step after the wave-cooperative physics kernel (results in StepArgs::DR):
// no physics code in the instance at all
enum { MODE_ALL = 0, MODE_QUIET = 1, MODE_BUSY = 2, MODE_POST = 3 };
// two copies of the busy bit, used alternately (StepArgs::phase): a launch reads bit 28 + phase and
// writes bit 28 + (phase ^ 1), so the QUIET launch cannot re-flag an env for the BUSY launch of
// the same step
constexpr uint32_t TS_BUSY_BIT = 1u << 28;
// The KIND of a busy env = which walks of the contact code its robot is within reach of - bit 0 the task object, bit 1 a static
// circle (pillar / button), bit 2 a vase or any body still moving - as a byte per env (StepArgs::kind, written with the NEXT step's
// busy bit).  A hint for the order of the busy list only: a busy wavefront executes every loop that ANY of its 64 lanes takes, in every
// substep, so wavefronts of one kind each skip the other kinds' walks (Car / push_box, 4 M envs: mean busy wavefront 416 -> 275 us
// when the list is sorted by bits 0 and 1, tools/busy_sort_probe.py).  The tests are the busy classification's own (reach bounds
// that hold for the whole step), so a wavefront of a kind can never meet a walk its kind excludes; results never depend on the kind.
constexpr int BUSY_CLASSES = 8;
// launch order of the classes, most expensive walks first (the cheap wavefronts then fill the last round)
__host__ __device__ constexpr int busy_class_order(int k) {
  constexpr int order[BUSY_CLASSES] = {7, 5, 3, 1, 6, 4, 2, 0};
  return order[k];
}

// Car: 1 / inertia of the spinning parts (joint damping implicit: I + h d); per floor-friction element (L / R the
// wheels' rolling directions, T their merged lateral element, X / Y the caster's: oracle car_floor_friction) the
// regulariser R = A (1 - d0) / d0 of A = J M^-1 J^T + r_w^2 / I_spin, 1 / (A + R), the weight 1 / D = 1 / (r_w^2 / I_spin + R)
// of the eliminated element, and the friction-augmented inertia K = M + sum J^T J / D of the direct solve.  Body
// frame: constants up to the h-dependent spin terms.
inline CarFricK car_fric_constants(float h) {
  CarFricK k;
  constexpr float CM = CAR_M, CIO = CAR_IO, COY = CAR_OY, CRW = 0.05f, AB = -CM * COY;
  constexpr float IDC = 1.0f / (CM * (CM * CIO - AB * AB));
  constexpr float M0 = (CM * CIO) * IDC, M2 = (-AB * CM) * IDC, M3 = (CM * CIO - AB * AB) * IDC, M5 = (CM * CM) * IDC;
  k.iIw = 1.0f / (CAR_IW + h * 0.001f); k.iIb = 1.0f / (CAR_IB + h * 0.001f);
  const float spw = CRW * CRW * k.iIw, spb = CRW * CRW * k.iIb;
  // d = body y at lever (rx, .): J = (0, 1, rx) -> A = M3 + M5 rx^2;  d = body x at (., ry): J = (1, 0, -ry) -> A = M0 - 2 M2 ry + M5 ry^2
  const float A_L = M3 + M5 * 0.13f * 0.13f + spw, A_T = M0 - 2 * M2 * 0.1f + M5 * 0.1f * 0.1f;
  const float A_X = M0 + 2 * M2 * 0.1f + M5 * 0.1f * 0.1f + spb, A_Y = M3 + spb;
  constexpr float RF = (1.0f - SOL_D0) / SOL_D0;
  k.R_L = A_L * RF; k.R_T = 0.5f * A_T * RF; k.R_X = A_X * RF; k.R_Y = A_Y * RF;   // T: two equal elements merged
  k.iA_L = 1.0f / (A_L + k.R_L); k.iA_T = 1.0f / (A_T + k.R_T); k.iA_X = 1.0f / (A_X + k.R_X); k.iA_Y = 1.0f / (A_Y + k.R_Y);
  k.w_L = 1.0f / (spw + k.R_L); k.w_T = 1.0f / k.R_T; k.w_X = 1.0f / (spb + k.R_X); k.w_Y = 1.0f / (spb + k.R_Y);
  k.K00 = CM + k.w_T + k.w_X; k.K02 = AB - 0.1f * k.w_T + 0.1f * k.w_X;
  k.K22 = CIO + 0.13f * 0.13f * (k.w_L + k.w_L) + 0.1f * 0.1f * k.w_T + 0.1f * 0.1f * k.w_X;
  k.idk = 1.0f / (k.K00 * k.K22 - k.K02 * k.K02);
  k.iK11 = 1.0f / (CM + k.w_L + k.w_L + k.w_Y);
  return k;
}

template <int ROBOT, bool HAS_BTN, bool HAS_TBOX, int MODE>
__device__ __forceinline__ void step_body(const StepArgs& p, float* lds, const int lane, const int i,
                                          const bool live, const int base_env, const int nvalid,
                                          const uint64_t skip_mask, const int* rows) {
  constexpr int NSTAT = SAG_MAX_PILLARS + (HAS_BTN ? SAG_MAX_BUTTONS : 0);
  constexpr bool CAR = ROBOT == SAG_ROBOT_CAR, DOGGO = ROBOT == SAG_ROBOT_DOGGO;
  static_assert(!DOGGO || MODE == MODE_ALL || MODE == MODE_POST, "Doggo runs the single-launch form");
  static_assert(DOGGO || MODE != MODE_POST, "MODE_POST is the Doggo post kernel");
  constexpr bool QUIET = MODE == MODE_QUIET;
  constexpr bool PARK = CAR && !QUIET && CAR_PARK_SLOTS > 0;   // (CPK slots above)
  constexpr int SH_ME = CAR ? SH_CAR : SH_ROBOT;
  constexpr int OBS_DIM = DOGGO ? 104 : (CAR ? 72 : 60);
  // sensor columns per staged chunk: Doggo's 56 go out as two chunks of 28
  constexpr int NSENS = DOGGO ? 28 : OBS_DIM - 48, NCHUNK = DOGGO ? 5 : 4;
  constexpr int STG_STRIDE = DOGGO ? 29 : (CAR ? 25 : 17);
  static_assert(STG_BASE + WAVE * STG_STRIDE <= LDS_FLOATS, "staging tile must fit");
  CYC_DECL
  const float my_bound = shape_bound(SH_ME, 0.f, 0.f);
  const int N = p.N;
  float* __restrict__ S = p.S;
  int32_t* __restrict__ I = p.I;
  const int capV = p.max_vases, capH = p.max_hazards, capP = p.max_pillars, capB = HAS_BTN ? p.max_buttons : 0;

  // ---- issue every load up front (bounds are context capacities, not per-env counts, so
  //      nothing waits on the meta word) ------------------------------------------------------
  // vector loads: one int4 and a handful of float4 per env (device layout: didx / iaddr)
  const float4* __restrict__ S4 = reinterpret_cast<const float4*>(S);
  auto G4 = [&](int g) { return S4[(size_t)g * N + i]; };
  const bool use_hot = MODE == MODE_BUSY && p.hot != nullptr;
  const float4* __restrict__ H4 = use_hot ? reinterpret_cast<const float4*>(p.hot) + (size_t)i * HOT_GROUPS : nullptr;
  // group g of this env: from the hot record (busy kernel; g < 14 covers 0-4 and the positions at
  // 5 + (g - DG_POS)) or from the group-major state
  auto GH = [&](int g, int hslot) { return use_hot ? H4[hslot] : G4(g); };
  int4 iw;
  uint32_t tstate;
  if (use_hot) {
    const float4 hi = H4[14], ht = H4[15];
    iw = make_int4(__float_as_int(hi.x), __float_as_int(hi.y), __float_as_int(hi.z), __float_as_int(hi.w));
    tstate = (uint32_t)__float_as_int(ht.x);
  } else {
    iw = reinterpret_cast<const int4*>(I + ipad((size_t)N))[i];
    tstate = (uint32_t)I[iaddr(DI_TSTATE, (size_t)N, (size_t)i)];
  }
  const uint32_t meta = (uint32_t)iw.x;
  int step = iw.y;
  const uint32_t env_id = (uint32_t)iw.z;
  const float4 gA = GH(0, 0), gB = GH(1, 1), gC = GH(2, 2), gD = GH(3, 3);
  BV R;
  R.x = gA.x; R.y = gA.y;
  float yaw = gA.z;
  R.vx = gA.w; R.vy = gB.x; R.w = gB.y;
  R.ax = R.ay = R.aw = 0; R.dyn = 1;
  const float gear = gC.y, damp = gC.z;
  const float vsz = gD.z, psz = gD.w;
  // car: wheel rates L, R; rear ball rate x, y, z (base axes); ball quaternion w, x, y, z
  float ext[9] = {0, 0, 0, 0, 0, 1, 0, 0, 0}, eacc[5] = {0, 0, 0, 0, 0};
  // (whole float4 groups: nine dword accesses at a 16-byte stride never cover a line, and the partial lines of the
  // write-back doubled the Car step's measured HBM traffic)
  constexpr int DG_EXT = (152 >> 2);
  static_assert(didx(SAG_F_ROBOT_EXT) == 152 && didx(SAG_F_ROBOT_EXT + 8) == 160, "car extension = device groups 38..40");
  float4 ext_tail = make_float4(0.f, 0.f, 0.f, 0.f);   // group 40: ext[8] and three floats the Car does not use (kept as they are)
  if constexpr (CAR) {
    const float4 e0 = G4(DG_EXT), e1 = G4(DG_EXT + 1);
    ext_tail = G4(DG_EXT + 2);
    ext[0] = e0.x; ext[1] = e0.y; ext[2] = e0.z; ext[3] = e0.w; ext[4] = e1.x; ext[5] = e1.y; ext[6] = e1.z; ext[7] = e1.w;
    ext[8] = ext_tail.x;
    if constexpr (PARK) {   // the quaternion and the three pass-through floats of its group leave the registers until they are needed
      CPK(CP_Q) = ext[5]; CPK(CP_Q + 1) = ext[6]; CPK(CP_Q + 2) = ext[7]; CPK(CP_Q + 3) = ext[8];
      CPK(CP_TAIL) = ext_tail.y; CPK(CP_TAIL + 1) = ext_tail.z; CPK(CP_TAIL + 2) = ext_tail.w;
    }
  }
  float goalx = gB.z, goaly = gB.w;
  float last0 = gC.x;
  float a0 = 0, a1 = 0, n0 = 0, n1 = 0;
  if (!p.observe_only) {
    if constexpr (!DOGGO) {
      const float2 a = reinterpret_cast<const float2*>(p.actions)[i];
      a0 = a.x; a1 = a.y;
      if (p.noise) { const float2 z = reinterpret_cast<const float2*>(p.noise)[i]; n0 = z.x; n1 = z.y; }
    }
  }
  {
    // positions (x y yaw) of the 11 free bodies: 9 consecutive groups of the device record
    float vpos[36];
    const int need = HAS_TBOX && p.has_box ? 9 : (3 * capV + 3) / 4;
#pragma unroll
    for (int g = 0; g < 9; g++) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g < need) v = GH(DG_POS + g, 5 + g);
      vpos[4 * g] = v.x; vpos[4 * g + 1] = v.y; vpos[4 * g + 2] = v.z; vpos[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < SAG_MAX_VASES; k++)
      if (k >= capV) { vpos[3 * k] = vpos[3 * k + 1] = vpos[3 * k + 2] = 0; }
    if (!(HAS_TBOX && p.has_box)) { vpos[3 * BOX_ID] = vpos[3 * BOX_ID + 1] = vpos[3 * BOX_ID + 2] = 0; }
#pragma unroll
    for (int k = 0; k < NBODY; k++) { LP(LS_X, k) = vpos[3 * k]; LP(LS_Y, k) = vpos[3 * k + 1]; LP(LS_YAW, k) = vpos[3 * k + 2]; }
  }
  // static colliders: pillars then buttons, in registers
  float stx[NSTAT], sty[NSTAT];
  {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (capP > 0) v = GH(DG_PILLARS, 4);
    stx[0] = v.x; sty[0] = v.y; stx[1] = capP > 1 ? v.z : 0.f; sty[1] = capP > 1 ? v.w : 0.f;
    static_assert(SAG_MAX_PILLARS == 2, "pillars fill one group");
  }
  if constexpr (HAS_BTN) {
    float bt[12];
#pragma unroll
    for (int g = 0; g < 3; g++) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (2 * g < capB) v = G4(DG_BUTTONS + g);
      bt[4 * g] = v.x; bt[4 * g + 1] = v.y; bt[4 * g + 2] = v.z; bt[4 * g + 3] = v.w;
    }
    static_assert(SAG_MAX_BUTTONS == 6, "buttons fill three groups");
#pragma unroll
    for (int k = 0; k < SAG_MAX_BUTTONS; k++) {
      stx[SAG_MAX_PILLARS + k] = k < capB ? bt[2 * k] : 0.f;
      sty[SAG_MAX_PILLARS + k] = k < capB ? bt[2 * k + 1] : 0.f;
    }
  }
  const int task = meta & 15, nH = meta >> 4 & 15, nV = meta >> 8 & 15, nP = meta >> 12 & 3,
            nB = HAS_BTN ? (meta >> 14 & 7) : 0, box_kind = HAS_TBOX ? (meta >> 17 & 3) : 0;
  const bool has_box = HAS_TBOX && box_kind != SAG_BOX_NONE;
  // body index k is the task object: never, at compile time, in an instance without one, so the shape
  // switches, geom loops and mass selects of the contact code fold to the vase case there
  auto is_box = [](int k) { return HAS_TBOX && k == BOX_ID; };
  int flags = 0;

  Rng rng;
  rng.tape = p.tape ? p.tape + (size_t)i * p.tape_len : nullptr;
  rng.len = p.tape_len; rng.pos = 0; rng.exhausted = 0;
  rng.k0 = p.key0; rng.k1 = p.key1; rng.env = env_id; rng.step = (uint32_t)step;
  rng.ep4 = ((uint32_t)iw.w >> FLAG_EPISODE_SHIFT) << 2;

  // ---- action noise + clip (safe_adaptation_gym.py:58-67) ---------------------
  float ctrl0 = 0, ctrl1 = 0;
  float ctrl12[DOGGO ? 12 : 1] = {};
  if constexpr (DOGGO) {
    if (!p.observe_only) {
      const float an = SF(SAG_F_ACTION_NOISE);
#pragma unroll 1
      for (int j = 0; j < 12; j += 2) {
        float z0, z1;
        if (p.noise) { z0 = p.noise[(size_t)i * 12 + j]; z1 = p.noise[(size_t)i * 12 + j + 1]; }
        else {
          uint32_t c[4] = {rng.env, rng.step, (uint32_t)(j >> 1), rng.ep4 | 1u};
          philox4x32_10(c, p.key0, p.key1);
          box_muller(c[0], c[1], z0, z1);
        }
        const float l0 = SF(SAG_F_CTRL_SCALE + j), l1 = SF(SAG_F_CTRL_SCALE + j + 1);
        ctrl12[j] = clampf(p.actions[(size_t)i * 12 + j] + an * z0, -l0, l0);
        ctrl12[j + 1] = clampf(p.actions[(size_t)i * 12 + j + 1] + an * z1, -l1, l1);
      }
    }
  }
  if (!p.observe_only) {
    if (!p.noise && !DOGGO) {
      uint32_t c[4] = {rng.env, rng.step, 0u, rng.ep4 | 1u};
      philox4x32_10(c, p.key0, p.key1);
      box_muller(c[0], c[1], n0, n1);
    }
    const float an = gC.w;
    const float cs0 = gD.x, cs1 = gD.y;
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

  // ---- free-body constants ---------------------------------------------------------
  const float h = p.h;
  const float tc = fmaxf(0.02f, 2.0f * h);
  Sol sol0; sol0.bcoef = 2.0f / (SOL_D1 * tc); sol0.kcoef = 1.0f / (SOL_D1 * SOL_D1 * tc * tc); sol0.mu = MU;
  Sol solb = sol0;  // pairs that involve the task object (priority-1 geoms of rod / ball win)
  if (box_kind == SAG_BOX_ROD || box_kind == SAG_BOX_BALL) solb.mu = 1.2f;
  if (box_kind == SAG_BOX_BALL) {
    const float tcb = fmaxf(0.018f, 2.0f * h);
    solb.bcoef = 2.0f / (SOL_D1 * tcb); solb.kcoef = 1.0f / (SOL_D1 * SOL_D1 * tcb * tcb * 0.2f * 0.2f);
  }
  BodyK vk, bk;
  vk.sh = SH_VASE; vk.m = VASE_DENSITY * 8 * vsz * vsz * vsz; vk.I = vk.m * (8 * vsz * vsz) / 12;
  vk.reff = vsz * 1.41421356237309504880f;
  bk.sh = SH_BOX; bk.m = 1; bk.I = 1; bk.reff = 0.2f * 1.41421356237309504880f;
  if (box_kind == SAG_BOX_BOX) {
    const float m0 = 0.001f * 0.064f, m1 = 0.001f * 0.016f;
    bk.m = m0 + 4 * m1; bk.I = m0 * 0.32f / 12 + 4 * (m1 * 0.08f / 12 + m1 * 0.08f);
  } else if (box_kind == SAG_BOX_ROD) {
    bk.sh = SH_ROD; bk.m = 0.0005f * (3.14159265358979323846f * 0.08f * 0.08f * 0.6f);
    bk.I = bk.m * (3 * 0.08f * 0.08f + 0.36f) / 12;
  } else if (box_kind == SAG_BOX_BALL) {
    bk.sh = SH_BALL; bk.m = 0.0005f * (4.0f / 3.0f * 3.14159265358979323846f * 0.14f * 0.14f * 0.14f);
    bk.I = 0.4f * bk.m * 0.14f * 0.14f;
  }
  const float vase_r = vk.reff, box_r = shape_bound(bk.sh, vsz, 0);
  // body view of free body k (positions from LDS, dynamics from the pool), its cos/sin
  auto load_body = [&](const Dyn& dy, int k, BV& V, float& c, float& s) {
    V.x = LP(LS_X, k); V.y = LP(LS_Y, k);
    dy.get(k, V.vx, V.vy, V.w, V.ax, V.ay, V.aw);
    sincosf(LP(LS_YAW, k), &s, &c);
    V.dyn = 1; V.m1 = 0; V.m2 = 0; V.m4 = 0;
    if (!is_box(k)) { V.m0 = V.m3 = 1.0f / vk.m; V.m5 = 1.0f / vk.I; }
    else {
      V.m5 = 1.0f / bk.I;
      if (bk.sh == SH_ROD) {  // rolls along local x (1.5 m), slides along its axis (m)
        const float ix = 1.0f / (1.5f * bk.m), iy = 1.0f / bk.m;
        V.m0 = c * c * ix + s * s * iy; V.m1 = c * s * (ix - iy); V.m3 = s * s * ix + c * c * iy;
      } else if (bk.sh == SH_BALL) { V.m0 = V.m3 = 1.0f / (1.4f * bk.m); }
      else { V.m0 = V.m3 = 1.0f / bk.m; }
    }
  };

  // ---- physics: nstep x (forward, integrate) + one forward at the final state ---
  // padding lanes (they mirror another env so that loads stay in bounds) run no physics at all: they
  // would widen the divergence union and, when the dynamic pool overflows, race on that env's state
  const int nsub = !live ? -1 : (p.observe_only ? 0 : (ABL(ABL_NSUB1) ? 1 : p.nstep));
  // statics occupy [0, capP) and [SAG_MAX_PILLARS, SAG_MAX_PILLARS + capB) of stx/sty
  const int n_static = capB ? SAG_MAX_PILLARS + capB : capP;
  const uint32_t fmask = ((1u << nV) - 1) | (has_box ? 1u << BOX_ID : 0u);
  uint32_t awake = (tstate >> TS_AWAKE_SHIFT) & fmask;
  Dyn dy; dy.lds = lds; dy.S = S; dy.G = p.G; dy.lane = lane; dy.N = N; dy.i = i; dy.dmap = 0; dy.ovf = 0; dy.nd = 0;
  if constexpr (QUIET) awake = 0;  // a quiet env has no moving body (classification)
  for (uint32_t m = awake; m; m &= m - 1) dy.ensure(__ffs(m) - 1, true);
  uint32_t dirty = 0;  // bodies whose state changed during this step
  int cost_contacts = 0;
  uint32_t btn_mask = 0;
  float cy = 1, sy = 0, yaw_turn = 0;
#ifdef SAG_WAVE_TIMES
  unsigned wt_work[8] = {};   // robot-static hits, robot-vase hits, robot-object hits, free-static hits, free-free pairs, integrated bodies, substeps with any, -
  unsigned wt_trips[8] = {};  // the same loops, passes of the wavefront
  unsigned wt_prev[8] = {};
#endif

  CYC(CY_LOAD);
  // Doggo: the physics kernel has run; state in private memory for the sensors
  DgState dgs;
  DgResult dgr;
  double dg_rot[DOGGO ? 9 : 1] = {};  // base rotation matrix (lidar, sensors)
  if constexpr (DOGGO) {
    DgWorld Wd;
    dg_load(dgs, S, (size_t)N, (size_t)i);
    for (int k = 0; k < NBODY; k++) {
      const bool isb = k == BOX_ID;
      const bool on = isb ? has_box : k < nV;
      const int f = DV_VEL + 3 * k;
      Wd.fb[k][0] = LP(LS_X, k); Wd.fb[k][1] = LP(LS_Y, k); Wd.fb[k][2] = LP(LS_YAW, k);
      Wd.fb[k][3] = on ? SD(f) : 0.f; Wd.fb[k][4] = on ? SD(f + 1) : 0.f; Wd.fb[k][5] = on ? SD(f + 2) : 0.f;
      for (int c = 0; c < 6; c++) Wd.minv[k][c] = 0;
      if (!isb) { Wd.minv[k][0] = Wd.minv[k][3] = 1.0f / vk.m; Wd.minv[k][5] = 1.0f / vk.I; }
      else {
        Wd.minv[k][5] = 1.0f / bk.I;
        Wd.minv[k][0] = Wd.minv[k][3] = bk.sh == SH_BALL ? 1.0f / (1.4f * bk.m) : 1.0f / bk.m;  // rod: per substep
      }
    }
    for (int k = 0; k < 3; k++) dgr.qacc_lin[k] = 0;
    for (int k = 0; k < 8; k++) dgr.touch[k] = 0;
    for (int k = 0; k < 4; k++) dgr.comvel[k] = 1;
    dgr.cost_contacts = 0; dgr.btn_mask = 0; dgr.overflow = 0;
    constexpr bool post = MODE == MODE_POST;   // cooperative form: k_doggo_physics has run, its results are in DR and S
    if constexpr (post) {
      const double* dr = p.DR + (size_t)i * DR_STRIDE;
      for (int k = 0; k < 3; k++) dgr.qacc_lin[k] = dr[k];
      for (int k = 0; k < 8; k++) dgr.touch[k] = dr[3 + k];
      for (int k = 0; k < 4; k++) dgr.comvel[k] = dr[11 + k];
      dgr.cost_contacts = (int)dr[15]; dgr.btn_mask = (uint32_t)dr[16]; dgr.overflow = (int)dr[17];
    } else {
      static_assert(post, "the Doggo physics is k_doggo_physics (sag_doggo_coop.hpp); the generic step only post-processes its results");
    }
    cost_contacts = dgr.cost_contacts; btn_mask = dgr.btn_mask;
    if (dgr.overflow) flags |= 4;   // SAG_I_FLAGS bit 2: a Doggo constraint did not fit the row budget
    float wz;
    if (!p.observe_only && live && !post) dg_store(dgs, S, (size_t)N, (size_t)i, yaw, wz);
    else { double Rm[9], ww[3]; dg_quat2mat(dgs.quat, Rm); dg_matvec(Rm, dgs.wloc, ww); yaw = (float)atan2(Rm[3], Rm[0]); wz = (float)ww[2]; }
    {
      double Rm[9];
      dg_quat2mat(dgs.quat, Rm);
      for (int k = 0; k < 9; k++) dg_rot[k] = Rm[k];
    }
    R.x = (float)dgs.pos[0]; R.y = (float)dgs.pos[1]; R.vx = (float)dgs.vlin[0]; R.vy = (float)dgs.vlin[1]; R.w = wz;
    R.ax = (float)dgr.qacc_lin[0]; R.ay = (float)dgr.qacc_lin[1];
    if (nsub > 0) awake = 0;   // (a step without substeps keeps the install-time flags, SAG_I_AWAKE)
    for (int k = 0; k < NBODY; k++) {
      const bool isb = k == BOX_ID;
      if (isb ? !has_box : k >= nV) continue;
      LP(LS_X, k) = Wd.fb[k][0]; LP(LS_Y, k) = Wd.fb[k][1]; LP(LS_YAW, k) = Wd.fb[k][2];
      if (Wd.fb[k][3] != 0 || Wd.fb[k][4] != 0 || Wd.fb[k][5] != 0) awake |= 1u << k;
      if (!p.observe_only && live && !post) {
        for (int c = 0; c < 3; c++) { SD(DV_POS + 3 * k + c) = Wd.fb[k][c]; SD(DV_VEL + 3 * k + c) = Wd.fb[k][3 + c]; }
      }
    }
  } else {
  float car_f0 = 0, car_f1 = 0, car_f2 = 0, car_px = 0, car_py = 0, car_pw = 0;   // friction's share of the base acceleration (body axes); R.a after the first solve
  float car_fL = 0, car_fT = 0, car_fR = 0, car_fX = 0, car_fY = 0;               // the five forces of the previous solve (warm start)
  bool car_warm = false;
  // Car floor friction solved to convergence (SPECIFICATION: oracle car_floor_friction): given the base
  // acceleration (a0, a1, a2) from everything else, in body axes, (1) the direct solve with all five elements
  // unclamped - eliminating f_i = (c_i - J_i a) / D_i leaves K a = M a_other + sum J_i^T c_i / D_i with the constant
  // K = M + sum J_i^T J_i / D_i, block diagonal {x, yaw} + {y} - and its forces clamped to their bounds, then (2)
  // CAR_FRICTION_SWEEPS projected Gauss-Seidel sweeps over the accumulated forces (order L, T, R, X, Y).  Only the
  // first solve of an env-step does (1): every later one starts from the previous solve's forces (warm start) and
  // runs CAR_FRICTION_SWEEPS_WARM sweeps.  Writes R.a (world axes), the spin accelerations eacc and the friction's
  // share car_f* of the base acceleration.
  auto car_floor = [&](const float a0, const float a1, const float a2) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
    constexpr float CM = CAR_M, CIO = CAR_IO, COY = CAR_OY, CJD = 0.001f, CRW = 0.05f, AB = -CM * COY;
    constexpr float CNL = CM * GRAV * (0.1f + COY) / 0.4f, CNC = CM * GRAV - 2 * CNL;
    constexpr float IDC = 1.0f / (CM * (CM * CIO - AB * AB));
    constexpr float M0 = (CM * CIO) * IDC, M2 = (-AB * CM) * IDC, M3 = (CM * CIO - AB * AB) * IDC, M5 = (CM * CM) * IDC;
    // M^-1 J^T: rolling elements J = (0, 1, -+.13), lateral (1, 0, -.1), caster x (1, 0, .1), caster y (0, 1, 0)
    constexpr float UL0 = -0.13f * M2, UL2 = -0.13f * M5, UT0 = M0 - 0.1f * M2, UT2 = M2 - 0.1f * M5, UX0 = M0 + 0.1f * M2, UX2 = M2 + 0.1f * M5;
    constexpr float LIML = MU * CNL, LIMT = 2 * MU * CNL, LIMC = MU * CNC;
    const float vbx = cy * R.vx + sy * R.vy, vby = cy * R.vy - sy * R.vx, bc = sol0.bcoef;
    // spin accelerations without floor friction (caster x couples to -ball_y), reference accelerations -b slip
    const float qL = (clampf(ctrl0, -0.02f, 0.02f) - CJD * ext[0]) * p.car.iIw, qR = (clampf(ctrl1, -0.02f, 0.02f) - CJD * ext[1]) * p.car.iIw;
    const float qX = CJD * ext[3] * p.car.iIb, qY = -CJD * ext[2] * p.car.iIb;
    const float rL = -bc * (vby - 0.13f * R.w + CRW * ext[0]), rT = -bc * (vbx - 0.1f * R.w), rR = -bc * (vby + 0.13f * R.w + CRW * ext[1]);
    const float rX = -bc * (vbx + 0.1f * R.w - CRW * ext[3]), rY = -bc * (vby + CRW * ext[2]);
    const float cL = rL - CRW * qL, cR = rR - CRW * qR, cX = rX - CRW * qX, cY = rY - CRW * qY;
    float fL, fT, fR, fX, fY;
    if constexpr (PARK) { fL = CPK(CP_FL); fT = CPK(CP_FL + 1); fR = CPK(CP_FL + 2); fX = CPK(CP_FL + 3); fY = CPK(CP_FL + 4); }
    else { fL = car_fL; fT = car_fT; fR = car_fR; fX = car_fX; fY = car_fY; }
#ifdef SAG_ABL_CARFRIC   // timing-only build: no floor friction at all
    car_warm = true;
#endif
    if (!car_warm) {   // (uniform: the first solve of the step)
      const float b0 = CM * a0 + AB * a2 + p.car.w_T * rT + p.car.w_X * cX;
      const float b1 = CM * a1 + p.car.w_L * cL + p.car.w_L * cR + p.car.w_Y * cY;
      const float b2 = AB * a0 + CIO * a2 + (-0.13f * p.car.w_L) * cL + (-0.1f * p.car.w_T) * rT + (0.13f * p.car.w_L) * cR + (0.1f * p.car.w_X) * cX;
      const float n0 = (p.car.K22 * b0 - p.car.K02 * b2) * p.car.idk, n1 = b1 * p.car.iK11, n2 = (p.car.K00 * b2 - p.car.K02 * b0) * p.car.idk;
      fL = clampf(p.car.w_L * (cL - (n1 - 0.13f * n2)), -LIML, LIML); fT = clampf(p.car.w_T * (rT - (n0 - 0.1f * n2)), -LIMT, LIMT);
      fR = clampf(p.car.w_L * (cR - (n1 + 0.13f * n2)), -LIML, LIML); fX = clampf(p.car.w_X * (cX - (n0 + 0.1f * n2)), -LIMC, LIMC);
      fY = clampf(p.car.w_Y * (cY - n1), -LIMC, LIMC);
    }
#ifdef SAG_ABL_CARFRIC
    const int nsweeps = 0;
#else
    const int nsweeps = car_warm ? CAR_FRICTION_SWEEPS_WARM : CAR_FRICTION_SWEEPS;
#endif
    float g0 = UL0 * fL + UT0 * fT - UL0 * fR + UX0 * fX, g1 = M3 * fL + M3 * fR + M3 * fY, g2 = UL2 * fL + UT2 * fT - UL2 * fR + UX2 * fX;
    float sL = qL + CRW * fL * p.car.iIw, sR = qR + CRW * fR * p.car.iIw, sX = qX + CRW * fX * p.car.iIb, sY = qY + CRW * fY * p.car.iIb;
#pragma unroll 1
    for (int sw = 0; sw < nsweeps; sw++) {
      {
        const float sa = (a1 + g1) - 0.13f * (a2 + g2) + CRW * sL, fn = clampf(fL + (rL - sa - p.car.R_L * fL) * p.car.iA_L, -LIML, LIML), df = fn - fL;
        fL = fn; g0 += UL0 * df; g1 += M3 * df; g2 += UL2 * df; sL += CRW * df * p.car.iIw;
      }
      {
        const float sa = (a0 + g0) - 0.1f * (a2 + g2), fn = clampf(fT + (rT - sa - p.car.R_T * fT) * p.car.iA_T, -LIMT, LIMT), df = fn - fT;
        fT = fn; g0 += UT0 * df; g2 += UT2 * df;
      }
      {
        const float sa = (a1 + g1) + 0.13f * (a2 + g2) + CRW * sR, fn = clampf(fR + (rR - sa - p.car.R_L * fR) * p.car.iA_L, -LIML, LIML), df = fn - fR;
        fR = fn; g0 -= UL0 * df; g1 += M3 * df; g2 -= UL2 * df; sR += CRW * df * p.car.iIw;
      }
      {
        const float sa = (a0 + g0) + 0.1f * (a2 + g2) + CRW * sX, fn = clampf(fX + (rX - sa - p.car.R_X * fX) * p.car.iA_X, -LIMC, LIMC), df = fn - fX;
        fX = fn; g0 += UX0 * df; g2 += UX2 * df; sX += CRW * df * p.car.iIb;
      }
      {
        const float sa = (a1 + g1) + CRW * sY, fn = clampf(fY + (rY - sa - p.car.R_Y * fY) * p.car.iA_Y, -LIMC, LIMC), df = fn - fY;
        fY = fn; g1 += M3 * df; sY += CRW * df * p.car.iIb;
      }
    }
    car_warm = true;
    if constexpr (PARK) {
      CPK(CP_F0) = g0; CPK(CP_F0 + 1) = g1; CPK(CP_F0 + 2) = g2;
      CPK(CP_FL) = fL; CPK(CP_FL + 1) = fT; CPK(CP_FL + 2) = fR; CPK(CP_FL + 3) = fX; CPK(CP_FL + 4) = fY;
      CPK(CP_EACC) = sL; CPK(CP_EACC + 1) = sR; CPK(CP_EACC + 2) = sY; CPK(CP_EACC + 3) = -sX; CPK(CP_EACC + 4) = -CJD * ext[4] * p.car.iIb;
    } else {
      car_f0 = g0; car_f1 = g1; car_f2 = g2;
      car_fL = fL; car_fT = fT; car_fR = fR; car_fX = fX; car_fY = fY;
      eacc[0] = sL; eacc[1] = sR; eacc[2] = sY; eacc[3] = -sX; eacc[4] = -CJD * ext[4] * p.car.iIb;
    }
    const float t0 = a0 + g0, t1 = a1 + g1;
    R.ax = cy * t0 - sy * t1; R.ay = sy * t0 + cy * t1; R.aw = a2 + g2;
  };
  // Point: what does not change over the substeps - the inverse inertia's determinant (a^2 + b^2 = mc^2 for any
  // heading), its yaw element, the yaw servo's denominator, the drive force - leaves the loop with its two divisions
  float pt_mIz = 0, pt_id = 0, pt_mid = 0, pt_m5 = 0, pt_iw = 0, pt_c1 = 0, pt_f0 = 0;
  if constexpr (!CAR) {
    const float m = PT_MASS + h * damp, Iz = PT_IO + h * PT_DAMP_Z;
    pt_mIz = m * Iz;
    pt_id = 1.0f / (m * (pt_mIz - PT_MC * PT_MC));
    pt_mid = m * pt_id; pt_m5 = m * pt_mid;
    const float g = h * pt_m5;
    pt_iw = 1.0f / (1 + g * (PT_GEAR_Z * PT_GEAR_Z));
    pt_c1 = g * PT_GEAR_Z * ctrl1;
    pt_f0 = gear * clampf(ctrl0, -PT_FLIM, PT_FLIM);
  }
#pragma unroll 1
  for (int sub = 0; sub <= nsub; sub++) {
    // heading: exact at the first substep and at the final forward evaluation (its cos / sin also serve the
    // lidar); in between the pair is rotated by the substep's own turn h w (|h w| < .2: series to h^7; an
    // implausible turn - non-finite state - falls back to the exact call)
    if (sub == 0 || sub == nsub || !(fabsf(yaw_turn) < 0.25f)) sincosf(yaw, &sy, &cy);
    else {
      const float d2 = yaw_turn * yaw_turn;
      const float sd = yaw_turn * __builtin_fmaf(d2, __builtin_fmaf(d2, __builtin_fmaf(d2, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
      const float cd = __builtin_fmaf(d2, __builtin_fmaf(d2, __builtin_fmaf(d2, -1.0f / 720.0f, 1.0f / 24.0f), -0.5f), 1.0f);
      const float c1 = __builtin_fmaf(cy, cd, -(sy * sd)), s1 = __builtin_fmaf(sy, cd, cy * sd);
      cy = c1; sy = s1;
    }
    if constexpr (!CAR) {
    // robot smooth dynamics (point.xml)
    {
      // (fused multiply-adds written out; the drive force and the centrifugal term of the offset COM share the
      // heading factor: (f0 + mc w^2) (cos, sin))
      const float fw = __builtin_fmaf(PT_MC * R.w, R.w, pt_f0);
      const float Fx = __builtin_fmaf(fw, cy, -(damp * R.vx)), Fy = __builtin_fmaf(fw, sy, -(damp * R.vy));
      const float Tz0 = -PT_DAMP_Z * R.w;
      const float a = -PT_MC * sy, b = PT_MC * cy;
      R.m0 = __builtin_fmaf(-b, b, pt_mIz) * pt_id; R.m1 = (a * b) * pt_id; R.m2 = -a * pt_mid;
      R.m3 = __builtin_fmaf(-a, a, pt_mIz) * pt_id; R.m4 = -b * pt_mid; R.m5 = pt_m5;
      // yaw servo evaluated at the end-of-substep rate (implicit, exact for the clipped-linear
      // law; see DESIGN.md "servo"): explicit feedback would have gain 12.7 per substep
      const float A = __builtin_fmaf(h, __builtin_fmaf(R.m2, Fx, __builtin_fmaf(R.m4, Fy, R.m5 * Tz0)), R.w);
      const float w_lin = (A + pt_c1) * pt_iw;
      const float Tz = __builtin_fmaf(PT_GEAR_Z, clampf(__builtin_fmaf(-PT_GEAR_Z, w_lin, ctrl1), -PT_FLIM, PT_FLIM), Tz0);
      R.ax = __builtin_fmaf(R.m0, Fx, __builtin_fmaf(R.m1, Fy, R.m2 * Tz));
      R.ay = __builtin_fmaf(R.m1, Fx, __builtin_fmaf(R.m3, Fy, R.m4 * Tz));
      R.aw = __builtin_fmaf(R.m2, Fx, __builtin_fmaf(R.m4, Fy, R.m5 * Tz));
    }
    } else {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
      // Car (car.xml; DESIGN.md "Car"): planar base with COM offset, two driven wheels and a
      // rear ball; floor contact = regularised Coulomb friction at the three contact points.
      // Evaluated in the BODY frame: there the generalised inverse inertia, the friction directions (body x and y at
      // the two wheels and the ball) and their levers are constants, so M^-1 J^T and J M^-1 J^T of every friction
      // element fold at compile time: car_floor above.
      constexpr float CM = CAR_M, CIO = CAR_IO, COY = CAR_OY;
      if constexpr (!QUIET) {
        // contacts of the footprint use the world-frame inverse inertia
        const float ox = -sy * COY, oy = cy * COY;
        const float a = -CM * oy, b = CM * ox;
        const float id = 1.0f / (CM * (CM * CIO - a * a - b * b));
        R.m0 = (CM * CIO - b * b) * id; R.m1 = (a * b) * id; R.m2 = (-a * CM) * id;
        R.m3 = (CM * CIO - a * a) * id; R.m4 = (-b * CM) * id; R.m5 = (CM * CM) * id;
      }
#ifdef SAG_CYC_SPLIT   // (diagnosis: what precedes the friction solve is booked under `load`)
      CYC(CY_LOAD);
#endif
      if (sub == nsub) car_warm = false;      // the forward evaluation behind the observation solves cold: obs = f(state)
      car_floor(0.f, R.w * R.w * COY, 0.f);   // M^-1 (centrifugal force of the offset COM) in body axes
      if constexpr (PARK) { CPK(CP_PX) = R.ax; CPK(CP_PX + 1) = R.ay; CPK(CP_PX + 2) = R.aw; }
      else { car_px = R.ax; car_py = R.ay; car_pw = R.aw; }
    }
    CYC(CY_ROBOT);
    cost_contacts = 0; btn_mask = 0;
    if constexpr (!QUIET) {
    // `active` = bodies that can have a non-zero acceleration or velocity this substep.
    // A sleeping body (v = 0, untouched) takes no part: the specification's rule (oracle world_forward, "sleeping
    // bodies"), which the oracle applies pair by pair in the same order.
    uint32_t active = awake;
    for (uint32_t m = active; m; m &= m - 1) dy.set_acc(__ffs(m) - 1, 0, 0, 0);
    // robot vs static circles (pillars, then buttons)
#pragma unroll 1
    for (int k = 0; k < n_static; k++) {
      if (k == capP) k = SAG_MAX_PILLARS;  // jump over unused pillar slots to the buttons
      const bool is_p = k < SAG_MAX_PILLARS;
      const bool on = (is_p ? (k < nP) : (k - SAG_MAX_PILLARS < nB)) && !ABL(ABL_NO_RS);
      const float sx = pick(stx, k), syy = pick(sty, k), sr = is_p ? psz : BUTTON_R;
      const float dx = sx - R.x, dyy = syy - R.y, rs = my_bound + sr;
      if (on && dx * dx + dyy * dyy <= rs * rs) {
        BV St; St.x = sx; St.y = syy; St.vx = St.vy = St.w = St.ax = St.ay = St.aw = 0;
        St.m0 = St.m1 = St.m2 = St.m3 = St.m4 = St.m5 = 0; St.dyn = 0;
        const int n = collide_list<SH_ME, false>(R, cy, sy, St, SH_STATIC, 1.f, 0.f, vsz, sr, sol0);
        WTW(0);
        if (is_p) cost_contacts += n;
        else if (n) btn_mask |= 1u << (k - SAG_MAX_PILLARS);
      }
    }
    CYC(CY_RS);
    // robot vs free bodies: cheap broadphase for all, then each lane walks ITS OWN hit list
    // (ascending index = the specification's order), so a wavefront runs the narrowphase
    // max-hits times, not once per body index that any lane happens to touch
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < SAG_MAX_VASES; k++) {  // unrolled: static LDS offsets, loads issued together
      const float dx = LP(LS_X, k) - R.x, dyy = LP(LS_Y, k) - R.y, rs = my_bound + vase_r;
      if (k < nV && dx * dx + dyy * dyy <= rs * rs) hits |= 1u << k;
    }
    if (has_box) {
      const float dx = LP(LS_X, BOX_ID) - R.x, dyy = LP(LS_Y, BOX_ID) - R.y, rs = my_bound + box_r;
      if (dx * dx + dyy * dyy <= rs * rs) hits |= 1u << BOX_ID;
    }
    if (ABL(ABL_NO_RV)) hits = 0;
    // (the task object is every lane's LAST hit - BOX_ID is the highest body index - but at different positions of
    // the lanes' lists: walked in one loop, every iteration ran both the 1-geom and the 5-geom form of the pair tests
    // for some lane.  Vases first for all lanes, then the object in one step of its own: the same order per lane.)
    for (uint32_t m = hits & ~(1u << BOX_ID); m; m &= m - 1) {
      const int k = __ffs(m) - 1;
      dy.ensure(k, false);
      BV V; float cv, sv; load_body(dy, k, V, cv, sv);
      const int n = collide_list<SH_ME, false>(R, cy, sy, V, SH_VASE, cv, sv, vsz, 0.f, sol0);
      WTW(1);
      cost_contacts += n;
      if (n) { dy.set_acc(k, V.ax, V.ay, V.aw); active |= 1u << k; }
    }
    if (HAS_TBOX && (hits >> BOX_ID & 1u)) {
      dy.ensure(BOX_ID, false);
      BV V; float cv, sv; load_body(dy, BOX_ID, V, cv, sv);
      const int n = collide_list<SH_ME, HAS_TBOX>(R, cy, sy, V, bk.sh, cv, sv, vsz, 0.f, solb);   // the task object is not an obstacle (consts.OBSTACLES)
      WTW(2);
      if (n) { dy.set_acc(BOX_ID, V.ax, V.ay, V.aw); active |= 1u << BOX_ID; }
    }
    // HaulBox tether (haul_box.py:21-29): spatial tendon robot site (z .1) <-> box site (z .2),
    // limited to [0, .75]; beyond the limit a soft constraint pulls the two together
    if (HAS_TBOX && task == SAG_TASK_HAUL_BOX) {
      const float bx = LP(LS_X, BOX_ID), by = LP(LS_Y, BOX_ID);
      const float dx = bx - R.x, dyy = by - R.y;
      const float d = sqrtf(dx * dx + dyy * dyy), L = sqrtf(d * d + 0.01f), viol = L - 0.75f;
      if (viol > 0 && d >= 1e-9f) {
        dy.ensure(BOX_ID, false);
        BV B; float cb_, sb_; load_body(dy, BOX_ID, B, cb_, sb_);
        const float jx = dx / L, jy = dyy / L;
        const float Ldot = jx * (B.vx - R.vx) + jy * (B.vy - R.vy);
        const float Lacc = jx * (B.ax - R.ax) + jy * (B.ay - R.ay);
        float ua[3], ub[3];
        const float Ainv = minv_apply(R, jx, jy, 0, ua) + minv_apply(B, jx, jy, 0, ub);
        const float f = impedance(viol) * ((-sol0.bcoef * Ldot - sol0.kcoef * viol) - Lacc) / Ainv;
        if (f < 0) {  // a tendon only pulls
          R.ax -= ua[0] * f; R.ay -= ua[1] * f; R.aw -= ua[2] * f;
          dy.set_acc(BOX_ID, B.ax + ub[0] * f, B.ay + ub[1] * f, B.aw + ub[2] * f);
          active |= 1u << BOX_ID;
        }
      }
    }
#ifdef SAG_CYC_SPLIT   // (diagnosis: the robot's contacts are booked under `robot-static`, the re-solve alone under `robot-free`)
    CYC(CY_RS);
#endif
    if constexpr (CAR) {
      // the floor friction once more when the contacts (or the tether) changed the base acceleration: what they
      // added now belongs to `everything else`
      if constexpr (PARK) {
        if (R.ax != CPK(CP_PX) || R.ay != CPK(CP_PX + 1) || R.aw != CPK(CP_PX + 2))
          car_floor(cy * R.ax + sy * R.ay - CPK(CP_F0), cy * R.ay - sy * R.ax - CPK(CP_F0 + 1), R.aw - CPK(CP_F0 + 2));
      } else {
        if (R.ax != car_px || R.ay != car_py || R.aw != car_pw)
          car_floor(cy * R.ax + sy * R.ay - car_f0, cy * R.ay - sy * R.ax - car_f1, R.aw - car_f2);
      }
    }
    CYC(CY_RV);
#ifdef SAG_WAVE_TIMES
    if (sub == nsub) for (int k_ = 0; k_ < 6; k_++) { WTT(k_, wt_work[k_] - wt_prev[k_]); wt_prev[k_] = wt_work[k_]; }
#endif
    if (sub == nsub) break;  // final forward: robot acceleration + contact flags only
    if (active && !ABL(ABL_NO_ACTIVE)) {
      // free bodies vs static circles (pillars then buttons), per active body
      for (uint32_t m = active; m; m &= m - 1) {
        const int k = __ffs(m) - 1;
        const bool isb = is_box(k);
        const float bx_ = LP(LS_X, k), by_ = LP(LS_Y, k), br = isb ? box_r : vase_r;
        uint32_t shit = 0;
#pragma unroll 1
        for (int q = 0; q < n_static; q++) {
          if (q == capP) q = SAG_MAX_PILLARS;
          const bool is_p = q < SAG_MAX_PILLARS;
          const bool on = (is_p ? (q < nP) : (q - SAG_MAX_PILLARS < nB)) && !ABL(ABL_NO_VS);
          const float dx = pick(stx, q) - bx_, dyy = pick(sty, q) - by_, rs = br + (is_p ? psz : BUTTON_R);
          if (on && dx * dx + dyy * dyy <= rs * rs) shit |= 1u << q;
        }
        for (uint32_t mq = shit; mq; mq &= mq - 1) {
          const int q = __ffs(mq) - 1;
          float sx = 0, syy = 0;  // per-lane q: select from the register arrays
#pragma unroll
          for (int z = 0; z < NSTAT; z++) if (z == q) { sx = stx[z]; syy = sty[z]; }
          const float sr = q < SAG_MAX_PILLARS ? psz : BUTTON_R;
          BV V; float cv, sv; load_body(dy, k, V, cv, sv);
          BV St; St.x = sx; St.y = syy; St.vx = St.vy = St.w = St.ax = St.ay = St.aw = 0;
          St.m0 = St.m1 = St.m2 = St.m3 = St.m4 = St.m5 = 0; St.dyn = 0;
          collide_shapes<!DOGGO>(V, isb ? bk.sh : SH_VASE, cv, sv, St, SH_STATIC, 1.f, 0.f, vsz, sr, isb ? solb : sol0);
          WTW(3);
          dy.set_acc(k, V.ax, V.ay, V.aw);
        }
      }
      CYC(CY_VS);
      // free body pairs (a < b) in lexicographic order, the task object being the last body; at
      // least one of the two active.  Broadphase per ACTIVE body (each lane walks its own one or
      // two) against every body index in a uniform unrolled loop: 11 x max-active iterations per
      // wavefront instead of the 55 a union over "for a: for b > a" costs once the lanes' active
      // indices differ.  Hits go into a 55-bit pair mask (bit = a (21 - a) / 2 + b - a - 1), walked
      // in ascending order = the specification's pair order.  A body that only becomes active inside
      // this loop was asleep: by the specification its other pairs with sleeping bodies stay skipped.
      uint64_t pairs = 0;
      if (!ABL(ABL_NO_VV)) {
        for (uint32_t m = active; m; m &= m - 1) {
          const int k = __ffs(m) - 1;
          const float kx = LP(LS_X, k), ky = LP(LS_Y, k), kr = is_box(k) ? box_r : vase_r;
          // (all positions first, then the tests without branches: read one by one under the hit branch of the body before,
          // every test waited out an LDS round trip - 2 k cycles per active body, the largest item of a small batch's slowest wavefront)
          constexpr int NJ = HAS_TBOX ? NBODY : SAG_MAX_VASES;
          float jx[NJ], jy[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++) { jx[j] = LP(LS_X, j); jy[j] = LP(LS_Y, j); }
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            const float dx = jx[j] - kx, dyy = jy[j] - ky, rs = kr + (j == BOX_ID ? box_r : vase_r);
            const bool hit = (fmask >> j & 1u) & (uint32_t)(j != k) & (uint32_t)(dx * dx + dyy * dyy <= rs * rs);
            const int lo = min(j, k), hi = max(j, k);
            pairs |= hit ? 1ull << (lo * (21 - lo) / 2 + hi - lo - 1) : 0ull;
          }
        }
      }
      CYC(CY_VV_BROAD);
      for (uint64_t m = pairs; m; m &= m - 1) {
        int idx = __ffsll((unsigned long long)m) - 1, a = 0;
        while (idx >= NBODY - 1 - a) { idx -= NBODY - 1 - a; a++; }
        const int b = a + 1 + idx;
        const bool isb = is_box(b);
        dy.ensure(a, false); dy.ensure(b, false);
        BV A, B; float ca, sa, cb, sb;
        load_body(dy, a, A, ca, sa);
        load_body(dy, b, B, cb, sb);
        const int n = collide_list<SH_VASE, HAS_TBOX>(A, ca, sa, B, isb ? bk.sh : SH_VASE, cb, sb, vsz, 0.f, isb ? solb : sol0);
        WTW(4);
        if (n) {
          dy.set_acc(a, A.ax, A.ay, A.aw); dy.set_acc(b, B.ax, B.ay, B.aw);
          active |= 1u << a | 1u << b;
        }
      }
      CYC(CY_VV_NARROW);
      CYC(CY_VV_BROAD);
      // floor friction + semi-implicit Euler + rest capture for the active bodies
      for (uint32_t m = active; m; m &= m - 1) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
        const int k = __ffs(m) - 1;
        float vx_, vy_, w_, ax_, ay_, aw_;
        WTW(5);
        dy.get(k, vx_, vy_, w_, ax_, ay_, aw_);
        const float bc = sol0.bcoef;
        if (!is_box(k) || bk.sh == SH_BOX) {
          const float mm = !is_box(k) ? vk.m : bk.m, II = !is_box(k) ? vk.I : bk.I, rr = !is_box(k) ? vk.reff : bk.reff;
          const float fmax_ = MU * GRAV * mm;
          float fx = -SOL_D0 * mm * (bc * vx_ + ax_), fy = -SOL_D0 * mm * (bc * vy_ + ay_);
          const float f2 = fx * fx + fy * fy;
          if (f2 > fmax_ * fmax_) { const float sc = fmax_ / sqrtf(f2); fx *= sc; fy *= sc; }
          ax_ += fx / mm; ay_ += fy / mm;
          const float t = clampf(-SOL_D0 * II * (bc * w_ + aw_), -fmax_ * rr, fmax_ * rr);
          aw_ += t / II;
        } else {
          // rod: rolling resistance across its axis, sliding friction along it; ball: rolling
          float c, s; sincosf(LP(LS_YAW, k), &s, &c);
          float bvx = c * vx_ + s * vy_, bvy = -s * vx_ + c * vy_, bax = c * ax_ + s * ay_, bay = -s * ax_ + c * ay_;
          const float mg = bk.m * GRAV;
          float tlim;
          if (bk.sh == SH_ROD) {
            const float mx = 1.5f * bk.m, lx = 0.05f * mg / 0.08f, ly = 1.2f * mg;
            const float fx = clampf(-SOL_D0 * mx * (bc * bvx + bax), -lx, lx);
            const float fy = clampf(-SOL_D0 * bk.m * (bc * bvy + bay), -ly, ly);
            bax += fx / mx; bay += fy / bk.m;
            tlim = 1.2f * mg * 0.15f;
          } else {
            const float me = 1.4f * bk.m, lim = 0.05f * mg / 0.14f;
            float fx = -SOL_D0 * me * (bc * bvx + bax), fy = -SOL_D0 * me * (bc * bvy + bay);
            const float f2 = fx * fx + fy * fy;
            if (f2 > lim * lim) { const float sc = lim / sqrtf(f2); fx *= sc; fy *= sc; }
            bax += fx / me; bay += fy / me;
            tlim = 0.003f * mg;
          }
          ax_ = c * bax - s * bay; ay_ = s * bax + c * bay;
          aw_ += clampf(-SOL_D0 * bk.I * (bc * w_ + aw_), -tlim, tlim) / bk.I;
        }
        vx_ += h * ax_; vy_ += h * ay_; w_ += h * aw_;
        // static friction capture (specification, oracle integrate_free)
        if (fabsf(vx_) < REST_V && fabsf(vy_) < REST_V && fabsf(w_) < REST_W && fabsf(h * ax_) < REST_V &&
            fabsf(h * ay_) < REST_V && fabsf(h * aw_) < REST_W) { vx_ = 0; vy_ = 0; w_ = 0; }
        dy.set_vel(k, vx_, vy_, w_);
        LP(LS_X, k) += h * vx_; LP(LS_Y, k) += h * vy_; LP(LS_YAW, k) += h * w_;
        if (vx_ == 0 && vy_ == 0 && w_ == 0) awake &= ~(1u << k); else awake |= 1u << k;
      }
      CYC(CY_INTEG);
      dirty |= active;
    }
    } else {
      if (sub == nsub) break;
    }
#ifdef SAG_WAVE_TIMES
    for (int k_ = 0; k_ < 6; k_++) { WTT(k_, wt_work[k_] - wt_prev[k_]); wt_prev[k_] = wt_work[k_]; }
#endif
    R.vx = __builtin_fmaf(h, R.ax, R.vx); R.vy = __builtin_fmaf(h, R.ay, R.vy); R.w = __builtin_fmaf(h, R.aw, R.w);
    R.x = __builtin_fmaf(h, R.vx, R.x); R.y = __builtin_fmaf(h, R.vy, R.y); yaw_turn = h * R.w; yaw += yaw_turn;
    if constexpr (CAR) {
#pragma clang fp contract(on)   // a * b + c of ONE expression fuses (frontend decision: the same in every instance)
#pragma unroll
      for (int k = 0; k < 5; k++) ext[k] += h * (PARK ? CPK(CP_EACC + k) : eacc[k]);
      if constexpr (PARK) { ext[5] = CPK(CP_Q); ext[6] = CPK(CP_Q + 1); ext[7] = CPK(CP_Q + 2); ext[8] = CPK(CP_Q + 3); }
      // ball quaternion: rate relative to the base, in base axes: q <- exp(h W / 2) q
      const float wx = ext[2], wy = ext[3], wz = ext[4];
      const float n2 = wx * wx + wy * wy + wz * wz;
      if (n2 > 0) {
        // exp(h W / 2) = (cos th, sin th W / |W|), th = h |W| / 2: for th < 1/2 the even series of cos th and
        // sin th / th in th^2 (no sqrt, no division; truncation < 2e-9), else the closed form
        const float th2 = 0.25f * h * h * n2;
        float sn, cs;
        if (th2 < 0.25f) {
          cs = 1.0f - th2 * 0.5f * (1.0f - th2 * (1.0f / 12.0f) * (1.0f - th2 * (1.0f / 30.0f) * (1.0f - th2 * (1.0f / 56.0f))));
          sn = 0.5f * h * (1.0f - th2 * (1.0f / 6.0f) * (1.0f - th2 * (1.0f / 20.0f) * (1.0f - th2 * (1.0f / 42.0f) * (1.0f - th2 * (1.0f / 72.0f)))));
        } else {
          const float n = sqrtf(n2);
          sincosf(0.5f * h * n, &sn, &cs);
          sn /= n;
        }
        const float dw = cs, dx = sn * wx, dy_ = sn * wy, dz = sn * wz;
        const float qw = dw * ext[5] - dx * ext[6] - dy_ * ext[7] - dz * ext[8];
        const float qx = dw * ext[6] + dx * ext[5] + dy_ * ext[8] - dz * ext[7];
        const float qy = dw * ext[7] - dx * ext[8] + dy_ * ext[5] + dz * ext[6];
        const float qz = dw * ext[8] + dx * ext[7] - dy_ * ext[6] + dz * ext[5];
        const float qn = 1.0f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        ext[5] = qw * qn; ext[6] = qx * qn; ext[7] = qy * qn; ext[8] = qz * qn;
        if constexpr (PARK) { CPK(CP_Q) = ext[5]; CPK(CP_Q + 1) = ext[6]; CPK(CP_Q + 2) = ext[7]; CPK(CP_Q + 3) = ext[8]; }
      }
    }
  }
  if constexpr (PARK) {   // back into the registers for the write-back and the sensors
    ext[5] = CPK(CP_Q); ext[6] = CPK(CP_Q + 1); ext[7] = CPK(CP_Q + 2); ext[8] = CPK(CP_Q + 3);
    ext_tail.y = CPK(CP_TAIL); ext_tail.z = CPK(CP_TAIL + 1); ext_tail.w = CPK(CP_TAIL + 2);
  }

  }  // !DOGGO
  if (p.ext_cc) {   // replayed episode: the contact results of the final state come from the caller (sag_set_ext_contacts)
    const int32_t v = p.ext_cc[i];
    if (v >= 0) { cost_contacts = v; btn_mask = p.ext_btn[i]; }
  }
  CYC(CY_ROBOT);
  // ---- write back dynamic state -------------------------------------------------
  const float boxx = LP(LS_X, BOX_ID), boxy = LP(LS_Y, BOX_ID);
  if (!p.observe_only && live) {
    if constexpr (!DOGGO) {
      // groups 0 and 1 whole (the goal half of group 1 is rewritten by the reward block if it moves)
      float4* __restrict__ W4 = reinterpret_cast<float4*>(S);
      W4[(size_t)0 * N + i] = make_float4(R.x, R.y, yaw, R.vx);   // (group 1 follows the reward block: it carries the goal)
    }
    if constexpr (CAR) {
      float4* __restrict__ W4 = reinterpret_cast<float4*>(S);
      W4[(size_t)DG_EXT * N + i] = make_float4(ext[0], ext[1], ext[2], ext[3]);
      W4[(size_t)(DG_EXT + 1) * N + i] = make_float4(ext[4], ext[5], ext[6], ext[7]);
      W4[(size_t)(DG_EXT + 2) * N + i] = make_float4(ext[8], ext_tail.y, ext_tail.z, ext_tail.w);
    }
    for (uint32_t m = dirty; m; m &= m - 1) {
      const int k = __ffs(m) - 1;
      const int f = DV_POS + 3 * k, fv = DV_VEL + 3 * k;
      SD(f) = LP(LS_X, k); SD(f + 1) = LP(LS_Y, k); SD(f + 2) = LP(LS_YAW, k);
      const int d = dy.slot(k);
      if (d >= 0) { SD(fv) = POOL(d, 0); SD(fv + 1) = POOL(d, 1); SD(fv + 2) = POOL(d, 2); }
    }
    step += 1;
    // whole int4 (meta, step, envid, flags): a 4-byte store at a 16-byte stride would leave partial lines
    reinterpret_cast<int4*>(I + ipad((size_t)N))[i] = make_int4(iw.x, step, iw.z, iw.w);
  }
  tstate = (tstate & ~(TS_AWAKE_BITS << TS_AWAKE_SHIFT)) | (awake & TS_AWAKE_BITS) << TS_AWAKE_SHIFT;
  // ---- classification for the NEXT step (QUIET / BUSY split): busy unless provably nothing can
  //      touch the robot or move.  Reach of the robot within one step from the final state:
  //      |v| T + a_max T^2 (drive + contact-free dynamics only raise |v| by at most a_max T)
  bool busy_next = false;
  if (!p.observe_only) {
    const float T = p.nstep_table * h;
    const float amax = CAR ? GRAV : 1.05f * gear * PT_FLIM / PT_MASS;
    const float reach = sqrtf(R.vx * R.vx + R.vy * R.vy) * T + amax * T * T + 0.005f;
    bool busy = awake != 0 || (HAS_TBOX && task == SAG_TASK_HAUL_BOX);
    bool near_obj = false, near_static = false;   // (the kind: BUSY_CLASSES)
    const bool moving = busy;
    const float rr = my_bound + reach;
    // Point: bounding circles first, then the footprint itself.  The robot is inside circle A (its sphere,
    // r .1) and circle B (around the arrow box at +.1 along the heading, r .05 sqrt 2); within the step
    // the origin travels <= reach and the heading turns <= dth (the yaw servo saturates at 3 rad/s).  A vase
    // (sleeping here, else the env is busy anyway) can only be touched if its box comes within those radii:
    // cb_contact needs the sphere centre closer than .1 to the box, bb_contact a vertex of one box inside the
    // other, i.e. the box within .05 sqrt 2 of B's centre.  A third fewer envs pass than with circles alone.
    constexpr bool FINE = !CAR && !DOGGO;
    const float dth = (fmaxf(fabsf(R.w), 3.0f) + 1.0f) * T;
    const float fa = 0.1f + reach, fb = 0.0707107f + reach + 0.1f * dth;
    const float Bx = R.x + 0.1f * cy, By = R.y + 0.1f * sy;   // cy, sy: heading at the final state
#pragma unroll 1
    for (int k = 0; k < nV; k++) {
      const float vx_ = LP(LS_X, k), vy_ = LP(LS_Y, k);
      const float dx = vx_ - R.x, dyy = vy_ - R.y, rs = rr + vase_r;
      if (dx * dx + dyy * dyy <= rs * rs) {
        if constexpr (FINE) {
          float sv, cv; sincosf(LP(LS_YAW, k), &sv, &cv);
          auto box_d2 = [&](float px, float py) {   // squared distance of a point to the vase's box
            const float wx = px - vx_, wy = py - vy_;
            const float ex = fmaxf(fabsf(cv * wx + sv * wy) - vsz, 0.f), ey = fmaxf(fabsf(cv * wy - sv * wx) - vsz, 0.f);
            return ex * ex + ey * ey;
          };
          busy |= box_d2(R.x, R.y) <= fa * fa || box_d2(Bx, By) <= fb * fb;
        } else busy = true;
      }
    }
    if (has_box) {
      const float dx = LP(LS_X, BOX_ID) - R.x, dyy = LP(LS_Y, BOX_ID) - R.y, rs = rr + box_r;
      near_obj = dx * dx + dyy * dyy <= rs * rs;
    }
#pragma unroll 1
    for (int k = 0; k < n_static; k++) {
      if (k == capP) k = SAG_MAX_PILLARS;
      const bool is_p = k < SAG_MAX_PILLARS;
      const bool on = is_p ? (k < nP) : (k - SAG_MAX_PILLARS < nB);
      const float sr = is_p ? psz : BUTTON_R;
      const float skx = pick(stx, k), sky = pick(sty, k);
      const float dx = skx - R.x, dyy = sky - R.y, rs = rr + sr;
      if constexpr (FINE) {
        const float ex = skx - Bx, ey = sky - By;
        near_static |= on && (dx * dx + dyy * dyy <= (fa + sr) * (fa + sr) || ex * ex + ey * ey <= (fb + sr) * (fb + sr));
      } else near_static |= on && dx * dx + dyy * dyy <= rs * rs;
    }
    const bool near_vase = busy && !moving;    // (only the vase loop above has raised `busy` since)
    busy |= near_obj || near_static;
    const uint32_t kind = p.busy_kinds ? (uint32_t)near_obj | (uint32_t)near_static << 1 | (uint32_t)(near_vase || moving) << 2 : 0u;
    const uint32_t nbit = TS_BUSY_BIT << (p.phase ^ 1);
    tstate = busy ? (tstate | nbit) : (tstate & ~nbit);
    if (busy && live && p.kind) p.kind[i] = (uint8_t)kind;
    busy_next = busy;
    // next step's busy list, built here instead of by a separate compaction pass: one atomic per
    // wavefront claims a contiguous chunk (a chunk keeps the env neighbourhood of its wavefront)
    if (MODE != MODE_ALL && p.rows_next) {
      for (int c = 0; c < BUSY_CLASSES; c++) {   // (per kind: the list is [BUSY_CLASSES][N], k_compact)
        const bool mine = busy && live && (int)kind == c;
        const uint64_t bm = __ballot(mine);
        if (bm) {
          int chunk = 0;
          if (lane == __ffsll((unsigned long long)bm) - 1) chunk = atomicAdd(p.count_next + c, __popcll(bm));
          chunk = __shfl(chunk, __ffsll((unsigned long long)bm) - 1);
          if (mine) p.rows_next[(size_t)c * p.N + chunk + __popcll(bm & ((1ull << lane) - 1))] = i;
        }
      }
    }
  }

  CYC(CY_WRITEBACK);
  // hazards are only needed from here on (cost, lidar): loading them late keeps 18 registers
  // free during the physics; the reward arithmetic below covers the latency
  float hzx[SAG_MAX_HAZARDS], hzy[SAG_MAX_HAZARDS];
  float hsz;
  {
    float hz[20];  // 9 x (x, y), hazard_size, bound: five groups
#pragma unroll
    for (int g = 0; g < 5; g++) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g == 4 || 2 * g < capH) v = use_hot ? reinterpret_cast<const float4*>(p.hot_haz)[(size_t)i * 5 + g] : G4(DG_HAZ + g);
      hz[4 * g] = v.x; hz[4 * g + 1] = v.y; hz[4 * g + 2] = v.z; hz[4 * g + 3] = v.w;
    }
    static_assert(SAG_MAX_HAZARDS == 9, "hazards + size fill five groups");
#pragma unroll
    for (int k = 0; k < SAG_MAX_HAZARDS; k++) { hzx[k] = k < capH ? hz[2 * k] : 0.f; hzy[k] = k < capH ? hz[2 * k + 1] : 0.f; }
    hsz = hz[18];
  }

  // ---- PhysicsError branch (safe_adaptation_gym.py:73-75) -------------------------
  bool bad = false;
  {
    float v[6] = {R.x, R.y, yaw, R.vx, R.vy, R.w};
#pragma unroll
    for (int k = 0; k < 6; k++) bad |= !(fabsf(v[k]) <= 1e10f);
    if constexpr (DOGGO) {
      bad |= !(fabs(dgs.pos[2]) <= 1e10) || !(fabs(dgs.vlin[2]) <= 1e10);
      for (int k = 0; k < 4; k++) bad |= !(fabs(dgs.quat[k]) <= 1e10);
      for (int k = 0; k < 3; k++) bad |= !(fabs(dgs.wloc[k]) <= 1e10);
      for (int k = 0; k < DG_NJ; k++) bad |= !(fabs(dgs.q[k]) <= 1e10) || !(fabs(dgs.qd[k]) <= 1e10);
    }
  }

  // ---- reward (tasks/<task>.py compute_reward), fp64 from the fp32 state -----------
  double rew0 = 0, rew1 = 0;
  int met = 0;
  const double rx = R.x, ry = R.y;
  // GoToGoal._resample_goal_position (:59-80) + utils.draw_placement (utils.py:28-70)
  bool need_goal = false;  // a goal was met: resample it (wave-cooperatively, below)
  float box_last1 = 0, box_last2 = 0;
  if (!p.observe_only && !bad) {
    if (task == SAG_TASK_PRESS_BUTTONS || task == SAG_TASK_PRESS_BUTTONS_SCARCE) {
      int gb = tstate & 7, bstate = tstate >> 3 & 1, timer = tstate >> 4 & 7;
      double gd = dist2d(rx, ry, SD(112 + 2 * gb), SD(113 + 2 * gb));
      double r = task == SAG_TASK_PRESS_BUTTONS ? (double)last0 - gd : 0.0;
      float last = (float)gd;
      if (btn_mask >> gb & 1) {
        r += 1.0; met = 1;
        gb = (int)(rng.word() & 3u);
        if (rng.exhausted) flags |= 2;
        timer = 5;
        last = (float)dist2d(rx, ry, SD(112 + 2 * gb), SD(113 + 2 * gb));
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
    } else if (!has_box) {
      // GoToGoal family (tasks/go_to_goal.py:31-45): 3-D distance incl. dz
      double gx = goalx, gy = goaly;
      double dx = rx - gx, dyy = ry - gy, dz = (DOGGO ? (double)(float)dgs.pos[2] : (double)PT_Z) - GOAL_Z;
      double dist = sqrt(dx * dx + dyy * dyy + dz * dz);
      double r = (double)last0 - dist;
      if (task == SAG_TASK_GO_TO_GOAL_SCARCE) r *= (dist <= GOAL_SIZE * 1.5) ? 1.0 : 0.0;
      float last = (float)dist;
      if (dist <= GOAL_SIZE) { met = 1; need_goal = true; r += 1.0; }
      last0 = last;
      if (task == SAG_TASK_UNSUPERVISED) {  // tasks/unsupervised.py:48-67
        // subtree COM of the robot in body axes: point (mc/m, 0); car (0, .0074)
        const double c = cos((double)yaw), s = sin((double)yaw);
        const double bx_ = CAR ? 0.0 : (double)PT_MC / (double)PT_MASS, by_ = CAR ? (double)CAR_OY : 0.0;
        const double lx = c * bx_ - s * by_, ly = s * bx_ + c * by_;
        double x = rx + lx, y = ry + ly, w = R.w;
        double u = (double)R.vx - w * ly, v = (double)R.vy + w * lx;
        if constexpr (DOGGO) { x = dgr.comvel[0]; y = dgr.comvel[1]; u = dgr.comvel[2]; v = dgr.comvel[3]; }
        double radius = sqrt(x * x + y * y);
        rew0 = (((-u * y + v * x) / radius) / (1 + fabs(radius - 1.5))) * 1e-1;
        rew1 = r;
      } else rew0 = r;
    } else {
      // PushBox family (push_box.py:74-100, push_box_scarce.py:24-50, haul_box.py:34-48): 2-D
      float last1 = SF(SAG_F_LAST + 1), last2 = SF(SAG_F_LAST + 2);
      const double bx = boxx, by = boxy;
      double r = 0;
      if (task != SAG_TASK_HAUL_BOX) {
        const double bd = dist2d(rx, ry, bx, by);
        double prog = (double)last1 - bd;
        if (task == SAG_TASK_PUSH_BOX_SCARCE) prog *= (bd <= GOAL_SIZE * 1.70) ? 1.0 : 0.0;
        r += prog;
        last1 = (float)bd;
      }
      const double bg = dist2d(bx, by, goalx, goaly);
      r += (double)last2 - bg;
      last2 = (float)bg;
      if (bg <= GOAL_SIZE) { met = 1; need_goal = true; r += 1.0; }
      box_last1 = last1; box_last2 = last2;
      rew0 = r;
    }
  }
  CYC(CY_REWARD);
  // ---- goal resampling (GoToGoal._resample_goal_position, tasks/go_to_goal.py:59-80 +
  //      utils.draw_placement, utils.py:28-70), wave-cooperative.  A met goal needs ~30 rejection
  //      tries on these crowded layouts, each testing 21 keep-out distances in fp64: done by the
  //      one lane that owns the env this stretched a wavefront by ~300k cycles and set the tail of
  //      the whole launch.  Candidate j of the reference's sequence is a pure function of j (the
  //      rectangle has grown 1.01^j, its four words sit at stream position pos + 4 j), so the 64
  //      lanes test candidates j = base + lane at once and the first accepted one wins.
  {
    // (only lanes that carry an env: the padding lanes of a small batch all mirror env N - 1 and would each repeat ITS resample -
    // 48 of them at 16 envs per wavefront, a 0.2-ms spike in every wavefront of the step in which that one env meets its goal)
    uint64_t need_mask = __ballot(need_goal && live && !ABL(ABL_NO_RESAMPLE));
    while (need_mask) {
      const int src = __ffsll((unsigned long long)need_mask) - 1;
      need_mask &= need_mask - 1;
      const int s_i = __shfl(i, src);
      const double s_rx = __shfl(R.x, src), s_ry = __shfl(R.y, src);
      const int s_nH = __shfl(nH, src), s_nV = __shfl(nV, src), s_nP = __shfl(nP, src);
      const int s_box = __shfl((int)has_box, src);
      const double s_bx = __shfl(boxx, src), s_by = __shfl(boxy, src);
      const int s_pos0 = __shfl(rng.pos, src), s_len = __shfl(rng.len, src);
      const uint32_t s_env = __shfl(rng.env, src), s_step = __shfl(rng.step, src), s_ep4 = __shfl(rng.ep4, src);
      const uint32_t* s_tape = p.tape ? p.tape + (size_t)s_i * p.tape_len : nullptr;
      const double k_robot = S[saddr(SAG_F_KEEPOUT, (size_t)N, (size_t)s_i)], k_haz = S[saddr(SAG_F_KEEPOUT + 1, (size_t)N, (size_t)s_i)],
                   k_vase = S[saddr(SAG_F_KEEPOUT + 2, (size_t)N, (size_t)s_i)], k_pil = S[saddr(SAG_F_KEEPOUT + 3, (size_t)N, (size_t)s_i)],
                   k_box = S[saddr(SAG_F_KEEPOUT + 4, (size_t)N, (size_t)s_i)];
      int winner = -1, exhausted = 0;
      float w_gx = 0, w_gy = 0;
#pragma unroll 1
      for (int base = 0; base < 10000 && winner < 0 && !exhausted; base += WAVE) {
        const int j = base + lane;
        double half = 1.5;  // one multiplication per earlier rejection, as the reference does
#pragma unroll 1
        for (int q = 0; q < j; q++) half *= 1.01;
        // four words at stream position pos0 + 4 j
        uint32_t wd[4];
        bool out_of_tape = false;
        const int w0 = s_pos0 + 4 * j;
        if (s_tape) {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            out_of_tape |= w0 + t >= s_len;
            wd[t] = w0 + t < s_len ? s_tape[w0 + t] : 0u;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            uint32_t c[4] = {s_env, s_step, (uint32_t)((w0 + t) >> 2), s_ep4};
            philox4x32_10(c, p.key0, p.key1);
            wd[t] = c[(w0 + t) & 3];
          }
        }
        const double lo = -half + GOAL_KEEPOUT, hi = half - GOAL_KEEPOUT;
        const double ngx = lo + (hi - lo) * (((wd[0] >> 5) * 67108864.0 + (wd[1] >> 6)) / 9007199254740992.0);
        const double ngy = lo + (hi - lo) * (((wd[2] >> 5) * 67108864.0 + (wd[3] >> 6)) / 9007199254740992.0);
        bool ok = !out_of_tape && j < 10000 && !(dist2d(ngx, ngy, s_rx, s_ry) < k_robot + GOAL_KEEPOUT);
#pragma unroll
        for (int k = 0; k < SAG_MAX_HAZARDS; k++) {
          const double hx = __shfl(hzx[k], src), hy = __shfl(hzy[k], src);
          if (k < s_nH && ok) ok = !(dist2d(ngx, ngy, hx, hy) < k_haz + GOAL_KEEPOUT);
        }
#pragma unroll 1
        for (int k = 0; k < s_nV; k++) {
          const double vx_ = lds[(LS_X + k) * WAVE + src], vy_ = lds[(LS_Y + k) * WAVE + src];
          if (ok) ok = !(dist2d(ngx, ngy, vx_, vy_) < k_vase + GOAL_KEEPOUT);
        }
#pragma unroll
        for (int k = 0; k < SAG_MAX_PILLARS; k++) {
          const double px_ = __shfl(stx[k], src), py_ = __shfl(sty[k], src);
          if (k < s_nP && ok) ok = !(dist2d(ngx, ngy, px_, py_) < k_pil + GOAL_KEEPOUT);
        }
        if (s_box && ok) ok = !(dist2d(ngx, ngy, s_bx, s_by) < k_box + GOAL_KEEPOUT);
        const uint64_t okm = __ballot(ok), exm = __ballot(out_of_tape);
        const int first_ok = okm ? __ffsll((unsigned long long)okm) - 1 : WAVE;
        const int first_ex = exm ? __ffsll((unsigned long long)exm) - 1 : WAVE;
        if (first_ok < first_ex) {
          winner = base + first_ok;
          w_gx = __shfl((float)ngx, first_ok); w_gy = __shfl((float)ngy, first_ok);
        } else if (first_ex < WAVE) {
          exhausted = 1; winner = base + first_ex;  // words consumed up to the failed draw
        }
      }
      if (lane == src) {
        if (winner >= 0 && !exhausted) { goalx = w_gx; goaly = w_gy; rng.pos = s_pos0 + 4 * (winner + 1); }
        else if (exhausted) { rng.exhausted = 1; rng.pos = s_pos0 + 4 * (winner + 1); flags |= 2; }
        else { rng.pos = s_pos0 + 4 * 10000; flags |= 1; }
      }
    }
  }
  CYC(CY_RESAMPLE);
  if (!p.observe_only && !bad) {
    if (need_goal) {
      // GoToGoal.reset (:50-57): last goal distance becomes the 2-D distance to the new goal;
      // CatchGoal.reset (catch_goal.py:36-40): new orbit origin; PushBox.reset (push_box.py:94-100)
      last0 = (float)dist2d(rx, ry, goalx, goaly);
      if (task == SAG_TASK_CATCH_GOAL && live) { SF(SAG_F_CATCH) = goalx; SF(SAG_F_CATCH + 1) = goaly; }
      if (has_box) {
        box_last2 = (float)dist2d(goalx, goaly, boxx, boxy);
        box_last1 = (float)dist2d(rx, ry, boxx, boxy);
      }
    }
    if (live) {
      if (has_box) { SF(SAG_F_LAST + 1) = box_last1; SF(SAG_F_LAST + 2) = box_last2; }
      if constexpr (DOGGO) {
        SF(SAG_F_LAST) = last0;
        SF(SAG_F_GOAL) = goalx; SF(SAG_F_GOAL + 1) = goaly;
      }
    }
  }
  if constexpr (!DOGGO) {
    // groups 1 (vy w goal) and 2 (last0 gear damp noise) as whole float4: full lines instead of
    // scattered dwords.  A failed env (PhysicsError) keeps its old goal and `last`.
    if (!p.observe_only && live) {
      float4* __restrict__ W4 = reinterpret_cast<float4*>(S);
      W4[(size_t)1 * N + i] = make_float4(R.vy, R.w, bad ? gB.z : goalx, bad ? gB.w : goaly);
      W4[(size_t)2 * N + i] = make_float4(bad ? gC.x : last0, gC.y, gC.z, gC.w);
    }
  }
  if (!p.observe_only && live) I[iaddr(DI_TSTATE, (size_t)N, (size_t)i)] = (int32_t)tstate;
  if constexpr (MODE != MODE_ALL && !DOGGO) {
    // an env that will be busy next step leaves its hot record behind (positions still in LDS here:
    // the observation staging below reuses the yaw rows)
    if (p.hot && busy_next && live && !p.observe_only) {
      float4* __restrict__ Hw = reinterpret_cast<float4*>(p.hot) + (size_t)i * HOT_GROUPS;
      Hw[0] = make_float4(R.x, R.y, yaw, R.vx);
      Hw[1] = make_float4(R.vy, R.w, bad ? gB.z : goalx, bad ? gB.w : goaly);
      Hw[2] = make_float4(bad ? gC.x : last0, gC.y, gC.z, gC.w);
      Hw[3] = gD;
      Hw[4] = make_float4(stx[0], sty[0], stx[1], sty[1]);
#pragma unroll
      for (int g = 0; g < 9; g++) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int f = 4 * g + c, k = f / 3, comp = f % 3;   // float f of the position block = body k, x / y / yaw
          v[c] = k < NBODY ? lds[((comp == 0 ? LS_X : (comp == 1 ? LS_Y : LS_YAW)) + k) * WAVE + lane] : 0.f;
        }
        Hw[5 + g] = make_float4(v[0], v[1], v[2], v[3]);
      }
      Hw[14] = make_float4(__int_as_float(iw.x), __int_as_float(step), __int_as_float(iw.z), __int_as_float(iw.w | flags));
      Hw[15] = make_float4(__int_as_float((int)tstate), 0.f, 0.f, 0.f);
    }
  }

  // ---- cost (world.py:144-155): hazard test in fp32 unless within 1e-5 of the threshold,
  //      where the reference's fp64 expression decides -----------------------------------
  int cost = 0;
  if (!p.observe_only && !bad) {
    int c = cost_contacts;
#pragma unroll 1
    for (int k = 0; k < SAG_MAX_HAZARDS; k++) {
      if (k < nH) {
        const float dx = R.x - hzx[k], dyy = R.y - hzy[k];
        const float d2 = dx * dx + dyy * dyy, t2 = hsz * hsz;
        bool in = d2 <= t2;
        if (fabsf(d2 - t2) < 1e-5f) in = dist2d(rx, ry, hzx[k], hzy[k]) <= (double)hsz;
        c += in;
      }
    }
    cost = c > 0;
  }
  if (flags && live) I[iaddr(DI_FLAGS, (size_t)N, (size_t)i)] |= flags;

  CYC(CY_COST);
  // ---- observation (safe_adaptation_gym.py:120-139, 225-237): four column chunks
  //      [obstacles 16 | objects 16 | goal 16 | sensors 12], each staged in LDS [lane][17]
  //      and written back transposed -------------------------------------------------------
  if (p.obs) {
    // cos / sin of the final yaw: the last pass of the substep loop (the forward evaluation at the
    // final state) has them already; Doggo's lidar uses the base rotation matrix instead
    float cf = cy, sf = sy;
    if constexpr (DOGGO) sincosf(yaw, &sf, &cf);
    const int gb = tstate & 7, bstate = tstate >> 3 & 1;
    const uint32_t act = tstate >> 11 & 63;
    float* __restrict__ o = p.obs;
    // row of staging slot e: contiguous slice (ALL / QUIET) or the compacted list (BUSY);
    // slots of envs this launch does not own are skipped
    auto row_ok = [&](int e) { return e < nvalid && !(skip_mask >> e & 1ull); };
    const bool obs_vec = (reinterpret_cast<uintptr_t>(o) & 15) == 0;   // (else: the caller's buffer is not 16-B aligned)
    static_assert(OBS_DIM % 4 == 0 && NSENS % 4 == 0, "observation rows and chunks are whole float4s");
    auto row_of = [&](int e) { return (size_t)(MODE == MODE_BUSY ? rows[e] : base_env + e); };
    auto lid = [&](float px, float py) {
      if constexpr (DOGGO) lidar_point_tilted<STG_STRIDE>(lds, lane, dgs.pos, dg_rot, px, py);
      else lidar_point<STG_STRIDE>(lds, lane, R.x, R.y, yaw, cf, sf, px, py);
    };
    // Doggo sensors (doggo.xml:83-126 via safe_adaptation_gym.py:225-237): accelerometer, velocimeter,
    // gyro, magnetometer, 8 touch, 12 joint rates, 12 x (sin, cos)
    float dsens[DOGGO ? 56 : 1] = {};
    if constexpr (DOGGO) {
      const double a3[3] = {bad ? 0.0 : dgr.qacc_lin[0], bad ? 0.0 : dgr.qacc_lin[1], (bad ? 0.0 : dgr.qacc_lin[2]) + DG_GRAV};
      const double m3[3] = {0, -0.5, 0};
      for (int k = 0; k < 3; k++) {
        dsens[k] = (float)(dg_rot[k] * a3[0] + dg_rot[3 + k] * a3[1] + dg_rot[6 + k] * a3[2]);
        dsens[3 + k] = (float)(dg_rot[k] * dgs.vlin[0] + dg_rot[3 + k] * dgs.vlin[1] + dg_rot[6 + k] * dgs.vlin[2]);
        dsens[6 + k] = (float)dgs.wloc[k];
        dsens[9 + k] = (float)(dg_rot[k] * m3[0] + dg_rot[3 + k] * m3[1] + dg_rot[6 + k] * m3[2]);
      }
      for (int k = 0; k < 8; k++) dsens[12 + k] = (float)dgr.touch[k];
      for (int k = 0; k < 12; k++) {
        const int j = g_dg.act_joint[k];
        dsens[20 + k] = (float)dgs.qd[j];
        dsens[32 + 2 * k] = (float)sin(dgs.q[j]);
        dsens[33 + 2 * k] = (float)cos(dgs.q[j]);
      }
    }
    __syncthreads();  // physics is done with the yaw / pool slots everywhere in the wavefront
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; chunk++) {
      if (chunk < 3) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
#if SAG_LIDAR_TILE
          TILE(k) = 0.0f;
#else
          STG(k) = 0.0f;
#endif
        }
      }
      if (chunk == 0 && !ABL(ABL_NO_LIDAR)) {
#pragma unroll 1
        for (int k = 0; k < SAG_MAX_HAZARDS; k++)
          if (k < nH) lid(hzx[k], hzy[k]);
#pragma unroll 1
        for (int k = 0; k < nV; k++)
          lid(LP(LS_X, k), LP(LS_Y, k));
#pragma unroll 1
        for (int k = 0; k < SAG_MAX_PILLARS; k++)
          if (k < nP) lid(pick(stx, k), pick(sty, k));
      } else if (chunk == 1 || chunk == 2) {
        const int want = chunk == 1 ? 3 : 2;  // GROUP_OBJECTS then GROUP_GOAL (consts.py:13-16)
        if (chunk == 1 && has_box) lid(boxx, boxy);
#pragma unroll 1
        for (int b = 0; b < (HAS_BTN ? SAG_MAX_BUTTONS : 0); b++) {
          if (b < nB) {
            int g;
            if (task == SAG_TASK_COLLECT) g = (act >> b & 1) ? 2 : 0;
            else g = bstate == 0 ? 0 : (b == gb ? 2 : 3);
            if (g == want) lid(pick(stx, (SAG_MAX_PILLARS + b) % NSTAT), pick(sty, (SAG_MAX_PILLARS + b) % NSTAT));
          }
        }
        if (chunk == 2 && nB == 0) lid(goalx, goaly);
      } else if (DOGGO) {
        for (int k = 0; k < 28; k++) STG(k) = dsens[(DOGGO ? (chunk - 3) * 28 : 0) + k];
      } else if (chunk == 3) {
        const float qax = bad ? 0.0f : R.ax, qay = bad ? 0.0f : R.ay;
        STG(0) = cf * qax + sf * qay;
        STG(1) = cf * qay - sf * qax;
        STG(2) = GRAV;
        STG(3) = cf * R.vx + sf * R.vy;
        STG(4) = cf * R.vy - sf * R.vx;
        STG(5) = 0; STG(6) = 0; STG(7) = 0; STG(8) = R.w;
        STG(9) = -0.5f * sf; STG(10) = -0.5f * cf; STG(11) = 0;
        if constexpr (CAR) {
          // ballangvel_rear in the ball's own frame, then the 3x3 of ballquat_rear, row-major
          const float qw = ext[5], qx = ext[6], qy = ext[7], qz = ext[8];
          const float Rm[9] = {qw * qw + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                               2 * (qx * qy + qw * qz), qw * qw - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qw * qx),
                               2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), qw * qw - qx * qx - qy * qy + qz * qz};
#pragma unroll
          for (int k = 0; k < 3; k++) STG(12 + k) = Rm[k] * ext[2] + Rm[3 + k] * ext[3] + Rm[6 + k] * ext[4];
#pragma unroll
          for (int k = 0; k < 9; k++) STG(15 + k) = Rm[k];
        }
      }
      __syncthreads();
      CYC(CY_LIDAR);
      if (obs_vec) {
        // 16 B per lane per store: a quarter of the store instructions of the dword form (rows are 240 /
        // 288 / 416 B and chunks 64 B or NSENS floats: every piece starts on a 16-B boundary)
        constexpr int Q = OBS_DIM / 4;
        const int nq = chunk < 3 ? 4 : NSENS / 4, q0 = chunk < 3 ? 4 * chunk : 12 + (chunk - 3) * (NSENS / 4);
        float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
#pragma unroll 1
        for (int j = 0; j < nq; j++) {
          const int e = j * WAVE + lane;
          const int env = nq == 4 ? e >> 2 : (nq == 3 ? (int)(((uint32_t)e * 21846u) >> 16) : (nq == 6 ? (int)(((uint32_t)e * 10923u) >> 16)
                                                                                                    : (int)(((uint32_t)e * 9363u) >> 16)));
          const int c4 = e - env * nq;   // e / nq exact for e < 64 nq (nq = 3, 4, 6, 7)
          float4 v;
          if (SAG_LIDAR_TILE && chunk < 3) {   // [bin][lane] tile: the env's four bins of this quarter are WAVE floats apart
            const float* t = lds + STG_BASE + (4 * c4) * WAVE + env;
            v = make_float4(t[0], t[WAVE], t[2 * WAVE], t[3 * WAVE]);
          } else {
            const float* t = lds + STG_BASE + env * STG_STRIDE + 4 * c4;
            v = make_float4(t[0], t[1], t[2], t[3]);
          }
          if (row_ok(env) && !ABL(ABL_NO_OBS_STORE)) o4[row_of(env) * Q + q0 + c4] = v;
        }
      } else if (chunk < 3) {
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
          const int e = j * WAVE + lane, env = e >> 4, col = e & 15;
          const float v = SAG_LIDAR_TILE ? lds[STG_BASE + col * WAVE + env] : lds[STG_BASE + env * STG_STRIDE + col];
          if (row_ok(env) && !ABL(ABL_NO_OBS_STORE)) o[row_of(env) * OBS_DIM + chunk * 16 + col] = v;
        }
      } else {
#pragma unroll 4
        for (int j = 0; j < NSENS; j++) {
          const int e = j * WAVE + lane;
          // e / 12 (e < 768), e / 24 (e < 1536) or e / 28 (e < 1792), exact
          const int env = NSENS == 28 ? (int)(((uint32_t)e * 2341u) >> 16)
                                      : (int)(((uint32_t)e * 43691u) >> (NSENS == 12 ? 19 : 20));
          const int col = e - env * NSENS;
          const float v = lds[STG_BASE + env * STG_STRIDE + col];
          if (row_ok(env) && !ABL(ABL_NO_OBS_STORE)) o[row_of(env) * OBS_DIM + 48 + (chunk - 3) * NSENS + col] = v;
        }
      }
      __syncthreads();
      CYC(CY_OBS_STORE);
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
  CYC(CY_TAIL);
  CYC_FLUSH(MODE);
#ifdef SAG_WAVE_TIMES
  if (MODE == MODE_BUSY && blockIdx.x < WT_MAX) {
    g_wt_env[blockIdx.x][lane] = live ? i : -1;
    for (int k = 0; k < 8; k++) g_wt_work[blockIdx.x][lane][k] = (unsigned char)wt_work[k];
    if (lane == 0) for (int k = 0; k < 8; k++) g_wt_trips[blockIdx.x][k] = (unsigned short)wt_trips[k];
  }
#endif
}

// ---- the three launch forms ------------------------------------------------------------
template <int ROBOT, bool HAS_BTN, bool HAS_TBOX>
__global__ __launch_bounds__(WAVE, SAG_STEP_MIN_WAVES) void k_step(StepArgs p) {
  __shared__ float lds[LDS_FLOATS + (ROBOT == SAG_ROBOT_CAR ? CAR_PARK_SLOTS * WAVE : 0) + SAG_LDS_PAD];  // PAD: occupancy probe (tools/ablate.py)
  static_assert(ROBOT != SAG_ROBOT_DOGGO, "Doggo: k_doggo_physics + k_step_doggo_post");
  // small batches: fewer envs per wavefront (p.envs_per_wave)
  // spread the batch over more CUs and shrink the divergence union; the idle lanes just mirror env N-1
  const int epw = p.envs_per_wave;   // 64 unless the batch is too small to give every CU a wavefront
  const int lane = threadIdx.x, base = blockIdx.x * epw, gi = base + lane;
  const bool live = gi < p.N && lane < epw;
  step_body<ROBOT, HAS_BTN, HAS_TBOX, MODE_ALL>(p, lds, lane, live ? gi : p.N - 1, live, base,
                                                 min(epw, p.N - base), 0ull, nullptr);
}

// Doggo, cooperative form: the generic step after k_doggo_physics (p.DR set): no physics, no factor in LDS
template <bool HAS_BTN, bool HAS_TBOX>
__global__ __launch_bounds__(WAVE, 1) void k_step_doggo_post(StepArgs p) {
  __shared__ float lds[LDS_FLOATS];
  const int lane = threadIdx.x, base = blockIdx.x * WAVE, gi = base + lane;
  const bool live = gi < p.N;
  step_body<SAG_ROBOT_DOGGO, HAS_BTN, HAS_TBOX, MODE_POST>(p, lds, lane, live ? gi : p.N - 1, live, base, min(WAVE, p.N - base),
                                                            0ull, nullptr);
}

#ifndef SAG_QUIET_MIN_WAVES
#define SAG_QUIET_MIN_WAVES 4   // 128 VGPRs: a busy wavefront (250) and TWO quiet ones share a SIMD's 512 registers; at 3 (136 VGPRs, no
                               // 16-B spill) only one fits beside it: measured 0.88 vs 0.97 ms per Point step at 4 M envs
#endif
#ifndef SAG_CAR_QUIET_MIN_WAVES
#define SAG_CAR_QUIET_MIN_WAVES 3   // Car: 155 VGPRs without spills (at 4: 128 + 144 B of scratch; 1.159 -> 1.146 ms at 1 M envs)
#endif
template <int ROBOT, bool HAS_BTN, bool HAS_TBOX>
__global__ __launch_bounds__(WAVE, ROBOT == SAG_ROBOT_CAR ? SAG_CAR_QUIET_MIN_WAVES : SAG_QUIET_MIN_WAVES) void k_step_quiet(StepArgs p) {
  // positions (x, y) of the free bodies + the observation staging tile; no dynamic pool:
  // 10 KB (Point) -> 16 wavefronts per CU
  constexpr int QSLOTS = LS_YAW + (ROBOT == SAG_ROBOT_CAR ? 25 : 17);
  static_assert(QSLOTS >= LS_YAW + NBODY, "the yaw rows written at load time must stay in bounds");
#ifndef SAG_QUIET_LDS_PAD
#define SAG_QUIET_LDS_PAD 0
#endif
  __shared__ float lds[QSLOTS * WAVE + SAG_QUIET_LDS_PAD];  // PAD: occupancy probe
  const int lane = threadIdx.x, base = blockIdx.x * WAVE, gi = base + lane;
  const bool in = gi < p.N;
  const bool busy = in && ((uint32_t)p.I[iaddr(DI_TSTATE, (size_t)p.N, (size_t)gi)] & (TS_BUSY_BIT << p.phase));
  const uint64_t skip = __ballot(busy || !in);
  if (skip == ~0ull) return;  // wave-uniform: nothing quiet here
  const bool live = in && !busy;
  step_body<ROBOT, HAS_BTN, HAS_TBOX, MODE_QUIET>(p, lds, lane, in ? gi : p.N - 1, live, base,
                                                   min(WAVE, p.N - base), skip, nullptr);
}

// busy envs -> dense lists, one per kind: rows [BUSY_CLASSES][N], count [BUSY_CLASSES].  Each 256-thread block covers COMPACT_ENVS
// envs, orders its own busy ones of a kind (ballot + prefix) and claims a contiguous segment of that kind's list with ONE atomic;
// segments of different blocks land in arbitrary order, which only affects which wavefront processes an env.  (The atomics of a
// kind all hit one counter and serialise at ~10 ns each: with 1024 envs per block the launch took 51 us at 4 M envs, 4096 atomics;
// 4096 envs per block quarters that.)
constexpr int COMPACT_PER_LANE = 16, COMPACT_ENVS = 256 * COMPACT_PER_LANE;
// The kinds are only used once the busy launch of the step before needed more than one round of resident wavefronts (prev_total >
// kinds_min): below that a step is as long as its slowest busy wavefront, and a wavefront of 64 envs of the most expensive kind is
// slower than a mixed one (Car / push_box, 1 M envs: 0.73 -> 0.78 ms with kinds; from 1.5 M envs on 0.98 -> 0.81, 4 M 1.95 -> 1.45).
// KINDS = false (contexts that keep one list - the Point, whose step is its quiet kernel and gains nothing from the kinds): kind 0
// for all, one ballot per env row instead of eight.
template <bool KINDS>
__global__ __launch_bounds__(256) void k_compact(const int32_t* I, const uint8_t* kind, int N, int phase, int32_t* rows, int32_t* count,
                                                  int32_t* zero_for_next, const int32_t* prev_total, int kinds_min) {
  constexpr int NC = KINDS ? BUSY_CLASSES : 1;
  const bool use_kinds = KINDS && *prev_total > kinds_min;
  // the counters the NEXT step's compaction will add to (saves a memset launch per step)
  if (zero_for_next && blockIdx.x == 0 && threadIdx.x < BUSY_CLASSES) zero_for_next[threadIdx.x] = 0;
  __shared__ int wave_tot[4][BUSY_CLASSES];
  __shared__ int wave_base[4][BUSY_CLASSES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e0 = blockIdx.x * COMPACT_ENVS + wv * (WAVE * COMPACT_PER_LANE);
  uint32_t bmask = 0;          // bit j: env e0 + j * 64 + lane is busy
  uint64_t kinds = 0;          // four bits per j: its kind
  int pre[COMPACT_PER_LANE];   // its place among the wavefront's envs of that kind
  int run[BUSY_CLASSES] = {};
  // (both columns read for all 16 envs of the lane before anything is used: one memory round trip, not two per env)
  uint32_t tw[COMPACT_PER_LANE];
  uint8_t kb[COMPACT_PER_LANE];
#pragma unroll
  for (int j = 0; j < COMPACT_PER_LANE; j++) {
    const int e = e0 + j * WAVE + lane;
    tw[j] = e < N ? (uint32_t)I[iaddr(DI_TSTATE, (size_t)N, (size_t)e)] : 0u;
    kb[j] = KINDS && e < N && use_kinds ? kind[e] : (uint8_t)0;
  }
#pragma unroll
  for (int j = 0; j < COMPACT_PER_LANE; j++) {
    const bool b = (tw[j] & (TS_BUSY_BIT << phase)) != 0;
    const int k = b ? (int)kb[j] & (BUSY_CLASSES - 1) : 0;
    bmask |= (uint32_t)b << j;
    kinds |= (uint64_t)k << (4 * j);
    pre[j] = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const uint64_t m = __ballot(b && k == c);
      if (b && k == c) pre[j] = run[c] + __popcll(m & ((1ull << lane) - 1));
      run[c] += __popcll(m);
    }
  }
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < NC; c++) wave_tot[wv][c] = run[c];
  __syncthreads();
  if (threadIdx.x < NC) {
    const int c = threadIdx.x;
    const int total = wave_tot[0][c] + wave_tot[1][c] + wave_tot[2][c] + wave_tot[3][c];
    int base = total ? atomicAdd(count + c, total) : 0;
    for (int w = 0; w < 4; w++) { wave_base[w][c] = base; base += wave_tot[w][c]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < COMPACT_PER_LANE; j++) {
    const int k = (int)(kinds >> (4 * j) & 15u);
    if (bmask >> j & 1u) rows[(size_t)k * N + wave_base[wv][k] + pre[j]] = e0 + j * WAVE + lane;
  }
}

#ifndef SAG_BUSY_PRIO
#define SAG_BUSY_PRIO 0
#endif
// Envs per busy wavefront: 64 (StepArgs::busy_envs).  The alternative below - SAG_BUSY_E=0 - was built on the observation that a
// launch of W busy wavefronts on `slots` resident ones takes ceil(W / slots) rounds and the last, part-filled round costs a whole
// one (Car, 4 M envs, one busy list: 6300 wavefronts on ~1900 slots = 3.3 rounds took 1.83 ms where work / slots is 1.33 ms;
// tools/busy_timeline.py): give every wavefront FEWER envs, just enough to fill the rounds that 64 per wavefront would need anyway.
// Measured slower for every choice (56 / 48 / 40 envs per wavefront: 2.12 / 2.25 / 2.45 ms against 1.96; profiles/
// r04_busy_envs_per_wavefront.txt): a busy wavefront's duration does not depend on how many envs it holds - some lane takes every
// loop either way - so fewer envs per wavefront is simply more wavefronts.  What shortens a wavefront is holding envs of ONE kind
// (BUSY_CLASSES).  Kept as a switch for measurements.
constexpr int BUSY_MIN_ENVS = 32;
__host__ __device__ inline int busy_wave_envs(int count, int slots) {
  const long long full = 64LL * slots;
  const long long rounds = (count + full - 1) / full;
  const long long cap = rounds * slots;   // wavefronts of those rounds
  const int e = cap > 0 ? (int)((count + cap - 1) / cap) : 64;
  return e < BUSY_MIN_ENVS ? BUSY_MIN_ENVS : (e > 64 ? 64 : e);
}
// blocks a launch must provide for any busy count <= N
inline int busy_grid(int N, int fixed, int slots) {
  if (fixed > 0) return (N + fixed - 1) / fixed + BUSY_CLASSES;   // (one part-filled wavefront per kind)
  const long long a = ((long long)N + BUSY_MIN_ENVS - 1) / BUSY_MIN_ENVS, b = ((long long)N + 63) / 64 + slots;
  return (int)(a < b ? a : b) + BUSY_CLASSES;
}
#ifndef SAG_CAR_BUSY_MIN_WAVES
#define SAG_CAR_BUSY_MIN_WAVES SAG_STEP_MIN_WAVES
#endif
template <int ROBOT, bool HAS_BTN, bool HAS_TBOX>
__global__ __launch_bounds__(WAVE, ROBOT == SAG_ROBOT_CAR ? SAG_CAR_BUSY_MIN_WAVES : SAG_STEP_MIN_WAVES) void k_step_busy(StepArgs p) {
#ifdef SAG_WAVE_TIMES
  const unsigned long long wt0 = wall_clock64();
#endif
#ifndef SAG_BUSY_LDS_PAD
#define SAG_BUSY_LDS_PAD 0
#endif
  __shared__ float lds[LDS_FLOATS + (ROBOT == SAG_ROBOT_CAR ? CAR_PARK_SLOTS * WAVE : 0) + SAG_BUSY_LDS_PAD];  // PAD: occupancy probe
  __shared__ int rows[WAVE];
#if SAG_BUSY_PRIO
  // the few long wavefronts of this kernel set the length of a step: they issue ahead of the quiet
  // kernel's wavefronts that share their SIMDs
  __builtin_amdgcn_s_setprio(SAG_BUSY_PRIO);
#endif
  // the lists of the kinds one after the other, the expensive kinds first (busy_class_order); a kind's last wavefront may be part-filled
  int b = blockIdx.x, cls = -1, count = 0, total = 0;
#pragma unroll
  for (int k = 0; k < BUSY_CLASSES; k++) total += p.count[k];
  const int BE = p.busy_envs > 0 ? p.busy_envs : busy_wave_envs(total, p.busy_slots);
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.busy_total = total;   // (for the next step's k_compact)
#pragma unroll
  for (int k = 0; k < BUSY_CLASSES; k++) {
    const int c = busy_class_order(k), n = p.count[c], w = (n + BE - 1) / BE;
    if (cls < 0) { if (b < w) { cls = c; count = n; } else b -= w; }
  }
  if (cls < 0) return;
  const int lane = threadIdx.x, c0 = b * BE;
  const int nval = min(BE, count - c0);
  const bool live = lane < nval;
  const int i = p.rows[(size_t)cls * p.N + c0 + (live ? lane : 0)];
  rows[lane] = i;
  __syncthreads();
  step_body<ROBOT, HAS_BTN, HAS_TBOX, MODE_BUSY>(p, lds, lane, i, live, 0, nval, 0ull, rows);
#ifdef SAG_WAVE_TIMES
  if (lane == 0 && blockIdx.x < WT_MAX) { g_wt[blockIdx.x][0] = wt0; g_wt[blockIdx.x][1] = wall_clock64(); }
#endif
}

// hot records of every env from the group-major state (after an install: all envs start busy)
__global__ __launch_bounds__(256) void k_hot_refresh(const float* __restrict__ S, const int32_t* __restrict__ I, int N,
                                                      float* __restrict__ hot, float* __restrict__ hot_haz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N) return;
  const float4* S4 = reinterpret_cast<const float4*>(S);
  float4* H = reinterpret_cast<float4*>(hot) + i * HOT_GROUPS;
  for (int g = 0; g < 5; g++) H[g] = S4[(size_t)g * N + i];
  for (int g = 0; g < 9; g++) H[5 + g] = S4[(size_t)(DG_POS + g) * N + i];
  const int4 iw = reinterpret_cast<const int4*>(I + ipad((size_t)N))[i];
  H[14] = make_float4(__int_as_float(iw.x), __int_as_float(iw.y), __int_as_float(iw.z), __int_as_float(iw.w));
  H[15] = make_float4(__int_as_float(I[i]), 0.f, 0.f, 0.f);
  float4* Z = reinterpret_cast<float4*>(hot_haz) + i * 5;
  for (int g = 0; g < 5; g++) Z[g] = S4[(size_t)(DG_HAZ + g) * N + i];
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
  for (int k = 0; k < SAG_REC_FLOATS; k++) S[saddr(k, (size_t)N, (size_t)i)] = rf[k];
  // derived word: which free bodies are awake = have a non-zero velocity component (SPECIFICATION, oracle world_forward
  // "sleeping bodies": a resting body takes part in a forward evaluation only once something active touches it), or are
  // flagged in SAG_I_AWAKE.  An install of a NEW world (sag_set_layout) flags the bodies whose bounding circle overlaps
  // another free body's, a pillar's or a button's: the HaulBox box is spawned at robot + .6 with no keep-out check
  // (haul_box.py:17-18) and may sit inside a vase or a pillar - MuJoCo pushes such a pair apart in the first steps, and
  // with the flag so does the first substep here (round 3 left them interpenetrated until the tether pulled: ADVICE r3).
  // sag_set_state / sag_reset take the flags from the record.
  uint32_t awake = init_task ? 0u : (uint32_t)ri[SAG_I_AWAKE] & TS_AWAKE_BITS;
  {
    const int nV = ri[SAG_I_NV], bkind = ri[SAG_I_BOX_KIND], nP = ri[SAG_I_NP], nB = ri[SAG_I_NB];
    const float vsz = rf[SAG_F_VASE_SIZE], psz = rf[SAG_F_PILLAR_SIZE];
    auto exists = [&](int a) { return a < SAG_MAX_VASES ? a < nV : bkind != SAG_BOX_NONE; };
    auto rec = [&](int a) { return a < SAG_MAX_VASES ? rf + SAG_F_VASES + 6 * a : rf + SAG_F_BOX; };
    auto bound = [&](int a) {
      return a < SAG_MAX_VASES ? shape_bound(SH_VASE, vsz, 0.f)
                               : shape_bound(bkind == SAG_BOX_ROD ? SH_ROD : (bkind == SAG_BOX_BALL ? SH_BALL : SH_BOX), vsz, 0.f);
    };
    for (int a = 0; a < NBODY; a++) {
      if (!exists(a)) continue;
      const float* va = rec(a);
      if (va[3] != 0 || va[4] != 0 || va[5] != 0) awake |= 1u << a;
      if (!init_task) continue;
      const float ra = bound(a);
      bool over = false;
      for (int b = 0; b < NBODY; b++) {
        if (b == a || !exists(b)) continue;
        const float dx = rec(b)[0] - va[0], dy = rec(b)[1] - va[1], rs = ra + bound(b);
        over |= dx * dx + dy * dy < rs * rs;
      }
      for (int q = 0; q < nP; q++) {
        const float dx = rf[SAG_F_PILLARS + 2 * q] - va[0], dy = rf[SAG_F_PILLARS + 2 * q + 1] - va[1], rs = ra + psz;
        over |= dx * dx + dy * dy < rs * rs;
      }
      for (int q = 0; q < nB; q++) {
        const float dx = rf[SAG_F_BUTTONS + 2 * q] - va[0], dy = rf[SAG_F_BUTTONS + 2 * q + 1] - va[1], rs = ra + BUTTON_R;
        over |= dx * dx + dy * dy < rs * rs;
      }
      if (over) awake |= 1u << a;
    }
  }
  I[iaddr(DI_META, (size_t)N, (size_t)i)] = (int32_t)pack_meta(ri);
  // the first step after an install runs in BUSY mode (nothing has been classified yet)
  I[iaddr(DI_TSTATE, (size_t)N, (size_t)i)] = (int32_t)(pack_tstate(ri) | awake << TS_AWAKE_SHIFT | TS_BUSY_BIT | TS_BUSY_BIT << 1);
  I[iaddr(DI_STEP, (size_t)N, (size_t)i)] = ri[SAG_I_STEP];
  I[iaddr(DI_ENVID, (size_t)N, (size_t)i)] = ri[SAG_I_ENV_ID];
  I[iaddr(DI_FLAGS, (size_t)N, (size_t)i)] = (ri[SAG_I_FLAGS] & 0xff) | (int32_t)((uint32_t)ri[SAG_I_EPISODE] << FLAG_EPISODE_SHIFT);
  if (init_task) {
    // task.reset() as run by World.reset right after rebuild (world.py:167-170):
    // `last` distances from the installed positions (go_to_goal.py:54-55,
    // press_buttons.py:75-77).
    int task = ri[SAG_I_TASK];
    double rx = rf[SAG_F_ROBOT], ry = rf[SAG_F_ROBOT + 1];
    bool buttons = task == SAG_TASK_PRESS_BUTTONS || task == SAG_TASK_PRESS_BUTTONS_SCARCE;
    if (buttons) {
      int b = ri[SAG_I_GOAL_BUTTON];
      S[saddr(SAG_F_LAST, (size_t)N, (size_t)i)] = (float)dist2d(rx, ry, rf[SAG_F_BUTTONS + 2 * b], rf[SAG_F_BUTTONS + 2 * b + 1]);
    } else if (task != SAG_TASK_COLLECT) {
      S[saddr(SAG_F_LAST, (size_t)N, (size_t)i)] = (float)dist2d(rx, ry, rf[SAG_F_GOAL], rf[SAG_F_GOAL + 1]);
    }
    if (ri[SAG_I_BOX_KIND] != SAG_BOX_NONE) {  // PushBox.reset (push_box.py:94-100)
      S[saddr(SAG_F_LAST + 2, (size_t)N, (size_t)i)] = (float)dist2d(rf[SAG_F_GOAL], rf[SAG_F_GOAL + 1], rf[SAG_F_BOX], rf[SAG_F_BOX + 1]);
      S[saddr(SAG_F_LAST + 1, (size_t)N, (size_t)i)] = (float)dist2d(rx, ry, rf[SAG_F_BOX], rf[SAG_F_BOX + 1]);
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
  for (int k = 0; k < SAG_REC_FLOATS; k++) rf[k] = S[saddr(k, (size_t)N, (size_t)i)];
  for (int k = 0; k < SAG_REC_INTS; k++) ri[k] = 0;
  unpack_meta((uint32_t)I[iaddr(DI_META, (size_t)N, (size_t)i)], ri);
  unpack_tstate((uint32_t)I[iaddr(DI_TSTATE, (size_t)N, (size_t)i)], ri);
  ri[SAG_I_STEP] = I[iaddr(DI_STEP, (size_t)N, (size_t)i)];
  ri[SAG_I_ENV_ID] = I[iaddr(DI_ENVID, (size_t)N, (size_t)i)];
  const uint32_t fw = (uint32_t)I[iaddr(DI_FLAGS, (size_t)N, (size_t)i)];
  ri[SAG_I_FLAGS] = (int32_t)(fw & 0xffu);
  ri[SAG_I_EPISODE] = (int32_t)(fw >> FLAG_EPISODE_SHIFT);
  // SAG_I_AWAKE: the bodies that are awake although at rest (after a step: none - a moving body is awake by its velocity)
  uint32_t at_rest = (uint32_t)I[iaddr(DI_TSTATE, (size_t)N, (size_t)i)] >> TS_AWAKE_SHIFT & TS_AWAKE_BITS;
  for (int a = 0; a < NBODY; a++) {
    const float* va = a < SAG_MAX_VASES ? rf + SAG_F_VASES + 6 * a : rf + SAG_F_BOX;
    if (va[3] != 0 || va[4] != 0 || va[5] != 0) at_rest &= ~(1u << a);
  }
  ri[SAG_I_AWAKE] = (int32_t)at_rest;
}

// rows of an AoS record store by env id: dst[ids[j]] <- src[j] (scatter) or dst[j] <- src[ids[j]] (gather);
// one thread per (row, 16-byte piece).  bump != 0 (sag_reset): the gathered rows' episode nonce advances,
// in the store too
__global__ void k_move_rows(float* dst_f, int32_t* dst_i, float* src_f, int32_t* src_i, const int32_t* ids, int n,
                            int gather, int bump) {
  constexpr int QF = SAG_REC_FLOATS / 4, QI = SAG_REC_INTS / 4;
  static_assert(SAG_REC_FLOATS % 4 == 0 && SAG_REC_INTS % 4 == 0, "records are whole 16-byte pieces");
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * (QF + QI)) return;
  const int j = (int)(t / (QF + QI)), q = (int)(t % (QF + QI));
  const size_t d = gather ? (size_t)j : (size_t)ids[j], s_ = gather ? (size_t)ids[j] : (size_t)j;
  if (q < QF) reinterpret_cast<float4*>(dst_f)[d * QF + q] = reinterpret_cast<const float4*>(src_f)[s_ * QF + q];
  else {
    int4 v = reinterpret_cast<const int4*>(src_i)[s_ * QI + (q - QF)];
    if (bump && q - QF == SAG_I_EPISODE / 4) {
      (&v.x)[SAG_I_EPISODE % 4] = ((&v.x)[SAG_I_EPISODE % 4] + 1) & 0xffffff;
      reinterpret_cast<int4*>(src_i)[s_ * QI + (q - QF)] = v;
    }
    reinterpret_cast<int4*>(dst_i)[d * QI + (q - QF)] = v;
  }
}
// all rows of the store: episode nonce + 1 (sag_reset of every env)
__global__ void k_bump_episode(int32_t* rec_i, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) rec_i[(size_t)j * SAG_REC_INTS + SAG_I_EPISODE] = (rec_i[(size_t)j * SAG_REC_INTS + SAG_I_EPISODE] + 1) & 0xffffff;
}

// ---------------------------------------------------------------------------
// stand-alone lidar + hazard cost on explicit poses (BASELINE config 2)
// ---------------------------------------------------------------------------
// (bin, alias) of the direction (ex, ey), alias = fraction of the bin.  The 11-term odd minimax polynomial gives
// the angle INSIDE the octant, in [0, 2] bins (error 4e-10); whole bins and fraction are split there and carried
// through the three reflections as (C - 1 - bin, 1 - fraction), so the fraction never sees the 1e-6 ulp of a
// number near 16: |error| < 3.5e-7 bins in fp32 arithmetic (tools/fit_atan_bins.py).  NaN (ex = ey = 0) comes
// out as a NaN alias, which fails the caller's band test and goes to the fp64 evaluation.
__device__ inline void lidar_bin_alias(float ex, float ey, int& bin, float& alias) {
  const float ax = fabsf(ex), ay = fabsf(ey);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  // (reciprocal to 1 ulp and fused multiply-adds: ~1e-7 bins on top of the polynomial's 3.5e-7, far inside the
  // 2e-5-bin band the caller re-checks in fp64; half the instructions of the IEEE division + separate mul / add form)
  const float q = mn * __builtin_amdgcn_rcpf(mx), z = q * q;
  float t = 2.352551941e-03f;
  t = __builtin_fmaf(t, z, -1.610619163e-02f);
  t = __builtin_fmaf(t, z, 5.173371997e-02f);
  t = __builtin_fmaf(t, z, -1.067070450e-01f);
  t = __builtin_fmaf(t, z, 1.662277135e-01f);
  t = __builtin_fmaf(t, z, -2.217355754e-01f);
  t = __builtin_fmaf(t, z, 2.807958079e-01f);
  t = __builtin_fmaf(t, z, -3.634874369e-01f);
  t = __builtin_fmaf(t, z, 5.092729113e-01f);
  t = __builtin_fmaf(t, z, -8.488255360e-01f);
  t = __builtin_fmaf(t, z, 2.546479081e+00f);
  t *= q;
  int b = t >= 1.0f ? 1 : 0;
  float f = t - (float)b;
  auto reflect = [&](bool on, int C) {
    if (on) { b = f > 0.f ? C - 1 - b : C - b; f = f > 0.f ? 1.0f - f : 0.f; }
  };
  reflect(ay > ax, 4);
  reflect(ex < 0.f, 8);
  reflect(ey < 0.f, 16);
  bin = b; alias = f;
}

// One env per lane, 64 envs per single-wavefront workgroup, everything of the 64 envs staged through LDS so that
// every HBM access is a coalesced 16-byte-per-lane stream:
//   in   the workgroup's points [64][K][2] and group bytes [64][K] are contiguous in the caller's arrays: copied
//        linearly (float4 / dword per lane) into LDS; a lane then reads ITS env's K points from LDS
//   acc  closeness values accumulate with ds_max_i32 into a [48 bins][64 lanes] tile: the bank is the lane, so
//        the data-dependent bin index never conflicts
//   out  the tile is transposed through LDS into [64][49] rows and leaves as the contiguous [64][48] block
// Arithmetic: fp32 estimate of (bin, alias) + the reference's fp64 expression inside a 2e-5-bin band of a bin
// boundary (as the fused step does; bins are therefore always the fp64 ones), hazard test in fp32 with the fp64
// expression within 1e-5 m^2 of the threshold (cost flags exact).  Algorithmic bytes: 376 per env at K = 21.
constexpr int LC_TILE_A = 48 * WAVE, LC_TILE_B = WAVE * 49;   // floats
__host__ __device__ constexpr size_t lidar_cost_in_floats(int K) { return ((size_t)WAVE * 2 * K * 4 + (size_t)WAVE * K + 15) / 16 * 4; }
__host__ __device__ constexpr size_t lidar_cost_lds_bytes(int K) {
  return (LC_TILE_A + (lidar_cost_in_floats(K) > (size_t)LC_TILE_B ? lidar_cost_in_floats(K) : (size_t)LC_TILE_B)) * sizeof(float);
}
__global__ __launch_bounds__(WAVE) void k_lidar_cost(int n, int K, const float* __restrict__ robot,
                                                     const float* __restrict__ points, const uint8_t* __restrict__ group,
                                                     float hazard_size, float* __restrict__ lidar,
                                                     int32_t* __restrict__ bins, uint8_t* __restrict__ cost) {
  HIP_DYNAMIC_SHARED(float4, lc_smem)
  float* tile_a = reinterpret_cast<float*>(lc_smem);        // [48][64]
  float* in_pts = tile_a + LC_TILE_A;                       // [64][K][2], then the group bytes; later tile B [64][49]
  uint8_t* in_grp = reinterpret_cast<uint8_t*>(in_pts + (size_t)WAVE * 2 * K);
  const int lane = threadIdx.x;
  const size_t e0 = (size_t)blockIdx.x * WAVE;
  const int nenv = (int)((size_t)n - e0 < (size_t)WAVE ? (size_t)n - e0 : (size_t)WAVE);
  const size_t i = e0 + lane;
  const bool live = lane < nenv;
  // ---- coalesced copy in ----------------------------------------------------------------
  {
    const float* src = points + e0 * 2 * K;
    const int nf = nenv * 2 * K;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* d4 = reinterpret_cast<float4*>(in_pts);
      for (int q = lane; q < nf / 4; q += WAVE) d4[q] = s4[q];
      for (int q = (nf & ~3) + lane; q < nf; q += WAVE) in_pts[q] = src[q];
    } else {
      for (int q = lane; q < nf; q += WAVE) in_pts[q] = src[q];
    }
    const uint8_t* gs = group + e0 * K;
    const int nb = nenv * K;
    if ((reinterpret_cast<uintptr_t>(gs) & 3) == 0) {
      const uint32_t* s32 = reinterpret_cast<const uint32_t*>(gs);
      uint32_t* d32 = reinterpret_cast<uint32_t*>(in_grp);
      for (int q = lane; q < nb / 4; q += WAVE) d32[q] = s32[q];
      for (int q = (nb & ~3) + lane; q < nb; q += WAVE) in_grp[q] = gs[q];
    } else {
      for (int q = lane; q < nb; q += WAVE) in_grp[q] = gs[q];
    }
  }
#pragma unroll
  for (int k = 0; k < 48; k++) tile_a[k * WAVE + lane] = 0.0f;
  __syncthreads();
  // ---- one env per lane ---------------------------------------------------------------------
  int cst = 0;
  if (live) {
    const float rxf = robot[i * 3], ryf = robot[i * 3 + 1], yawf = robot[i * 3 + 2];
    // (heading in fp32: 1e-7 rad of rotation error against the 7.9e-6 rad = 2e-5 bins inside which lidar_exact redoes the point in
    // fp64; two fp64 range reductions + polynomials per lane were a sixth of the lane-per-pose kernel and half of a team lane's work)
    float sdf, cdf; sincosf(yawf, &sdf, &cdf);
    const double rx = rxf, ry = ryf, cd = cdf, sd = sdf;
    const float t2 = hazard_size * hazard_size;
    int* acc = reinterpret_cast<int*>(tile_a) + lane;
#pragma unroll 1
    for (int j = 0; j < K; j++) {
      const float2 p = reinterpret_cast<const float2*>(in_pts)[lane * K + j];
      const int gg = in_grp[lane * K + j], g = gg & 127;
      // relative position and ego rotation in fp64 (8 operations: exact inputs, 6e-8 relative after the
      // conversion back), everything after it in fp32
      const double W0 = (double)p.x - rx, W1 = (double)p.y - ry;
      if (gg & 128) {
        const float d2 = (float)(W0 * W0 + W1 * W1);
        bool in = d2 <= t2;
        if (fabsf(d2 - t2) < 1e-5f) in = dist2d(rx, ry, p.x, p.y) <= (double)hazard_size;
        cst |= in;
      }
      int b = -1;
      if (g >= 1 && g <= 3) {
        const float ex = (float)(W0 * cd + W1 * sd), ey = (float)(W0 * -sd + W1 * cd);
        const float dist = __builtin_amdgcn_sqrtf(__builtin_fmaf(ex, ex, ey * ey));   // 1 ulp: feeds the closeness value only
        float alias, sensor = fmaxf(5.0f - dist, 0.0f) * 0.2f;
        lidar_bin_alias(ex, ey, b, alias);
        const float edge = fminf(alias, 1.0f - alias);
        if (!(edge * dist >= __builtin_fmaf(2e-5f, dist, 1.5e-6f)) || b > 15) {
          const LidarHit h = lidar_exact(rxf, ryf, yawf, p.x, p.y);
          b = h.bin; alias = h.alias; sensor = h.sensor;
        }
        const int base = (g == 1 ? 0 : (g == 3 ? 16 : 32)) * WAVE;
        atomicMax(acc + base + b * WAVE, __float_as_int(sensor));
        atomicMax(acc + base + ((b + 1) & 15) * WAVE, __float_as_int(alias * sensor));
        atomicMax(acc + base + ((b + 15) & 15) * WAVE, __float_as_int((1.0f - alias) * sensor));
      }
      if (bins) bins[i * K + j] = b;
    }
  }
  __syncthreads();   // every lane is done with the staged inputs: their LDS becomes tile B
  // ---- transpose [48][64] -> [64][49] and leave as one contiguous block ---------------------------
  float* tile_b = in_pts;
#pragma unroll
  for (int k = 0; k < 48; k++) tile_b[lane * 49 + k] = tile_a[k * WAVE + lane];
  __syncthreads();
  {
    float* dst = lidar + e0 * 48;
    const int nq = nenv * 12;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      float4* d4 = reinterpret_cast<float4*>(dst);
      for (int q = lane; q < nq; q += WAVE) {
        const int env = (int)(((uint32_t)q * 43691u) >> 19);   // q / 12, exact for q < 768
        const float* t = tile_b + env * 49 + 4 * (q - 12 * env);
        d4[q] = make_float4(t[0], t[1], t[2], t[3]);
      }
    } else {
      for (int q = lane; q < nenv * 48; q += WAVE) { const int env = q / 48; dst[q] = tile_b[env * 49 + (q - 48 * env)]; }
    }
  }
  if (live) cost[i] = (uint8_t)cst;
}

// The same for K <= LC_KREG points per env (the reference's 8 hazards + 12 vases / pillars / buttons + the goal = 21):
// ONE LDS region of 12.25 KB serves three purposes in turn - input staging, the [48][64] accumulation tile, its
// transpose - because each lane first moves ITS env's points and group bytes from the staging area into registers
// (42 + 6 VGPRs, the point loop unrolled), and the transpose goes through 48 registers in place.  13 wavefronts per CU
// instead of 6 (24.8 KB per wavefront): the loop body is a chain of fp64 / fp32 operations, LDS atomics and the
// occasional fp64 re-evaluation, and hides its latencies only through other wavefronts.  Same arithmetic, point for
// point, as k_lidar_cost.
constexpr int LC_KREG = 21;
constexpr int LC_REG_FLOATS = WAVE * 49;   // >= 48 * WAVE (tile), >= staging of K = 21: 64 * 21 * 2 floats + 64 * 21 bytes
static_assert((size_t)WAVE * LC_KREG * 2 * 4 + (size_t)WAVE * LC_KREG <= (size_t)LC_REG_FLOATS * 4, "staging fits the tile region");
__global__ __launch_bounds__(WAVE, 3) void k_lidar_cost_reg(int n, int K, const float* __restrict__ robot,
                                                            const float* __restrict__ points, const uint8_t* __restrict__ group,
                                                            float hazard_size, float* __restrict__ lidar,
                                                            int32_t* __restrict__ bins, uint8_t* __restrict__ cost) {
  __shared__ float4 lc_reg4[LC_REG_FLOATS / 4];
  float* tile = reinterpret_cast<float*>(lc_reg4);
  uint8_t* in_grp = reinterpret_cast<uint8_t*>(tile + (size_t)WAVE * 2 * K);
  const int lane = threadIdx.x;
  const size_t e0 = (size_t)blockIdx.x * WAVE;
  const int nenv = (int)((size_t)n - e0 < (size_t)WAVE ? (size_t)n - e0 : (size_t)WAVE);
  const size_t i = e0 + lane;
  const bool live = lane < nenv;
  // ---- coalesced copy in, then the lane's own env into registers ------------------------------
  {
    const float* src = points + e0 * 2 * K;
    const int nf = nenv * 2 * K;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      for (int q = lane; q < nf / 4; q += WAVE) lc_reg4[q] = s4[q];
      for (int q = (nf & ~3) + lane; q < nf; q += WAVE) tile[q] = src[q];
    } else {
      for (int q = lane; q < nf; q += WAVE) tile[q] = src[q];
    }
    const uint8_t* gs = group + e0 * K;
    const int nb = nenv * K;
    for (int q = lane; q < nb; q += WAVE) in_grp[q] = gs[q];
  }
  __syncthreads();
  float px[LC_KREG], py[LC_KREG];
  uint32_t gpk[(LC_KREG + 3) / 4] = {};
#pragma unroll
  for (int j = 0; j < LC_KREG; j++) {
    px[j] = 0.f; py[j] = 0.f;
    if (j < K && live) {
      const float2 p = reinterpret_cast<const float2*>(tile)[lane * K + j];
      px[j] = p.x; py[j] = p.y;
      gpk[j >> 2] |= (uint32_t)in_grp[lane * K + j] << (8 * (j & 3));
    }
  }
  __syncthreads();   // the staging area becomes the accumulation tile
#pragma unroll
  for (int k = 0; k < 48; k++) tile[k * WAVE + lane] = 0.0f;
  int cst = 0;
  if (live) {
    const float rxf = robot[i * 3], ryf = robot[i * 3 + 1], yawf = robot[i * 3 + 2];
    // (heading in fp32: 1e-7 rad of rotation error against the 7.9e-6 rad = 2e-5 bins inside which lidar_exact redoes the point in
    // fp64; two fp64 range reductions + polynomials per lane were a sixth of the lane-per-pose kernel and half of a team lane's work)
    float sdf, cdf; sincosf(yawf, &sdf, &cdf);
    const double rx = rxf, ry = ryf, cd = cdf, sd = sdf;
    const float t2 = hazard_size * hazard_size;
    int* acc = reinterpret_cast<int*>(tile) + lane;   // (a lane only touches its own column: no barrier before the loop)
#pragma unroll
    for (int j = 0; j < LC_KREG; j++) {
      if (j >= K) continue;   // (K is uniform)
      const int gg = (int)(gpk[j >> 2] >> (8 * (j & 3)) & 255u), g = gg & 127;
      const double W0 = (double)px[j] - rx, W1 = (double)py[j] - ry;
      if (gg & 128) {
        const float d2 = (float)(W0 * W0 + W1 * W1);
        bool in = d2 <= t2;
        if (fabsf(d2 - t2) < 1e-5f) in = dist2d(rx, ry, px[j], py[j]) <= (double)hazard_size;
        cst |= in;
      }
      int b = -1;
      if (g >= 1 && g <= 3) {
        const float ex = (float)(W0 * cd + W1 * sd), ey = (float)(W0 * -sd + W1 * cd);
        const float dist = __builtin_amdgcn_sqrtf(__builtin_fmaf(ex, ex, ey * ey));
        float alias, sensor = fmaxf(5.0f - dist, 0.0f) * 0.2f;
        lidar_bin_alias(ex, ey, b, alias);
        const float edge = fminf(alias, 1.0f - alias);
        if (!(edge * dist >= __builtin_fmaf(2e-5f, dist, 1.5e-6f)) || b > 15) {
          const LidarHit h = lidar_exact(rxf, ryf, yawf, px[j], py[j]);
          b = h.bin; alias = h.alias; sensor = h.sensor;
        }
        const int base = (g == 1 ? 0 : (g == 3 ? 16 : 32)) * WAVE;
        atomicMax(acc + base + b * WAVE, __float_as_int(sensor));
        atomicMax(acc + base + ((b + 1) & 15) * WAVE, __float_as_int(alias * sensor));
        atomicMax(acc + base + ((b + 15) & 15) * WAVE, __float_as_int((1.0f - alias) * sensor));
      }
      if (bins) bins[i * K + j] = b;
    }
  }
  // ---- transpose [48][64] -> [64][49] in place through registers, leave as one contiguous block ----
  float col[48];
#pragma unroll
  for (int k = 0; k < 48; k++) col[k] = tile[k * WAVE + lane];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 48; k++) tile[lane * 49 + k] = col[k];
  __syncthreads();
  {
    float* dst = lidar + e0 * 48;
    const int nq = nenv * 12;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      float4* d4 = reinterpret_cast<float4*>(dst);
      for (int q = lane; q < nq; q += WAVE) {
        const int env = (int)(((uint32_t)q * 43691u) >> 19);   // q / 12, exact for q < 768
        const float* t = tile + env * 49 + 4 * (q - 12 * env);
        d4[q] = make_float4(t[0], t[1], t[2], t[3]);
      }
    } else {
      for (int q = lane; q < nenv * 48; q += WAVE) { const int env = q / 48; dst[q] = tile[env * 49 + (q - 48 * env)]; }
    }
  }
  if (live) cost[i] = (uint8_t)cst;
}

// Few poses (a learner's what-if queries at the reference's own batch - BASELINE config 2: 4096): a lane per pose
// would leave three quarters of the chip idle and run the K points of a pose as ONE serial chain.  Here a team of LPP
// lanes shares a pose: lane s takes the points s, s + LPP, ...; the team's closeness values meet in the pose's column
// of a [48 bins][P poses] tile (ds_max_i32: the maximum does not depend on the order, hence bit-identical to the
// lane-per-pose kernels), the hazard flags in one word per pose (ds_or).  Any K; no staging: the team's lanes read
// consecutive points.  Same arithmetic, point for point, as k_lidar_cost.
template <int LPP>
__global__ __launch_bounds__(WAVE) void k_lidar_cost_team(int n, int K, const float* __restrict__ robot,
                                                          const float* __restrict__ points, const uint8_t* __restrict__ group,
                                                          float hazard_size, float* __restrict__ lidar,
                                                          int32_t* __restrict__ bins, uint8_t* __restrict__ cost) {
  constexpr int P = WAVE / LPP;
  static_assert(LPP >= 2 && (LPP & (LPP - 1)) == 0 && (48 * P) % WAVE == 0, "teams tile the wavefront; the tile is whole rounds of it");
  __shared__ int tile[48 * P];
  __shared__ int cflag[P];
  const int lane = threadIdx.x, pose = lane / LPP, sub = lane % LPP;
  const size_t e0 = (size_t)blockIdx.x * P;
  const int nenv = (int)((size_t)n - e0 < (size_t)P ? (size_t)n - e0 : (size_t)P);
  const size_t i = e0 + pose;
  const bool live = pose < nenv;
#pragma unroll
  for (int q = 0; q < 48 * P / WAVE; q++) tile[q * WAVE + lane] = 0;
  if (lane < P) cflag[lane] = 0;
  __syncthreads();
  if (live) {
    const float rxf = robot[i * 3], ryf = robot[i * 3 + 1], yawf = robot[i * 3 + 2];
    // (heading in fp32: 1e-7 rad of rotation error against the 7.9e-6 rad = 2e-5 bins inside which lidar_exact redoes the point in
    // fp64; two fp64 range reductions + polynomials per lane were a sixth of the lane-per-pose kernel and half of a team lane's work)
    float sdf, cdf; sincosf(yawf, &sdf, &cdf);
    const double rx = rxf, ry = ryf, cd = cdf, sd = sdf;
    const float t2 = hazard_size * hazard_size;
    int* acc = tile + pose;
    int cst = 0;
#pragma unroll 1
    for (int j = sub; j < K; j += LPP) {
      const float2 pt = reinterpret_cast<const float2*>(points)[i * K + j];
      const int gg = group[i * K + j], g = gg & 127;
      const double W0 = (double)pt.x - rx, W1 = (double)pt.y - ry;
      if (gg & 128) {
        const float d2 = (float)(W0 * W0 + W1 * W1);
        bool in = d2 <= t2;
        if (fabsf(d2 - t2) < 1e-5f) in = dist2d(rx, ry, pt.x, pt.y) <= (double)hazard_size;
        cst |= in;
      }
      int b = -1;
      if (g >= 1 && g <= 3) {
        const float ex = (float)(W0 * cd + W1 * sd), ey = (float)(W0 * -sd + W1 * cd);
        const float dist = __builtin_amdgcn_sqrtf(__builtin_fmaf(ex, ex, ey * ey));
        float alias, sensor = fmaxf(5.0f - dist, 0.0f) * 0.2f;
        lidar_bin_alias(ex, ey, b, alias);
        const float edge = fminf(alias, 1.0f - alias);
        if (!(edge * dist >= __builtin_fmaf(2e-5f, dist, 1.5e-6f)) || b > 15) {
          const LidarHit h = lidar_exact(rxf, ryf, yawf, pt.x, pt.y);
          b = h.bin; alias = h.alias; sensor = h.sensor;
        }
        const int base = (g == 1 ? 0 : (g == 3 ? 16 : 32)) * P;
        atomicMax(acc + base + b * P, __float_as_int(sensor));
        atomicMax(acc + base + ((b + 1) & 15) * P, __float_as_int(alias * sensor));
        atomicMax(acc + base + ((b + 15) & 15) * P, __float_as_int((1.0f - alias) * sensor));
      }
      if (bins) bins[i * K + j] = b;
    }
    if (cst) atomicOr(cflag + pose, 1);
  }
  __syncthreads();
  // the block's poses are one contiguous piece of the output: nenv * 48 floats
  float* dst = lidar + e0 * 48;
  for (int q = lane; q < nenv * 48; q += WAVE) {
    const int env = (int)(((uint32_t)q * 43691u) >> 21);   // q / 48, exact for q < 1536
    dst[q] = __int_as_float(tile[(q - 48 * env) * P + env]);
  }
  if (live && sub == 0) cost[i] = (uint8_t)(cflag[pose] != 0);
}

// synthetic policy: U(-1,1)^nu from Philox stream 2
__global__ void k_fill_actions(float* actions, const int32_t* I, int N, int nu, uint32_t k0,
                               uint32_t k1, uint32_t step_index) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint32_t env = (uint32_t)I[iaddr(DI_ENVID, (size_t)N, (size_t)i)];
  for (int j = 0; j < nu; j += 4) {
    uint32_t c[4] = {env, step_index, (uint32_t)(j >> 2), 2u};
    philox4x32_10(c, k0, k1);
    for (int k = 0; k < 4 && j + k < nu; k++)
      actions[(size_t)i * nu + j + k] = ((float)(c[k] >> 8) + 0.5f) * (2.0f / 16777216.0f) - 1.0f;
  }
}

}  // namespace sag
#include "sag_render.hpp"
#include "sag_doggo_coop.hpp"
