/*
 * sag_oracle.c - CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (safe_adaptation_gym_amd + libsag.so) never does.
 *
 * Scalar, one env at a time, IEEE double (build with -DSAGO_F32 for a float
 * variant used to separate precision effects from logic differences).
 *
 * Parity status
 *   PINNED by fixtures generated from the reference's own NumPy code
 *   (oracle/gen_golden.py -> tests/golden/): lidar, lidar grouping, hazard cost,
 *   contact-count cost rule, every task's compute_reward / reset / set_mocaps,
 *   goal resampling and its RNG draw order, button state machine.
 *   UNPINNED ("parity unpinned"): the rigid-body dynamics.  The reference
 *   delegates them to the MuJoCo C library (dm_control>=0.0.403778684, no pinned
 *   version, not vendored, not installable here), so sago_substep() is a
 *   specification written from the MJCF (assets/xmls/point.xml) and MuJoCo's
 *   documented semantics, pinned only by analytic known answers
 *   (tests/test_oracle_physics.py).
 *
 * Reference lines followed are cited at each function as file:line relative to
 * /root/reference/safe_adaptation_gym/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sag.h"

#ifdef SAGO_F32
typedef float real;
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_COS cosf
#define R_FABS fabsf
#else
typedef double real;
#define R_SQRT sqrt
#define R_SIN sin
#define R_COS cos
#define R_FABS fabs
#endif

#define PI 3.14159265358979323846

/* ------------------------------------------------------------------------ */
/* constants from the MJCF / primitive_objects.py                           */
/* ------------------------------------------------------------------------ */
/* point.xml:3 timestep, safe_adaptation_gym.py:15-19 substeps */
static const double DT[3] = {0.004, 0.008, 0.012};
static const int NSTEP[3] = {5, 10, 12};
static const int NU[3] = {2, 2, 12};
static const int OBS_DIM[3] = {60, 72, 104};

/* point.xml:18-19, density 1 (:5): sphere r .1 + box half .05 at (.1,0,0) */
#define PT_M_SPHERE (4.0 / 3.0 * PI * 0.001)
#define PT_M_BOX 0.001
#define PT_MASS (PT_M_SPHERE + PT_M_BOX)
#define PT_MC (PT_M_BOX * 0.1) /* mass * com offset along local +x */
#define PT_IO (0.4 * PT_M_SPHERE * 0.01 + PT_M_BOX * (0.01 + 0.01) / 12.0 + PT_M_BOX * 0.01)
#define PT_DAMP_Z 0.005      /* point.xml:17 */
#define PT_FORCE_LIM 0.05    /* point.xml:7-8 forcerange */
#define PT_GEAR_Z 0.3        /* point.xml:37 */
#define PT_KV 1.0            /* MuJoCo velocity actuator default kv */
#define PT_Z 0.1             /* point.xml:13 body height (no z DoF) */
#define GOAL_Z (0.3 / 2.0 + 1e-2) /* primitive_objects.py:141 size/2 + 1e-2 */
#define GOAL_SIZE 0.3        /* tasks/go_to_goal.py:12 */
#define GOAL_KEEPOUT 0.4     /* tasks/go_to_goal.py:13 */
#define GOAL_RECT 1.5        /* tasks/go_to_goal.py:9 */
#define BUTTON_R 0.1         /* tasks/press_buttons.py:16 */
#define GRAVITY 9.81
#define LIDAR_MAX 5.0        /* safe_adaptation_gym.py:23 */

/* MuJoCo default contact softness: solref (0.02, 1), solimp (0.9, 0.95, 0.001, 0.5, 2);
 * time constant clamped to >= 2*dt (refsafe). */
#define SOL_TC 0.02
#define SOL_D0 0.9
#define SOL_D1 0.95
#define SOL_WIDTH 0.001
#define FRICTION_MU 1.0 /* max(geom frictions) = 1 for every default pair */

/* vases: free box, density .001 (consts.py:20-21, primitive_objects.py:39-54) */
#define VASE_DENSITY 0.001

#define MAX_RESAMPLE_TRIES 10000 /* inner loop of tasks/go_to_goal.py:63-64 */

/* ------------------------------------------------------------------------ */
/* env state: the sag.h record, widened to `real`                           */
/* ------------------------------------------------------------------------ */
typedef struct {
  real f[SAG_REC_FLOATS];
  int32_t i[SAG_REC_INTS];
} OEnv;

typedef struct {
  real obs[104];
  real reward[2];
  int cost, done, goal_met, tape_used;
  /* diagnostics for parity tests */
  real qacc[3];
  real cost_margin; /* smallest |distance to a flag threshold| seen */
  uint32_t btn_contact_mask;
  real touch[8];    /* doggo touch sensors of the final forward pass */
  real comvel[4];   /* doggo subtree COM xy and its velocity (Unsupervised) */
} OOut;

int sago_sizeof_env(void) { return (int)sizeof(OEnv); }
int sago_sizeof_out(void) { return (int)sizeof(OOut); }
int sago_real_bytes(void) { return (int)sizeof(real); }

void sago_load(OEnv* e, const float* rf, const int32_t* ri) {
  for (int k = 0; k < SAG_REC_FLOATS; k++) e->f[k] = (real)rf[k];
  memcpy(e->i, ri, sizeof(e->i));
}
void sago_load_f64(OEnv* e, const double* rf, const int32_t* ri) {
  for (int k = 0; k < SAG_REC_FLOATS; k++) e->f[k] = (real)rf[k];
  memcpy(e->i, ri, sizeof(e->i));
}
void sago_store(const OEnv* e, float* rf, int32_t* ri) {
  for (int k = 0; k < SAG_REC_FLOATS; k++) rf[k] = (float)e->f[k];
  memcpy(ri, e->i, sizeof(e->i));
}
void sago_store_f64(const OEnv* e, double* rf, int32_t* ri) {
  for (int k = 0; k < SAG_REC_FLOATS; k++) rf[k] = (double)e->f[k];
  memcpy(ri, e->i, sizeof(e->i));
}

/* ------------------------------------------------------------------------ */
/* random words: tape (parity mode) or Philox4x32-10 (throughput mode)       */
/* ------------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
void sago_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  memcpy(out, ctr, 16);
  philox4x32_10(out, key[0], key[1]);
}

typedef struct {
  const uint32_t* tape;
  int len, pos;
  int exhausted;
  uint32_t key0, key1, env_id, step;
  uint32_t ep4; /* episode nonce (SAG_I_EPISODE) << 2: counter word 3 = nonce | stream */
} Rng;

/* stream 0: in-step draws (words indexed by pos); stream 1: action noise;
 * stream 2: synthetic actions (sag_dev_fill_actions). */
static uint32_t rng_word(Rng* g) {
  if (g->tape) {
    if (g->pos >= g->len) { g->exhausted = 1; g->pos++; return 0; }
    return g->tape[g->pos++];
  }
  uint32_t c[4] = {g->env_id, g->step, (uint32_t)(g->pos >> 2), g->ep4};
  philox4x32_10(c, g->key0, g->key1);
  return c[g->pos++ & 3];
}
/* numpy legacy random_sample(): 53-bit double from two 32-bit words */
static double rng_double(Rng* g) {
  uint32_t a = rng_word(g) >> 5, b = rng_word(g) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
/* rs.uniform(lo, hi) = lo + (hi - lo) * random_sample() */
static double rng_uniform(Rng* g, double lo, double hi) { return lo + (hi - lo) * rng_double(g); }

/* counter-based standard normals for action noise (throughput mode only; the
 * reference draws rs.normal(size=nu), safe_adaptation_gym.py:63-65). */
void sago_noise_ep(uint32_t key0, uint32_t key1, uint32_t env_id, uint32_t step, uint32_t episode, int nu, float* out) {
  for (int j = 0; j < nu; j += 2) {
    uint32_t c[4] = {env_id, step, (uint32_t)(j >> 1), episode << 2 | 1u};
    philox4x32_10(c, key0, key1);
    float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float r = sqrtf(-2.0f * logf(u1));
    float a = 6.28318530717958647692f * u2;
    out[j] = r * cosf(a);
    if (j + 1 < nu) out[j + 1] = r * sinf(a);
  }
}
void sago_noise(uint32_t key0, uint32_t key1, uint32_t env_id, uint32_t step, int nu, float* out) {
  sago_noise_ep(key0, key1, env_id, step, 0u, nu, out);
}
/* synthetic policy: U(-1,1) (stream 2) */
void sago_actions(uint32_t key0, uint32_t key1, uint32_t env_id, uint32_t step, int nu, float* out) {
  for (int j = 0; j < nu; j += 4) {
    uint32_t c[4] = {env_id, step, (uint32_t)(j >> 2), 2u};
    philox4x32_10(c, key0, key1);
    for (int k = 0; k < 4 && j + k < nu; k++)
      out[j + k] = ((float)(c[k] >> 8) + 0.5f) * (2.0f / 16777216.0f) - 1.0f;
  }
}

/* ------------------------------------------------------------------------ */
/* lidar: safe_adaptation_gym.py:174-223, fp64 exactly as NumPy evaluates it  */
/* ------------------------------------------------------------------------ */
/* Python float % : fmod, then shift into the divisor's sign (npy_divmod). */
static double py_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0 && ((b < 0) != (m < 0))) m += b;
  return m;
}

/* One object into a 16-bin reading. ex, ey = ego_xy (:197-202). Returns bin. */
static int lidar_accumulate(double ex, double ey, double* obs) {
  const double two_pi = PI * 2;
  double dist = hypot(ex, ey);              /* np.abs(complex) :209            */
  double angle = py_mod(atan2(ey, ex), two_pi); /* np.angle(z) % (2 pi) :210   */
  double bin_size = two_pi / SAG_LIDAR_BINS;    /* :211                         */
  int bin = (int)(angle / bin_size);            /* :212                         */
  /* Reference quirk: angle % 2pi can round to exactly 2pi for a
   * tiny negative angle -> bin 16 -> IndexError in the reference.  Documented
   * deviation: wrap to bin 0. */
  if (bin >= SAG_LIDAR_BINS) bin -= SAG_LIDAR_BINS;
  double bin_angle = bin_size * bin;            /* :213 */
  double sensor = (LIDAR_MAX - dist > 0 ? LIDAR_MAX - dist : 0.0) / LIDAR_MAX; /* :214 */
  if (sensor > obs[bin]) obs[bin] = sensor;     /* :215 */
  double alias = (angle - bin_angle) / bin_size; /* :216 */
  int bp = (bin + 1) % SAG_LIDAR_BINS, bm = (bin + SAG_LIDAR_BINS - 1) % SAG_LIDAR_BINS;
  if (alias * sensor > obs[bp]) obs[bp] = alias * sensor;             /* :221 */
  if ((1 - alias) * sensor > obs[bm]) obs[bm] = (1 - alias) * sensor; /* :222 */
  return bin;
}

/* General form (tilted base: car/doggo): robot_pos[3], row-major robot_mat[9],
 * pts [n][2]; obs16 zero-initialised here; bins may be NULL.
 * ego = ([p,0] - robot_pos) @ R  -> first two components (:200-202). */
void sago_lidar(const double* robot_pos, const double* mat, const double* pts, int n,
                double* obs16, int* bins) {
  for (int k = 0; k < SAG_LIDAR_BINS; k++) obs16[k] = 0.0;
  for (int j = 0; j < n; j++) {
    double w0 = pts[2 * j] - robot_pos[0], w1 = pts[2 * j + 1] - robot_pos[1],
           w2 = 0.0 - robot_pos[2];
    /* np.matmul(world_3vec, robot_mat)[:2]: column k = sum_i w_i * R[i][k] */
    double ex = w0 * mat[0] + w1 * mat[3] + w2 * mat[6];
    double ey = w0 * mat[1] + w1 * mat[4] + w2 * mat[7];
    int b = lidar_accumulate(ex, ey, obs16);
    if (bins) bins[j] = b;
  }
}

/* Planar robot (Point): R = Rz(yaw), so the z term vanishes exactly. */
static int lidar_planar(double rx, double ry, double c, double s, double px, double py,
                        double* obs) {
  double w0 = px - rx, w1 = py - ry;
  double ex = w0 * c + w1 * s;  /* w0*R[0][0] + w1*R[1][0] */
  double ey = w0 * -s + w1 * c; /* w0*R[0][1] + w1*R[1][1] */
  return lidar_accumulate(ex, ey, obs);
}

/* The stand-alone lidar + hazard-cost path of the C ABI (sag_lidar_cost):
 * float inputs promoted to double, same arithmetic. */
void sago_lidar_cost(int K, const float* robot3, const float* points, const uint8_t* group,
                     float hazard_size, float* lidar48, int32_t* bins, uint8_t* cost) {
  double obs[48];
  for (int k = 0; k < 48; k++) obs[k] = 0;
  double rx = robot3[0], ry = robot3[1], yaw = robot3[2];
  double c = cos(yaw), s = sin(yaw);
  int cst = 0;
  for (int j = 0; j < K; j++) {
    int g = group[j] & 127;
    double px = points[2 * j], py = points[2 * j + 1];
    if (group[j] & 128) { /* world.py:148-153 */
      double dx = rx - px, dy = ry - py;
      if (sqrt(dx * dx + dy * dy) <= (double)hazard_size) cst = 1;
    }
    int b = -1;
    /* output order [obstacles, objects, goal] (safe_adaptation_gym.py:136-139) */
    if (g == 1) b = lidar_planar(rx, ry, c, s, px, py, obs);
    else if (g == 3) b = lidar_planar(rx, ry, c, s, px, py, obs + 16);
    else if (g == 2) b = lidar_planar(rx, ry, c, s, px, py, obs + 32);
    if (bins) bins[j] = b;
  }
  for (int k = 0; k < 48; k++) lidar48[k] = (float)obs[k];
  *cost = (uint8_t)cst;
}

/* ------------------------------------------------------------------------ */
/* planar geometry                                                           */
/* ------------------------------------------------------------------------ */
typedef struct { int type; real ox, oy, a, b; } Geom; /* type 0 circle (a=r), 1 box (a=hx,b=hy) */

typedef struct Body_ {
  real x, y, yaw, vx, vy, w;   /* state */
  real ax, ay, aw;             /* accumulated acceleration this substep */
  real minv[6];                /* symmetric inverse inertia: xx, xy, xw, yy, yw, ww (0 for static) */
  int ngeom;
  Geom g[8];
  int dynamic;
  /* contact material (MuJoCo geom friction[0], priority, solref): the higher priority geom's
   * values win, equal priority takes max friction / equal solref (all defaults are equal) */
  real mu, tc, dr;
  int prio;
} Body;

typedef struct { real nx, ny, px, py, depth; } Contact;

static real clampr(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* world centre + rotation of a geom */
static void geom_world(const Body* b, const Geom* g, real* cx, real* cy, real* c, real* s) {
  *c = R_COS(b->yaw); *s = R_SIN(b->yaw);
  *cx = b->x + *c * g->ox - *s * g->oy;
  *cy = b->y + *s * g->ox + *c * g->oy;
}

/* Each narrowphase returns contacts with normal pointing from A to B.  Touching
 * (depth == 0) is not a contact (MuJoCo: dist < margin = 0). */
static int circle_circle(real ax, real ay, real ra, real bx, real by, real rb, Contact* out) {
  real dx = bx - ax, dy = by - ay;
  real d2 = dx * dx + dy * dy, rs = ra + rb;
  if (d2 >= rs * rs) return 0;
  real d = R_SQRT(d2);
  real nx = 1, ny = 0;
  if (d > (real)1e-12) { nx = dx / d; ny = dy / d; }
  out->nx = nx; out->ny = ny; out->depth = rs - d;
  out->px = ax + nx * (ra - (real)0.5 * out->depth);
  out->py = ay + ny * (ra - (real)0.5 * out->depth);
  return 1;
}

/* circle A vs oriented box B (centre bx,by; cos/sin cb,sb; half hx,hy).
 * g_box_tie (0 except inside the Doggo's dg_collide_body): a circle centre INSIDE the box takes the x face also when the
 * y face is nearer by less than this.  The Doggo's capsules need it: the deepest point of an axis that crosses a box is
 * where the two face distances are EQUAL (a kink of the signed distance, dg_seg_box_t), so the plain `<` would be decided
 * by rounding there - structurally, not by coincidence. */
static _Thread_local real g_box_tie = 0;
static int circle_box(real ax, real ay, real ra, real bx, real by, real cb, real sb, real hx,
                      real hy, Contact* out) {
  real wx = ax - bx, wy = ay - by;
  real lx = cb * wx + sb * wy, ly = -sb * wx + cb * wy; /* circle centre in box frame */
  real qx = clampr(lx, -hx, hx), qy = clampr(ly, -hy, hy);
  real ddx = lx - qx, ddy = ly - qy;
  real onx, ony, depth; /* outward normal of the box in its frame */
  if (ddx == 0 && ddy == 0) { /* centre inside the box */
    real px = hx - R_FABS(lx), py = hy - R_FABS(ly);
    /* (likewise the + face when the centre is within g_box_tie of the mid-plane: the other kink of the signed distance) */
    if (px < py || (g_box_tie > 0 && px <= py + g_box_tie)) { onx = lx >= -g_box_tie ? 1 : -1; ony = 0; depth = ra + px; qx = onx * hx; }
    else { onx = 0; ony = ly >= -g_box_tie ? 1 : -1; depth = ra + py; qy = ony * hy; }
  } else {
    real d2 = ddx * ddx + ddy * ddy;
    if (d2 >= ra * ra) return 0;
    real d = R_SQRT(d2);
    onx = ddx / d; ony = ddy / d; depth = ra - d;
  }
  /* to world; normal from A (circle) to B (box) = -outward */
  out->nx = -(cb * onx - sb * ony);
  out->ny = -(sb * onx + cb * ony);
  out->px = bx + cb * qx - sb * qy;
  out->py = by + sb * qx + cb * qy;
  out->depth = depth;
  return 1;
}

/* vertices of box P strictly inside box Q -> contacts with the outward normal
 * of Q's least-penetrated face.  sign = +1 if Q is body A (normal A->B is Q's
 * outward), -1 if Q is body B. Vertex order (+,+), (-,+), (-,-), (+,-). */
static int verts_in_box(real pxc, real pyc, real cp, real sp, real phx, real phy, real qxc,
                        real qyc, real cq, real sq, real qhx, real qhy, real sign, Contact* out) {
  static const int SX[4] = {1, -1, -1, 1}, SY[4] = {1, 1, -1, -1};
  int n = 0;
  for (int k = 0; k < 4; k++) {
    real vx = pxc + cp * (SX[k] * phx) - sp * (SY[k] * phy);
    real vy = pyc + sp * (SX[k] * phx) + cp * (SY[k] * phy);
    real wx = vx - qxc, wy = vy - qyc;
    real lx = cq * wx + sq * wy, ly = -sq * wx + cq * wy;
    real dx = qhx - R_FABS(lx), dy = qhy - R_FABS(ly);
    if (dx <= 0 || dy <= 0) continue;
    real onx, ony, depth;
    if (dx < dy) { onx = lx >= 0 ? 1 : -1; ony = 0; depth = dx; }
    else { onx = 0; ony = ly >= 0 ? 1 : -1; depth = dy; }
    out[n].nx = sign * (cq * onx - sq * ony);
    out[n].ny = sign * (sq * onx + cq * ony);
    out[n].px = vx; out[n].py = vy; out[n].depth = depth;
    n++;
  }
  return n;
}

static int geom_pair(const Body* A, const Geom* ga, const Body* B, const Geom* gb, Contact* out) {
  real ax, ay, ca, sa, bx, by, cb, sb;
  geom_world(A, ga, &ax, &ay, &ca, &sa);
  geom_world(B, gb, &bx, &by, &cb, &sb);
  if (ga->type == 0 && gb->type == 0) return circle_circle(ax, ay, ga->a, bx, by, gb->a, out);
  if (ga->type == 0 && gb->type == 1) return circle_box(ax, ay, ga->a, bx, by, cb, sb, gb->a, gb->b, out);
  if (ga->type == 1 && gb->type == 0) {
    int n = circle_box(bx, by, gb->a, ax, ay, ca, sa, ga->a, ga->b, out);
    if (n) { out->nx = -out->nx; out->ny = -out->ny; }
    return n;
  }
  int n = verts_in_box(ax, ay, ca, sa, ga->a, ga->b, bx, by, cb, sb, gb->a, gb->b, (real)-1, out);
  n += verts_in_box(bx, by, cb, sb, gb->a, gb->b, ax, ay, ca, sa, ga->a, ga->b, (real)1, out + n);
  return n;
}

static real bound_radius(const Body* b) {
  real r = 0;
  for (int k = 0; k < b->ngeom; k++) {
    const Geom* g = &b->g[k];
    real off = R_SQRT(g->ox * g->ox + g->oy * g->oy);
    real e = g->type == 0 ? g->a : R_SQRT(g->a * g->a + g->b * g->b);
    if (off + e > r) r = off + e;
  }
  return r;
}

/* ------------------------------------------------------------------------ */
/* soft contact (MuJoCo-style reference acceleration, one Gauss-Seidel sweep) */
/* ------------------------------------------------------------------------ */
typedef struct { real kcoef, bcoef, h, mu; } Sol;

static Sol make_sol_tc(double h, double tc0, double dr) {
  double tc = tc0 < 2 * h ? 2 * h : tc0; /* refsafe */
  Sol s;
  s.bcoef = (real)(2.0 / (SOL_D1 * tc));
  s.kcoef = (real)(1.0 / (SOL_D1 * SOL_D1 * tc * tc * dr * dr));
  s.h = (real)h;
  s.mu = (real)FRICTION_MU;
  return s;
}
static Sol make_sol(double h) { return make_sol_tc(h, SOL_TC, 1.0); }



static real impedance(real depth) {
  real x = depth / (real)SOL_WIDTH;
  if (x > 1) x = 1;
  real y = x < (real)0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x); /* power 2, midpoint .5 */
  return (real)SOL_D0 + ((real)SOL_D1 - (real)SOL_D0) * y;
}

/* u = Minv * J^T for J = [dx, dy, r x d]; returns J u */
/* ---- N-sweep mode (diagnosis; DESIGN.md "how far is one sweep from the converged solution") -------------
 * The specification solves every scalar constraint ONCE per forward evaluation, in a fixed order (one
 * Gauss-Seidel sweep).  sago_set_sweeps(n > 1) repeats the sweep n times over the same constraint set as a
 * proper projected Gauss-Seidel: every constraint keeps its accumulated force f, an update is
 *   f <- clamp(f + (aref - a - R f) / (A + R)),  R = A (1 - d) / d   (the soft-constraint regulariser),
 * and only the increment is applied.  With f = 0 this is the specification's one-shot formula
 * d (aref - a) / A, so sweep 0 of the N-sweep mode IS the specification; n = 1 runs the original code. */
static int g_sweeps = 1;
void sago_set_sweeps(int n) { g_sweeps = n < 1 ? 1 : n; }
int sago_get_sweeps(void) { return g_sweeps; }
#define ACC_SLOTS 8192
static _Thread_local real g_acc[ACC_SLOTS];
static _Thread_local int g_ai, g_sweep;
static real* acc_slot(void) {
  real* f = &g_acc[g_ai < ACC_SLOTS ? g_ai : ACC_SLOTS - 1];
  g_ai++;
  if (g_sweep == 0) *f = 0;
  return f;
}
/* PGS update of one accumulated force; returns the increment */
static real pgs_step(real* f, real aref, real a, real A, real d, real lo, real hi) {
  real R = A * (1 - d) / d;
  real fn = *f + (aref - a - R * *f) / (A + R);
  fn = fn < lo ? lo : (fn > hi ? hi : fn);
  real df = fn - *f;
  *f = fn;
  return df;
}

static real minv_apply(const Body* b, real dx, real dy, real rxd, real u[3]) {
  const real* m = b->minv;
  u[0] = m[0] * dx + m[1] * dy + m[2] * rxd;
  u[1] = m[1] * dx + m[3] * dy + m[4] * rxd;
  u[2] = m[2] * dx + m[4] * dy + m[5] * rxd;
  return dx * u[0] + dy * u[1] + rxd * u[2];
}

static void rel_at(const Body* A, const Body* B, real rax, real ray, real rbx, real rby,
                   real* vx, real* vy, real* ax, real* ay) {
  /* point velocity v + w x r, point acceleration a + alpha x r (centripetal
   * term omitted by specification) */
  *vx = (B->vx - B->w * rby) - (A->vx - A->w * ray);
  *vy = (B->vy + B->w * rbx) - (A->vy + A->w * rax);
  *ax = (B->ax - B->aw * rby) - (A->ax - A->aw * ray);
  *ay = (B->ay + B->aw * rbx) - (A->ay + A->aw * rax);
}

static void apply_dir(Body* A, Body* B, real dx, real dy, real rax, real ray, real rbx,
                      real rby, real f, const real ua[3], const real ub[3]) {
  (void)dx; (void)dy; (void)rax; (void)ray; (void)rbx; (void)rby;
  if (A->dynamic) { A->ax -= ua[0] * f; A->ay -= ua[1] * f; A->aw -= ua[2] * f; }
  if (B->dynamic) { B->ax += ub[0] * f; B->ay += ub[1] * f; B->aw += ub[2] * f; }
}

static void solve_contact(Body* A, Body* B, const Contact* c, const Sol* sol) {
  real rax = c->px - A->x, ray = c->py - A->y, rbx = c->px - B->x, rby = c->py - B->y;
  real nx = c->nx, ny = c->ny;
  real vx, vy, ax, ay;
  real ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
  /* normal */
  rel_at(A, B, rax, ray, rbx, rby, &vx, &vy, &ax, &ay);
  real vn = vx * nx + vy * ny, an = ax * nx + ay * ny;
  real An = 0;
  if (A->dynamic) An += minv_apply(A, nx, ny, rax * ny - ray * nx, ua);
  if (B->dynamic) An += minv_apply(B, nx, ny, rbx * ny - rby * nx, ub);
  if (An <= 0) return;
  real d = impedance(c->depth);
  /* separating acceleration wanted: aref = -b*vn + k*depth (vn<0 approaching) */
  real aref = -sol->bcoef * vn + sol->kcoef * c->depth;
  if (g_sweeps > 1) {
    real* fN = acc_slot(); real* fT = acc_slot();
    real dfn = pgs_step(fN, aref, an, An, d, 0, (real)1e30);
    apply_dir(A, B, nx, ny, rax, ray, rbx, rby, dfn, ua, ub);
    real tx = -ny, ty = nx;
    rel_at(A, B, rax, ray, rbx, rby, &vx, &vy, &ax, &ay);
    real vt = vx * tx + vy * ty, at = ax * tx + ay * ty, At = 0;
    if (A->dynamic) At += minv_apply(A, tx, ty, rax * ty - ray * tx, ua);
    if (B->dynamic) At += minv_apply(B, tx, ty, rbx * ty - rby * tx, ub);
    if (At <= 0) return;
    real dft = pgs_step(fT, -sol->bcoef * vt, at, At, d, -sol->mu * *fN, sol->mu * *fN);
    apply_dir(A, B, tx, ty, rax, ray, rbx, rby, dft, ua, ub);
    return;
  }
  real fn = d * (aref - an) / An;
  if (fn <= 0) return;
  apply_dir(A, B, nx, ny, rax, ray, rbx, rby, fn, ua, ub);
  /* friction along t = (-ny, nx), regularised Coulomb */
  real tx = -ny, ty = nx;
  rel_at(A, B, rax, ray, rbx, rby, &vx, &vy, &ax, &ay);
  real vt = vx * tx + vy * ty, at = ax * tx + ay * ty;
  real At = 0;
  if (A->dynamic) At += minv_apply(A, tx, ty, rax * ty - ray * tx, ua);
  if (B->dynamic) At += minv_apply(B, tx, ty, rbx * ty - rby * tx, ub);
  if (At <= 0) return;
  real ft = d * (-sol->bcoef * vt - at) / At;
  ft = clampr(ft, -sol->mu * fn, sol->mu * fn);
  apply_dir(A, B, tx, ty, rax, ray, rbx, rby, ft, ua, ub);
}

static Sol pair_sol(const Body* A, const Body* B, const Sol* base) {
  const Body* P = A->prio > B->prio ? A : (B->prio > A->prio ? B : NULL);
  real mu = P ? P->mu : (A->mu > B->mu ? A->mu : B->mu);
  real tc = P ? P->tc : A->tc, dr = P ? P->dr : A->dr;
  Sol s = make_sol_tc(base->h, tc, dr);
  s.mu = mu;
  return s;
}

/* all contacts of one body pair; returns number of contacts (penetrations) */
static int collide_pair(Body* A, Body* B, real ra, real rb, const Sol* base, int solve) {
  real dx = B->x - A->x, dy = B->y - A->y, rs = ra + rb;
  if (dx * dx + dy * dy > rs * rs) return 0; /* broadphase */
  Sol psol = pair_sol(A, B, base);
  const Sol* sol = &psol;
  int total = 0;
  Contact c[8];
  for (int i = 0; i < A->ngeom; i++)
    for (int j = 0; j < B->ngeom; j++) {
      int n = geom_pair(A, &A->g[i], B, &B->g[j], c);
      total += n;
      if (solve)
        for (int k = 0; k < n; k++) solve_contact(A, B, &c[k], sol);
    }
  return total;
}

/* floor friction of a free body resting on the plane (condim-6 floor contact
 * reduced to translation + torsion about the vertical, decoupled):
 *   f = clamp_norm(-d0 m (b v + a), mu m g),  tau = clamp(-d0 I (b w + alpha), mu m g r_eff) */
static void floor_friction(Body* b, real mass, real inertia, real reff, const Sol* sol) {
  real d0 = (real)SOL_D0;
  if (g_sweeps > 1) {   /* accumulated (fx, fy) with the norm bound, accumulated torsion */
    real* ax_ = acc_slot(); real* ay_ = acc_slot(); real* at_ = acc_slot();
    real fmax_ = (real)(FRICTION_MU * GRAVITY) * mass, A = 1 / mass, R = A * (1 - d0) / d0;
    real nx_ = *ax_ + (-sol->bcoef * b->vx - b->ax - R * *ax_) / (A + R);
    real ny_ = *ay_ + (-sol->bcoef * b->vy - b->ay - R * *ay_) / (A + R);
    real n2 = nx_ * nx_ + ny_ * ny_;
    if (n2 > fmax_ * fmax_) { real sc = fmax_ / R_SQRT(n2); nx_ *= sc; ny_ *= sc; }
    b->ax += (nx_ - *ax_) / mass; b->ay += (ny_ - *ay_) / mass;
    *ax_ = nx_; *ay_ = ny_;
    real dt_ = pgs_step(at_, -sol->bcoef * b->w, b->aw, 1 / inertia, d0, -fmax_ * reff, fmax_ * reff);
    b->aw += dt_ / inertia;
    return;
  }
  real fx = -d0 * mass * (sol->bcoef * b->vx + b->ax);
  real fy = -d0 * mass * (sol->bcoef * b->vy + b->ay);
  real fmax = (real)(FRICTION_MU * GRAVITY) * mass;
  real fn2 = fx * fx + fy * fy;
  if (fn2 > fmax * fmax) { real sc = fmax / R_SQRT(fn2); fx *= sc; fy *= sc; }
  b->ax += fx / mass; b->ay += fy / mass;
  real t = -d0 * inertia * (sol->bcoef * b->w + b->aw);
  t = clampr(t, -fmax * reff, fmax * reff);
  b->aw += t / inertia;
}

/* ------------------------------------------------------------------------ */
/* world assembly from the record                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
  Body robot, vase[SAG_MAX_VASES], pillar[SAG_MAX_PILLARS], button[SAG_MAX_BUTTONS], box;
  real r_robot, r_vase, r_pillar, r_button, r_box;
  real vase_m, vase_I, vase_reff;
  real box_m, box_I;
  int nV, nP, nB, box_kind, haul;
  int robot_id;
  real ext[SAG_ROBOT_EXT_FLOATS], ext_acc[5]; /* car: wheel rates L,R; ball rate x,y,z; ball quat w,x,y,z */
  int act_v[SAG_MAX_VASES], act_box;          /* free bodies that take part in this forward evaluation (see world_forward) */
  real ext_acc0[5], car_fa[3];                /* car: spin accelerations without floor friction; the friction's share of the base acceleration (body axes) */
  void* dg;                                   /* doggo: Doggo* articulated state (sag_oracle_doggo.inc) */
  uint32_t awake0;   /* SAG_I_AWAKE: free bodies that take part in the NEXT forward evaluation although at rest (install-time overlap) */
} World;

static void material_default(Body* b) { b->mu = (real)FRICTION_MU; b->tc = (real)SOL_TC; b->dr = 1; b->prio = 0; }

static void body_static_circle(Body* b, real x, real y, real r) {
  memset(b, 0, sizeof(*b));
  b->x = x; b->y = y; b->ngeom = 1; b->g[0].type = 0; b->g[0].a = r; b->dynamic = 0;
  material_default(b);
}

#include "sag_oracle_doggo.inc"

/* The task object of the PushBox family, as a planar free body.
 *  BOX  (push_box.py:28-72): box half .2 + four corner columns half .1 at (+-.2, +-.2), density
 *       .001 each (masses add, overlaps included, as MuJoCo does)
 *  ROD  (roll_rod.py:19-43): cylinder r .08, half length .3 lying along local y (euler 90 0 0),
 *       density .0005, friction 1.2 / rolling .05, priority 1.  It rolls along local x
 *       (translational inertia m + I_axis/r^2 = 1.5 m) and slides along its axis.
 *  BALL (dribble_ball.py:18-41): sphere r .14, density .0005, friction 1.2 / rolling .05,
 *       solref (.018, .2), priority 1; rolls in every direction (inertia 1.4 m). */
#define BOX_DENSITY 0.001
static void box_from_env(const OEnv* e, World* w) {
  const real* f = e->f;
  Body* b = &w->box;
  memset(b, 0, sizeof(*b));
  material_default(b);
  w->box_kind = e->i[SAG_I_BOX_KIND];
  w->haul = e->i[SAG_I_TASK] == SAG_TASK_HAUL_BOX;
  if (w->box_kind == SAG_BOX_NONE) return;
  const real* v = f + SAG_F_BOX;
  b->x = v[0]; b->y = v[1]; b->yaw = v[2]; b->vx = v[3]; b->vy = v[4]; b->w = v[5];
  b->dynamic = 1;
  real c = R_COS(b->yaw), s = R_SIN(b->yaw);
  if (w->box_kind == SAG_BOX_BOX) {
    static const real OX[5] = {0, 0.2, -0.2, 0.2, -0.2}, OY[5] = {0, 0.2, 0.2, -0.2, -0.2};
    b->ngeom = 5;
    for (int k = 0; k < 5; k++) {
      b->g[k].type = 1; b->g[k].ox = OX[k]; b->g[k].oy = OY[k];
      b->g[k].a = b->g[k].b = k == 0 ? (real)0.2 : (real)0.1;
    }
    real m0 = (real)BOX_DENSITY * (real)(0.4 * 0.4 * 0.4), m1 = (real)BOX_DENSITY * (real)(0.2 * 0.2 * 0.4);
    w->box_m = m0 + 4 * m1;
    w->box_I = m0 * (real)(0.16 + 0.16) / 12 + 4 * (m1 * (real)(0.04 + 0.04) / 12 + m1 * (real)0.08);
    b->minv[0] = b->minv[3] = 1 / w->box_m; b->minv[5] = 1 / w->box_I;
  } else if (w->box_kind == SAG_BOX_ROD) {
    b->ngeom = 1; b->g[0].type = 1; b->g[0].a = (real)0.08; b->g[0].b = (real)0.3;
    real m = (real)(BOX_DENSITY / 2) * (real)(PI * 0.08 * 0.08 * 0.6);
    w->box_m = m;
    w->box_I = m * (real)(3 * 0.08 * 0.08 + 0.6 * 0.6) / 12;
    /* world-frame inverse of R diag(1.5 m, m) R^T */
    real ix = 1 / ((real)1.5 * m), iy = 1 / m;
    b->minv[0] = c * c * ix + s * s * iy; b->minv[1] = c * s * (ix - iy);
    b->minv[3] = s * s * ix + c * c * iy; b->minv[5] = 1 / w->box_I;
    b->mu = (real)1.2; b->prio = 1;
  } else {
    b->ngeom = 1; b->g[0].type = 0; b->g[0].a = (real)0.14;
    real m = (real)(BOX_DENSITY / 2) * (real)(4.0 / 3.0 * PI * 0.14 * 0.14 * 0.14);
    w->box_m = m;
    w->box_I = (real)0.4 * m * (real)(0.14 * 0.14);
    b->minv[0] = b->minv[3] = 1 / ((real)1.4 * m); b->minv[5] = 1 / w->box_I;
    b->mu = (real)1.2; b->prio = 1; b->tc = (real)0.018; b->dr = (real)0.2;
  }
  w->r_box = bound_radius(b);
}

/* floor interaction of the task object (see floor_friction for vases) */
static void box_floor_friction(World* w, const Sol* sol) {
  Body* b = &w->box;
  real d0 = (real)SOL_D0, m = w->box_m, g = (real)GRAVITY;
  if (w->box_kind == SAG_BOX_BOX) { floor_friction(b, m, w->box_I, (real)(0.2 * 1.41421356237309504880), sol); return; }
  real c = R_COS(b->yaw), s = R_SIN(b->yaw);
  /* body-frame velocity / acceleration */
  real vx = c * b->vx + s * b->vy, vy = -s * b->vx + c * b->vy;
  real ax = c * b->ax + s * b->ay, ay = -s * b->ax + c * b->ay;
  real fx, fy, tlim;
  if (g_sweeps > 1) {
    real* fx_ = acc_slot(); real* fy_ = acc_slot(); real* ft_ = acc_slot();
    if (w->box_kind == SAG_BOX_ROD) {
      real mx = (real)1.5 * m, lx = (real)0.05 * m * g / (real)0.08, ly = (real)1.2 * m * g;
      ax += pgs_step(fx_, -sol->bcoef * vx, ax, 1 / mx, d0, -lx, lx) / mx;
      ay += pgs_step(fy_, -sol->bcoef * vy, ay, 1 / m, d0, -ly, ly) / m;
      tlim = (real)1.2 * m * g * (real)0.15;
    } else {
      real me = (real)1.4 * m, lim = (real)0.05 * m * g / (real)0.14, A = 1 / me, R = A * (1 - d0) / d0;
      real nx_ = *fx_ + (-sol->bcoef * vx - ax - R * *fx_) / (A + R), ny_ = *fy_ + (-sol->bcoef * vy - ay - R * *fy_) / (A + R);
      real n2 = nx_ * nx_ + ny_ * ny_;
      if (n2 > lim * lim) { real sc = lim / R_SQRT(n2); nx_ *= sc; ny_ *= sc; }
      ax += (nx_ - *fx_) / me; ay += (ny_ - *fy_) / me;
      *fx_ = nx_; *fy_ = ny_;
      tlim = (real)0.003 * m * g;
    }
    b->ax = c * ax - s * ay; b->ay = s * ax + c * ay;
    b->aw += pgs_step(ft_, -sol->bcoef * b->w, b->aw, 1 / w->box_I, d0, -tlim, tlim) / w->box_I;
    return;
  }
  if (w->box_kind == SAG_BOX_ROD) {
    real mx = (real)1.5 * m;
    fx = clampr(-d0 * mx * (sol->bcoef * vx + ax), -(real)0.05 * m * g / (real)0.08, (real)0.05 * m * g / (real)0.08);
    fy = clampr(-d0 * m * (sol->bcoef * vy + ay), -(real)1.2 * m * g, (real)1.2 * m * g);
    ax += fx / mx; ay += fy / m;
    tlim = (real)1.2 * m * g * (real)0.15;
  } else {
    real me = (real)1.4 * m, lim = (real)0.05 * m * g / (real)0.14;
    fx = -d0 * me * (sol->bcoef * vx + ax); fy = -d0 * me * (sol->bcoef * vy + ay);
    real f2 = fx * fx + fy * fy;
    if (f2 > lim * lim) { real sc = lim / R_SQRT(f2); fx *= sc; fy *= sc; }
    ax += fx / me; ay += fy / me;
    tlim = (real)0.003 * m * g;
  }
  b->ax = c * ax - s * ay; b->ay = s * ax + c * ay;
  real t = clampr(-d0 * w->box_I * (sol->bcoef * b->w + b->aw), -tlim, tlim);
  b->aw += t / w->box_I;
}

/* HaulBox (haul_box.py:21-29): spatial tendon robot site <-> box site, limited to [0, .75].
 * Sites are the robot origin (z .1) and the box centre (z .2): length L = sqrt(d^2 + .1^2).
 * Beyond the limit it acts as a soft constraint (default solref/solimp) pulling the two
 * together along the tendon; its planar Jacobian is (d/L) times the unit vector. */
static int haul_tendon(World* w, const Sol* sol) {
  Body* A = &w->robot; Body* B = &w->box;
  real dx = B->x - A->x, dy = B->y - A->y;
  real d = R_SQRT(dx * dx + dy * dy), L = R_SQRT(d * d + (real)0.01);
  real viol = L - (real)0.75;
  if (viol <= 0 || d < (real)1e-9) return 0;
  real jx = dx / L, jy = dy / L; /* d(L)/d(pB) ; d(L)/d(pA) = -j */
  /* rate and acceleration of L along the constraint */
  real Ldot = jx * (B->vx - A->vx) + jy * (B->vy - A->vy);
  real Lacc = jx * (B->ax - A->ax) + jy * (B->ay - A->ay);
  real ua[3], ub[3];
  real Ainv = minv_apply(A, jx, jy, 0, ua) + minv_apply(B, jx, jy, 0, ub);
  real dimp = impedance(viol);
  /* want Lacc -> aref = -b*Ldot - k*viol */
  if (g_sweeps > 1) {
    real f = pgs_step(acc_slot(), -sol->bcoef * Ldot - sol->kcoef * viol, Lacc, Ainv, dimp, -(real)1e30, 0);
    A->ax -= ua[0] * f; A->ay -= ua[1] * f; A->aw -= ua[2] * f;
    B->ax += ub[0] * f; B->ay += ub[1] * f; B->aw += ub[2] * f;
    return 1;
  }
  real f = dimp * ((-sol->bcoef * Ldot - sol->kcoef * viol) - Lacc) / Ainv;
  if (f >= 0) return 0; /* a tendon only pulls */
  A->ax -= ua[0] * f; A->ay -= ua[1] * f; A->aw -= ua[2] * f;
  B->ax += ub[0] * f; B->ay += ub[1] * f; B->aw += ub[2] * f;
  return 1;
}

static void point_minv(Body* b, real damp, real h) {
  /* (M + h D)^-1 for M(yaw) of the Point robot about the body origin */
  real c = R_COS(b->yaw), s = R_SIN(b->yaw);
  real m = (real)PT_MASS + h * damp, I = (real)PT_IO + h * (real)PT_DAMP_Z;
  real a = -(real)PT_MC * s, bb = (real)PT_MC * c;
  /* M = [[m,0,a],[0,m,bb],[a,bb,I]] ; det = m (m I - a^2 - bb^2) */
  real det = m * (m * I - a * a - bb * bb);
  real id = 1 / det;
  b->minv[0] = (m * I - bb * bb) * id; /* xx */
  b->minv[1] = (a * bb) * id;          /* xy */
  b->minv[2] = (-a * m) * id;          /* xw */
  b->minv[3] = (m * I - a * a) * id;   /* yy */
  b->minv[4] = (-bb * m) * id;         /* yw */
  b->minv[5] = (m * m) * id;           /* ww */
}

static void car_body(Body* r);
static void world_from_env_r(const OEnv* e, World* w, int robot) {
  memset(w, 0, sizeof(*w));
  w->robot_id = robot;
  for (int k = 0; k < SAG_ROBOT_EXT_FLOATS; k++) w->ext[k] = e->f[SAG_F_ROBOT_EXT + k];
  const real* f = e->f;
  Body* r = &w->robot;
  r->x = f[SAG_F_ROBOT]; r->y = f[SAG_F_ROBOT + 1]; r->yaw = f[SAG_F_ROBOT + 2];
  r->vx = f[SAG_F_ROBOT + 3]; r->vy = f[SAG_F_ROBOT + 4]; r->w = f[SAG_F_ROBOT + 5];
  r->dynamic = 1; r->ngeom = 2;
  material_default(r);
  r->g[0].type = 0; r->g[0].a = (real)0.1;                       /* point.xml:18 */
  r->g[1].type = 1; r->g[1].ox = (real)0.1; r->g[1].a = (real)0.05; r->g[1].b = (real)0.05; /* :19 */
  if (robot == SAG_ROBOT_CAR) car_body(r);
  w->r_robot = bound_radius(r);
  w->nV = e->i[SAG_I_NV]; w->nP = e->i[SAG_I_NP]; w->nB = e->i[SAG_I_NB];
  real vs = f[SAG_F_VASE_SIZE];
  for (int k = 0; k < w->nV; k++) {
    Body* b = &w->vase[k];
    const real* v = f + SAG_F_VASES + 6 * k;
    b->x = v[0]; b->y = v[1]; b->yaw = v[2]; b->vx = v[3]; b->vy = v[4]; b->w = v[5];
    b->dynamic = 1; b->ngeom = 1; b->g[0].type = 1; b->g[0].a = vs; b->g[0].b = vs;
    material_default(b);
  }
  /* vase mass properties: box of half extent vs, density .001 */
  w->vase_m = (real)VASE_DENSITY * 8 * vs * vs * vs;
  w->vase_I = w->vase_m * (4 * vs * vs + 4 * vs * vs) / 12;
  w->vase_reff = vs * (real)1.41421356237309504880;
  for (int k = 0; k < w->nV; k++) {
    Body* b = &w->vase[k];
    b->minv[0] = 1 / w->vase_m; b->minv[3] = 1 / w->vase_m; b->minv[5] = 1 / w->vase_I;
  }
  w->r_vase = vs * (real)1.41421356237309504880;
  for (int k = 0; k < w->nP; k++)
    body_static_circle(&w->pillar[k], f[SAG_F_PILLARS + 2 * k], f[SAG_F_PILLARS + 2 * k + 1],
                       f[SAG_F_PILLAR_SIZE]);
  w->r_pillar = f[SAG_F_PILLAR_SIZE];
  for (int k = 0; k < w->nB; k++)
    body_static_circle(&w->button[k], f[SAG_F_BUTTONS + 2 * k], f[SAG_F_BUTTONS + 2 * k + 1],
                       (real)BUTTON_R);
  w->r_button = (real)BUTTON_R;
  box_from_env(e, w);
  w->awake0 = (uint32_t)e->i[SAG_I_AWAKE];
}

static void world_to_env(const World* w, OEnv* e) {
  real* f = e->f;
  const Body* r = &w->robot;
  if (w->robot_id == SAG_ROBOT_DOGGO && w->dg) {
    dg_store((const Doggo*)w->dg, f + SAG_F_ROBOT, f + SAG_F_ROBOT_EXT);
  } else {
    for (int k = 0; k < SAG_ROBOT_EXT_FLOATS; k++) f[SAG_F_ROBOT_EXT + k] = w->ext[k];
  }
  if (w->robot_id != SAG_ROBOT_DOGGO) {
    f[SAG_F_ROBOT] = r->x; f[SAG_F_ROBOT + 1] = r->y; f[SAG_F_ROBOT + 2] = r->yaw;
    f[SAG_F_ROBOT + 3] = r->vx; f[SAG_F_ROBOT + 4] = r->vy; f[SAG_F_ROBOT + 5] = r->w;
  }
  for (int k = 0; k < w->nV; k++) {
    const Body* b = &w->vase[k];
    real* v = f + SAG_F_VASES + 6 * k;
    v[0] = b->x; v[1] = b->y; v[2] = b->yaw; v[3] = b->vx; v[4] = b->vy; v[5] = b->w;
  }
  if (w->box_kind != SAG_BOX_NONE) {
    const Body* b = &w->box;
    real* v = f + SAG_F_BOX;
    v[0] = b->x; v[1] = b->y; v[2] = b->yaw; v[3] = b->vx; v[4] = b->vy; v[5] = b->w;
  }
  /* SAG_I_AWAKE as the device exports it: awake although at rest */
  uint32_t at_rest = w->awake0;
  for (int k = 0; k < w->nV; k++)
    if (w->vase[k].vx != 0 || w->vase[k].vy != 0 || w->vase[k].w != 0) at_rest &= ~(1u << k);
  if (w->box_kind == SAG_BOX_NONE || w->box.vx != 0 || w->box.vy != 0 || w->box.w != 0) at_rest &= ~(1u << SAG_MAX_VASES);
  e->i[SAG_I_AWAKE] = (int32_t)at_rest;
}

/* ------------------------------------------------------------------------ */
/* Point robot forward dynamics (point.xml)                  */
/* ------------------------------------------------------------------------ */
/* ctrl: already noised + clipped to ctrlrange (safe_adaptation_gym.py:63-67).
 * Generalised coordinates (x, y, yaw) of the body origin in the world frame:
 * slide x, slide y precede the hinge, so their axes stay fixed (rotated by the
 * constant rot0), and isotropic damping makes the world frame equivalent.
 *   M(q) qacc + bias = tau_act - D qvel + J^T f_contact, integrated with MuJoCo's
 *   Euler: damping implicit ((M + h D) qacc = ...), velocity then position. */
static void point_smooth(Body* r, const real ctrl[2], real gear, real damp, real h) {
  real c = R_COS(r->yaw), s = R_SIN(r->yaw);
  /* motor on site "robot": force gear*clip(ctrl, forcerange) along body +x (point.xml:36) */
  real f0 = gear * clampr(ctrl[0], -(real)PT_FORCE_LIM, (real)PT_FORCE_LIM);
  /* tau - bias - D v;  bias = (-mc c w^2, -mc s w^2, 0) */
  real Fx = f0 * c - damp * r->vx + (real)PT_MC * c * r->w * r->w;
  real Fy = f0 * s - damp * r->vy + (real)PT_MC * s * r->w * r->w;
  real Tz0 = -(real)PT_DAMP_Z * r->w;
  point_minv(r, damp, h);
  const real* m = r->minv;
  /* velocity servo on hinge z (point.xml:37): torque gear*clip(kv*(ctrl - gear*w), forcerange).
   * Its velocity feedback is stiff: h*kv*gear^2/Izz = 12.7 per substep, far outside the
   * stability region of an explicit update (MuJoCo's Euler treats only joint damping
   * implicitly and would limit-cycle at +-1 rad/s here).  SPECIFICATION: the servo is
   * evaluated at the end-of-substep yaw rate, solved exactly for the clipped-linear law:
   * w_lin = implicit linear prediction; the clip of its error gives the torque, which is the
   * exact root in the linear zone and the correctly signed limit outside it (the map is
   * monotone). */
  real A = r->w + h * (m[2] * Fx + m[4] * Fy + m[5] * Tz0);
  real g = h * m[5];
  real w_lin = (A + g * (real)(PT_GEAR_Z * PT_KV) * ctrl[1]) / (1 + g * (real)(PT_GEAR_Z * PT_GEAR_Z * PT_KV));
  real f1 = (real)PT_GEAR_Z *
            clampr((real)PT_KV * (ctrl[1] - (real)PT_GEAR_Z * w_lin), -(real)PT_FORCE_LIM, (real)PT_FORCE_LIM);
  real Tz = f1 + Tz0;
  r->ax = m[0] * Fx + m[1] * Fy + m[2] * Tz;
  r->ay = m[1] * Fx + m[3] * Fy + m[4] * Tz;
  r->aw = m[2] * Fx + m[4] * Fy + m[5] * Tz;
}


/* ------------------------------------------------------------------------ */
/* Car robot (assets/xmls/car.xml) - planar reduction        */
/* ------------------------------------------------------------------------ */
/* SPECIFICATION.  The MJCF car is a free-joint base on two driven wheels (hinge about the
 * body x axis, armature 2.5e-4, motor gear 1, force range +-.02) and a rear ball caster.  Its
 * three floor contacts hold height, pitch and roll; those three stiff DoF are frozen here and
 * the remaining eight are integrated: base (x, y, yaw), the two wheel rates and the ball's
 * angular velocity (plus the ball quaternion, which the observation needs).  Floor contact is
 * reduced to regularised Coulomb friction at the three contact points under their static
 * normal loads, with the same soft-constraint reference (b = 2 / (dmax tc)) as body contacts:
 * wheels resist lateral slip and couple longitudinal slip to the wheel rate (radius .05); the
 * ball couples both slip components to its spin.  Geoms keep their planar footprints. */
#define CAR_DENS 5.0
#define CAR_RW 0.05
#define CAR_ARMATURE 0.00025
#define CAR_JDAMP 0.001
#define CAR_FLIM 0.02
typedef struct { real m, Io, ox, oy, Iw, Ib, N[3]; } CarK;

static void car_body(Body* r) {
  /* car.xml:16-32: footprints of the eight geoms in the base frame */
  static const real G[8][5] = {/* type, ox, oy, a, b */
      {1, 0, 0, 0.1, 0.1},        {1, 0, 0.15, 0.1, 0.01},     {1, 0, 0.125, 0.01, 0.025},
      {1, 0, -0.165, 0.05, 0.01}, {1, 0, -0.13, 0.05, 0.03},   {1, -0.13, 0.1, 0.025, 0.05},
      {1, 0.13, 0.1, 0.025, 0.05}, {0, 0, -0.1, 0.05, 0}};
  r->ngeom = 8;
  for (int k = 0; k < 8; k++) {
    r->g[k].type = (int)G[k][0]; r->g[k].ox = G[k][1]; r->g[k].oy = G[k][2];
    r->g[k].a = G[k][3]; r->g[k].b = G[k][4];
  }
}

static CarK car_constants(void) {
  /* masses from the 3-D geoms (density 5): boxes 8 hx hy hz, wheels pi r^2 L, ball 4/3 pi r^3 */
  static const double HZ[5] = {0.05, 0.05, 0.03, 0.05, 0.01};
  Body b; memset(&b, 0, sizeof(b)); car_body(&b);
  double m = 0, mx = 0, my = 0, Io = 0;
  for (int k = 0; k < 8; k++) {
    const Geom* g = &b.g[k];
    double mk, Ik;
    if (k < 5) { mk = CAR_DENS * 8 * g->a * g->b * HZ[k]; Ik = mk * (g->a * g->a + g->b * g->b) / 3; }
    else if (k < 7) { mk = CAR_DENS * PI * 0.05 * 0.05 * 0.05; Ik = mk * (3 * 0.05 * 0.05 + 0.05 * 0.05) / 12; }
    else { mk = CAR_DENS * 4.0 / 3.0 * PI * 0.05 * 0.05 * 0.05; Ik = 0.4 * mk * 0.05 * 0.05; }
    m += mk; mx += mk * g->ox; my += mk * g->oy;
    Io += Ik + mk * (g->ox * g->ox + g->oy * g->oy);
  }
  CarK c;
  c.m = (real)m; c.Io = (real)Io; c.ox = (real)(mx / m); c.oy = (real)(my / m);
  double mw = CAR_DENS * PI * 0.05 * 0.05 * 0.05, mb = CAR_DENS * 4.0 / 3.0 * PI * 0.05 * 0.05 * 0.05;
  c.Iw = (real)(0.5 * mw * 0.05 * 0.05 + CAR_ARMATURE);
  c.Ib = (real)(0.4 * mb * 0.05 * 0.05);
  /* static loads on (left, right, caster) at y = .1, .1, -.1, x = -.13, .13, 0 */
  double W = m * GRAVITY, yc = my / m;
  double NL = W * (0.1 + yc) / 0.4;
  c.N[0] = (real)NL; c.N[1] = (real)NL; c.N[2] = (real)(W - 2 * NL);
  return c;
}

void sago_car_constants(double out[9]) {
  CarK c = car_constants();
  out[0] = c.m; out[1] = c.Io; out[2] = c.ox; out[3] = c.oy; out[4] = c.Iw; out[5] = c.Ib;
  out[6] = c.N[0]; out[7] = c.N[1]; out[8] = c.N[2];
}

/* one regularised friction direction at a contact point of the base, optionally coupled to a
 * spinning part (wheel / ball) through lever `rw` and inertia `Ispin` */
static real car_friction(Body* r, real dx, real dy, real rx, real ry, real spin_rate, real* spin_acc,
                         real rw, real Ispin, real limit, real bcoef, real* facc) {
  (void)facc;
  real u[3];
  real A = minv_apply(r, dx, dy, rx * dy - ry * dx, u);
  real slip = (r->vx - r->w * ry) * dx + (r->vy + r->w * rx) * dy + rw * spin_rate;
  real sacc = (r->ax - r->aw * ry) * dx + (r->ay + r->aw * rx) * dy + rw * (spin_acc ? *spin_acc : 0);
  if (spin_acc) A += rw * rw / Ispin;
  real f;
  if (g_sweeps > 1)   /* converged reference: accumulated force, PGS increment */
    f = pgs_step(acc_slot(), -bcoef * slip, sacc, A, (real)SOL_D0, -limit, limit);
  else f = clampr((real)SOL_D0 * (-bcoef * slip - sacc) / A, -limit, limit);
  r->ax += u[0] * f; r->ay += u[1] * f; r->aw += u[2] * f;
  if (spin_acc) *spin_acc += rw * f / Ispin;
  return f;
}

/* part 1: inverse inertia, centrifugal term, motor / damping of the spinning parts; part 2: the six floor-friction
 * elements as one plain sweep inside the N-sweep reference mode (sago_set_sweeps; the specification itself solves
 * them with car_floor_friction below) */
static void car_smooth(World* w, const real ctrl[2], const Sol* sol, int part) {
  Body* r = &w->robot;
  const CarK k = car_constants();
  real h = sol->h;
  real c = R_COS(r->yaw), s = R_SIN(r->yaw);
  real Iw = k.Iw + h * (real)CAR_JDAMP, Ib = k.Ib + h * (real)CAR_JDAMP;
  real* acc = w->ext_acc;
  if (part & 1) {
  /* COM offset in world axes, M = [[m,0,-m oy],[0,m,m ox],[.,.,Io]], bias = -m w^2 o */
  real ox = c * k.ox - s * k.oy, oy = s * k.ox + c * k.oy;
  real a = -k.m * oy, b = k.m * ox, m = k.m, I = k.Io;
  real id = 1 / (m * (m * I - a * a - b * b));
  r->minv[0] = (m * I - b * b) * id; r->minv[1] = (a * b) * id; r->minv[2] = (-a * m) * id;
  r->minv[3] = (m * I - a * a) * id; r->minv[4] = (-b * m) * id; r->minv[5] = (m * m) * id;
  real Fx = k.m * r->w * r->w * ox, Fy = k.m * r->w * r->w * oy;
  r->ax = r->minv[0] * Fx + r->minv[1] * Fy; r->ay = r->minv[1] * Fx + r->minv[3] * Fy;
  r->aw = r->minv[2] * Fx + r->minv[4] * Fy;
  /* wheels: motor torque clip(ctrl, +-.02) (gear 1), joint damping implicit */
  for (int i = 0; i < 2; i++)
    acc[i] = (clampr(ctrl[i], -(real)CAR_FLIM, (real)CAR_FLIM) - (real)CAR_JDAMP * w->ext[i]) / Iw;
  for (int i = 0; i < 3; i++) acc[2 + i] = -(real)CAR_JDAMP * w->ext[2 + i] / Ib;
  for (int i = 0; i < 5; i++) w->ext_acc0[i] = acc[i];
  w->car_fa[0] = w->car_fa[1] = w->car_fa[2] = 0;
  }
  if (!(part & 2)) return;
  /* floor friction: left, right (longitudinal = body y, coupled to the wheel; lateral = body x),
   * then the caster (x coupled to -ball_y spin, y to +ball_x spin) */
  static const real PX[3] = {-0.13, 0.13, 0}, PY[3] = {0.1, 0.1, -0.1};
  real xbx = c, xby = s, ybx = -s, yby = c;
  real facc[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 3; i++) {
    real rx = c * PX[i] - s * PY[i], ry = s * PX[i] + c * PY[i];
    real lim = (real)FRICTION_MU * k.N[i];
    if (i < 2) {
      car_friction(r, ybx, yby, rx, ry, w->ext[i], &acc[i], (real)CAR_RW, Iw, lim, sol->bcoef, &facc[2 * i]);
      car_friction(r, xbx, xby, rx, ry, 0, NULL, 0, 1, lim, sol->bcoef, &facc[2 * i + 1]);
    } else {
      /* x slip = v.x - r * ball_y  -> spin variable -ball_y;  y slip = v.y + r * ball_x */
      real sy_rate = -w->ext[3], sy_acc = -acc[3];
      car_friction(r, xbx, xby, rx, ry, sy_rate, &sy_acc, (real)CAR_RW, Ib, lim, sol->bcoef, &facc[4]);
      acc[3] = -sy_acc;
      car_friction(r, ybx, yby, rx, ry, w->ext[2], &acc[2], (real)CAR_RW, Ib, lim, sol->bcoef, &facc[5]);
    }
  }
}

/* SPECIFICATION (round 3): the car's floor friction solved to convergence in every forward
 * evaluation instead of one Gauss-Seidel sweep over six elements.
 *  - FIVE elements in the body frame, where their Jacobians, M^-1 J^T and J M^-1 J^T are constants: rolling
 *    direction of the left wheel, the two wheels' LATERAL elements merged, rolling direction of the right wheel,
 *    caster x, caster y.  (The wheels' lateral elements share Jacobian (body x at y = .1, whatever x), bound and
 *    reference, so they carry equal forces: one element with bound 2 mu N and regulariser R / 2.  As two rows they
 *    were the sweep's slow mode - factor d0^2 = .81 per sweep.)
 *  - start: the direct solve with every element unclamped.  Eliminating f_i = (c_i - J_i a) / D_i from
 *    J_i a + r_i alpha_i + R_i f_i = aref_i (alpha_i = alpha0_i + r_i f_i / I_i the spin acceleration the element drives,
 *    c_i = aref_i - r_i alpha0_i, D_i = r_i^2 / I_i + R_i) leaves (M + sum_i J_i^T J_i / D_i) a = M a_other + sum_i J_i^T c_i / D_i,
 *    a constant 3 x 3 matrix (block diagonal {x, yaw} + {y} by the car's symmetry); forces clamped to +-mu N;
 *  - then CAR_FRICTION_SWEEPS projected Gauss-Seidel sweeps over the five accumulated forces (exact start when
 *    nothing clamps);
 *  - WARM START (as MuJoCo's solver): only the first solve of an env-step is done that way; every later one - the
 *    re-solve after contacts, the next substeps - starts from the previous solve's five forces and runs
 *    CAR_FRICTION_SWEEPS_WARM sweeps (the forces change little from one substep to the next); the forward evaluation
 *    at the final state (mj_forward, safe_adaptation_gym.py:76: accelerometer, contact flags) starts cold again, so
 *    that the observation is a function of the state alone;
 *    together 6e-5 m per env-step (median) from the converged solution, where one cold sweep of the six elements
 *    left 3e-3: profiles/r03_sweep_convergence.txt;
 *  - solved before the robot's contacts and, when these changed the base acceleration, once more after them
 *    (`a_other` then includes the contact forces).
 * R_i = A_ii (1 - d0) / d0 with A_ii = J_i M^-1 J_i^T + r_i^2 / I_i, bounds mu N_i under the static loads: as before. */
#define CAR_FRICTION_SWEEPS 4        /* after the direct solve (the first solve of an env-step) */
#define CAR_FRICTION_SWEEPS_WARM 2   /* a solve that starts from the previous solve's forces */
/* the five forces of the previous solve of this env-step (set by sago_step around its forward evaluations; a lone
 * forward evaluation - sago_observe - has none and solves cold) */
typedef struct { int have; real f[5]; } CarWarm;
static _Thread_local CarWarm* g_carws = NULL;
static void car_floor_friction(World* w, const Sol* sol) {
  Body* r = &w->robot;
  const CarK k = car_constants();
  const real h = sol->h, c = R_COS(r->yaw), s = R_SIN(r->yaw);
  const real iIw = 1 / (k.Iw + h * (real)CAR_JDAMP), iIb = 1 / (k.Ib + h * (real)CAR_JDAMP), RF = (1 - (real)SOL_D0) / (real)SOL_D0;
  /* inertia about the base origin, body axes: [[m, 0, ma], [0, m, 0], [ma, 0, I]] (the COM lies on the body y axis) */
  const real m = k.m, ma = -k.m * k.oy, I = k.Io;
  const real idm = 1 / (m * I - ma * ma);
  const real Mi0 = I * idm, Mi2 = -ma * idm, Mi3 = 1 / m, Mi5 = m * idm;          /* M^-1: xx, xw, yy, ww (xy = yw = 0) */
  /* element: direction (body x or y), J = (dx, dy, jw), spin lever r_i / inverse spin inertia, bound, R scale */
  static const int ISX[5] = {0, 1, 0, 1, 0};
  static const real JW[5] = {-0.13, -0.1, 0.13, 0.1, 0};                          /* rx dy - ry dx */
  const real rw[5] = {(real)CAR_RW, 0, (real)CAR_RW, (real)CAR_RW, (real)CAR_RW};
  const real iIs[5] = {iIw, 0, iIw, iIb, iIb};
  const real lim[5] = {(real)FRICTION_MU * k.N[0], 2 * (real)FRICTION_MU * k.N[0], (real)FRICTION_MU * k.N[1],
                       (real)FRICTION_MU * k.N[2], (real)FRICTION_MU * k.N[2]};
  static const real RSC[5] = {1, 0.5, 1, 1, 1};
  const real rate[5] = {w->ext[0], 0, w->ext[1], -w->ext[3], w->ext[2]};           /* caster x slip <-> -ball_y spin */
  const real al0[5] = {w->ext_acc0[0], 0, w->ext_acc0[1], -w->ext_acc0[3], w->ext_acc0[2]};
  real U0[5], U1[5], U2[5], Rr[5], iAR[5], wD[5];
  for (int i = 0; i < 5; i++) {
    if (ISX[i]) { U0[i] = Mi0 + Mi2 * JW[i]; U1[i] = 0; U2[i] = Mi2 + Mi5 * JW[i]; }
    else { U0[i] = Mi2 * JW[i]; U1[i] = Mi3; U2[i] = Mi5 * JW[i]; }
    const real sp = rw[i] * rw[i] * iIs[i];
    const real A = (ISX[i] ? U0[i] : U1[i]) + JW[i] * U2[i] + sp;
    Rr[i] = RSC[i] * A * RF;
    iAR[i] = 1 / (A + Rr[i]);
    wD[i] = 1 / (sp + Rr[i]);
  }
  const real K00 = m + wD[1] + wD[3], K02 = ma + JW[1] * wD[1] + JW[3] * wD[3], K11 = m + wD[0] + wD[2] + wD[4];
  const real K22 = I + JW[0] * JW[0] * (wD[0] + wD[2]) + JW[1] * JW[1] * wD[1] + JW[3] * JW[3] * wD[3];
  const real idk = 1 / (K00 * K22 - K02 * K02);
  /* base velocity, and the base acceleration from everything but the floor friction, in body axes */
  const real vbx = c * r->vx + s * r->vy, vby = c * r->vy - s * r->vx;
  real a0 = c * r->ax + s * r->ay - w->car_fa[0], a1 = c * r->ay - s * r->ax - w->car_fa[1], a2 = r->aw - w->car_fa[2];
  real aref[5], cc[5];
  for (int i = 0; i < 5; i++) {
    const real slip = (ISX[i] ? vbx : vby) + JW[i] * r->w + rw[i] * rate[i];
    aref[i] = -sol->bcoef * slip;
    cc[i] = aref[i] - rw[i] * al0[i];
  }
  const real b0 = m * a0 + ma * a2 + wD[1] * cc[1] + wD[3] * cc[3];
  const real b1 = m * a1 + wD[0] * cc[0] + wD[2] * cc[2] + wD[4] * cc[4];
  real b2 = ma * a0 + I * a2;
  for (int i = 0; i < 5; i++) b2 += wD[i] * cc[i] * JW[i];
  const real n0 = (K22 * b0 - K02 * b2) * idk, n1 = b1 / K11, n2 = (K00 * b2 - K02 * b0) * idk;
  real f[5], al[5], fa0 = 0, fa1 = 0, fa2 = 0;
  const int warm = g_carws && g_carws->have;
  for (int i = 0; i < 5; i++) {
    f[i] = warm ? g_carws->f[i] : clampr(wD[i] * (cc[i] - ((ISX[i] ? n0 : n1) + JW[i] * n2)), -lim[i], lim[i]);
    fa0 += U0[i] * f[i]; fa1 += U1[i] * f[i]; fa2 += U2[i] * f[i];
    al[i] = al0[i] + rw[i] * f[i] * iIs[i];
  }
  for (int sw = 0; sw < (warm ? CAR_FRICTION_SWEEPS_WARM : CAR_FRICTION_SWEEPS); sw++)
    for (int i = 0; i < 5; i++) {
      const real sa = (ISX[i] ? a0 + fa0 : a1 + fa1) + JW[i] * (a2 + fa2) + rw[i] * al[i];
      const real fn = clampr(f[i] + (aref[i] - sa - Rr[i] * f[i]) * iAR[i], -lim[i], lim[i]);
      const real df = fn - f[i];
      f[i] = fn;
      fa0 += U0[i] * df; fa1 += U1[i] * df; fa2 += U2[i] * df;
      al[i] += rw[i] * df * iIs[i];
    }
  if (g_carws) { g_carws->have = 1; for (int i = 0; i < 5; i++) g_carws->f[i] = f[i]; }
  a0 += fa0; a1 += fa1; a2 += fa2;
  w->car_fa[0] = fa0; w->car_fa[1] = fa1; w->car_fa[2] = fa2;
  r->ax = c * a0 - s * a1; r->ay = s * a0 + c * a1; r->aw = a2;
  real* acc = w->ext_acc;
  acc[0] = al[0]; acc[1] = al[2]; acc[3] = -al[3]; acc[2] = al[4]; acc[4] = w->ext_acc0[4];
}

static void car_integrate_ext(World* w, real h) {
  for (int i = 0; i < 5; i++) w->ext[i] += h * w->ext_acc[i];
  /* ball quaternion: rate is relative to the base, in base (parent) axes: q <- exp(h W / 2) q */
  real wx = w->ext[2], wy = w->ext[3], wz = w->ext[4];
  real n = R_SQRT(wx * wx + wy * wy + wz * wz);
  if (n > 0) {
    real half = (real)0.5 * h * n, sn = R_SIN(half) / n, cs = R_COS(half);
    real dw = cs, dx = sn * wx, dy = sn * wy, dz = sn * wz;
    real* q = &w->ext[5];
    real qw = dw * q[0] - dx * q[1] - dy * q[2] - dz * q[3];
    real qx = dw * q[1] + dx * q[0] + dy * q[3] - dz * q[2];
    real qy = dw * q[2] - dx * q[3] + dy * q[0] + dz * q[1];
    real qz = dw * q[3] + dx * q[2] - dy * q[1] + dz * q[0];
    real qn = R_SQRT(qw * qw + qx * qx + qy * qy + qz * qz);
    q[0] = qw / qn; q[1] = qx / qn; q[2] = qy / qn; q[3] = qz / qn;
  }
}

/* One forward-dynamics evaluation: accelerations of every body at the current
 * state.  Fixed pair order (specification):
 *   robot-pillars, robot-buttons, robot-vases, robot-box, haul tendon,
 *   vase-pillars/buttons, box-pillars/buttons, free-body pairs (i<j, lexicographic, the task
 *   object counting as the last body), then floor friction of each vase and of the box.
 * (Round 4, ADVICE r3: bodies that OVERLAP something when a new world is installed are flagged awake for the first
 * forward evaluation - SAG_I_AWAKE, set by sag_set_layout's bounding-circle test; the oracle takes the flags from the
 * record - so a HaulBox box spawned over a vase or pillar is pushed clear in the first steps, as MuJoCo does.  The
 * rest-capture rule below concerns bodies that came to rest DURING the simulation.)
 * SLEEPING BODIES (specification; the device kernels skip them): a free body takes part in a forward evaluation
 * only if it is ACTIVE - it has a non-zero velocity component, or a contact with the robot (or the pull of the
 * tether) was found in this evaluation, or a contact with an active free body was found in the pair walk.  Its
 * contacts with the static circles are evaluated only if it is active when that phase starts, a pair of free bodies
 * only if one of the two is active when the pair walk starts.  A body at rest that nothing active touches therefore
 * stays exactly at rest, also while it still overlaps another resting body by the depth at which the rest capture
 * caught it (evaluating such pairs made the two creep apart at ~2e-5 m/s).  The N-sweep reference mode
 * (sago_set_sweeps) evaluates every pair.
 * Returns robot/obstacle penetration count (cost rule, mujoco_bridge.py:177-191
 * with prefixes consts.OBSTACLES; the box is not an obstacle) and the button contact mask. */
static int world_forward(World* w, const OEnv* e, const real* ctrl, const Sol* sol,
                         uint32_t* btn_mask) {
  if (w->robot_id == SAG_ROBOT_CAR) car_smooth(w, ctrl, sol, 1);
  else if (w->robot_id == SAG_ROBOT_POINT) point_smooth(&w->robot, ctrl, e->f[SAG_F_GEAR], e->f[SAG_F_DAMP], sol->h);
  for (int k = 0; k < w->nV; k++) { w->vase[k].ax = w->vase[k].ay = w->vase[k].aw = 0; }
  w->box.ax = w->box.ay = w->box.aw = 0;
  if (w->box_kind == SAG_BOX_ROD) { /* orientation-dependent inverse inertia */
    real c = R_COS(w->box.yaw), s = R_SIN(w->box.yaw), ix = 1 / ((real)1.5 * w->box_m), iy = 1 / w->box_m;
    w->box.minv[0] = c * c * ix + s * s * iy; w->box.minv[1] = c * s * (ix - iy);
    w->box.minv[3] = s * s * ix + c * c * iy;
  }
  const int has_box = w->box_kind != SAG_BOX_NONE;
  for (int k = 0; k < w->nV; k++) w->act_v[k] = g_sweeps > 1 || (w->awake0 >> k & 1u) || w->vase[k].vx != 0 || w->vase[k].vy != 0 || w->vase[k].w != 0;
  w->act_box = has_box && (g_sweeps > 1 || (w->awake0 >> SAG_MAX_VASES & 1u) || w->box.vx != 0 || w->box.vy != 0 || w->box.w != 0);
  int cost_contacts = 0;
  uint32_t mask = 0;
  /* every robot row of the Doggo (limits, floor, pillars, buttons, vases, box, tether) in one PGS of its own */
  if (w->robot_id == SAG_ROBOT_DOGGO) cost_contacts = dg_forward(w, (Doggo*)w->dg, ctrl, sol, &mask);
  /* the planar constraints: one sweep (the specification) or, in the study mode, g_sweeps over accumulated forces */
  for (g_sweep = 0; g_sweep < g_sweeps; g_sweep++) {
  g_ai = 0;
  int cc = 0;
  uint32_t mk = 0;
  if (w->robot_id == SAG_ROBOT_DOGGO) goto free_bodies;
  real car_pre[3] = {0, 0, 0};
  if (w->robot_id == SAG_ROBOT_CAR) {
    if (g_sweeps > 1) car_smooth(w, ctrl, sol, 2);   /* converged reference: the six elements inside the N sweeps */
    else { car_floor_friction(w, sol); car_pre[0] = w->robot.ax; car_pre[1] = w->robot.ay; car_pre[2] = w->robot.aw; }
  }
  for (int p = 0; p < w->nP; p++)
    cc += collide_pair(&w->robot, &w->pillar[p], w->r_robot, w->r_pillar, sol, 1);
  for (int b = 0; b < w->nB; b++)
    if (collide_pair(&w->robot, &w->button[b], w->r_robot, w->r_button, sol, 1)) mk |= 1u << b;
  for (int k = 0; k < w->nV; k++) {
    const int nc = collide_pair(&w->robot, &w->vase[k], w->r_robot, w->r_vase, sol, 1);
    cc += nc;
    if (nc) w->act_v[k] = 1;
  }
  if (has_box) {
    if (w->box_kind == SAG_BOX_BALL) {
      /* the ball's centre is .04 above the robot sphere's: they touch at horizontal distance
       * sqrt(.24^2 - .04^2) - the ball presents radius .1366 to the sphere, .14 to the arrow */
      Body sphere = w->robot, arrow = w->robot;
      sphere.ngeom = 1; arrow.ngeom = 1; arrow.g[0] = w->robot.g[1];
      Body ball = w->box;
      ball.g[0].a = (real)(0.23664319132398464 - 0.1);
      if (collide_pair(&sphere, &ball, w->r_robot, w->r_box, sol, 1)) w->act_box = 1;
      w->robot.ax = sphere.ax; w->robot.ay = sphere.ay; w->robot.aw = sphere.aw;
      w->box.ax = ball.ax; w->box.ay = ball.ay; w->box.aw = ball.aw;
      arrow.ax = sphere.ax; arrow.ay = sphere.ay; arrow.aw = sphere.aw;
      if (collide_pair(&arrow, &w->box, w->r_robot, w->r_box, sol, 1)) w->act_box = 1;
      w->robot.ax = arrow.ax; w->robot.ay = arrow.ay; w->robot.aw = arrow.aw;
    } else {
      if (collide_pair(&w->robot, &w->box, w->r_robot, w->r_box, sol, 1)) w->act_box = 1;
    }
    if (w->haul && haul_tendon(w, sol)) w->act_box = 1;
  }
  /* the car's floor friction once more when its contacts (or the tether) changed the base acceleration */
  if (w->robot_id == SAG_ROBOT_CAR && g_sweeps == 1 &&
      (w->robot.ax != car_pre[0] || w->robot.ay != car_pre[1] || w->robot.aw != car_pre[2])) car_floor_friction(w, sol);
  if (g_sweep == 0) { cost_contacts = cc; mask = mk; }
free_bodies:
  for (int k = 0; k < w->nV; k++) {
    if (!w->act_v[k]) continue;
    for (int p = 0; p < w->nP; p++)
      collide_pair(&w->vase[k], &w->pillar[p], w->r_vase, w->r_pillar, sol, 1);
    for (int b = 0; b < w->nB; b++)
      collide_pair(&w->vase[k], &w->button[b], w->r_vase, w->r_button, sol, 1);
  }
  if (w->act_box) {
    for (int p = 0; p < w->nP; p++) collide_pair(&w->box, &w->pillar[p], w->r_box, w->r_pillar, sol, 1);
    for (int b = 0; b < w->nB; b++) collide_pair(&w->box, &w->button[b], w->r_box, w->r_button, sol, 1);
  }
  /* free body pairs in lexicographic order, the task object being the last body; which pairs are walked is decided
   * by who is active NOW (a body woken inside the walk was asleep: its other pairs with sleepers stay skipped) */
  {
    int was_v[SAG_MAX_VASES], was_box = w->act_box;
    for (int k = 0; k < w->nV; k++) was_v[k] = w->act_v[k];
    for (int i = 0; i < w->nV; i++) {
      for (int j = i + 1; j < w->nV; j++)
        if ((was_v[i] || was_v[j]) && collide_pair(&w->vase[i], &w->vase[j], w->r_vase, w->r_vase, sol, 1))
          w->act_v[i] = w->act_v[j] = 1;
      if (has_box && (was_v[i] || was_box) && collide_pair(&w->vase[i], &w->box, w->r_vase, w->r_box, sol, 1))
        w->act_v[i] = w->act_box = 1;
    }
  }
  for (int k = 0; k < w->nV; k++)
    if (w->act_v[k]) floor_friction(&w->vase[k], w->vase_m, w->vase_I, w->vase_reff, sol);
  if (w->act_box) box_floor_friction(w, sol);
  }
  g_sweep = 0;
  if (btn_mask) *btn_mask = mask;
  return cost_contacts;
}

static void integrate(Body* b, real h) {
  b->vx += h * b->ax; b->vy += h * b->ay; b->w += h * b->aw;
  b->x += h * b->vx; b->y += h * b->vy; b->yaw += h * b->w;
}

/* Free bodies on the floor come to rest (static friction capture; SPECIFICATION): when the
 * new velocity and this substep's velocity change are both below REST_V (linear, m/s) /
 * REST_W (angular, rad/s) the body stops exactly, before the position update. */
#define REST_V 1e-5
#define REST_W 1e-4
static void integrate_free(Body* b, real h) {
  b->vx += h * b->ax; b->vy += h * b->ay; b->w += h * b->aw;
  if (R_FABS(b->vx) < (real)REST_V && R_FABS(b->vy) < (real)REST_V && R_FABS(b->w) < (real)REST_W &&
      R_FABS(h * b->ax) < (real)REST_V && R_FABS(h * b->ay) < (real)REST_V &&
      R_FABS(h * b->aw) < (real)REST_W) {
    b->vx = 0; b->vy = 0; b->w = 0;
  }
  b->x += h * b->vx; b->y += h * b->vy; b->yaw += h * b->w;
}

/* nstep x mj_step (safe_adaptation_gym.py:72) */
static void substeps_r(OEnv* e, const real* ctrl, int nstep, double h, int robot) {
  World w;
  Doggo dg;
  world_from_env_r(e, &w, robot);
  if (robot == SAG_ROBOT_DOGGO) { w.dg = &dg; dg_load(&dg, e->f + SAG_F_ROBOT, e->f + SAG_F_ROBOT_EXT); }
  Sol sol = make_sol(h);
  for (int s = 0; s < nstep; s++) {
    world_forward(&w, e, ctrl, &sol, NULL);
    if (robot == SAG_ROBOT_DOGGO) dg_integrate(&dg, h);
    else integrate(&w.robot, sol.h);
    if (robot == SAG_ROBOT_CAR) car_integrate_ext(&w, sol.h);
    for (int k = 0; k < w.nV; k++) integrate_free(&w.vase[k], sol.h);
    if (w.box_kind != SAG_BOX_NONE) integrate_free(&w.box, sol.h);
    w.awake0 = 0;   /* the install-time flags last for one forward evaluation + integration: from here on a body is awake by its velocity */
  }
  world_to_env(&w, e);
}
void sago_substeps(OEnv* e, const real ctrl[2], int nstep, double h) { substeps_r(e, ctrl, nstep, h, SAG_ROBOT_POINT); }
/* doggo: `nstep` substeps with controls as doubles (tests) */
void sago_doggo_substeps(OEnv* e, const double* ctrl12, int nstep, double h) {
  real c[12];
  for (int k = 0; k < 12; k++) c[k] = (real)ctrl12[k];
  DgWarm dgws; dgws.n = 0; dgws.evals = 0;
  g_dgws = &dgws;
  substeps_r(e, c, nstep, h, SAG_ROBOT_DOGGO);
  g_dgws = NULL;
}
/* doggo diagnostics for the analytic tests: total energy; mass matrix [19x19], bias [19], collision geometry
 * (the 16 floor points' lowest points [16x3], then the 14 geoms' axis end points [14x6]), total mass,
 * contact-free qacc [19] at zero control */
double sago_doggo_energy(const OEnv* e) {
  Doggo D;
  dg_load(&D, e->f + SAG_F_ROBOT, e->f + SAG_F_ROBOT_EXT);
  return dg_energy(&D);
}
void sago_doggo_debug(const OEnv* e, double* M, double* bias, double* sph, double* mass, double* qacc0) {
  Doggo D;
  dg_load(&D, e->f + SAG_F_ROBOT, e->f + SAG_F_ROBOT_EXT);
  dg_kinematics(&D);
  dg_mass_matrix(&D);
  for (int i = 0; i < DG_NV; i++) for (int j = 0; j < DG_NV; j++) M[i * DG_NV + j] = D.L[i][j];
  dg_bias(&D, bias);
  for (int s = 0; s < DG_NFP; s++) for (int k = 0; k < 3; k++) sph[3 * s + k] = D.fpt[s][k];
  for (int g = 0; g < DG_NCG; g++)
    for (int k = 0; k < 3; k++) { sph[3 * DG_NFP + 6 * g + k] = D.ga[g][k]; sph[3 * DG_NFP + 6 * g + 3 + k] = D.gb[g][k]; }
  *mass = 0;
  for (int b = 0; b < DG_NB; b++) *mass += D.Ib[b].m;
  double tau[DG_NV];
  for (int i = 0; i < DG_NV; i++) tau[i] = -bias[i];
  for (int j = 0; j < DG_NJ; j++) tau[6 + j] += -DG_STIFF * (D.q[j] - DG_SPRINGREF_DEG[j] * PI / 180);
  if (dg_cholesky(D.L)) dg_solve(D.L, tau, qacc0);
}

/* dg_seg_box_t / dg_box_sd for the tests: the parameter of the segment a + t d (box frame) nearest to the box in signed
 * distance, and that distance */
double sago_seg_box_t(double ax, double ay, double dx, double dy, double hx, double hy, double t0, double t1, double* sd) {
  const real t = dg_seg_box_t((real)ax, (real)ay, (real)dx, (real)dy, (real)hx, (real)hy, (real)t0, (real)t1);
  if (sd) *sd = (double)dg_box_sd((real)(ax + (double)t * dx), (real)(ay + (double)t * dy), (real)hx, (real)hy);
  return (double)t;
}

/* the contacts (normal rows) of ONE forward evaluation of a Doggo env at zero control: out[k] = key, point xyz,
 * normal xyz (into the robot), depth, solved normal force; returns their number, *cost_contacts = the count the
 * cost rule sees (robot geoms x vases / pillars).  Keys: sag_oracle_doggo.inc (floor 0x1000 + 4 point,
 * object 0x10000 + 4 (32 (8 (8 obj + geom of the object) + contact) + robot geom)) */
int sago_doggo_contacts(const OEnv* e, int max, double* out, int* cost_contacts) {
  World w;
  Doggo dg;
  real ctrl[12] = {0};
  world_from_env_r(e, &w, SAG_ROBOT_DOGGO);
  w.dg = &dg;
  dg_load(&dg, e->f + SAG_F_ROBOT, e->f + SAG_F_ROBOT_EXT);
  Sol sol = make_sol(DT[SAG_ROBOT_DOGGO]);
  g_dg_dump = out; g_dg_dump_max = max; g_dg_dump_n = 0;
  const int cc = world_forward(&w, e, ctrl, &sol, NULL);
  g_dg_dump = NULL;
  if (cost_contacts) *cost_contacts = cc;
  return g_dg_dump_n;
}

/* ------------------------------------------------------------------------ */
/* goal resampling: tasks/go_to_goal.py:59-80 + utils.py:22-70                */
/* ------------------------------------------------------------------------ */
static real kind_keepout(const OEnv* e, int which) { return e->f[SAG_F_KEEPOUT + which]; }

/* returns 1 and writes goal if a valid placement was drawn */
static int resample_goal(OEnv* e, Rng* g) {
  const real* f = e->f;
  double xmin = -GOAL_RECT, ymin = -GOAL_RECT, xmax = GOAL_RECT, ymax = GOAL_RECT;
  int nH = e->i[SAG_I_NH], nV = e->i[SAG_I_NV], nP = e->i[SAG_I_NP], nB = e->i[SAG_I_NB];
  for (int t = 0; t < MAX_RESAMPLE_TRIES; t++) {
    /* utils.draw_placement: constrain by keepout, then two uniforms (:51-70) */
    double cx0 = xmin + GOAL_KEEPOUT, cy0 = ymin + GOAL_KEEPOUT, cx1 = xmax - GOAL_KEEPOUT,
           cy1 = ymax - GOAL_KEEPOUT;
    double gx = rng_uniform(g, cx0, cx1);
    double gy = rng_uniform(g, cy0, cy1);
    if (g->exhausted) return 0;
    int ok = 1;
    /* every other layout entry, dist < keepout_other + GOAL_KEEPOUT rejects (:68-73) */
#define CHECK(px, py, ko)                                                  \
  do {                                                                     \
    double ddx = gx - (double)(px), ddy = gy - (double)(py);               \
    if (sqrt(ddx * ddx + ddy * ddy) < (double)(ko) + GOAL_KEEPOUT) ok = 0; \
  } while (0)
    CHECK(f[SAG_F_ROBOT], f[SAG_F_ROBOT + 1], kind_keepout(e, 0));
    for (int k = 0; k < nH && ok; k++) CHECK(f[SAG_F_HAZARDS + 2 * k], f[SAG_F_HAZARDS + 2 * k + 1], kind_keepout(e, 1));
    for (int k = 0; k < nV && ok; k++) CHECK(f[SAG_F_VASES + 6 * k], f[SAG_F_VASES + 6 * k + 1], kind_keepout(e, 2));
    for (int k = 0; k < nP && ok; k++) CHECK(f[SAG_F_PILLARS + 2 * k], f[SAG_F_PILLARS + 2 * k + 1], kind_keepout(e, 3));
    if (e->i[SAG_I_BOX_KIND] != SAG_BOX_NONE && ok) CHECK(f[SAG_F_BOX], f[SAG_F_BOX + 1], kind_keepout(e, 4));
    (void)nB; /* no task has both a goal body and buttons */
#undef CHECK
    if (ok) {
      e->f[SAG_F_GOAL] = (real)gx;
      e->f[SAG_F_GOAL + 1] = (real)gy;
      return 1;
    }
    /* every rejected draw enlarges the rectangle by 1 % (:76-79, utils.py:118-119) */
    xmin *= 1.01; ymin *= 1.01; xmax *= 1.01; ymax *= 1.01;
  }
  return 0;
}

static double dist2d(double ax, double ay, double bx, double by) {
  double dx = ax - bx, dy = ay - by;
  return sqrt(dx * dx + dy * dy);
}
static double dist3d(double ax, double ay, double az, double bx, double by, double bz) {
  double dx = ax - bx, dy = ay - by, dz = az - bz;
  return sqrt(dx * dx + dy * dy + dz * dz);
}

/* GoToGoal.reset (tasks/go_to_goal.py:50-57): resample, last <- 2-D distance */
static void goal_reset(OEnv* e, Rng* g) {
  if (!resample_goal(e, g)) e->i[SAG_I_FLAGS] |= g->exhausted ? 2 : 1;
  e->f[SAG_F_LAST] = (real)dist2d(e->f[SAG_F_ROBOT], e->f[SAG_F_ROBOT + 1], e->f[SAG_F_GOAL],
                                  e->f[SAG_F_GOAL + 1]);
}

/* PressButtons._sample_goal_button (tasks/press_buttons.py:71-77) */
static void sample_goal_button(OEnv* e, Rng* g) {
  /* legacy rs.choice(4) -> randint(0,4): masked rejection on a 32-bit word, mask 3 */
  int b = (int)(rng_word(g) & 3u);
  if (g->exhausted) { e->i[SAG_I_FLAGS] |= 2; return; }
  e->i[SAG_I_GOAL_BUTTON] = b;
  e->i[SAG_I_BTN_TIMER] = 5; /* BUTTON_TICKING_DELAY */
  e->f[SAG_F_LAST] = (real)dist2d(e->f[SAG_F_ROBOT], e->f[SAG_F_ROBOT + 1],
                                  e->f[SAG_F_BUTTONS + 2 * b], e->f[SAG_F_BUTTONS + 2 * b + 1]);
}

/* ------------------------------------------------------------------------ */
/* CatchGoal.set_mocaps (tasks/catch_goal.py:20-31), before physics           */
/* ------------------------------------------------------------------------ */
static void catch_goal_mocap(OEnv* e, Rng* g, double time) {
  int t = e->i[SAG_I_CATCH_TIMER];
  t = t - 1 > 0 ? t - 1 : 0; /* Timer.tick */
  if (t == 0) {
    e->f[SAG_F_CATCH + 2] = e->f[SAG_F_CATCH + 3];
    e->f[SAG_F_CATCH + 3] = (real)rng_uniform(g, 0.2, 1.0);
    if (g->exhausted) e->i[SAG_I_FLAGS] |= 2;
    t = 10; /* SAMPLE_POINTS */
  }
  e->i[SAG_I_CATCH_TIMER] = t;
  double progress = (10 - t) / 10.0;
  double cur = e->f[SAG_F_CATCH + 2], nxt = e->f[SAG_F_CATCH + 3];
  double radius = progress * (nxt - cur) + cur;
  e->f[SAG_F_GOAL] = (real)(e->f[SAG_F_CATCH] + sin(time) * radius);
  e->f[SAG_F_GOAL + 1] = (real)(e->f[SAG_F_CATCH + 1] + cos(time) * radius);
}

/* ------------------------------------------------------------------------ */
/* per-task reward (tasks/<task>.py compute_reward)                                */
/* ------------------------------------------------------------------------ */
static void task_reward(OEnv* e, int robot, Rng* g, uint32_t btn_mask, OOut* out) {
  real* f = e->f;
  int task = e->i[SAG_I_TASK];
  double rx = f[SAG_F_ROBOT], ry = f[SAG_F_ROBOT + 1];
  out->reward[0] = out->reward[1] = 0;
  out->goal_met = 0;
  switch (task) {
    case SAG_TASK_GO_TO_GOAL:
    case SAG_TASK_GO_TO_GOAL_DAMPING:
    case SAG_TASK_GO_TO_GOAL_MOTOR:
    case SAG_TASK_CATCH_GOAL:
    case SAG_TASK_GO_TO_GOAL_SCARCE:
    case SAG_TASK_UNSUPERVISED: {
      /* tasks/go_to_goal.py:31-45: 3-D distance robot xpos - goal xpos */
      double dist = dist3d(rx, ry, robot == SAG_ROBOT_DOGGO ? (double)f[SAG_F_ROBOT_EXT] : PT_Z, f[SAG_F_GOAL], f[SAG_F_GOAL + 1], GOAL_Z);
      double r = (double)f[SAG_F_LAST] - dist;
      if (task == SAG_TASK_GO_TO_GOAL_SCARCE) /* go_to_goal_scarce.py:26-32: indicator(0<=d<=.45) */
        r *= (0 <= dist && dist <= GOAL_SIZE * 1.5) ? 1.0 : 0.0;
      f[SAG_F_LAST] = (real)dist;
      if (dist <= GOAL_SIZE) {
        out->goal_met = 1;
        goal_reset(e, g); /* update_layout is implicit: the record holds current poses */
        if (task == SAG_TASK_CATCH_GOAL) { /* catch_goal.py:36-40: origin <- new goal */
          f[SAG_F_CATCH] = f[SAG_F_GOAL];
          f[SAG_F_CATCH + 1] = f[SAG_F_GOAL + 1];
        }
        r += 1.0;
      }
      if (task == SAG_TASK_UNSUPERVISED) { /* tasks/unsupervised.py:48-67 */
        double c = cos((double)f[SAG_F_ROBOT + 2]), s = sin((double)f[SAG_F_ROBOT + 2]);
        /* subtree COM in body axes: point (mc/m, 0) along +x; car (0, oy) */
        double bx = PT_MC / PT_MASS, by = 0;
        if (robot == SAG_ROBOT_CAR) { CarK k = car_constants(); bx = k.ox; by = k.oy; }
        double lx = c * bx - s * by, ly = s * bx + c * by;
        double x = rx + lx, y = ry + ly;
        double w = f[SAG_F_ROBOT + 5];
        double u = f[SAG_F_ROBOT + 3] - w * ly, v = f[SAG_F_ROBOT + 4] + w * lx;
        if (robot == SAG_ROBOT_DOGGO) { x = out->comvel[0]; y = out->comvel[1]; u = out->comvel[2]; v = out->comvel[3]; }
        double radius = sqrt(x * x + y * y);
        out->reward[0] = (real)((((-u * y + v * x) / radius) / (1 + fabs(radius - 1.5))) * 1e-1);
        out->reward[1] = (real)r;
      } else {
        out->reward[0] = (real)r;
      }
      break;
    }
    case SAG_TASK_PRESS_BUTTONS:
    case SAG_TASK_PRESS_BUTTONS_SCARCE: {
      /* tasks/press_buttons.py:42-64, press_buttons_scarce.py:22-55 */
      int gb = e->i[SAG_I_GOAL_BUTTON];
      double gd = dist2d(rx, ry, f[SAG_F_BUTTONS + 2 * gb], f[SAG_F_BUTTONS + 2 * gb + 1]);
      double r = task == SAG_TASK_PRESS_BUTTONS ? (double)f[SAG_F_LAST] - gd : 0.0;
      f[SAG_F_LAST] = (real)gd;
      if (btn_mask & (1u << gb)) {
        r += 1.0;
        out->goal_met = 1;
        sample_goal_button(e, g);
        e->i[SAG_I_BTN_STATE] = 0; /* BUTTON_CHANGE */
      }
      /* (scarce: the wrong-button branch tests names that never match - App. C) */
      if (e->i[SAG_I_BTN_STATE] == 0) {
        if (e->i[SAG_I_BTN_TIMER] != 0) e->i[SAG_I_BTN_TIMER] -= 1; /* tick */
        else { e->i[SAG_I_BTN_STATE] = 1; e->i[SAG_I_BTN_TIMER] = 5; }
      }
      out->reward[0] = (real)r;
      break;
    }
    case SAG_TASK_COLLECT: {
      /* tasks/collect.py:24-47 */
      int nB = e->i[SAG_I_NB];
      if (e->i[SAG_I_ACTIVE_MASK] == 0) e->i[SAG_I_ACTIVE_MASK] = (1 << nB) - 1; /* reset */
      double r = 0;
      /* the reference iterates a Python set of names; which touched button is taken
       * first when two are touched in one step is hash-order dependent.  Specified
       * here as lowest index first. */
      for (int b = 0; b < nB; b++)
        if ((e->i[SAG_I_ACTIVE_MASK] >> b & 1) && (btn_mask >> b & 1)) {
          r += 1.0;
          out->goal_met = 1;
          e->i[SAG_I_ACTIVE_MASK] &= ~(1 << b);
          break;
        }
      out->reward[0] = (real)r;
      break;
    }
    case SAG_TASK_PUSH_BOX:
    case SAG_TASK_PUSH_BOX_SCARCE:
    case SAG_TASK_ROLL_ROD:
    case SAG_TASK_DRIBBLE_BALL:
    case SAG_TASK_HAUL_BOX: {
      /* tasks/push_box.py:74-100 (RollRod, DribbleBall inherit it), push_box_scarce.py:24-50,
       * haul_box.py:34-48: 2-D distances */
      double bx = f[SAG_F_BOX], by = f[SAG_F_BOX + 1], gx = f[SAG_F_GOAL], gy = f[SAG_F_GOAL + 1];
      double r = 0;
      if (task != SAG_TASK_HAUL_BOX) {
        double bd = dist2d(rx, ry, bx, by);
        double prog = (double)f[SAG_F_LAST + 1] - bd;
        if (task == SAG_TASK_PUSH_BOX_SCARCE) /* indicator(0 <= d <= GOAL_SIZE * 1.70) */
          prog *= (0 <= bd && bd <= GOAL_SIZE * 1.70) ? 1.0 : 0.0;
        r += prog;
        f[SAG_F_LAST + 1] = (real)bd;
      }
      double bg = dist2d(bx, by, gx, gy);
      r += (double)f[SAG_F_LAST + 2] - bg;
      f[SAG_F_LAST + 2] = (real)bg;
      if (bg <= GOAL_SIZE) {
        out->goal_met = 1;
        goal_reset(e, g); /* GoToGoal.reset: new goal, last goal distance (2-D) */
        /* PushBox.reset (push_box.py:94-100) */
        f[SAG_F_LAST + 2] = (real)dist2d(f[SAG_F_GOAL], f[SAG_F_GOAL + 1], bx, by);
        f[SAG_F_LAST + 1] = (real)dist2d(rx, ry, bx, by);
        r += 1.0;
      }
      out->reward[0] = (real)r;
      break;
    }
    default:
      break;
  }
}

/* lidar group of button b after the task's _update_goal_button / collect logic
 * (press_buttons.py:78-91, collect.py:35-36,43-46) */
static int button_group(const OEnv* e, int b) {
  int task = e->i[SAG_I_TASK];
  if (task == SAG_TASK_COLLECT) return (e->i[SAG_I_ACTIVE_MASK] >> b & 1) ? 2 : 0;
  if (e->i[SAG_I_BTN_STATE] == 0) return 0;
  return b == e->i[SAG_I_GOAL_BUTTON] ? 2 : 3;
}

static int has_goal_body(int task) {
  return !(task == SAG_TASK_PRESS_BUTTONS || task == SAG_TASK_PRESS_BUTTONS_SCARCE ||
           task == SAG_TASK_COLLECT);
}

/* ------------------------------------------------------------------------ */
/* observation: safe_adaptation_gym.py:120-139,225-237 (Point: 48 + 12)       */
/* ------------------------------------------------------------------------ */
static void observe_doggo(const OEnv* e, const real qacc[3], const real* touch, real* obs);
static void observe(const OEnv* e, int robot, const real qacc[3], real* obs) {
  const real* f = e->f;
  double lid[48];
  for (int k = 0; k < 48; k++) lid[k] = 0;
  double rx = f[SAG_F_ROBOT], ry = f[SAG_F_ROBOT + 1], yaw = f[SAG_F_ROBOT + 2];
  double c = cos(yaw), s = sin(yaw);
  int nH = e->i[SAG_I_NH], nV = e->i[SAG_I_NV], nP = e->i[SAG_I_NP], nB = e->i[SAG_I_NB];
  for (int k = 0; k < nH; k++) lidar_planar(rx, ry, c, s, f[SAG_F_HAZARDS + 2 * k], f[SAG_F_HAZARDS + 2 * k + 1], lid);
  for (int k = 0; k < nV; k++) lidar_planar(rx, ry, c, s, f[SAG_F_VASES + 6 * k], f[SAG_F_VASES + 6 * k + 1], lid);
  for (int k = 0; k < nP; k++) lidar_planar(rx, ry, c, s, f[SAG_F_PILLARS + 2 * k], f[SAG_F_PILLARS + 2 * k + 1], lid);
  if (e->i[SAG_I_BOX_KIND] != SAG_BOX_NONE) lidar_planar(rx, ry, c, s, f[SAG_F_BOX], f[SAG_F_BOX + 1], lid + 16);
  for (int b = 0; b < nB; b++) {
    int g = button_group(e, b);
    if (g == 3) lidar_planar(rx, ry, c, s, f[SAG_F_BUTTONS + 2 * b], f[SAG_F_BUTTONS + 2 * b + 1], lid + 16);
    if (g == 2) lidar_planar(rx, ry, c, s, f[SAG_F_BUTTONS + 2 * b], f[SAG_F_BUTTONS + 2 * b + 1], lid + 32);
  }
  if (has_goal_body(e->i[SAG_I_TASK])) lidar_planar(rx, ry, c, s, f[SAG_F_GOAL], f[SAG_F_GOAL + 1], lid + 32);
  for (int k = 0; k < 48; k++) obs[k] = (real)lid[k];
  /* sensors (point.xml:23-28) in the site frame = body frame, R = Rz(yaw):
   * accelerometer = R^T (a_origin - g), velocimeter = R^T v, gyro = (0,0,w),
   * magnetometer = R^T (0,-0.5,0). */
  double ax = qacc[0], ay = qacc[1];
  obs[48] = (real)(c * ax + s * ay);
  obs[49] = (real)(-s * ax + c * ay);
  obs[50] = (real)GRAVITY;
  double vx = f[SAG_F_ROBOT + 3], vy = f[SAG_F_ROBOT + 4];
  obs[51] = (real)(c * vx + s * vy);
  obs[52] = (real)(-s * vx + c * vy);
  obs[53] = 0;
  obs[54] = 0; obs[55] = 0; obs[56] = f[SAG_F_ROBOT + 5];
  obs[57] = (real)(-0.5 * s);
  obs[58] = (real)(-0.5 * c);
  obs[59] = 0;
  if (robot == SAG_ROBOT_CAR) {
    /* car.xml:37-38 via safe_adaptation_gym.py:228-236: ballangvel_rear (joint rate in the ball's
     * own frame), then the 3x3 of ballquat_rear, row-major (utils.py:110-116) */
    const real* q = f + SAG_F_ROBOT_EXT + 5;
    double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    double R[9] = {qw * qw + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                   2 * (qx * qy + qw * qz), qw * qw - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qw * qx),
                   2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), qw * qw - qx * qx - qy * qy + qz * qz};
    double W[3] = {f[SAG_F_ROBOT_EXT + 2], f[SAG_F_ROBOT_EXT + 3], f[SAG_F_ROBOT_EXT + 4]};
    for (int k = 0; k < 3; k++) obs[60 + k] = (real)(R[k] * W[0] + R[3 + k] * W[1] + R[6 + k] * W[2]);
    for (int k = 0; k < 9; k++) obs[63 + k] = (real)R[k];
  }
}

#include "sag_oracle_render.inc"

/* hazard part of World.compute_cost (world.py:147-153), fp64 on the stored poses */
static int hazard_cost(const OEnv* e, double* margin) {
  const real* f = e->f;
  int n = 0;
  double hs = f[SAG_F_HAZARD_SIZE];
  for (int k = 0; k < e->i[SAG_I_NH]; k++) {
    double d = dist2d(f[SAG_F_ROBOT], f[SAG_F_ROBOT + 1], f[SAG_F_HAZARDS + 2 * k], f[SAG_F_HAZARDS + 2 * k + 1]);
    if (d <= hs) n++;
    if (fabs(d - hs) < *margin) *margin = fabs(d - hs);
  }
  return n;
}

static int state_bad(const OEnv* e) {
  for (int k = 0; k < 6; k++) {
    real v = e->f[SAG_F_ROBOT + k];
    if (!(v == v) || R_FABS(v) > (real)1e10) return 1;
  }
  return 0;
}
static int state_bad_r(const OEnv* e, int robot) {
  if (state_bad(e)) return 1;
  if (robot == SAG_ROBOT_DOGGO)
    for (int k = 0; k < 35; k++) {
      real v = e->f[SAG_F_ROBOT_EXT + k];
      if (!(v == v) || R_FABS(v) > (real)1e10) return 1;
    }
  return 0;
}

/* Doggo observation: lidar with the tilted base (safe_adaptation_gym.py:197-216: e = (d @ R)[:2]
 * with d = [p_xy, 0] - robot_xpos), then accelerometer, velocimeter, gyro, magnetometer, 8 touch,
 * 12 joint rates, 12 x (sin, cos) of the joint angles (:225-237, doggo.xml:83-126) = 104 */
static void dg_lidar(const double pos[3], const double R[9], double px, double py, double* lid) {
  const double dx = px - pos[0], dy = py - pos[1], dz = -pos[2];
  lidar_accumulate(R[0] * dx + R[3] * dy + R[6] * dz, R[1] * dx + R[4] * dy + R[7] * dz, lid);
}
static void observe_doggo(const OEnv* e, const real qacc[3], const real* touch, real* obs) {
  const real* f = e->f;
  Doggo D;
  dg_load(&D, f + SAG_F_ROBOT, f + SAG_F_ROBOT_EXT);
  double R[9], lid[48];
  dg_quat2mat(D.quat, R);
  for (int k = 0; k < 48; k++) lid[k] = 0;
  int nH = e->i[SAG_I_NH], nV = e->i[SAG_I_NV], nP = e->i[SAG_I_NP], nB = e->i[SAG_I_NB];
  for (int k = 0; k < nH; k++) dg_lidar(D.pos, R, f[SAG_F_HAZARDS + 2 * k], f[SAG_F_HAZARDS + 2 * k + 1], lid);
  for (int k = 0; k < nV; k++) dg_lidar(D.pos, R, f[SAG_F_VASES + 6 * k], f[SAG_F_VASES + 6 * k + 1], lid);
  for (int k = 0; k < nP; k++) dg_lidar(D.pos, R, f[SAG_F_PILLARS + 2 * k], f[SAG_F_PILLARS + 2 * k + 1], lid);
  if (e->i[SAG_I_BOX_KIND] != SAG_BOX_NONE) dg_lidar(D.pos, R, f[SAG_F_BOX], f[SAG_F_BOX + 1], lid + 16);
  for (int b = 0; b < nB; b++) {
    int g = button_group(e, b);
    if (g == 3) dg_lidar(D.pos, R, f[SAG_F_BUTTONS + 2 * b], f[SAG_F_BUTTONS + 2 * b + 1], lid + 16);
    if (g == 2) dg_lidar(D.pos, R, f[SAG_F_BUTTONS + 2 * b], f[SAG_F_BUTTONS + 2 * b + 1], lid + 32);
  }
  if (has_goal_body(e->i[SAG_I_TASK])) dg_lidar(D.pos, R, f[SAG_F_GOAL], f[SAG_F_GOAL + 1], lid + 32);
  for (int k = 0; k < 48; k++) obs[k] = (real)lid[k];
  const double a[3] = {qacc[0], qacc[1], (double)qacc[2] + GRAVITY}, m[3] = {0, -0.5, 0};
  for (int k = 0; k < 3; k++) {
    obs[48 + k] = (real)(R[k] * a[0] + R[3 + k] * a[1] + R[6 + k] * a[2]);
    obs[51 + k] = (real)(R[k] * D.vlin[0] + R[3 + k] * D.vlin[1] + R[6 + k] * D.vlin[2]);
    obs[54 + k] = (real)D.wloc[k];
    obs[57 + k] = (real)(R[k] * m[0] + R[3 + k] * m[1] + R[6 + k] * m[2]);
  }
  for (int k = 0; k < 8; k++) obs[60 + k] = touch ? touch[k] : 0;
  for (int k = 0; k < 12; k++) {
    const int j = DG_ACT_JOINT[k];
    obs[68 + k] = (real)D.qd[j];
    obs[80 + 2 * k] = (real)sin(D.q[j]);
    obs[81 + 2 * k] = (real)cos(D.q[j]);
  }
}

/* ------------------------------------------------------------------------ */
/* SafeAdaptationGym.step (safe_adaptation_gym.py:56-83)                      */
/* ------------------------------------------------------------------------ */
/* action[nu]; noise[nu] standard normals or NULL (-> Philox stream 1);
 * tape or NULL (-> Philox stream 0); nstep<0 -> table; ext_contacts >= 0
 * overrides the geometric contact results (golden episodes script contacts):
 * ext_cost_contacts = robot/obstacle contact count, ext_btn_mask = touched buttons. */
void sago_step(OEnv* e, int robot, const float* action, const float* noise, const uint32_t* tape,
               int tape_len, uint32_t key0, uint32_t key1, int nstep, int ext_cost_contacts,
               uint32_t ext_btn_mask, OOut* out) {
  int nu = NU[robot];
  double h = DT[robot];
  if (nstep < 0) nstep = NSTEP[robot];
  Rng g = {tape, tape_len, 0, 0, key0, key1, (uint32_t)e->i[SAG_I_ENV_ID], (uint32_t)e->i[SAG_I_STEP],
           (uint32_t)e->i[SAG_I_EPISODE] << 2};
  float nz[SAG_MAX_NU];
  if (!noise) { sago_noise_ep(key0, key1, g.env_id, g.step, (uint32_t)e->i[SAG_I_EPISODE], nu, nz); noise = nz; }
  real ctrl[SAG_MAX_NU];
  for (int j = 0; j < nu; j++) {
    /* action + action_noise * normal, clipped to ctrlrange * scale (:63-67, mujoco_bridge.py:164-166) */
    real a = (real)action[j] + e->f[SAG_F_ACTION_NOISE] * (real)noise[j];
    real lim = e->f[SAG_F_CTRL_SCALE + j];
    ctrl[j] = clampr(a, -lim, lim);
  }
  memset(out, 0, sizeof(*out));
  out->cost_margin = 1e30;
  /* World.set_mocaps (:71): only CatchGoal acts; uses time BEFORE the step */
  double time = (double)e->i[SAG_I_STEP] * NSTEP[robot] * h;
  if (e->i[SAG_I_TASK] == SAG_TASK_CATCH_GOAL) catch_goal_mocap(e, &g, time);
  DgWarm dgws; dgws.n = 0; dgws.evals = 0;
  g_dgws = &dgws;   /* doggo: constraint forces carried from one forward evaluation of this env-step to the next */
  CarWarm carws; carws.have = 0;
  g_carws = &carws; /* car: the same for its floor friction */
  substeps_r(e, ctrl, nstep, h, robot);
  e->i[SAG_I_STEP] += 1;
  if (state_bad_r(e, robot)) { /* PhysicsError branch (:73-75) */
    real z[3] = {0, 0, 0};
    if (robot == SAG_ROBOT_DOGGO) memset(out->obs, 0, sizeof(out->obs));
    else observe(e, robot, z, out->obs);
    out->reward[0] = -10; out->done = 1; out->cost = 0;
    out->tape_used = g.pos;
    g_dgws = NULL; g_carws = NULL;
    return;
  }
  /* mj_forward at the final state (:76): contacts + qacc */
  World w;
  Doggo dg;
  world_from_env_r(e, &w, robot);
  if (robot == SAG_ROBOT_DOGGO) { w.dg = &dg; dg_load(&dg, e->f + SAG_F_ROBOT, e->f + SAG_F_ROBOT_EXT); }
  Sol sol = make_sol(h);
  uint32_t mask = 0;
  carws.have = 0;   /* the forward evaluation behind the observation solves the car's friction cold: the accelerometer stays a function of the state */
  dgws.n = 0; dgws.evals = 0;   /* ... and the Doggo's rows (DG_PGS_ITERS0 sweeps from zero): accelerometer and touch columns of step() == sago_observe() of the same state */
  int cc = world_forward(&w, e, ctrl, &sol, &mask);
  g_dgws = NULL; g_carws = NULL;
  if (ext_cost_contacts >= 0) { cc = ext_cost_contacts; mask = ext_btn_mask; }
  out->qacc[0] = w.robot.ax; out->qacc[1] = w.robot.ay; out->qacc[2] = w.robot.aw;
  if (robot == SAG_ROBOT_DOGGO) {
    double cv[4];
    dg_com_vel(&dg, cv);
    for (int k = 0; k < 3; k++) out->qacc[k] = (real)dg.qacc[k];
    for (int k = 0; k < 8; k++) out->touch[k] = (real)dg.touch[k];
    for (int k = 0; k < 4; k++) out->comvel[k] = (real)cv[k];
  }
  out->btn_contact_mask = mask;
  task_reward(e, robot, &g, mask, out);         /* :77 */
  double margin = 1e30;
  int cost = cc + hazard_cost(e, &margin);      /* :78, world.py:144-155 */
  out->cost = cost > 0;
  out->cost_margin = (real)margin;
  if (robot == SAG_ROBOT_DOGGO) observe_doggo(e, out->qacc, out->touch, out->obs);
  else observe(e, robot, out->qacc, out->obs);  /* :80 */
  out->tape_used = g.pos;
}

/* observation after reset (safe_adaptation_gym.py:104,107): forward + observe, ctrl = 0 */
void sago_observe(OEnv* e, int robot, OOut* out) {
  memset(out, 0, sizeof(*out));
  World w;
  Doggo dg;
  world_from_env_r(e, &w, robot);
  if (robot == SAG_ROBOT_DOGGO) { w.dg = &dg; dg_load(&dg, e->f + SAG_F_ROBOT, e->f + SAG_F_ROBOT_EXT); }
  Sol sol = make_sol(DT[robot]);
  real ctrl[SAG_MAX_NU] = {0};
  CarWarm carws; carws.have = 0;
  g_carws = &carws;   /* one cold solve; the re-solve after contacts starts from it, as in a step */
  world_forward(&w, e, ctrl, &sol, &out->btn_contact_mask);
  g_carws = NULL;
  out->qacc[0] = w.robot.ax; out->qacc[1] = w.robot.ay; out->qacc[2] = w.robot.aw;
  if (robot == SAG_ROBOT_DOGGO) {
    for (int k = 0; k < 3; k++) out->qacc[k] = (real)dg.qacc[k];
    for (int k = 0; k < 8; k++) out->touch[k] = (real)dg.touch[k];
    observe_doggo(e, out->qacc, out->touch, out->obs);
    return;
  }
  observe(e, robot, out->qacc, out->obs);
}

/* task.reset as run by World.reset right after rebuild (world.py:167-170):
 * installs the `last` distances from the positions in the record.  (The goal
 * resample / button choice of reset is drawn host-side, App. B.5.) */
void sago_task_reset(OEnv* e) {
  real* f = e->f;
  int task = e->i[SAG_I_TASK];
  if (has_goal_body(task))
    f[SAG_F_LAST] = (real)dist2d(f[SAG_F_ROBOT], f[SAG_F_ROBOT + 1], f[SAG_F_GOAL], f[SAG_F_GOAL + 1]);
  else if (task != SAG_TASK_COLLECT) {
    int b = e->i[SAG_I_GOAL_BUTTON];
    f[SAG_F_LAST] = (real)dist2d(f[SAG_F_ROBOT], f[SAG_F_ROBOT + 1], f[SAG_F_BUTTONS + 2 * b], f[SAG_F_BUTTONS + 2 * b + 1]);
  }
  if (e->i[SAG_I_BOX_KIND] != SAG_BOX_NONE) { /* PushBox.reset (push_box.py:94-100) */
    f[SAG_F_LAST + 2] = (real)dist2d(f[SAG_F_GOAL], f[SAG_F_GOAL + 1], f[SAG_F_BOX], f[SAG_F_BOX + 1]);
    f[SAG_F_LAST + 1] = (real)dist2d(f[SAG_F_ROBOT], f[SAG_F_ROBOT + 1], f[SAG_F_BOX], f[SAG_F_BOX + 1]);
  }
}

/* ------------------------------------------------------------------------ */
/* batch driver (cpu_baseline leg of bench.py; OpenMP over envs)              */
/* ------------------------------------------------------------------------ */
/* envs: n contiguous OEnv; actions [n][nu]; outputs per env.  Philox mode. */
void sago_step_batch(OEnv* envs, int n, int robot, const float* actions, uint32_t key0,
                     uint32_t key1, float* obs, float* reward, uint8_t* cost, uint8_t* done,
                     uint8_t* goal_met, int nthreads) {
  int nu = NU[robot], od = OBS_DIM[robot];
  dg_init();   /* shared model tables: before the threads start (see dg_init) */
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int i = 0; i < n; i++) {
    OOut o;
    sago_step(&envs[i], robot, actions + (size_t)i * nu, NULL, NULL, 0, key0, key1, -1, -1, 0, &o);
    if (obs) for (int k = 0; k < od; k++) obs[(size_t)i * od + k] = (float)o.obs[k];
    if (reward) { reward[2 * i] = (float)o.reward[0]; reward[2 * i + 1] = (float)o.reward[1]; }
    if (cost) cost[i] = (uint8_t)o.cost;
    if (done) done[i] = (uint8_t)o.done;
    if (goal_met) goal_met[i] = (uint8_t)o.goal_met;
  }
}

/* lockstep parity driver: explicit noise / tape per env (either may be NULL), all
 * outputs incl. diagnostics.  Serial unless sago_set_threads(n > 1) (envs are independent). */
static int g_full_threads = 1;
void sago_set_threads(int n) { g_full_threads = n < 1 ? 1 : n; }
void sago_step_batch_full(OEnv* envs, int n, int robot, const float* actions, const float* noise,
                          const uint32_t* tape, int tape_len, uint32_t key0, uint32_t key1,
                          int nstep, float* obs, float* reward, uint8_t* cost, uint8_t* done,
                          uint8_t* goal_met, int32_t* tape_used, double* cost_margin) {
  int nu = NU[robot], od = OBS_DIM[robot];
  dg_init();
#pragma omp parallel for num_threads(g_full_threads) schedule(dynamic, 4)
  for (int i = 0; i < n; i++) {
    OOut o;
    sago_step(&envs[i], robot, actions + (size_t)i * nu, noise ? noise + (size_t)i * nu : NULL,
              tape ? tape + (size_t)i * tape_len : NULL, tape_len, key0, key1, nstep, -1, 0, &o);
    for (int k = 0; k < od; k++) obs[(size_t)i * od + k] = (float)o.obs[k];
    reward[2 * i] = (float)o.reward[0]; reward[2 * i + 1] = (float)o.reward[1];
    cost[i] = (uint8_t)o.cost; done[i] = (uint8_t)o.done; goal_met[i] = (uint8_t)o.goal_met;
    tape_used[i] = o.tape_used;
    if (cost_margin) cost_margin[i] = (double)o.cost_margin;
  }
}

void sago_observe_batch(OEnv* envs, int n, int robot, float* obs) {
  int od = OBS_DIM[robot];
  for (int i = 0; i < n; i++) {
    OOut o;
    sago_observe(&envs[i], robot, &o);
    for (int k = 0; k < od; k++) obs[(size_t)i * od + k] = (float)o.obs[k];
  }
}

void sago_robot_info(int robot, int out[5], double* dt) {
  static const int NQ[3] = {3, 13, 20}, NV[3] = {3, 11, 19};
  out[0] = NU[robot]; out[1] = OBS_DIM[robot]; out[2] = NSTEP[robot]; out[3] = NQ[robot]; out[4] = NV[robot];
  *dt = DT[robot];
}
