// Doggo physics, wave-cooperative form: 32 lanes per env (two envs per wavefront), every working
// array in LDS, the dense linear algebra spread over the lanes.
//
// Specification: oracle/sag_oracle_doggo.inc / DESIGN.md §4 "Doggo" (same row order, same warm-started PGS).  A
// lane-per-env kernel would walk every dependent chain (Cholesky, 40+ triangular solves, row construction) through
// private memory at ~100 clocks per step (round 1: 17 ms per step for 4096 envs).  Here lane i owns dof i (row i of
// the mass matrix, component i of every generalised vector), lane b body b, lane s floor point s, lane g collision geom g:
//   kinematics / inertias / RNEA   each lane walks its own root-to-body chain (<= 3 joints)
//   mass matrix                    lane i computes row i from the composite inertias
//   Cholesky                       right-looking, 19 rank-1 steps, lane i updates row i
//   M^-1                           19 right-hand sides at once (lane i: component i of each)
//   constraint rows                the contact's owner lane builds J (sparse) and W = M^-1 J^T;
//                                  all contacts of a phase in parallel
//   PGS                            one row per lane on the Delassus matrix (fp32), warm-started from the previous
//                                  forward evaluation of the env-step, DG_PGS_ITERS sweeps (DG_PGS_ITERS0 cold)
//   planar world                   body k on lane k for statics / friction / integration
// The generic step kernel then runs with `doggo_post` set: it skips the physics and does reward,
// cost, lidar and the observation from the stored state and the per-env result block.
#pragma once

namespace sag {

constexpr int DC_ROWS = DG_MAXROWS;              // max constraint rows per env
constexpr int DC_PGS_LANES = 32;                // fast PGS path: one constraint row per lane
#ifndef SAG_DC_FAST_ROWS
#define SAG_DC_FAST_ROWS DC_PGS_LANES           // (test builds lower it so that the 64-lane path runs on ordinary states)
#endif
struct DcContact {                              // a contact as its owner lane found it; its three rows are built by all lanes
  double dir[3][3], c[3], depth;                // normal + two tangents, point, penetration
  float bcoef, kcoef, mu;
  uint32_t key;                                 // identity of the contact's normal row (+1, +2: its tangents): warm start
  short body, other;                            // robot body of the geom, planar body (-1: static / floor)
  short touch, dbl;                             // touch slot (+ 8: the sensor reads half of the row) or -1; merged row (half the regulariser)
};
struct DcEnv {                                  // one env's working set in LDS
  double pos[3], quat[4], q[DG_NJ], vlin[3], wloc[3], qd[DG_NJ];
  // Two regions are reused within a substep - 18.5 KB per env instead of 25.6, so that FOUR wavefronts (one per SIMD)
  // fit a CU's 160 KB of LDS instead of three:
  union {
    struct {                                    // kinematics .. constraint Jacobians (dc_jac is the last reader)
      double R[DG_NB][9], p[DG_NB][3];
      double S[DG_NV][6];                       // [a; l]
      double Ib[DG_NB][10];                     // m, mc[3], I[6]
      union {
        struct {
          double Ic[DG_NB][10];
          double F[DG_NV][6];                   // CRBA: Ic[body(j)] S_j ; RNEA: scratch
          double fb[DG_NB][6];
        };
        DcContact ct[DC_ROWS / 3];              // (collision phase: Ic, F, fb are dead)
      };
    };
    float A[DC_PGS_LANES * DC_PGS_LANES];       // Delassus matrix J M^-1 J^T of the fast PGS path (symmetric, no padding:
  };                                            // lane u reads its row as the column A[r][u] - consecutive words)
  union {
    struct {
      double M[DG_NV][DG_NV + 1];               // mass matrix -> Cholesky factor L (lower) -> L and L^-1 -> M^-1
      double bias[DG_NV], tau[DG_NV];
    };
    float rW[DC_ROWS][DG_NV];                   // W = J M^-1 (fp32 like J): written once M^-1 sits in registers
  };
  double qacc[DG_NV], qdv[DG_NV], col[DG_NV];
  double touch[8];
  double com[2], mom[2];                        // centre of mass, momentum / mass (outputs; from Ib and M rows 0, 1)
  float rJ[DC_ROWS][DG_NV];
  double rA[DC_ROWS], rAref[DC_ROWS], rF[DC_ROWS];
  float rImp[DC_ROWS], rMu[DC_ROWS];             // impedance (0.9 .. 0.95), friction coefficient (two wavefronts' envs must fit 40 KB: four per CU)
  double rReg[DC_ROWS], rInv[DC_ROWS];           // A (1 - d) / d and 1 / (A + reg): no division inside the PGS sweeps
  float rOu[DC_ROWS][3], rOd[DC_ROWS][2], rOrx[DC_ROWS], rOry[DC_ROWS];
  short rParent[DC_ROWS], rOther[DC_ROWS], rTouch[DC_ROWS];
  // warm start (oracle DgWarm): identity of every row (0 = none; key scheme in oracle/sag_oracle_doggo.inc), and the
  // rows + forces of the previous forward evaluation of this env-step
  uint32_t rKey[DC_ROWS], wsKey[DC_ROWS];
  float wsF[DC_ROWS];
  int wsN;
  uint32_t act;                                 // planar bodies that take part in this forward evaluation (oracle world_forward, "sleeping bodies")
  float wfb[NBODY][9], wminv[NBODY][6];         // planar free bodies
  int nrows, flag;
};
static_assert(sizeof(float) * DC_PGS_LANES * DC_PGS_LANES <= sizeof(double) * (DG_NB * 38 + DG_NV * 12), "A fits the kinematics region");

// Both envs of the wavefront.  File scope, so that the (non-inlined) phase functions address it as LDS:
// through a `DcEnv&` parameter every access would be a flat instruction.
// DC_EPW envs per wavefront (32 lanes each): 2 = 37 KB of LDS per wavefront, four wavefronts per CU.
// 1 (a 32-thread workgroup per env, six per CU) was measured slower: 10.4 vs 8.4 ms at 4096 envs.
#ifndef DC_EPW
#define DC_EPW 2
#endif
__shared__ DcEnv g_dc_env[DC_EPW];
#define DC_ENV DcEnv& E = g_dc_env[hf]
// The model tables in LDS, one copy per workgroup.  From constant memory every lookup with a lane-dependent index
// (dof -> body, body -> ancestors, sphere -> body ...) is a global load the lone wavefront of a SIMD waits out:
// ten of them in a row made the bias sum of RNEA cost as much as the whole PGS.
__shared__ DgPhys g_dc_phys;
__shared__ int g_dc_first_dof[DG_NB], g_dc_ndof[DG_NB];
__device__ inline void dc_load_model() {   // all threads of the workgroup; the caller's barrier publishes it
  static_assert(sizeof(DgPhys) % 4 == 0, "word copy");
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&g_dg);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&g_dc_phys);
  for (int w = threadIdx.x; w < (int)(sizeof(DgPhys) / 4); w += blockDim.x) dst[w] = src[w];
  if (threadIdx.x < DG_NB) {
    int first = 0, n = 0;
    for (int d = DG_NV - 1; d >= 0; d--)
      if (g_dg.dof_body[d] == (int)threadIdx.x) { first = d; n++; }
    g_dc_first_dof[threadIdx.x] = first; g_dc_ndof[threadIdx.x] = n;
  }
}
// section profile of k_doggo_physics (tools/cycles_doggo.py, -DSAG_CYCLES): lane 0's clock per section -> g_cyc[1][.]
enum { DCY_LOAD = 0, DCY_KIN, DCY_BODIES, DCY_CRBA_RNEA, DCY_CHOL, DCY_INV, DCY_ROWS_SELF, DCY_ROWS_WORLD, DCY_FINISH, DCY_PGS,
       DCY_AFTER, DCY_PLANAR, DCY_STORE, DCY_CRBA_F, DCY_CRBA_ROWS, DCY_N };
#ifdef SAG_CYCLES
static_assert(DCY_N <= CY_N, "sections fit g_cyc");
__shared__ unsigned long long g_dc_cyc[DCY_N], g_dc_cyc_t;
#define DCC(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                    g_dc_cyc[k] += t_ - g_dc_cyc_t; g_dc_cyc_t = t_; } } while (0)
#else
#define DCC(k) do {} while (0)
#endif

// paths of the kinematic tree (static): bodies from the root's child down to b
struct DcPath { int n, c0, c1, c2; __device__ int at(int s) const { return s == 0 ? c0 : (s == 1 ? c1 : c2); } };
__device__ inline int dc_path(int b, DcPath& path) {  // bodies from the root's child down to b (depth <= 3), in registers
  const int a1 = b, a2 = b > 0 ? g_dc_phys.parent[a1] : 0, a3 = a2 > 0 ? g_dc_phys.parent[a2] : 0;
  if (b <= 0) { path.n = 0; path.c0 = path.c1 = path.c2 = 0; }
  else if (a2 <= 0) { path.n = 1; path.c0 = a1; path.c1 = path.c2 = 0; }
  else if (a3 <= 0) { path.n = 2; path.c0 = a2; path.c1 = a1; path.c2 = 0; }
  else { path.n = 3; path.c0 = a3; path.c1 = a2; path.c2 = a1; }
  return path.n;
}
__device__ inline int dc_first_dof(int b) { return g_dc_first_dof[b]; }  // first dof of body b
__device__ inline int dc_ndof(int b) { return g_dc_ndof[b]; }

// rotation matrix (3x3, registers) times axis-angle rotation
__device__ inline void dc_rot_apply(double* R, const double* axis, double ang) {
  double Rj[9];
  dg_axis_rot(axis, ang, Rj);
  dg_matmul(R, Rj, R);
}

// phase 1: frames, motion vectors, inertias about O = base origin, sphere centres.  `u` = lane in the half.
__device__ __attribute__((noinline)) void dc_kinematics(int hf, int u) {
  DC_ENV;
  const DgPhys& M = g_dc_phys;
  // the 13 joint rotations once (lane j: two fp64 sin / cos, ~300 instructions), parked in LDS over F / fb, which are
  // written only after this phase; every chain walk below then multiplies ready matrices - before, each lane
  // re-derived up to three of them in each of its two roles
  static_assert(sizeof(double) * DG_NJ * 9 <= sizeof(double) * (DG_NV * 6 + DG_NB * 6), "joint rotations fit into F + fb");
  double* RJ = &E.F[0][0];
  if (u < DG_NJ) {
    double Rj[9];
    dg_axis_rot(M.axis[u], E.q[u], Rj);
#pragma unroll
    for (int k = 0; k < 9; k++) RJ[u * 9 + k] = Rj[k];
  }
  __syncthreads();
  double R0[9];
  dg_quat2mat(E.quat, R0);
  // dof role
  if (u < DG_NV) {
    double a[3] = {0, 0, 0}, l[3] = {0, 0, 0};
    if (u < 3) l[u] = 1;
    else if (u < 6) { a[0] = R0[u - 3]; a[1] = R0[3 + u - 3]; a[2] = R0[6 + u - 3]; }
    else {
      const int b = M.dof_body[u];
      DcPath path;
      const int n = dc_path(b, path);
      double R[9], p[3] = {E.pos[0], E.pos[1], E.pos[2]};
      for (int k = 0; k < 9; k++) R[k] = R0[k];
      for (int s = 0; s < n; s++) {
        const int c = path.at(s);
        double off[3];
        dg_matvec(R, M.bpos[c], off);
        for (int k = 0; k < 3; k++) p[k] += off[k];
        const int d0 = dc_first_dof(c), nd = dc_ndof(c);
        for (int d = d0; d < d0 + nd && d < u; d++) dg_matmul(R, RJ + (d - 6) * 9, R);
      }
      dg_matvec(R, M.axis[u - 6], a);
      const double r[3] = {p[0] - E.pos[0], p[1] - E.pos[1], p[2] - E.pos[2]};
      dg_cross(r, a, l);
    }
    for (int k = 0; k < 3; k++) { E.S[u][k] = a[k]; E.S[u][3 + k] = l[k]; }
  }
  // body role
  if (u < DG_NB) {
    DcPath path;
    const int n = dc_path(u, path);
    double R[9], p[3] = {E.pos[0], E.pos[1], E.pos[2]};
    for (int k = 0; k < 9; k++) R[k] = R0[k];
    for (int s = 0; s < n; s++) {
      const int c = path.at(s);
      double off[3];
      dg_matvec(R, M.bpos[c], off);
      for (int k = 0; k < 3; k++) p[k] += off[k];
      const int d0 = dc_first_dof(c), nd = dc_ndof(c);
      for (int d = d0; d < d0 + nd; d++) dg_matmul(R, RJ + (d - 6) * 9, R);
    }
    for (int k = 0; k < 9; k++) E.R[u][k] = R[k];
    for (int k = 0; k < 3; k++) E.p[u][k] = p[k];
    double c[3], T[9], Rt[9], Iw[9];
    dg_matvec(R, M.com[u], c);
    for (int k = 0; k < 3; k++) c[k] += p[k] - E.pos[k];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) Rt[3 * r + cc] = R[3 * cc + r];
    dg_matmul(R, M.I[u], T);
    dg_matmul(T, Rt, Iw);
    const double d2 = dg_dot(c, c), m = M.m[u];
    double* I = E.Ib[u];
    I[0] = m; I[1] = m * c[0]; I[2] = m * c[1]; I[3] = m * c[2];
    I[4] = Iw[0] + m * (d2 - c[0] * c[0]); I[5] = Iw[1] - m * c[0] * c[1]; I[6] = Iw[2] - m * c[0] * c[2];
    I[7] = Iw[4] + m * (d2 - c[1] * c[1]); I[8] = Iw[5] - m * c[1] * c[2]; I[9] = Iw[8] + m * (d2 - c[2] * c[2]);
  }
}
// f = I v for the 10-double inertia layout
__device__ inline void dc_inertia_apply(const double* I, const double* v, double* f) {
  const double* w = v; const double* l = v + 3;
  const double mc[3] = {I[1], I[2], I[3]};
  double t[3];
  dg_cross(w, mc, t);
  for (int k = 0; k < 3; k++) f[3 + k] = I[0] * l[k] + t[k];
  dg_cross(mc, l, t);
  f[0] = I[4] * w[0] + I[5] * w[1] + I[6] * w[2] + t[0];
  f[1] = I[5] * w[0] + I[7] * w[1] + I[8] * w[2] + t[1];
  f[2] = I[6] * w[0] + I[8] * w[1] + I[9] * w[2] + t[2];
}
__device__ inline double dc_dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// motion cross product o = v x s
__device__ inline void dc_mcross(const double* v, const double* s, double* o) {
  double t1[3], t2[3];
  dg_cross(v, s, o);
  dg_cross(v, s + 3, t1);
  dg_cross(v + 3, s, t2);
  for (int k = 0; k < 3; k++) o[3 + k] = t1[k] + t2[k];
}

// phases 3-6 need barriers between them; the caller owns the barriers (one wavefront per workgroup)
__device__ __attribute__((noinline)) void dc_composite(int hf, int u) {
  DC_ENV;
  if (u < DG_NB) {
    double acc[10];
    for (int k = 0; k < 10; k++) acc[k] = 0;
    for (int d = u; d < DG_NB; d++)
      if (g_dc_phys.anc[d] >> u & 1u)
        for (int k = 0; k < 10; k++) acc[k] += E.Ib[d][k];
    for (int k = 0; k < 10; k++) E.Ic[u][k] = acc[k];
  }
}
__device__ __attribute__((noinline)) void dc_crba_f(int hf, int u) {
  DC_ENV;
  if (u < DG_NV) dc_inertia_apply(E.Ic[g_dc_phys.dof_body[u]], E.S[u], E.F[u]);
}
__device__ __attribute__((noinline)) void dc_crba_rows(int hf, int u) {
  DC_ENV;
  if (u < DG_NV) {
    const int bi = g_dc_phys.dof_body[u];
    for (int j = 0; j < DG_NV; j++) {
      const int bj = g_dc_phys.dof_body[j];
      double v = 0;
      if ((g_dc_phys.anc[bj] >> bi & 1u) && (bi != bj || u <= j)) v = dc_dot6(E.S[u], E.F[j]);       // i above (or beside, i <= j)
      else if (g_dc_phys.anc[bi] >> bj & 1u) v = dc_dot6(E.S[j], E.F[u]);                           // j above i
      E.M[u][j] = v;
    }
  }
}
// body velocities / accelerations (qacc = 0) and the body wrenches
__device__ __attribute__((noinline)) void dc_rnea_bodies(int hf, int u) {
  DC_ENV;
  if (u < DG_NB) {
    double v[6], a[6];
    for (int k = 0; k < 3; k++) { v[k] = 0; v[3 + k] = E.vlin[k]; a[k] = 0; }
    for (int k = 0; k < 3; k++)
      for (int c = 0; c < 3; c++) v[c] += E.S[3 + k][c] * E.wloc[k];
    dg_cross(E.vlin, v, a + 3);
    a[5] += DG_GRAV;
    DcPath path;
    const int n = dc_path(u, path);
    for (int s = 0; s < n; s++) {
      const int c = path.at(s), d0 = dc_first_dof(c), nd = dc_ndof(c);
      for (int d = d0; d < d0 + nd; d++) {
        double sd[6];
        dc_mcross(v, E.S[d], sd);
        const double qd = E.qd[d - 6];
        for (int k = 0; k < 6; k++) { a[k] += sd[k] * qd; }
        for (int k = 0; k < 6; k++) { v[k] += E.S[d][k] * qd; }
      }
    }
    double f[6], h[6], t1[3], t2[3], t3[3];
    dc_inertia_apply(E.Ib[u], a, f);
    dc_inertia_apply(E.Ib[u], v, h);
    dg_cross(v, h, t1); dg_cross(v + 3, h + 3, t2); dg_cross(v, h + 3, t3);
    for (int k = 0; k < 3; k++) { f[k] += t1[k] + t2[k]; f[3 + k] += t3[k]; }
    for (int k = 0; k < 6; k++) E.fb[u][k] = f[k];
  }
}
__device__ __attribute__((noinline)) void dc_rnea_bias(int hf, int u) {
  DC_ENV;
  if (u < DG_NV) {
    const int b = g_dc_phys.dof_body[u];
    double F[6] = {0, 0, 0, 0, 0, 0};
    for (int d = b; d < DG_NB; d++)
      if (g_dc_phys.anc[d] >> b & 1u)
        for (int k = 0; k < 6; k++) F[k] += E.fb[d][k];
    E.bias[u] = dc_dot6(E.S[u], F);
  }
}

// Cholesky of E.M in place (lower), right-looking; lane i holds row i in registers (static indices,
// k loops unrolled) and publishes column j to LDS for the rank-1 update.  Barriers inside.
__device__ __attribute__((noinline)) void dc_cholesky(int hf, int u) {
  DC_ENV;
  double row[DG_NV];
  const int ur = u < DG_NV ? u : 0;
#pragma unroll
  for (int k = 0; k < DG_NV; k++) row[k] = E.M[ur][k];
#pragma unroll
  for (int j = 0; j < DG_NV; j++) {   // unrolled: static register indices, the updates stop at the matrix edge
    const double rj = row[j];
    if (u == j) {
      if (!(rj > 0)) E.flag |= 1;
      E.col[0] = sqrt(rj);
    }
    __syncthreads();
    const double d = E.col[0];
    const double lij = u == j ? d : rj / d;     // column j of L, entry of this lane's row
    if (u >= j && u < DG_NV) E.M[u][j] = lij;   // published: E.M[k][j] for k >= j is final
    __syncthreads();
    if (u > j && u < DG_NV) {
#pragma unroll
      for (int k = 0; k < DG_NV; k++)
        if (k > j && k <= u) row[k] -= lij * E.M[k][j];
    }
    __syncthreads();
  }
}
// x <- (L L^T)^-1 b, component i of b / x in lane i's register `x`; E.col is the broadcast slot
__device__ __attribute__((noinline)) double dc_solve1(int hf, int u, double x) {
  DC_ENV;
#pragma unroll 1
  for (int j = 0; j < DG_NV; j++) {   // forward: L y = b
    if (u == j) { x = x / E.M[j][j]; E.col[0] = x; }
    __syncthreads();
    if (u > j && u < DG_NV) x -= E.M[u][j] * E.col[0];
    __syncthreads();
  }
#pragma unroll 1
  for (int j = DG_NV - 1; j >= 0; j--) {   // backward: L^T x = y
    if (u == j) { x = x / E.M[j][j]; E.col[0] = x; }
    __syncthreads();
    if (u < j) x -= E.M[j][u] * E.col[0];
    __syncthreads();
  }
  return x;
}
// M^-1 = L^-T L^-1 from the Cholesky factor in E.M, and qacc0 = M^-1 tau on the way (returned: component u).
// Lane j forward-substitutes COLUMN j of L^-1 on its own (171 multiply-adds on rows of L that every lane reads at the
// same address: LDS broadcasts, no exchange between lanes), parks it beside L in E.M, then lane a forms row a of
// M^-1 = sum_k Linv[k][a] Linv[k][:].  4 barriers; the previous form - 19 right-hand sides through 2 x 19
// substitution steps with two barriers each - was 11 % of the kernel.
__device__ __attribute__((noinline)) double dc_inverse(int hf, int u) {
  DC_ENV;
  if (u < DG_NV) E.col[u] = 1.0 / E.M[u][u];
  __syncthreads();
  double x[DG_NV];   // column u of L^-1 (zero above the diagonal)
  const int uc = u < DG_NV ? u : DG_NV - 1;
#pragma unroll
  for (int i = 0; i < DG_NV; i++) {
    double sacc = 0;
#pragma unroll
    for (int k = 0; k < i; k++) sacc += E.M[i][k] * x[k];
    x[i] = i < uc ? 0.0 : (i == uc ? E.col[i] : -sacc * E.col[i]);
  }
  if (u < DG_NV) {   // Linv[i][u] (i >= u) -> E.M[u][i + 1]: the strict upper part of the 19 x 20 array, L keeps the lower
#pragma unroll
    for (int i = 0; i < DG_NV; i++)
      if (i >= u) E.M[u][i + 1] = x[i];
  }
  __syncthreads();
  double row[DG_NV], q0 = 0;
#pragma unroll
  for (int b = 0; b < DG_NV; b++) {
    double v = 0;
#pragma unroll
    for (int k = b; k < DG_NV; k++) v += x[k] * E.M[b][k + 1];   // Linv[k][b]; (x[k] = 0 for k < u)
    row[b] = v;
    q0 += v * E.tau[b];
  }
  __syncthreads();   // every lane has read the parked columns; L is dead: M^-1 takes the array
  if (u < DG_NV) {
#pragma unroll
    for (int b = 0; b < DG_NV; b++) E.M[u][b] = row[b];
  }
  __syncthreads();
  return q0;
}

}  // namespace sag

namespace sag {

// load the env's robot state into LDS (lane 0 of the half), zero flags
__device__ inline void dc_load_state(int hf, int u, const float* S, size_t N, size_t i) {
  DC_ENV;
  if (u == 0) {
    DgState D;
    dg_load(D, S, N, i);
    for (int k = 0; k < 3; k++) { E.pos[k] = D.pos[k]; E.vlin[k] = D.vlin[k]; E.wloc[k] = D.wloc[k]; }
    for (int k = 0; k < 4; k++) E.quat[k] = D.quat[k];
    for (int j = 0; j < DG_NJ; j++) { E.q[j] = D.q[j]; E.qd[j] = D.qd[j]; }
    E.flag = 0; E.nrows = 0;
  }
}

// smooth dynamics: everything up to qacc0 = M^-1 (tau - bias).  ctrl12: LDS/global pointer or nullptr (zero)
__device__ inline void dc_smooth(int hf, int u, const float* ctrl12) {
  DC_ENV;
  dc_kinematics(hf, u);
  __syncthreads();
  DCC(DCY_KIN);
  dc_composite(hf, u);
  dc_rnea_bodies(hf, u);
  __syncthreads();
  DCC(DCY_BODIES);
  dc_crba_f(hf, u);
  dc_rnea_bias(hf, u);
  __syncthreads();
  DCC(DCY_CRBA_F);
  dc_crba_rows(hf, u);
  DCC(DCY_CRBA_ROWS);
  if (u < DG_NV) {
    double t = 0;
    if (u >= 6) {
      const int j = u - 6;
      t = -DG_STIFF * (E.q[j] - g_dc_phys.springref[j]);
      if (ctrl12)
        for (int k = 0; k < 12; k++)
          if (g_dc_phys.act_joint[k] == j) t += DG_GEAR * (double)ctrl12[k];
    }
    E.tau[u] = t - E.bias[u];
    E.qdv[u] = u < 3 ? E.vlin[u] : (u < 6 ? E.wloc[u - 3] : E.qd[u - 6]);
  }
  __syncthreads();
  DCC(DCY_CRBA_RNEA);
}

// debug: mass matrix [19x19], bias [19], qacc0 [19], Minv [19x19] per env -> out[N][19*19*2 + 38]
__global__ __launch_bounds__(32 * DC_EPW) void k_doggo_coop_debug(const float* __restrict__ S, int N, double* __restrict__ out) {
  const int lane = threadIdx.x, half = lane >> 5, u = lane & 31;
  const size_t i = (size_t)blockIdx.x * DC_EPW + half;
  const bool live = i < (size_t)N;
  const int hf = half;
  DC_ENV;
  dc_load_model();
  dc_load_state(hf, u, S, (size_t)N, live ? i : (size_t)N - 1);
  __syncthreads();
  dc_smooth(hf, u, nullptr);
  double* o = out + (live ? i : 0) * (size_t)(2 * DG_NV * DG_NV + 2 * DG_NV);
  if (live && u < DG_NV) {
    for (int j = 0; j < DG_NV; j++) o[u * DG_NV + j] = E.M[u][j];
    o[DG_NV * DG_NV + u] = E.bias[u];
  }
  __syncthreads();
  dc_cholesky(hf, u);
  const double x = dc_solve1(hf, u, u < DG_NV ? E.tau[u] : 0.0);
  dc_inverse(hf, u);
  if (live && u < DG_NV) {
    o[DG_NV * DG_NV + DG_NV + u] = x;
    for (int j = 0; j < DG_NV; j++) o[DG_NV * DG_NV + 2 * DG_NV + u * DG_NV + j] = E.M[u][j];
  }
}

}  // namespace sag

// =====================================================================================
// constraint rows, PGS, planar world, integration: the full cooperative physics kernel
// =====================================================================================
namespace sag {

// exclusive prefix sum over the 32 lanes of a half (u = lane in the half)
__device__ inline int dc_scan32(int c, int u, int& total) {
  int x = c;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const int t = __shfl_up(x, d, 32);
    if (u >= d) x += t;
  }
  total = __shfl(x, 31, 32);
  return x - c;
}
// sum over the 32 lanes of a half, result in every lane
__device__ inline double dc_sum32(double v) {
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 32);
  return v;
}

// Row r whose Jacobian the owner lane has already written to E.rJ[r] (fp32): the owner's part of the row - the other
// (planar) body's terms and the velocity along the row (returned).  W = M^-1 J^T, A = J W and the regulariser are
// computed afterwards by all lanes together (dc_rows_finish): one lane doing 19 x 19 products per row while 28
// idle was 32 % of the kernel.
__device__ __attribute__((noinline)) double dc_build_row(int hf, int r, int other, double dx, double dy, double px, double py) {
  DC_ENV;
  double vel = 0;
#pragma unroll
  for (int k = 0; k < DG_NV; k++) vel += (double)E.rJ[r][k] * E.qdv[k];
  E.rOther[r] = (short)other; E.rParent[r] = -1; E.rTouch[r] = -1; E.rF[r] = 0; E.rMu[r] = 0; E.rKey[r] = 0;
  float ou0 = 0, ou1 = 0, ou2 = 0, od0 = 0, od1 = 0, orx = 0, ory = 0;
  if (other >= 0) {
    const float* m = E.wminv[other];
    const float* B = E.wfb[other];
    orx = (float)(px - (double)B[0]); ory = (float)(py - (double)B[1]);
    od0 = (float)dx; od1 = (float)dy;
    const float rxd = orx * od1 - ory * od0;
    ou0 = m[0] * od0 + m[1] * od1 + m[2] * rxd;
    ou1 = m[1] * od0 + m[3] * od1 + m[4] * rxd;
    ou2 = m[2] * od0 + m[4] * od1 + m[5] * rxd;
    vel += (double)((B[3] - B[5] * ory) * od0 + (B[4] + B[5] * orx) * od1);
  }
  E.rOu[r][0] = ou0; E.rOu[r][1] = ou1; E.rOu[r][2] = ou2; E.rOd[r][0] = od0; E.rOd[r][1] = od1;
  E.rOrx[r] = orx; E.rOry[r] = ory;
  return vel;
}

// fast PGS path: all rows of an env on its 32 lanes (one row per lane), the Delassus matrix A = J M^-1 J^T (+ the
// coupling through a shared planar body) in LDS as fp32 [DC_PGS_LANES][DC_PGS_LANES] over the kinematics region

// contribution of a unit force increment of row s to the acceleration along row r through the planar body they share
__device__ inline double dc_body_coupling(const DcEnv& E, int r, int s) {
  if (E.rOther[r] < 0 || E.rOther[r] != E.rOther[s]) return 0.0;
  return (double)((E.rOu[s][0] - E.rOu[s][2] * E.rOry[r]) * E.rOd[r][0] + (E.rOu[s][1] + E.rOu[s][2] * E.rOrx[r]) * E.rOd[r][1]);
}

// All lanes: W = J M^-1 for every row, the diagonal A_rr with regulariser and 1 / (A + R), and - when every row has a
// lane (fast) - the full Delassus matrix and each row's initial constraint acceleration J qacc0 (returned for the
// lane's own row).  Barriers inside; qacc0 must be in E.qacc.
__device__ __attribute__((noinline)) double dc_rows_finish(int hf, int u, int nrows, bool fast) {
  DC_ENV;
  double mi[DG_NV];   // row u of M^-1 (symmetric) in registers
  {
    const int uc = u < DG_NV ? u : DG_NV - 1;
#pragma unroll
    for (int k = 0; k < DG_NV; k++) mi[k] = E.M[uc][k];
  }
  __syncthreads();   // every lane holds its row: W may overwrite M^-1
  if (u < DG_NV) {
#pragma unroll 1
    for (int r = 0; r < nrows; r++) {
      double w = 0;
#pragma unroll
      for (int k = 0; k < DG_NV; k++) w += mi[k] * (double)E.rJ[r][k];   // (the same address in every lane: LDS broadcast)
      E.rW[r][u] = (float)w;
    }
  }
  __syncthreads();   // W complete; the Jacobians are final: A may overwrite the kinematics region
  float* Abuf = E.A;
  double a0 = 0;
#pragma unroll 1
  for (int rr = u; rr < nrows; rr += 32) {
    double Jr[DG_NV];
#pragma unroll
    for (int k = 0; k < DG_NV; k++) Jr[k] = (double)E.rJ[rr][k];
    double Arr = 0;
#pragma unroll
    for (int k = 0; k < DG_NV; k++) Arr += Jr[k] * (double)E.rW[rr][k];
    Arr += dc_body_coupling(E, rr, rr);
    const double imp = (double)E.rImp[rr], reg = Arr * (1 - imp) / imp;
    E.rA[rr] = Arr; E.rReg[rr] = reg; E.rInv[rr] = 1.0 / (Arr + reg);
    if (fast) {
#pragma unroll
      for (int k = 0; k < DG_NV; k++) a0 += Jr[k] * E.qacc[k];
      // A is symmetric: lane rr computes the entries (rr, rr + k mod R) for k = 0 .. R / 2 and stores both copies
      // (for even R the k = R / 2 entries would be computed twice: the lower half of the lanes takes them)
      const int half_r = nrows >> 1;
#pragma unroll 1
      for (int kk = 0; kk <= half_r; kk++) {
        if (kk == half_r && !(nrows & 1) && rr >= half_r) break;
        int sr = rr + kk;
        if (sr >= nrows) sr -= nrows;
        double v = 0;
#pragma unroll
        for (int k = 0; k < DG_NV; k++) v += Jr[k] * (double)E.rW[sr][k];
        const float av = (float)(v + dc_body_coupling(E, rr, sr));
        Abuf[rr * DC_PGS_LANES + sr] = av;
        Abuf[sr * DC_PGS_LANES + rr] = av;
      }
    }
  }
  __syncthreads();
  return a0;
}

// a value the optimiser must treat as new at this point (no instruction is emitted)
#ifndef DC_OPAQUE
#define DC_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
// lower half of the wavefront: a, upper half: b.  The lane mask is a CONSTANT (0xffffffff00000000) - written as a
// select on `half`, it is a computed wave mask that lives in a spilled SGPR pair inside the PGS loop and costs two
// v_readlane per use.
#ifndef DC_HALF_SELECT
__device__ inline int dc_half_select(int a, int b) {
  int out;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(a), "v"(b), "s"(0xffffffff00000000ull));
  return out;
}
#define DC_HALF_SELECT(a, b, half) dc_half_select(a, b)
#endif
// the same for a float
__device__ inline float dc_bcastf(float v, int lane_in_half, int half) {
  const int b = __float_as_int(v);
  const int v0 = __builtin_amdgcn_readlane(b, lane_in_half);
  if (DC_EPW == 2) { const int v1 = __builtin_amdgcn_readlane(b, lane_in_half + 32); return __int_as_float(DC_HALF_SELECT(v0, v1, half)); }
  return __int_as_float(v0);
}
// v of lane `lane_in_half` of this lane's own half, for every lane (lane index uniform across the wavefront)
__device__ inline double dc_bcast(double v, int lane_in_half, int half) {
  const long long b = __double_as_longlong(v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  const int lo0 = __builtin_amdgcn_readlane(lo, lane_in_half), hi0 = __builtin_amdgcn_readlane(hi, lane_in_half);
  int l = lo0, h = hi0;
  if (DC_EPW == 2) {
    const int lo1 = __builtin_amdgcn_readlane(lo, lane_in_half + 32), hi1 = __builtin_amdgcn_readlane(hi, lane_in_half + 32);
    l = DC_HALF_SELECT(lo0, lo1, half); h = DC_HALF_SELECT(hi0, hi1, half);
  }
  return __longlong_as_double((long long)(((unsigned long long)(unsigned int)h << 32) | (unsigned int)l));
}

// Jacobian of point c on body b along d, written to E.rJ[r] (fp32)
__device__ inline void dc_jac(int hf, int r, int b, const double* c, const double* d) {
  DC_ENV;
  const double rr[3] = {c[0] - E.pos[0], c[1] - E.pos[1], c[2] - E.pos[2]};
#pragma unroll 1
  for (int i = 0; i < DG_NV; i++) {
    double v = 0;
    if (g_dc_phys.anc[b] >> g_dc_phys.dof_body[i] & 1u) {
      double t[3];
      dg_cross(E.S[i], rr, t);
      v = d[0] * (E.S[i][3] + t[0]) + d[1] * (E.S[i][4] + t[1]) + d[2] * (E.S[i][5] + t[2]);
    }
    E.rJ[r][i] = (float)v;
  }
}

// Contact j of the substep, registered by its owner lane (floor point / collision geom): directions, point and coefficients only.  Its
// three rows - 57 Jacobian entries - are computed by all lanes together in dc_contacts_finish: one lane per contact
// doing them while the others idle was 17 % of the kernel.
__device__ inline void dc_contact_add(int hf, int j, int body, int touch, int dbl, const double* n, const double* c, double depth, int other,
                                      double bcoef, double kcoef, double mu, uint32_t key) {
  DC_ENV;
  DcContact& C = E.ct[j];
  double dir[3][3];
  for (int k = 0; k < 3; k++) dir[0][k] = n[k];
  if (fabs(n[2]) > 0.5) { dir[1][0] = 1; dir[1][1] = 0; dir[1][2] = 0; }
  else { dir[1][0] = -n[1]; dir[1][1] = n[0]; dir[1][2] = 0; }
  dg_cross(dir[0], dir[1], dir[2]);
  for (int k = 0; k < 3; k++)
    for (int a = 0; a < 3; a++) C.dir[k][a] = dir[k][a];
  for (int a = 0; a < 3; a++) C.c[a] = c[a];
  C.depth = depth; C.bcoef = (float)bcoef; C.kcoef = (float)kcoef; C.mu = (float)mu;   // (float-valued: exact)
  C.body = (short)body; C.other = (short)other; C.touch = (short)touch; C.dbl = (short)dbl; C.key = key;
}

// rows [r0, r0 + 3 nct) of the nct registered contacts: Jacobians entry by entry across the lanes, then one row per lane
__device__ __attribute__((noinline)) void dc_contacts_finish(int hf, int u, int r0, int nct) {
  DC_ENV;
  __syncthreads();   // the owners' descriptors are visible
  const int total = nct * (3 * DG_NV);
#pragma unroll 1
  for (int e = u; e < total; e += 32) {
    const int j = e / (3 * DG_NV), rem = e - j * (3 * DG_NV), k = rem / DG_NV, i = rem - k * DG_NV;
    const DcContact& C = E.ct[j];
    const int b = C.body;
    double v = 0;
    if (g_dc_phys.anc[b] >> g_dc_phys.dof_body[i] & 1u) {
      const double rr[3] = {C.c[0] - E.pos[0], C.c[1] - E.pos[1], C.c[2] - E.pos[2]};
      const double* d = C.dir[k];
      double t[3];
      dg_cross(E.S[i], rr, t);
      v = d[0] * (E.S[i][3] + t[0]) + d[1] * (E.S[i][4] + t[1]) + d[2] * (E.S[i][5] + t[2]);
    }
    E.rJ[r0 + 3 * j + k][i] = (float)v;
  }
  __syncthreads();
#pragma unroll 1
  for (int q = u; q < 3 * nct; q += 32) {
    const int j = q / 3, k = q - 3 * j, r = r0 + q, base = r0 + 3 * j;
    const DcContact& C = E.ct[j];
    const double vel = dc_build_row(hf, r, C.other, -C.dir[k][0], -C.dir[k][1], C.c[0], C.c[1]);
    double imp = dg_impedance(C.depth);
    if (C.dbl) imp = 2 * imp / (1 + imp);   // two coincident contacts in one row: half the regulariser
    E.rImp[r] = (float)imp;
    E.rKey[r] = C.key + (uint32_t)k;
    if (k == 0) { E.rAref[r] = -(double)C.bcoef * vel + (double)C.kcoef * C.depth; E.rTouch[r] = C.touch; }
    else { E.rAref[r] = -(double)C.bcoef * vel; E.rParent[r] = (short)base; E.rMu[r] = C.mu; }
  }
}

// the robot's collision geoms (one per lane) vs one planar body (oracle dg_collide_body): appends the contacts in
// (robot geom, geom of the body, contact) order; returns the number of contacts (uniform in the half)
__device__ __attribute__((noinline)) int dc_collide_body(int hf, int u, int r0, int& nrows, int fbi, int obj, int shape, float bx, float by, float byaw,
                                      float rbound, float vsz, float rstatic, double top, double bcoef, double kcoef, double mu) {
  DC_ENV;
  DgHit hits[DG_PAIR_CAP];
  int hgk[DG_PAIR_CAP] = {0, 0, 0, 0, 0};   // 8 * geom of the body + contact of the pair (row identity)
  float hz[DG_PAIR_CAP] = {0, 0, 0, 0, 0};  // height of each contact
  int cnt = 0;
  double A3[3] = {0, 0, 0}, B3[3] = {0, 0, 0};
  if (u < DG_NGEOM) {
    const int b = g_dc_phys.geom_body[u];
    double o[3];
    dg_matvec(E.R[b], g_dc_phys.geom_a[u], o);
    for (int k = 0; k < 3; k++) A3[k] = E.p[b][k] + o[k];
    dg_matvec(E.R[b], g_dc_phys.geom_b[u], o);
    for (int k = 0; k < 3; k++) B3[k] = E.p[b][k] + o[k];
    const double r = g_dc_phys.geom_r[u];
    // the part [t0, t1] of the axis at or below the object's top; all of it when all of it is above
    const bool a_in = A3[2] <= top, b_in = B3[2] <= top;
    bool go = a_in || b_in || fmin(A3[2], B3[2]) - r < top;
    float t0 = 0.f, t1 = 1.f;
    if (a_in != b_in) {
      const float tc = (float)((top - A3[2]) / (B3[2] - A3[2]));
      if (a_in) t1 = tc; else t0 = tc;
    }
    const float ax = (float)(A3[0] - (double)bx), ay = (float)(A3[1] - (double)by);
    const float dx = (float)(B3[0] - A3[0]), dy = (float)(B3[1] - A3[1]), dz = (float)(B3[2] - A3[2]);
    const float dd = dx * dx + dy * dy;
    if (go) {   // cull (conservative): the body's bounding circle against the axis
      float t = dd > 0 ? -(ax * dx + ay * dy) / dd : 0.f;
      t = t < 0 ? 0.f : (t > 1 ? 1.f : t);
      const float ex = ax + t * dx, ey = ay + t * dy, rs = (float)(1.5 * r) + rbound + 1e-4f;
      go = !(ex * ex + ey * ey > rs * rs);
    }
    if (go) {
      float cb = 1, sb = 0;
      if (fbi >= 0) sincosf(byaw, &sb, &cb);
      const int ng = shape_ngeom(shape);
      if (!g_dc_phys.geom_capsule[u]) {
        // cylinder: the rectangle under [t0, t1] of its axis
        const float len = sqrtf(dd), tm = 0.5f * (t0 + t1), zm = (float)A3[2] + tm * dz;
        float hw = (float)r;
        bool ok = true;
        if (zm > (float)top) {
          const float q = (float)(r * r) - (zm - (float)top) * (zm - (float)top);
          ok = q > 0;
          hw = sqrtf(fmaxf(q, 0.f));
        }
        if (ok) {
          const float l3 = sqrtf(dd + dz * dz), ext = l3 > 0 ? (float)r * fabsf(dz) / l3 : (float)r;
          const float ux = len > 0 ? dx / len : 1.f, uy = len > 0 ? dy / len : 0.f;
          float s0 = t0 * len, s1 = t1 * len;
          if (t0 == 0) s0 -= ext;
          if (t1 == 1) s1 += ext;
          const float rcx = (float)A3[0] + ux * 0.5f * (s0 + s1), rcy = (float)A3[1] + uy * 0.5f * (s0 + s1);
          const float rc = ux, rsn = uy;   // (the oracle stores the rectangle's yaw = atan2(uy, ux) and takes cos / sin of it: the same up to rounding)
#pragma unroll 1
          for (int g = 0; g < ng && cnt < DG_PAIR_CAP; g++) {
            const Geom ge = shape_geom(shape, g, vsz, rstatic);
            const float gx = bx + cb * ge.ox - sb * ge.oy, gy = by + sb * ge.ox + cb * ge.oy;
            const int c0 = cnt;
            const int n = dg_rect_geom(rcx, rcy, rc, rsn, 0.5f * (s1 - s0), hw, ge, gx, gy, cb, sb, hits, cnt, DG_PAIR_CAP);
            cnt = min(cnt + n, DG_PAIR_CAP);
            for (int k = c0; k < cnt; k++) {
              float tcn = len > 0 ? ((hits[k].px - (float)A3[0]) * ux + (hits[k].py - (float)A3[1]) * uy) / len : tm;
              tcn = tcn < t0 ? t0 : (tcn > t1 ? t1 : tcn);
              hz[k] = (float)A3[2] + tcn * dz;
              hgk[k] = 8 * g + (k - c0);
            }
          }
        }
      } else {
#pragma unroll 1
        for (int g = 0; g < ng && cnt < DG_PAIR_CAP; g++) {
          const Geom ge = shape_geom(shape, g, vsz, rstatic);
          const float gox = cb * ge.ox - sb * ge.oy, goy = sb * ge.ox + cb * ge.oy;
          const float wx = ax - gox, wy = ay - goy;
          float ts;
          if (!ge.box) {
            ts = dd > 0 ? -(wx * dx + wy * dy) / dd : t0;
            if (!(ts > t0)) ts = t0;
            if (ts > t1) ts = t1;
          } else {
            ts = dg_seg_box_t(cb * wx + sb * wy, -sb * wx + cb * wy, cb * dx + sb * dy, -sb * dx + cb * dy, ge.a, ge.b, t0, t1);
          }
          const double cz = A3[2] + (double)ts * (B3[2] - A3[2]);
          float rr = (float)r;
          if (cz > top) {
            const double q = r * r - (cz - top) * (cz - top);
            if (!(q > 0)) continue;
            rr = (float)sqrt(q);
          }
          const float cx = (float)(A3[0] + (double)ts * (B3[0] - A3[0])), cy = (float)(A3[1] + (double)ts * (B3[1] - A3[1]));
          DgHit h;
          if (dg_circle_geom(cx, cy, rr, ge, bx + gox, by + goy, cb, sb, h)) { hits[cnt] = h; hgk[cnt] = 8 * g; hz[cnt] = (float)cz; cnt++; }
        }
      }
    }
  }
  int total;
  const int excl = dc_scan32(cnt, u, total);
  if (total == 0) return 0;
  const int fit = min(total, (DC_ROWS - nrows) / 3);
  if (fit < total && u == 0) E.flag |= 2;   // a contact did not fit the row budget (reported, not silent)
  for (int k = 0; k < cnt; k++) {
    const int idx = excl + k;
    if (idx < fit) {
      const double nn[3] = {-(double)hits[k].nx, -(double)hits[k].ny, 0}, pc[3] = {(double)hits[k].px, (double)hits[k].py, (double)hz[k]};
      int touch = -1;
      const int t0s = g_dc_phys.geom_touch[u];
      if (t0s >= 0) {   // an ankle capsule: is the contact point inside the knee's / the foot's site?
        const double da[3] = {pc[0] - A3[0], pc[1] - A3[1], pc[2] - A3[2]}, db[3] = {pc[0] - B3[0], pc[1] - B3[1], pc[2] - B3[2]};
        if (dg_dot(da, da) <= DG_SITE_R * DG_SITE_R) touch = t0s;
        else if (dg_dot(db, db) <= DG_SITE_R * DG_SITE_R) touch = t0s + 4;
      }
      dc_contact_add(hf, (nrows - r0) / 3 + idx, g_dc_phys.geom_body[u], touch, 0, nn, pc, (double)hits[k].depth, fbi, bcoef, kcoef, mu,
                     0x10000u + 4u * (uint32_t)(32 * (8 * 8 * obj + hgk[k]) + u));
    }
  }
  nrows += 3 * fit;
  return total;
}

// Projected Gauss-Seidel of ONE env on all 64 lanes of the wavefront (an env with more than 32 rows; the other env
// of the wavefront waits its turn): lane r owns row r as in the 32-lane form of k_doggo_physics, but computes its row
// of the Delassus matrix A[r][s] = J_r W_s^T (+ the coupling through a shared planar body) straight into registers.
// Needs W, rA / rReg / rInv (dc_rows_finish) and qacc0 in E.qacc; leaves the forces in E.rF.
__device__ __attribute__((noinline)) void dc_pgs_wide(const int e, const int lane, const int nrows, const int iters, const bool warm) {
  DcEnv& E = g_dc_env[e];
  const bool mine = lane < nrows;
  const int ur = mine ? lane : 0;
  float Ar[DC_ROWS];
  float acc;
  {
    double Jr[DG_NV], a0 = 0;
#pragma unroll
    for (int k = 0; k < DG_NV; k++) { Jr[k] = (double)E.rJ[ur][k]; a0 += Jr[k] * E.qacc[k]; }
    acc = (float)a0;
#pragma unroll
    for (int s = 0; s < DC_ROWS; s++) {
      float av = 0.f;
      if (s < nrows) {   // (uniform)
        double v = 0;
#pragma unroll
        for (int k = 0; k < DG_NV; k++) v += Jr[k] * (double)E.rW[s][k];
        av = (float)(v + dc_body_coupling(E, ur, s));
      }
      Ar[s] = av;
    }
  }
  const float aref = (float)E.rAref[ur], reg = (float)E.rReg[ur], inv = (float)E.rInv[ur], mu = E.rMu[ur];
  const int parent = E.rParent[ur];
  const bool okA = mine && E.rA[ur] > 0, isfric = parent >= 0;
  float f = 0.f;
  if (warm) {
    const uint32_t key = E.rKey[ur];
    const int nprev = E.wsN;
    float raw = 0.f;
#pragma unroll 1
    for (int q = 0; q < nprev; q++) raw = E.wsKey[q] == key ? E.wsF[q] : raw;
    if (!okA || key == 0) raw = 0.f;
    const float fn0 = fmaxf(raw, 0.f);
    const float fpar = __shfl(fn0, isfric ? parent : 0, 64);
    f = isfric ? fminf(fmaxf(raw, -mu * fpar), mu * fpar) : fn0;
  }
  const float fpar0 = __shfl(f, isfric ? parent : 0, 64);
  float fn_contact = isfric ? fpar0 : 0.f;
  if (warm) {
#pragma unroll
    for (int r = 0; r < DC_ROWS; r++)
      if (r < __builtin_amdgcn_readfirstlane(nrows)) acc += Ar[r] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), r));   // (no `break`: the loop must unroll - Ar is a register array)
  }
  // (the same short dependent chain as the 32-lane form in k_doggo_physics: see there)
  const float ninv = okA ? -inv : 0.f, c1 = okA ? 1.f - reg * inv : 1.f, c0 = okA ? aref * inv : 0.f;
  const float mu_eff = isfric ? mu : 0.f, hi0 = isfric ? 0.f : 3.0e38f;
  const int nrows_u = __builtin_amdgcn_readfirstlane(nrows);
  int uu = lane, pp = parent;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    DC_OPAQUE(uu); DC_OPAQUE(pp);
#pragma unroll
    for (int r = 0; r < DC_ROWS; r++)
      if (r < nrows_u) {
        const float hi = fmaf(mu_eff, __builtin_amdgcn_fmed3f(fn_contact, 0.f, 3.0e38f), hi0);   // (as in dc_pgs_fast)
        const float fc = __builtin_amdgcn_fmed3f(fmaf(ninv, acc, fmaf(c1, f, c0)), isfric ? -hi : 0.f, hi);
        const float df = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fc - f), r));
        acc = fmaf(Ar[r], df, acc);
        f = uu == r ? fc : f;
        fn_contact += pp == r ? df : 0.f;
      }
  }
  if (mine) E.rF[lane] = (double)f;
}

// The 32-lane PGS path: every row of an env on its own lane of the env's half (`part`: this half takes part in the pass;
// npass: rows of the pass = the larger count of the halves taking part, uniform; a0: the lane's row's J qacc0).  Leaves
// the forces in E.rF.  Not inlined: the kernel calls it once for both halves together or once per half (k_doggo_physics).
__device__ __attribute__((noinline)) void dc_pgs_fast(const int hf, const int u, const int half, const int nrows, const int npass, const bool part,
                                                      const double a0, const int iters, const bool warm) {
  DC_ENV;
  // ---- projected Gauss-Seidel on the Delassus matrix: lane r owns row r (its accumulated force f and its
  //      constraint acceleration acc = J_r qacc + planar-body term).  Updating row r changes every row's acc by
  //      A[.][r] df: one broadcast (v_readlane: r is uniform) and one multiply-add per lane, no barrier, no
  //      19-term dot product.  fp32 like A (round 2 ran these sweeps in fp64: 4 of them; the specification now runs
  //      24, and a sweep is a contraction, so rounding does not accumulate).  Same sweep order and bounds as the oracle.
  const bool mine = part && u < nrows;
  const int ur = mine ? u : 0;
  float Ar[DC_PGS_LANES];         // the lane's row of A in registers (A[r][u] = A[u][r]: consecutive words across lanes)
#pragma unroll
  for (int r = 0; r < DC_PGS_LANES; r++) Ar[r] = mine && r < nrows ? E.A[r * DC_PGS_LANES + ur] : 0.f;
  const float aref = (float)E.rAref[ur], reg = (float)E.rReg[ur], inv = (float)E.rInv[ur], mu = E.rMu[ur];
  const int parent = E.rParent[ur];
  const bool okA = mine && E.rA[ur] > 0, isfric = parent >= 0;
  // warm start: the force this row had in the previous forward evaluation (matched by identity), inside its bounds
  float f = 0.f;
  if (warm) {
    const uint32_t key = E.rKey[ur];
    const int nprev = E.wsN;
    float raw = 0.f;
#pragma unroll 1
    for (int q = 0; q < nprev; q++) raw = E.wsKey[q] == key ? E.wsF[q] : raw;   // (same address in every lane: LDS broadcast)
    if (!okA || key == 0) raw = 0.f;
    const float fn0 = fmaxf(raw, 0.f);                       // a normal / limit row: [0, inf)
    const float fpar = __shfl(fn0, isfric ? parent : 0, 32);  // a friction row: +-mu x its normal row's force
    f = isfric ? fminf(fmaxf(raw, -mu * fpar), mu * fpar) : fn0;
  }
  const float fpar0 = __shfl(f, isfric ? parent : 0, 32);   // (every lane takes part in the exchange)
  float fn_contact = isfric ? fpar0 : 0.f;
  float acc = (float)a0;
  if (warm) {
#pragma unroll
    for (int r = 0; r < DC_PGS_LANES; r++)
      if (r < __builtin_amdgcn_readfirstlane(npass)) acc += Ar[r] * dc_bcastf(f, r, half);   // (no `break`: the loop must unroll - Ar is a register array)
  }
  // The sweep is ONE dependent chain through all rows (row r+1 needs the acc that row r's increment changed); a
  // wavefront alone on its SIMD pays every link's latency, so the chain is kept short: the update
  //   f <- clamp(f + (aref - acc - reg f) inv)  =  clamp(c1 f + c0 - inv acc),  c1 = 1 - reg inv, c0 = aref inv
  // has its own-force part t = c1 f + c0 precomputed (f changes only at the row's own turn), every lane computes a
  // candidate increment and the row's is picked by the broadcast itself (no select in front of it); rows that take
  // no part (A <= 0, lanes beyond the last row) carry inv = 0, c1 = 1, c0 = 0: their candidate is 0.  Chain per
  // row: fma, max, min, sub, readlane, (half select), fma.
  // The lane tests `u == r`, `parent == r` are made on the spot (a compare + a select): hoisted out of the sweep
  // loop they are 64 wave masks that live in spilled SGPRs and come back through two v_readlane each per row.
  const float ninv = okA ? -inv : 0.f, c1 = okA ? 1.f - reg * inv : 1.f, c0 = okA ? aref * inv : 0.f;
  const float mu_eff = isfric ? mu : 0.f, hi0 = isfric ? 0.f : 3.0e38f;   // bounds: [-mu fn, mu fn] / [0, inf)
  const int nmax_u = __builtin_amdgcn_readfirstlane(npass);              // (uniform: the row tests are scalar branches)
  int uu = u, pp = parent;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    DC_OPAQUE(uu); DC_OPAQUE(pp);   // (once per sweep, in place: keeps the row's two compares inside the sweep loop without a copy per row)
#pragma unroll
    for (int r = 0; r < DC_PGS_LANES; r++)   // unrolled: lane index and register index of the row are constants
      if (r < nmax_u) {
        // (the running sum of increments may end a hair below zero when the normal force clamps to 0: the bound must not
        //  turn negative; med3 instead of fmaxf: one instruction, no canonicalising copy in front of it)
        const float hi = fmaf(mu_eff, __builtin_amdgcn_fmed3f(fn_contact, 0.f, 3.0e38f), hi0);
        const float fc = __builtin_amdgcn_fmed3f(fmaf(ninv, acc, fmaf(c1, f, c0)), isfric ? -hi : 0.f, hi);
        const float df = dc_bcastf(fc - f, r, half);
        acc = fmaf(Ar[r], df, acc);
        f = uu == r ? fc : f;
        // a friction row tracks its contact's normal force by the same increments
        fn_contact += pp == r ? df : 0.f;
      }
  }
  if (mine) E.rF[u] = (double)f;
}

// Longest-first launch order.  The kernel is as long as its last wavefront, and a wavefront's time goes with its envs'
// constraint rows (the PGS is one dependent chain through all of them): 15 rows is the median; a robot lying on the floor
// has 25 - 35; one entangled with the HaulBox box 50, on the 64-lane path - 2.5 ms alone where an ordinary pair takes
// 1.4.  ONE such env among 4096, started in the second round of wavefronts, made the multitask step 4.0 instead of
// 3.0 ms.  So every env files itself, at the end of a step, under the cost class of the rows it had (atomic append to
// one of DC_NCLS lists), and the next step's workgroups take the lists in order, costliest first (the mapping is at
// the head of k_doggo_physics).  sched: [0 .. 12) three rotating
// sets of DC_NCLS counters (this step's lists, the ones being built, the ones being cleared), [16 ..) two buffers of
// DC_NCLS lists of N env ids.  The order inside a list is the order of arrival: it changes from run to run, the results
// do not - an env's arithmetic depends on its own row count only (which PGS path it takes is decided per env).
constexpr int DC_NCLS = 4;
#ifndef SAG_DC_CLS2
#define SAG_DC_CLS2 26   // rows from which an env is filed in class 2 / class 1
#define SAG_DC_CLS1 20
#endif
__host__ __device__ inline size_t dc_sched_ints(size_t N) { return 16 + 2 * DC_NCLS * N; }
__host__ __device__ inline int dc_sched_blocks(int N) { return (N + DC_EPW - 1) / DC_EPW; }

// inputs of the planar world shared by the lanes of a half
struct DcWorldK {
  int nV, nP, nB, task;
  bool has_box, haul;
  float vsz, psz;
  BodyK vk, bk;
  Sol sol0, solb;
};

__global__ __launch_bounds__(32 * DC_EPW) void k_doggo_physics(StepArgs p, double* __restrict__ DR) {
  __shared__ float ctrl_s[DC_EPW][12];
  __shared__ float stx_s[DC_EPW][SAG_MAX_PILLARS + SAG_MAX_BUTTONS], sty_s[DC_EPW][SAG_MAX_PILLARS + SAG_MAX_BUTTONS];
  const int lane = threadIdx.x, half = lane >> 5, u = lane & 31;
  const size_t N = (size_t)p.N;
  size_t gi = (size_t)blockIdx.x * DC_EPW + half;
  bool live = gi < N;
  const bool sched = p.dg_sched != nullptr && p.dg_phase >= 0;
  int32_t* const sc = p.dg_sched;
  if (sched) {   // longest-first order (above)
    const int32_t* cnt = sc + (p.dg_phase % 3) * DC_NCLS;
    if (blockIdx.x == 0 && lane < DC_NCLS) sc[((p.dg_phase + 2) % 3) * DC_NCLS + lane] = 0;
    const int n3 = cnt[3], n2 = cnt[2], n1 = cnt[1], n0 = cnt[0];
    int b = (int)blockIdx.x;
    if ((size_t)n3 + n2 + n1 + n0 != N) {   // no complete set of lists yet (first step): plain order
      gi = (size_t)b * DC_EPW + half; live = gi < N;
    } else {
      // The envs as ONE sequence, costliest first: class 3, 2, 1, 0, each list read from its END (envs arrive in the order
      // their wavefronts finished).  Workgroup b < n3 takes the b-th env of the sequence AND the b-th from its end - an env
      // on the 64-lane path with one of the cheapest (the two are solved one after the other) -, the others take
      // consecutive pairs of what lies between.  Exactly ceil(N / 2) workgroups have work: with 4096 envs the chip runs
      // two full rounds of them, and a single workgroup more starts a third.
      const int32_t* lists = sc + 16 + (size_t)(p.dg_phase & 1) * DC_NCLS * N;
      const int heavy = min(n3, (int)(N / 2));
      int q;   // this half's position in the sequence
      if (b < heavy) q = half == 0 ? b : (int)N - 1 - b;
      else q = heavy + (b - heavy) * DC_EPW + half;
      live = q < (int)N - (b < heavy ? 0 : heavy) && b < (int)((N + DC_EPW - 1) / DC_EPW);
      int cls = 3, off = q;
      if (off >= n3) { off -= n3; cls = 2; if (off >= n2) { off -= n2; cls = 1; if (off >= n1) { off -= n1; cls = 0; } } }
      const int end = cls == 3 ? n3 : (cls == 2 ? n2 : (cls == 1 ? n1 : n0));
      gi = live ? (size_t)lists[(size_t)cls * N + (end - 1 - off)] : 0;
    }
    if (__ballot(live) == 0) return;   // (one wavefront per workgroup: uniform)
  }
  const size_t i = live ? gi : N - 1;
  const int hf = half;
  DC_ENV;
  int rows_max = 0;
#ifdef SAG_CYCLES
  if (lane == 0) { for (int k = 0; k < DCY_N; k++) g_dc_cyc[k] = 0; g_dc_cyc_t = __builtin_readcyclecounter(); }
#endif
  float* __restrict__ S = p.S;
  const float* stx = stx_s[half];
  const float* sty = sty_s[half];

  // ---- per-env constants (every lane reads the same words: broadcast loads) ----------------
  const uint32_t meta = (uint32_t)p.I[iaddr(DI_META, N, i)];
  DcWorldK W;
  W.task = meta & 15; W.nV = meta >> 8 & 15; W.nP = meta >> 12 & 3; W.nB = meta >> 14 & 7;
  const int box_kind = meta >> 17 & 3;
  W.has_box = box_kind != SAG_BOX_NONE;
  W.haul = W.task == SAG_TASK_HAUL_BOX;
  W.vsz = S[saddr(SAG_F_VASE_SIZE, N, i)]; W.psz = S[saddr(SAG_F_PILLAR_SIZE, N, i)];
  const float h = p.h;
  {
    const float tc = fmaxf(0.02f, 2.0f * h);
    W.sol0.bcoef = 2.0f / (SOL_D1 * tc); W.sol0.kcoef = 1.0f / (SOL_D1 * SOL_D1 * tc * tc); W.sol0.mu = MU;
    W.solb = W.sol0;
    if (box_kind == SAG_BOX_ROD || box_kind == SAG_BOX_BALL) W.solb.mu = 1.2f;
    if (box_kind == SAG_BOX_BALL) {
      const float tcb = fmaxf(0.018f, 2.0f * h);
      W.solb.bcoef = 2.0f / (SOL_D1 * tcb); W.solb.kcoef = 1.0f / (SOL_D1 * SOL_D1 * tcb * tcb * 0.2f * 0.2f);
    }
    const float vsz = W.vsz;
    W.vk.sh = SH_VASE; W.vk.m = VASE_DENSITY * 8 * vsz * vsz * vsz; W.vk.I = W.vk.m * (8 * vsz * vsz) / 12;
    W.vk.reff = vsz * 1.41421356237309504880f;
    W.bk.sh = SH_BOX; W.bk.m = 1; W.bk.I = 1; W.bk.reff = 0.2f * 1.41421356237309504880f;
    if (box_kind == SAG_BOX_BOX) {
      const float m0 = 0.001f * 0.064f, m1 = 0.001f * 0.016f;
      W.bk.m = m0 + 4 * m1; W.bk.I = m0 * 0.32f / 12 + 4 * (m1 * 0.08f / 12 + m1 * 0.08f);
    } else if (box_kind == SAG_BOX_ROD) {
      W.bk.sh = SH_ROD; W.bk.m = 0.0005f * (3.14159265358979323846f * 0.08f * 0.08f * 0.6f);
      W.bk.I = W.bk.m * (3 * 0.08f * 0.08f + 0.36f) / 12;
    } else if (box_kind == SAG_BOX_BALL) {
      W.bk.sh = SH_BALL; W.bk.m = 0.0005f * (4.0f / 3.0f * 3.14159265358979323846f * 0.14f * 0.14f * 0.14f);
      W.bk.I = 0.4f * W.bk.m * 0.14f * 0.14f;
    }
  }
  const float vase_r = W.vk.reff, box_r = shape_bound(W.bk.sh, W.vsz, 0);
  const double top_vase = 2.0 * (double)W.vsz;
  const double top_box = W.bk.sh == SH_ROD ? 0.16 : (W.bk.sh == SH_BALL ? 0.28 : 0.4);

  // ---- load: model tables, robot state (lane 0), planar bodies (lane k), statics, controls ----
  dc_load_model();
  dc_load_state(hf, u, S, N, i);
  if (u < NBODY) {
    const int k = u;
    const bool isb = k == BOX_ID, on = isb ? W.has_box : k < W.nV;
    for (int c = 0; c < 3; c++) {
      E.wfb[k][c] = on ? S[((size_t)((DV_POS + 3 * k + c) >> 2) * N + i) * 4 + ((DV_POS + 3 * k + c) & 3)] : 0.f;
      E.wfb[k][3 + c] = on ? S[((size_t)((DV_VEL + 3 * k + c) >> 2) * N + i) * 4 + ((DV_VEL + 3 * k + c) & 3)] : 0.f;
      E.wfb[k][6 + c] = 0.f;
    }
    for (int c = 0; c < 6; c++) E.wminv[k][c] = 0;
    if (!isb) { E.wminv[k][0] = E.wminv[k][3] = 1.0f / W.vk.m; E.wminv[k][5] = 1.0f / W.vk.I; }
    else {
      E.wminv[k][5] = 1.0f / W.bk.I;
      E.wminv[k][0] = E.wminv[k][3] = W.bk.sh == SH_BALL ? 1.0f / (1.4f * W.bk.m) : 1.0f / W.bk.m;
    }
  }
  if (u < SAG_MAX_PILLARS + SAG_MAX_BUTTONS) {
    const int f = u < SAG_MAX_PILLARS ? SAG_F_PILLARS + 2 * u : SAG_F_BUTTONS + 2 * (u - SAG_MAX_PILLARS);
    stx_s[half][u] = S[saddr(f, N, i)]; sty_s[half][u] = S[saddr(f + 1, N, i)];
  }
  if (u < 12) {
    float c = 0;
    if (!p.observe_only) {
      float z;
      if (p.noise) z = p.noise[i * 12 + u];
      else {
        const uint32_t env_id = (uint32_t)p.I[iaddr(DI_ENVID, N, i)], step = (uint32_t)p.I[iaddr(DI_STEP, N, i)];
        const uint32_t ep4 = ((uint32_t)p.I[iaddr(DI_FLAGS, N, i)] >> FLAG_EPISODE_SHIFT) << 2;
        uint32_t cc[4] = {env_id, step, (uint32_t)(u >> 1), ep4 | 1u};
        philox4x32_10(cc, p.key0, p.key1);
        float z0, z1;
        box_muller(cc[0], cc[1], z0, z1);
        z = (u & 1) ? z1 : z0;
      }
      const float an = S[saddr(SAG_F_ACTION_NOISE, N, i)], lim = S[saddr(SAG_F_CTRL_SCALE + u, N, i)];
      c = clampf(p.actions[i * 12 + u] + an * z, -lim, lim);
    }
    ctrl_s[half][u] = c;
  }
  __syncthreads();
  DCC(DCY_LOAD);

  const int nsub = p.observe_only ? 0 : p.nstep;
  const uint32_t awake0 = (uint32_t)p.I[iaddr(DI_TSTATE, N, i)] >> TS_AWAKE_SHIFT & TS_AWAKE_BITS;
  int cost_contacts = 0;
  uint32_t btn_mask = 0;
  double qacc_u = 0;
#pragma unroll 1
  for (int sub = 0; sub <= nsub; sub++) {
    if (u < NBODY) { E.wfb[u][6] = 0; E.wfb[u][7] = 0; E.wfb[u][8] = 0; }
    // planar bodies that take part in this forward evaluation: the moving ones, + below those the robot touches
    // (specification: oracle world_forward, "sleeping bodies"); uniform in the half
    uint32_t act = (uint32_t)(__ballot(u < NBODY && (E.wfb[u][3] != 0 || E.wfb[u][4] != 0 || E.wfb[u][5] != 0)) >> (32 * half)) & ((1u << NBODY) - 1);
    if (sub == 0) act |= awake0;   // SAG_I_AWAKE: bodies installed overlapping something take part in the first evaluation
    if (u == BOX_ID && W.has_box && W.bk.sh == SH_ROD) {
      float c, s; sincosf(E.wfb[BOX_ID][2], &s, &c);
      const float ix = 1.0f / (1.5f * W.bk.m), iy = 1.0f / W.bk.m;
      E.wminv[BOX_ID][0] = c * c * ix + s * s * iy; E.wminv[BOX_ID][1] = c * s * (ix - iy);
      E.wminv[BOX_ID][3] = s * s * ix + c * c * iy;
    }
    dc_smooth(hf, u, ctrl_s[half]);
    if (sub == nsub) {   // outputs of the final state: centre of mass and momentum / mass (M's rows 0, 1 before they turn into L)
      double m = 0;
      for (int b = 0; b < DG_NB; b++) m += E.Ib[b][0];
      if (u < 2) {
        double mc = 0, P = m * E.qdv[u];
        for (int b = 0; b < DG_NB; b++) mc += E.Ib[b][1 + u];
        for (int j = 3; j < DG_NV; j++) P += E.M[u][j] * E.qdv[j];
        E.com[u] = E.pos[u] + mc / m; E.mom[u] = P / m;
      }
    }
    dc_cholesky(hf, u);
    DCC(DCY_CHOL);
    qacc_u = dc_inverse(hf, u);   // M^-1 and qacc0 = M^-1 tau
    DCC(DCY_INV);
    if (u < 8) E.touch[u] = 0;
    int nrows = 0;
    // ---- rows: joint limits (lane j), in ascending joint order ---------------------------------
    {
      double depth = 0, sign = 0;
      if (u < DG_NJ) {
        if (E.q[u] < g_dc_phys.lo[u]) { depth = g_dc_phys.lo[u] - E.q[u]; sign = 1; }
        else if (E.q[u] > g_dc_phys.hi[u]) { depth = E.q[u] - g_dc_phys.hi[u]; sign = -1; }
      }
      int total;
      const int excl = dc_scan32(sign != 0 ? 1 : 0, u, total);
      const int fit = min(total, DC_ROWS - nrows);
      if (fit < total && u == 0) E.flag |= 2;
      if (sign != 0 && excl < fit) {
        const int r = nrows + excl;
        for (int k = 0; k < DG_NV; k++) E.rJ[r][k] = k == 6 + u ? (float)sign : 0.f;
        const double vel = dc_build_row(hf, r, -1, 0, 0, 0, 0);
        E.rImp[r] = (float)dg_impedance(depth);
        E.rAref[r] = -(double)W.sol0.bcoef * vel + (double)W.sol0.kcoef * depth;
        E.rKey[r] = 0x100u + (uint32_t)(2 * u + (sign < 0 ? 1 : 0));
      }
      nrows += fit;
    }
    // ---- floor (lane s: floor point s = an end of a geom's axis) ------------------------------------
    const int r0 = nrows;   // the contacts' rows start here
    {
      double depth = 0, lp[3] = {0, 0, 0};
      int code = 0;
      if (u < DG_NFP) {
        code = g_dc_phys.fp_code[u];
        const int g = code & 15, b = g_dc_phys.geom_body[g];
        const double* R = E.R[b];
        const double r = g_dc_phys.geom_r[g];
        double o[3];
        dg_matvec(R, (code >> 4 & 1) ? g_dc_phys.geom_b[g] : g_dc_phys.geom_a[g], o);
        for (int k = 0; k < 3; k++) lp[k] = E.p[b][k] + o[k];
        if (!(code >> 5 & 1)) lp[2] -= r;   // a capsule's end sphere: the point under its centre
        else {
          // lowest point of the cylinder's rim: centre + r (-z + u_z u) / |..|, u = the axis (the body's x); upright: the centre
          const double ux = R[0], uy = R[3], uz = R[6], n2 = 1 - uz * uz;
          if (n2 > 1e-12) {
            const double sc = r / sqrt(n2);
            lp[0] += sc * uz * ux; lp[1] += sc * uz * uy; lp[2] += sc * (uz * uz - 1);
          }
        }
        depth = -lp[2];
      }
      int total;
      const int excl = dc_scan32(depth > 0 ? 1 : 0, u, total);
      const int fit = min(total, (DC_ROWS - nrows) / 3);
      if (fit < total && u == 0) E.flag |= 2;
      if (depth > 0 && excl < fit) {
        const double n[3] = {0, 0, 1}, c[3] = {lp[0], lp[1], 0.5 * lp[2]};
        const int slot = (code >> 7 & 15) - 1;
        dc_contact_add(hf, excl, g_dc_phys.geom_body[code & 15], slot < 0 ? -1 : slot + 8 * (code >> 11 & 1), code >> 6 & 1, n, c, depth, -1,
                       (double)W.sol0.bcoef, (double)W.sol0.kcoef, (double)MU, 0x1000u + 4u * (uint32_t)u);
      }
      nrows += 3 * fit;
    }
    DCC(DCY_ROWS_SELF);
    // ---- world objects, in the specification's order -------------------------------------------
    int cc = 0;
    uint32_t mask = 0;
    auto near = [&](float bx, float by, float rb) {
      const double dx = (double)bx - E.pos[0], dy = (double)by - E.pos[1], rs = 0.6 + (double)rb;
      return dx * dx + dy * dy <= rs * rs;
    };
#pragma unroll 1
    for (int q = 0; q < W.nP; q++)
      if (near(stx[q], sty[q], W.psz))
        cc += dc_collide_body(hf, u, r0, nrows, -1, q, SH_STATIC, stx[q], sty[q], 0.f, W.psz, W.vsz, W.psz, 1.0,
                              (double)W.sol0.bcoef, (double)W.sol0.kcoef, (double)W.sol0.mu);
#pragma unroll 1
    for (int b = 0; b < W.nB; b++)
      if (near(stx[SAG_MAX_PILLARS + b], sty[SAG_MAX_PILLARS + b], BUTTON_R) &&
          dc_collide_body(hf, u, r0, nrows, -1, SAG_MAX_PILLARS + b, SH_STATIC, stx[SAG_MAX_PILLARS + b], sty[SAG_MAX_PILLARS + b], 0.f, BUTTON_R, W.vsz,
                          BUTTON_R, 0.2, (double)W.sol0.bcoef, (double)W.sol0.kcoef, (double)W.sol0.mu))
        mask |= 1u << b;
#pragma unroll 1
    for (int k = 0; k < W.nV; k++)
      if (near(E.wfb[k][0], E.wfb[k][1], vase_r)) {
        const int nc = dc_collide_body(hf, u, r0, nrows, k, SAG_MAX_PILLARS + SAG_MAX_BUTTONS + k, SH_VASE, E.wfb[k][0], E.wfb[k][1], E.wfb[k][2],
                                       vase_r, W.vsz, 0.f, top_vase, (double)W.sol0.bcoef, (double)W.sol0.kcoef, (double)W.sol0.mu);
        cc += nc;
        if (nc) act |= 1u << k;   // a body the robot touches takes part in this evaluation
      }
    if (W.has_box) {
      if (near(E.wfb[BOX_ID][0], E.wfb[BOX_ID][1], box_r) &&
          dc_collide_body(hf, u, r0, nrows, BOX_ID, SAG_MAX_PILLARS + SAG_MAX_BUTTONS + SAG_MAX_VASES, W.bk.sh, E.wfb[BOX_ID][0], E.wfb[BOX_ID][1],
                          E.wfb[BOX_ID][2], box_r, W.vsz, 0.f, top_box, (double)W.solb.bcoef, (double)W.solb.kcoef, (double)W.solb.mu))
        act |= 1u << BOX_ID;
    }
    dc_contacts_finish(hf, u, r0, (nrows - r0) / 3);
    if (W.has_box) {
      if (W.haul && nrows >= DC_ROWS && u == 0) E.flag |= 2;
      if (W.haul && nrows < DC_ROWS) {   // haul_box.py:21-29: tendon base site <-> box site (z .2), range [0, .75]
        const double dx = (double)E.wfb[BOX_ID][0] - E.pos[0], dy = (double)E.wfb[BOX_ID][1] - E.pos[1], dz = 0.2 - E.pos[2];
        const double d2 = dx * dx + dy * dy, Lt = sqrt(d2 + dz * dz), viol = Lt - 0.75;
        if (viol > 0 && d2 >= 1e-18) {
          if (u == 0) {
            const double j[3] = {dx / Lt, dy / Lt, dz / Lt};
            dc_jac(hf, nrows, 0, E.pos, j);
            const double vel = dc_build_row(hf, nrows, BOX_ID, -j[0], -j[1], (double)E.wfb[BOX_ID][0], (double)E.wfb[BOX_ID][1]);
            E.rImp[nrows] = (float)dg_impedance(viol);
            E.rAref[nrows] = -(double)W.sol0.bcoef * vel + (double)W.sol0.kcoef * viol;
            E.rKey[nrows] = 0x80u;
          }
          nrows += 1;
          act |= 1u << BOX_ID;
        }
      }
    }
    if (u < DG_NV) E.qacc[u] = qacc_u;   // qacc0 for the rows' initial constraint accelerations
    __syncthreads();
    DCC(DCY_ROWS_WORLD);
    // both envs of the wavefront take the same path: every row on its own lane of the env's half (<= 32 rows: the
    // usual case), or one env after the other on all 64 lanes
    if (!live) nrows = 0;   // an idle half (odd batch, an env that runs in a front workgroup) takes no part in the solve
    const int nother = DC_EPW == 2 ? __shfl(nrows, (lane + 32) & 63) : nrows;
    const int nmax = max(nrows, nother);
    rows_max = max(rows_max, nrows);
    // which path an env takes is decided by ITS row count alone (its arithmetic must not depend on its partner in the
    // wavefront: the launch order pairs envs differently from run to run)
    const bool fast = nrows <= SAG_DC_FAST_ROWS;
#ifdef SAG_CYCLES
    if (lane == 0) atomicAdd(&g_cyc[0][min(nmax >> 2, CY_N)], 1ull);   // histogram of the wavefront's row count per evaluation, buckets of 4
#endif
    const double a0 = dc_rows_finish(hf, u, nrows, fast);
    DCC(DCY_FINISH);
    // the first forward evaluation of an env-step starts cold, and so does the one at the final state behind the observation
    // (accelerometer and touch are functions of the state alone: a step's columns equal sag_observe's of the same state)
    const bool warm = sub > 0 && sub < nsub;
    const int iters = warm ? DG_PGS_ITERS : DG_PGS_ITERS0;
    // both envs on the 32-lane path: one pass for the two halves together; otherwise one env after the other, each on its path
    if (nmax <= SAG_DC_FAST_ROWS) dc_pgs_fast(hf, u, half, nrows, nmax, true, a0, iters, warm);
    else {
      // more than 32 rows (a robot lying on the floor, or deep inside the task box): on all 64 lanes.  (Three barriers
      // per row update made such a wavefront the kernel's tail: 13 ms.)
      const int n0 = __shfl(nrows, 0), n1 = DC_EPW == 2 ? __shfl(nrows, 32) : 0;   // (uniform)
      if (n0 > SAG_DC_FAST_ROWS) dc_pgs_wide(0, lane, n0, iters, warm);
      else if (n0 > 0) dc_pgs_fast(hf, u, half, nrows, n0, half == 0, a0, iters, warm);
      if (n1 > SAG_DC_FAST_ROWS) dc_pgs_wide(1, lane, n1, iters, warm);
      else if (n1 > 0) dc_pgs_fast(hf, u, half, nrows, n1, half == 1, a0, iters, warm);
    }
    __syncthreads();
    DCC(DCY_PGS);
    if (u < DG_NV) {
      double q = qacc_u;
#pragma unroll 1
      for (int r = 0; r < nrows; r++) q += (double)E.rW[r][u] * E.rF[r];
      qacc_u = q;
    }
    if (u < NBODY) {   // planar bodies: accelerations from the rows that act on them
      float b6 = E.wfb[u][6], b7 = E.wfb[u][7], b8 = E.wfb[u][8];
      for (int r = 0; r < nrows; r++)
        if (E.rOther[r] == u) {
          const double fr = E.rF[r];
          b6 += (float)((double)E.rOu[r][0] * fr); b7 += (float)((double)E.rOu[r][1] * fr); b8 += (float)((double)E.rOu[r][2] * fr);
        }
      E.wfb[u][6] = b6; E.wfb[u][7] = b7; E.wfb[u][8] = b8;
    }
    __syncthreads();
    if (u < DG_NV) E.qacc[u] = qacc_u;
    if (u < 8) {
      double t = 0;
      for (int r = 0; r < nrows; r++)
        if (E.rTouch[r] >= 0 && (E.rTouch[r] & 7) == u && E.rParent[r] < 0) t += E.rTouch[r] >= 8 ? 0.5 * E.rF[r] : E.rF[r];
      E.touch[u] = t;
    }
    // this evaluation's rows and forces for the next one's warm start
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int r = u + 32 * t;
      if (r < nrows) { E.wsKey[r] = E.rKey[r]; E.wsF[r] = (float)E.rF[r]; }
    }
    if (u == 0) E.wsN = nrows;
    cost_contacts = cc; btn_mask = mask;
    __syncthreads();
    DCC(DCY_AFTER);
    if (sub == nsub) break;
    // ---- planar world: body k on lane k vs the statics; pairs on lane 0; friction + integration --
    //      (active bodies only: the moving ones and those the robot touched in this evaluation)
    if (u < NBODY) {
      const int k = u;
      const bool isb = k == BOX_ID;
      if ((isb ? W.has_box : k < W.nV) && (act >> k & 1u)) {
        const float br = isb ? box_r : vase_r;
        for (int q = 0; q < SAG_MAX_PILLARS + W.nB; q++) {
          if (q == W.nP && q < SAG_MAX_PILLARS) q = SAG_MAX_PILLARS;
          if (q >= SAG_MAX_PILLARS + W.nB) break;
          const bool is_p = q < SAG_MAX_PILLARS;
          const float sr = is_p ? W.psz : BUTTON_R;
          const float dx = stx[q] - E.wfb[k][0], dyy = sty[q] - E.wfb[k][1], rs = br + sr;
          if (dx * dx + dyy * dyy > rs * rs) continue;
          BV V; float cv, sv;
          {
            const float* B = E.wfb[k]; const float* m = E.wminv[k];
            V.x = B[0]; V.y = B[1]; V.vx = B[3]; V.vy = B[4]; V.w = B[5]; V.ax = B[6]; V.ay = B[7]; V.aw = B[8];
            sincosf(B[2], &sv, &cv);
            V.m0 = m[0]; V.m1 = m[1]; V.m2 = m[2]; V.m3 = m[3]; V.m4 = m[4]; V.m5 = m[5]; V.dyn = 1;
          }
          BV St; St.x = stx[q]; St.y = sty[q]; St.vx = St.vy = St.w = St.ax = St.ay = St.aw = 0;
          St.m0 = St.m1 = St.m2 = St.m3 = St.m4 = St.m5 = 0; St.dyn = 0;
          collide_shapes(V, isb ? W.bk.sh : SH_VASE, cv, sv, St, SH_STATIC, 1.f, 0.f, W.vsz, sr, isb ? W.solb : W.sol0);
          E.wfb[k][6] = V.ax; E.wfb[k][7] = V.ay; E.wfb[k][8] = V.aw;
        }
      }
    }
    __syncthreads();
    // pairs of planar bodies: lane a tests its pairs' bounding circles; lane 0 walks the pairs only when one overlaps
    bool pair_near = false;
    if (u < W.nV)
      for (int b = u + 1; b < NBODY; b++) {
        const bool isb = b == BOX_ID;
        if (isb ? !W.has_box : b >= W.nV) continue;
        if (!((act >> u | act >> b) & 1u)) continue;   // two sleeping bodies: the pair is not evaluated
        const float dx = E.wfb[b][0] - E.wfb[u][0], dyy = E.wfb[b][1] - E.wfb[u][1], rs = vase_r + (isb ? box_r : vase_r);
        pair_near |= !(dx * dx + dyy * dyy > rs * rs);
      }
    const unsigned long long near_lanes = __ballot(pair_near);
    if (u == 0 && (near_lanes >> (32 * half) & 0xffffffffull)) {
      // (which pairs are walked is decided by who was active when the walk started: `act` is not updated inside it)
      for (int a = 0; a < W.nV; a++)
        for (int b = a + 1; b < NBODY; b++) {
          const bool isb = b == BOX_ID;
          if (isb ? !W.has_box : b >= W.nV) continue;
          if (!((act >> a | act >> b) & 1u)) continue;
          const float dx = E.wfb[b][0] - E.wfb[a][0], dyy = E.wfb[b][1] - E.wfb[a][1], rs = vase_r + (isb ? box_r : vase_r);
          if (dx * dx + dyy * dyy > rs * rs) continue;
          BV A, B; float ca, sa, cb, sb;
          {
            const float* P = E.wfb[a]; const float* m = E.wminv[a];
            A.x = P[0]; A.y = P[1]; A.vx = P[3]; A.vy = P[4]; A.w = P[5]; A.ax = P[6]; A.ay = P[7]; A.aw = P[8];
            sincosf(P[2], &sa, &ca);
            A.m0 = m[0]; A.m1 = m[1]; A.m2 = m[2]; A.m3 = m[3]; A.m4 = m[4]; A.m5 = m[5]; A.dyn = 1;
          }
          {
            const float* P = E.wfb[b]; const float* m = E.wminv[b];
            B.x = P[0]; B.y = P[1]; B.vx = P[3]; B.vy = P[4]; B.w = P[5]; B.ax = P[6]; B.ay = P[7]; B.aw = P[8];
            sincosf(P[2], &sb, &cb);
            B.m0 = m[0]; B.m1 = m[1]; B.m2 = m[2]; B.m3 = m[3]; B.m4 = m[4]; B.m5 = m[5]; B.dyn = 1;
          }
          if (collide_shapes(A, SH_VASE, ca, sa, B, isb ? W.bk.sh : SH_VASE, cb, sb, W.vsz, 0.f, isb ? W.solb : W.sol0)) {
            E.wfb[a][6] = A.ax; E.wfb[a][7] = A.ay; E.wfb[a][8] = A.aw;
            E.wfb[b][6] = B.ax; E.wfb[b][7] = B.ay; E.wfb[b][8] = B.aw;
          }
        }
    }
    __syncthreads();
    if (u < NBODY) {
      const bool isb = u == BOX_ID;
      if (isb ? W.has_box : u < W.nV) {
        float B[9];
        for (int c = 0; c < 9; c++) B[c] = E.wfb[u][c];
        dg_free_body_finish(B, isb, W.vk, W.bk, W.sol0.bcoef, h);
        for (int c = 0; c < 6; c++) E.wfb[u][c] = B[c];
      }
    }
    // ---- robot: semi-implicit Euler --------------------------------------------------------
    if (u < DG_NV) {
      const double v = E.qdv[u] + (double)h * qacc_u;
      if (u < 3) { E.vlin[u] = v; E.pos[u] += (double)h * v; }
      else if (u < 6) E.wloc[u - 3] = v;
      else { E.qd[u - 6] = v; E.q[u - 6] += (double)h * v; }
    }
    __syncthreads();
    if (u == 0) {
      const double wn = sqrt(dg_dot(E.wloc, E.wloc));
      if (wn > 0) {
        double sn, c;
        sincos(0.5 * (double)h * wn, &sn, &c);
        const double s = sn / wn;
        const double dq[4] = {c, s * E.wloc[0], s * E.wloc[1], s * E.wloc[2]};
        const double q0 = E.quat[0], q1 = E.quat[1], q2 = E.quat[2], q3 = E.quat[3];
        const double o[4] = {q0 * dq[0] - q1 * dq[1] - q2 * dq[2] - q3 * dq[3], q0 * dq[1] + q1 * dq[0] + q2 * dq[3] - q3 * dq[2],
                             q0 * dq[2] - q1 * dq[3] + q2 * dq[0] + q3 * dq[1], q0 * dq[3] + q1 * dq[2] - q2 * dq[1] + q3 * dq[0]};
        const double nn = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        for (int k = 0; k < 4; k++) E.quat[k] = o[k] / nn;
      }
      if (E.flag & 1) E.pos[0] = __longlong_as_double(0x7ff8000000000000ll);   // not positive definite: PhysicsError
    }
    __syncthreads();
    DCC(DCY_PLANAR);
  }

  // ---- results: per-env block for the post kernel, state back to HBM ---------------------------
  if (sched && live && u == 0) {   // file the env under its cost class for the next step's order
    const int cls = rows_max > SAG_DC_FAST_ROWS ? 3 : (rows_max > SAG_DC_CLS2 ? 2 : (rows_max > SAG_DC_CLS1 ? 1 : 0));
    const int slot = atomicAdd(&sc[((p.dg_phase + 1) % 3) * DC_NCLS + cls], 1);
    sc[16 + ((size_t)((p.dg_phase + 1) & 1) * DC_NCLS + cls) * N + slot] = (int32_t)gi;
  }
  if (live) {
    double* dr = DR + gi * DR_STRIDE;
    if (u < 3) dr[u] = E.qacc[u];
    if (u < 8) dr[3 + u] = E.touch[u];
    if (u == 0) {
      dr[11] = E.com[0]; dr[12] = E.com[1]; dr[13] = E.mom[0]; dr[14] = E.mom[1];
      dr[15] = (double)cost_contacts; dr[16] = (double)btn_mask; dr[17] = (E.flag & 2) ? 1.0 : 0.0;
    }
    if (!p.observe_only) {
      if (u == 0) {
        DgState D;
        for (int k = 0; k < 3; k++) { D.pos[k] = E.pos[k]; D.vlin[k] = E.vlin[k]; D.wloc[k] = E.wloc[k]; }
        for (int k = 0; k < 4; k++) D.quat[k] = E.quat[k];
        for (int j = 0; j < DG_NJ; j++) { D.q[j] = E.q[j]; D.qd[j] = E.qd[j]; }
        float yaw, wz;
        dg_store(D, S, N, gi, yaw, wz);
      }
      if (u < NBODY) {
        const bool isb = u == BOX_ID;
        if (isb ? W.has_box : u < W.nV)
          for (int c = 0; c < 3; c++) {
            S[((size_t)((DV_POS + 3 * u + c) >> 2) * N + gi) * 4 + ((DV_POS + 3 * u + c) & 3)] = E.wfb[u][c];
            S[((size_t)((DV_VEL + 3 * u + c) >> 2) * N + gi) * 4 + ((DV_VEL + 3 * u + c) & 3)] = E.wfb[u][3 + c];
          }
      }
    }
  }
#ifdef SAG_CYCLES
  DCC(DCY_STORE);
  if (lane == 0) {
    unsigned long long tot = 0;
    for (int k = 0; k < DCY_N; k++) { atomicAdd(&g_cyc[1][k], g_dc_cyc[k]); tot += g_dc_cyc[k]; }
    atomicAdd(&g_cyc[1][CY_N], 1ull);
    atomicAdd(&g_cyc[2][min((int)(tot >> 19), CY_N)], 1ull);   // histogram of the wavefronts' total ticks, buckets of 2^19
  }
#endif
}

}  // namespace sag
