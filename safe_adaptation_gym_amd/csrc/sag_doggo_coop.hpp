// Doggo physics, wave-cooperative form: 32 lanes per env (two envs per wavefront), every working
// array in LDS, the dense linear algebra spread over the lanes.
//
// Same specification as sag_doggo.hpp / DESIGN.md §4 "Doggo" (same row order, same PGS sweeps), other
// mapping onto the machine: the lane-per-env form of sag_doggo.hpp walks every dependent chain
// (Cholesky, 40+ triangular solves, row construction) through private memory at ~100 clocks per
// step, 17 ms per step for 4096 envs.  Here lane i owns dof i (row i of the mass matrix, component
// i of every generalised vector), lane b body b, lane s collision sphere s:
//   kinematics / inertias / RNEA   each lane walks its own root-to-body chain (<= 3 joints)
//   mass matrix                    lane i computes row i from the composite inertias
//   Cholesky                       right-looking, 19 rank-1 steps, lane i updates row i
//   M^-1                           19 right-hand sides at once (lane i: component i of each)
//   constraint rows                the contact's owner lane builds J (sparse) and W = M^-1 J^T;
//                                  all contacts of a phase in parallel
//   PGS                            rows in sequence, J.qacc by a 32-lane butterfly reduction
//   planar world                   body k on lane k for statics / friction / integration
// The generic step kernel then runs with `doggo_post` set: it skips the physics and does reward,
// cost, lidar and the observation from the stored state and the per-env result block.
#pragma once

namespace sag {

constexpr int DC_ROWS = 48;                     // max constraint rows per env
struct DcEnv {                                  // one env's working set in LDS
  double pos[3], quat[4], q[DG_NJ], vlin[3], wloc[3], qd[DG_NJ];
  double R[DG_NB][9], p[DG_NB][3];
  double S[DG_NV][6];                           // [a; l]
  double Ib[DG_NB][10], Ic[DG_NB][10];          // m, mc[3], I[6]
  double F[DG_NV][6];                           // CRBA: Ic[body(j)] S_j ; RNEA: scratch
  double vb[DG_NB][6], ab[DG_NB][6], fb[DG_NB][6];
  double M[DG_NV][DG_NV + 1];                   // mass matrix, then its Cholesky factor (lower); +1: bank padding
  double Minv[DG_NV][DG_NV + 1];
  double bias[DG_NV], tau[DG_NV], qacc[DG_NV], qdv[DG_NV], col[DG_NV];
  double sph[DG_NS][3];
  double touch[8];
  float rJ[DC_ROWS][DG_NV], rW[DC_ROWS][DG_NV];
  double rA[DC_ROWS], rAref[DC_ROWS], rImp[DC_ROWS], rF[DC_ROWS], rMu[DC_ROWS];
  float rOu[DC_ROWS][3], rOd[DC_ROWS][2], rOrx[DC_ROWS], rOry[DC_ROWS];
  short rParent[DC_ROWS], rOther[DC_ROWS], rTouch[DC_ROWS];
  float wfb[NBODY][9], wminv[NBODY][6];         // planar free bodies
  int nrows, flag;
};

// paths of the kinematic tree (static): bodies from the root's child down to b
__device__ inline int dc_path(int b, int* path) {  // returns length; path excludes the root
  int n = 0, tmp[4];
  for (int c = b; c > 0; c = g_dg.parent[c]) tmp[n++] = c;
  for (int k = 0; k < n; k++) path[k] = tmp[n - 1 - k];
  return n;
}
__device__ inline int dc_first_dof(int b) {  // first dof of body b (b >= 1)
  const int T[DG_NB] = {0, 6, 8, 9, 11, 12, 13, 15, 16, 18};
  return T[b];
}
__device__ inline int dc_ndof(int b) {
  const int T[DG_NB] = {6, 2, 1, 2, 1, 1, 2, 1, 2, 1};
  return T[b];
}

// rotation matrix (3x3, registers) times axis-angle rotation
__device__ inline void dc_rot_apply(double* R, const double* axis, double ang) {
  double Rj[9];
  dg_axis_rot(axis, ang, Rj);
  dg_matmul(R, Rj, R);
}

// phase 1: frames, motion vectors, inertias about O = base origin, sphere centres.  `u` = lane in the half.
__device__ inline void dc_kinematics(DcEnv& E, int u) {
  const DgModel& M = g_dg;
  double R0[9];
  dg_quat2mat(E.quat, R0);
  // dof role
  if (u < DG_NV) {
    double a[3] = {0, 0, 0}, l[3] = {0, 0, 0};
    if (u < 3) l[u] = 1;
    else if (u < 6) { a[0] = R0[u - 3]; a[1] = R0[3 + u - 3]; a[2] = R0[6 + u - 3]; }
    else {
      const int b = M.dof_body[u];
      int path[4];
      const int n = dc_path(b, path);
      double R[9], p[3] = {E.pos[0], E.pos[1], E.pos[2]};
      for (int k = 0; k < 9; k++) R[k] = R0[k];
      for (int s = 0; s < n; s++) {
        const int c = path[s];
        double off[3];
        dg_matvec(R, M.bpos[c], off);
        for (int k = 0; k < 3; k++) p[k] += off[k];
        const int d0 = dc_first_dof(c), nd = dc_ndof(c);
        for (int d = d0; d < d0 + nd && d < u; d++) dc_rot_apply(R, M.axis[d - 6], E.q[d - 6]);
      }
      dg_matvec(R, M.axis[u - 6], a);
      const double r[3] = {p[0] - E.pos[0], p[1] - E.pos[1], p[2] - E.pos[2]};
      dg_cross(r, a, l);
    }
    for (int k = 0; k < 3; k++) { E.S[u][k] = a[k]; E.S[u][3 + k] = l[k]; }
  }
  // body role
  if (u < DG_NB) {
    int path[4];
    const int n = dc_path(u, path);
    double R[9], p[3] = {E.pos[0], E.pos[1], E.pos[2]};
    for (int k = 0; k < 9; k++) R[k] = R0[k];
    for (int s = 0; s < n; s++) {
      const int c = path[s];
      double off[3];
      dg_matvec(R, M.bpos[c], off);
      for (int k = 0; k < 3; k++) p[k] += off[k];
      const int d0 = dc_first_dof(c), nd = dc_ndof(c);
      for (int d = d0; d < d0 + nd; d++) dc_rot_apply(R, M.axis[d - 6], E.q[d - 6]);
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
// sphere centres (needs E.R / E.p)
__device__ inline void dc_spheres(DcEnv& E, int u) {
  if (u < DG_NS) {
    const int b = g_dg.sph_body[u];
    double o[3];
    dg_matvec(E.R[b], g_dg.sph_p[u], o);
    for (int k = 0; k < 3; k++) E.sph[u][k] = E.p[b][k] + o[k];
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
__device__ inline void dc_composite(DcEnv& E, int u) {
  if (u < DG_NB) {
    double acc[10];
    for (int k = 0; k < 10; k++) acc[k] = 0;
    for (int d = u; d < DG_NB; d++)
      if (g_dg.anc[d] >> u & 1u)
        for (int k = 0; k < 10; k++) acc[k] += E.Ib[d][k];
    for (int k = 0; k < 10; k++) E.Ic[u][k] = acc[k];
  }
}
__device__ inline void dc_crba_f(DcEnv& E, int u) {
  if (u < DG_NV) dc_inertia_apply(E.Ic[g_dg.dof_body[u]], E.S[u], E.F[u]);
}
__device__ inline void dc_crba_rows(DcEnv& E, int u) {
  if (u < DG_NV) {
    const int bi = g_dg.dof_body[u];
    for (int j = 0; j < DG_NV; j++) {
      const int bj = g_dg.dof_body[j];
      double v = 0;
      if ((g_dg.anc[bj] >> bi & 1u) && (bi != bj || u <= j)) v = dc_dot6(E.S[u], E.F[j]);       // i above (or beside, i <= j)
      else if (g_dg.anc[bi] >> bj & 1u) v = dc_dot6(E.S[j], E.F[u]);                           // j above i
      E.M[u][j] = v;
    }
  }
}
// body velocities / accelerations (qacc = 0) and the body wrenches
__device__ inline void dc_rnea_bodies(DcEnv& E, int u) {
  if (u < DG_NB) {
    double v[6], a[6];
    for (int k = 0; k < 3; k++) { v[k] = 0; v[3 + k] = E.vlin[k]; a[k] = 0; }
    for (int k = 0; k < 3; k++)
      for (int c = 0; c < 3; c++) v[c] += E.S[3 + k][c] * E.wloc[k];
    dg_cross(E.vlin, v, a + 3);
    a[5] += DG_GRAV;
    int path[4];
    const int n = dc_path(u, path);
    for (int s = 0; s < n; s++) {
      const int c = path[s], d0 = dc_first_dof(c), nd = dc_ndof(c);
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
__device__ inline void dc_rnea_bias(DcEnv& E, int u) {
  if (u < DG_NV) {
    const int b = g_dg.dof_body[u];
    double F[6] = {0, 0, 0, 0, 0, 0};
    for (int d = b; d < DG_NB; d++)
      if (g_dg.anc[d] >> b & 1u)
        for (int k = 0; k < 6; k++) F[k] += E.fb[d][k];
    E.bias[u] = dc_dot6(E.S[u], F);
  }
}

// Cholesky of E.M in place (lower), right-looking; lane i updates row i.  Barriers inside.
__device__ inline void dc_cholesky(DcEnv& E, int u) {
#pragma unroll 1
  for (int j = 0; j < DG_NV; j++) {
    if (u == j) {
      const double d = E.M[j][j];
      if (!(d > 0)) E.flag = 1;
      E.M[j][j] = sqrt(d);
    }
    __syncthreads();
    if (u > j && u < DG_NV) E.M[u][j] = E.M[u][j] / E.M[j][j];
    __syncthreads();
    if (u > j && u < DG_NV) {
      const double lij = E.M[u][j];
      for (int k = j + 1; k <= u; k++) E.M[u][k] -= lij * E.M[k][j];
    }
    __syncthreads();
  }
}
// x <- (L L^T)^-1 b, component i of b / x in lane i's register `x`; E.col is the broadcast slot
__device__ inline double dc_solve1(DcEnv& E, int u, double x) {
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
// M^-1: 19 right-hand sides (identity), lane i holds component i of each; result row i -> E.Minv[i][:]
__device__ inline void dc_inverse(DcEnv& E, int u) {
  double x[DG_NV];
#pragma unroll
  for (int r = 0; r < DG_NV; r++) x[r] = (u == r) ? 1.0 : 0.0;
#pragma unroll 1
  for (int j = 0; j < DG_NV; j++) {
    if (u == j) {
      const double inv = 1.0 / E.M[j][j];
#pragma unroll
      for (int r = 0; r < DG_NV; r++) { x[r] *= inv; E.Minv[0][r] = x[r]; }   // row 0 of Minv doubles as the broadcast buffer
    }
    __syncthreads();
    if (u > j && u < DG_NV) {
      const double l = E.M[u][j];
#pragma unroll
      for (int r = 0; r < DG_NV; r++) x[r] -= l * E.Minv[0][r];
    }
    __syncthreads();
  }
#pragma unroll 1
  for (int j = DG_NV - 1; j >= 0; j--) {
    if (u == j) {
      const double inv = 1.0 / E.M[j][j];
#pragma unroll
      for (int r = 0; r < DG_NV; r++) { x[r] *= inv; E.Minv[0][r] = x[r]; }
    }
    __syncthreads();
    if (u < j) {
      const double l = E.M[j][u];
#pragma unroll
      for (int r = 0; r < DG_NV; r++) x[r] -= l * E.Minv[0][r];
    }
    __syncthreads();
  }
  if (u < DG_NV) {
#pragma unroll
    for (int r = 0; r < DG_NV; r++) E.Minv[u][r] = x[r];
  }
  __syncthreads();
}

}  // namespace sag

namespace sag {

// load the env's robot state into LDS (lane 0 of the half), zero flags
__device__ inline void dc_load_state(DcEnv& E, int u, const float* S, size_t N, size_t i) {
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
__device__ inline void dc_smooth(DcEnv& E, int u, const float* ctrl12) {
  dc_kinematics(E, u);
  __syncthreads();
  dc_spheres(E, u);
  dc_composite(E, u);
  dc_rnea_bodies(E, u);
  __syncthreads();
  dc_crba_f(E, u);
  dc_rnea_bias(E, u);
  __syncthreads();
  dc_crba_rows(E, u);
  if (u < DG_NV) {
    double t = 0;
    if (u >= 6) {
      const int j = u - 6;
      t = -DG_STIFF * (E.q[j] - g_dg.springref[j]);
      if (ctrl12)
        for (int k = 0; k < 12; k++)
          if (g_dg.act_joint[k] == j) t += DG_GEAR * (double)ctrl12[k];
    }
    E.tau[u] = t - E.bias[u];
    E.qdv[u] = u < 3 ? E.vlin[u] : (u < 6 ? E.wloc[u - 3] : E.qd[u - 6]);
  }
  __syncthreads();
}

// debug: mass matrix [19x19], bias [19], qacc0 [19], Minv [19x19] per env -> out[N][19*19*2 + 38]
__global__ __launch_bounds__(64) void k_doggo_coop_debug(const float* __restrict__ S, int N, double* __restrict__ out) {
  __shared__ DcEnv env[2];
  const int lane = threadIdx.x, half = lane >> 5, u = lane & 31;
  const size_t i = (size_t)blockIdx.x * 2 + half;
  const bool live = i < (size_t)N;
  DcEnv& E = env[half];
  dc_load_state(E, u, S, (size_t)N, live ? i : (size_t)N - 1);
  __syncthreads();
  dc_smooth(E, u, nullptr);
  double* o = out + (live ? i : 0) * (size_t)(2 * DG_NV * DG_NV + 2 * DG_NV);
  if (live && u < DG_NV) {
    for (int j = 0; j < DG_NV; j++) o[u * DG_NV + j] = E.M[u][j];
    o[DG_NV * DG_NV + u] = E.bias[u];
  }
  __syncthreads();
  dc_cholesky(E, u);
  const double x = dc_solve1(E, u, u < DG_NV ? E.tau[u] : 0.0);
  dc_inverse(E, u);
  if (live && u < DG_NV) {
    o[DG_NV * DG_NV + DG_NV + u] = x;
    for (int j = 0; j < DG_NV; j++) o[DG_NV * DG_NV + 2 * DG_NV + u * DG_NV + j] = E.Minv[u][j];
  }
}

}  // namespace sag
