// Doggo (assets/xmls/doggo.xml): 3-D articulated quadruped, one env per lane.
//
// Specification: DESIGN.md §4 "Doggo" (the CPU restatement used by the tests states the same model):
// free base + waist hinge + 4 x (hip_z, hip_y, ankle); composite-rigid-body mass matrix and
// recursive Newton-Euler bias in world axes about the base origin; dense Cholesky; joint limits,
// floor and object contacts as MuJoCo-style soft constraints solved by projected Gauss-Seidel;
// 17 collision spheres; horizontal contacts against the extruded planar world.
//
// Arithmetic: fp64.  The mass matrix spans base mass 4e-2 kg to ankle inertia 1e-5 kg m^2 and the
// contact rows need M^-1 J^T: in fp32 the Cholesky solve loses 4-5 of its 7 digits.  State in HBM
// stays fp32 like every other field.  Working set per lane (~15 KB: factor, motion vectors, the
// constraint rows) lives in private (scratch) memory; at the batch sizes Doggo is run at
// (BASELINE config 4: 4096 envs per GPU = 64 wavefronts) the kernel is latency-bound anyway.
#pragma once

namespace sag {

constexpr int DG_NB = 10, DG_NV = 19, DG_NJ = 13, DG_NS = 17, DG_MAXROWS = 3 * 12 + DG_NJ + 1, DG_NGEOM = 14;
constexpr int DG_PGS_ITERS = 4;
constexpr double DG_GEAR = 0.0125, DG_STIFF = 0.01, DG_Z0 = 0.22, DG_PI = 3.14159265358979323846, DG_GRAV = 9.81;

// model tables, built on the host in fp64 (sag_api.hip: dg_build_model) from the XML numbers
struct DgPhys {                    // what the dynamics read (the cooperative kernel keeps a copy in LDS)
  int parent[DG_NB];
  double bpos[DG_NB][3];
  int dof_body[DG_NV];
  double axis[DG_NJ][3];          // normalised
  double lo[DG_NJ], hi[DG_NJ], springref[DG_NJ];
  int act_joint[12];
  double m[DG_NB], com[DG_NB][3], I[DG_NB][9];  // inertia about the com, body axes
  int sph_body[DG_NS], sph_touch[DG_NS];
  double sph_p[DG_NS][3], sph_r[DG_NS];
  unsigned anc[DG_NB];            // bit a set: body a is on the path root -> b (inclusive)
};
struct DgModel : DgPhys {
  // the XML's geoms (rendering): body, end points in the body frame, radius, capsule (1) / cylinder (0),
  // ankle colour class (0 default red, 1 front ankles blue, 2 rear ankles green: doggo.xml:26,40,58,72)
  int geom_body[DG_NGEOM], geom_capsule[DG_NGEOM], geom_ankle[DG_NGEOM];
  double geom_a[DG_NGEOM][3], geom_b[DG_NGEOM][3], geom_r[DG_NGEOM];
};
// one copy per translation unit (sag_api.hip: cooperative form + post kernel; sag_doggo_lane.hip: lane-per-env
// form); sag_create uploads the model to both
static __constant__ DgModel g_dg;

// LDS pointers keep their address space through the (non-inlined) function boundaries: a generic
// `double*` turns every access into a flat load that takes the slow path to the LDS aperture
typedef __attribute__((address_space(3))) double lds_f64;

struct DgState {
  double pos[3], quat[4], q[DG_NJ];
  double vlin[3], wloc[3], qd[DG_NJ];
};
struct DgVec { double a[3], l[3]; };
struct DgInertia { double m, mc[3], I[6]; };  // about O, world axes; I: xx xy xz yy yz zz

struct DgWork {
  double R[DG_NB][9], p[DG_NB][3];
  DgVec S[DG_NV];
  DgInertia Ib[DG_NB];
  lds_f64* L;                         // LDS, [packed lower index][lane]: mass matrix, then its Cholesky factor
  lds_f64* xs;                        // LDS, [3][19][lane]: right-hand sides / solutions of a solve
  lds_f64* dinv;                      // LDS, [19][lane]: 1 / diagonal of the factor
  double Mlin[2][DG_NV];              // rows 0, 1 of the mass matrix (momentum, Unsupervised)
  double qacc[DG_NV];
  double sph[DG_NS][3];
  double touch[8];
};

// one scalar constraint row
struct DgRow {
  double W[DG_NV];
  float J[DG_NV];
  double A, aref, imp, f, mu;
  float ou[3], od[2], orx, ory;  // other (planar) body: Minv J^T, direction, lever
  short parent, other, touch;    // normal row of a friction row / free-body index / touch slot; -1 = none
};

constexpr int DG_NTRI = DG_NV * (DG_NV + 1) / 2;
// packed lower triangle, one column of 64 lanes per entry (bank = lane: conflict-free); the
// pointer handed around is already offset by the lane
__device__ __host__ constexpr int dg_tri(int i, int j) { return (i * (i + 1) / 2 + j) * 64; }  // i >= j
__device__ inline void dg_cross(const double* a, const double* b, double* o) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ inline double dg_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline void dg_matvec(const double* R, const double* v, double* o) {
  const double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
               z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ inline void dg_matmul(const double* A, const double* B, double* O) {
  double T[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) T[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
#pragma unroll
  for (int k = 0; k < 9; k++) O[k] = T[k];
}
__device__ inline void dg_quat2mat(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
__device__ inline void dg_axis_rot(const double* u, double ang, double* R) {
  const double c = cos(ang), s = sin(ang), t = 1 - c;
  R[0] = c + t * u[0] * u[0]; R[1] = t * u[0] * u[1] - s * u[2]; R[2] = t * u[0] * u[2] + s * u[1];
  R[3] = t * u[0] * u[1] + s * u[2]; R[4] = c + t * u[1] * u[1]; R[5] = t * u[1] * u[2] - s * u[0];
  R[6] = t * u[0] * u[2] - s * u[1]; R[7] = t * u[1] * u[2] + s * u[0]; R[8] = c + t * u[2] * u[2];
}

// record fields -> state.  A zero quaternion means "upright at ROBOT yaw, z = .22, joints 0" (sag.h)
__device__ inline void dg_load(DgState& D, const float* S, size_t N, size_t i) {
  auto F = [&](int k) { return (double)S[saddr(k, N, i)]; };
  D.pos[0] = F(SAG_F_ROBOT); D.pos[1] = F(SAG_F_ROBOT + 1); D.pos[2] = F(SAG_F_ROBOT_EXT);
  for (int k = 0; k < 4; k++) D.quat[k] = F(SAG_F_ROBOT_EXT + 1 + k);
  const double n2 = D.quat[0] * D.quat[0] + D.quat[1] * D.quat[1] + D.quat[2] * D.quat[2] + D.quat[3] * D.quat[3];
  const bool fresh = n2 < 0.25;
  if (fresh) {
    const double yaw = F(SAG_F_ROBOT + 2);
    D.pos[2] = DG_Z0;
    D.quat[0] = cos(0.5 * yaw); D.quat[1] = 0; D.quat[2] = 0; D.quat[3] = sin(0.5 * yaw);
  }
  D.vlin[0] = F(SAG_F_ROBOT + 3); D.vlin[1] = F(SAG_F_ROBOT + 4); D.vlin[2] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 5);
  for (int k = 0; k < 3; k++) D.wloc[k] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 6 + k);
  for (int j = 0; j < DG_NJ; j++) {
    D.q[j] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 9 + j);
    D.qd[j] = fresh ? 0.0 : F(SAG_F_ROBOT_EXT + 22 + j);
  }
}
__device__ inline void dg_store(const DgState& D, float* S, size_t N, size_t i, float& yaw_out, float& wz_out) {
  double R[9], ww[3];
  dg_quat2mat(D.quat, R);
  dg_matvec(R, D.wloc, ww);
  auto W = [&](int k, double v) { S[saddr(k, N, i)] = (float)v; };
  yaw_out = (float)atan2(R[3], R[0]); wz_out = (float)ww[2];
  W(SAG_F_ROBOT, D.pos[0]); W(SAG_F_ROBOT + 1, D.pos[1]); W(SAG_F_ROBOT + 2, yaw_out);
  W(SAG_F_ROBOT + 3, D.vlin[0]); W(SAG_F_ROBOT + 4, D.vlin[1]); W(SAG_F_ROBOT + 5, wz_out);
  W(SAG_F_ROBOT_EXT, D.pos[2]);
  for (int k = 0; k < 4; k++) W(SAG_F_ROBOT_EXT + 1 + k, D.quat[k]);
  W(SAG_F_ROBOT_EXT + 5, D.vlin[2]);
  for (int k = 0; k < 3; k++) W(SAG_F_ROBOT_EXT + 6 + k, D.wloc[k]);
  for (int j = 0; j < DG_NJ; j++) { W(SAG_F_ROBOT_EXT + 9 + j, D.q[j]); W(SAG_F_ROBOT_EXT + 22 + j, D.qd[j]); }
}

__device__ inline void dg_qd(const DgState& D, double* qd) {
  for (int k = 0; k < 3; k++) { qd[k] = D.vlin[k]; qd[3 + k] = D.wloc[k]; }
  for (int j = 0; j < DG_NJ; j++) qd[6 + j] = D.qd[j];
}

__device__ __attribute__((noinline)) void dg_kinematics(const DgState& D, DgWork& K) {
  const DgModel& M = g_dg;
  dg_quat2mat(D.quat, K.R[0]);
  for (int k = 0; k < 3; k++) K.p[0][k] = D.pos[k];
  for (int i = 0; i < DG_NV; i++)
    for (int k = 0; k < 3; k++) { K.S[i].a[k] = 0; K.S[i].l[k] = 0; }
  for (int k = 0; k < 3; k++) {
    K.S[k].l[k] = 1;
    const double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
    dg_matvec(K.R[0], e, K.S[3 + k].a);
  }
  int dof = 6;
#pragma unroll 1
  for (int b = 1; b < DG_NB; b++) {
    const int P = M.parent[b];
    double off[3];
    dg_matvec(K.R[P], M.bpos[b], off);
    for (int k = 0; k < 3; k++) K.p[b][k] = K.p[P][k] + off[k];
    for (int k = 0; k < 9; k++) K.R[b][k] = K.R[P][k];
    while (dof < DG_NV && M.dof_body[dof] == b) {
      const int j = dof - 6;
      double Rj[9], r[3];
      dg_matvec(K.R[b], M.axis[j], K.S[dof].a);
      for (int k = 0; k < 3; k++) r[k] = K.p[b][k] - D.pos[k];
      dg_cross(r, K.S[dof].a, K.S[dof].l);
      dg_axis_rot(M.axis[j], D.q[j], Rj);
      dg_matmul(K.R[b], Rj, K.R[b]);
      dof++;
    }
  }
#pragma unroll 1
  for (int b = 0; b < DG_NB; b++) {
    DgInertia& I = K.Ib[b];
    double c[3], T[9], Rt[9], Iw[9];
    dg_matvec(K.R[b], M.com[b], c);
    for (int k = 0; k < 3; k++) c[k] += K.p[b][k] - D.pos[k];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) Rt[3 * r + cc] = K.R[b][3 * cc + r];
    dg_matmul(K.R[b], M.I[b], T);
    dg_matmul(T, Rt, Iw);
    const double d2 = dg_dot(c, c), m = M.m[b];
    I.I[0] = Iw[0] + m * (d2 - c[0] * c[0]); I.I[1] = Iw[1] - m * c[0] * c[1]; I.I[2] = Iw[2] - m * c[0] * c[2];
    I.I[3] = Iw[4] + m * (d2 - c[1] * c[1]); I.I[4] = Iw[5] - m * c[1] * c[2]; I.I[5] = Iw[8] + m * (d2 - c[2] * c[2]);
    I.m = m;
    for (int k = 0; k < 3; k++) I.mc[k] = m * c[k];
  }
#pragma unroll 1
  for (int s = 0; s < DG_NS; s++) {
    double o[3];
    dg_matvec(K.R[M.sph_body[s]], M.sph_p[s], o);
    for (int k = 0; k < 3; k++) K.sph[s][k] = K.p[M.sph_body[s]][k] + o[k];
  }
}

__device__ inline void dg_inertia_apply(const DgInertia& I, const DgVec& v, DgVec& f) {
  double t[3];
  dg_cross(v.a, I.mc, t);
  for (int k = 0; k < 3; k++) f.l[k] = I.m * v.l[k] + t[k];
  const double n0 = I.I[0] * v.a[0] + I.I[1] * v.a[1] + I.I[2] * v.a[2],
               n1 = I.I[1] * v.a[0] + I.I[3] * v.a[1] + I.I[4] * v.a[2],
               n2 = I.I[2] * v.a[0] + I.I[4] * v.a[1] + I.I[5] * v.a[2];
  dg_cross(I.mc, v.l, t);
  f.a[0] = n0 + t[0]; f.a[1] = n1 + t[1]; f.a[2] = n2 + t[2];
}
__device__ inline double dg_sdot(const DgVec& s, const DgVec& f) { return dg_dot(s.a, f.a) + dg_dot(s.l, f.l); }

__device__ __attribute__((noinline)) void dg_mass_matrix(DgWork& K) {
  const DgModel& M = g_dg;
  DgInertia Ic[DG_NB];
  for (int b = 0; b < DG_NB; b++) Ic[b] = K.Ib[b];
#pragma unroll 1
  for (int b = DG_NB - 1; b >= 1; b--) {
    DgInertia& P = Ic[M.parent[b]];
    P.m += Ic[b].m;
    for (int k = 0; k < 3; k++) P.mc[k] += Ic[b].mc[k];
    for (int k = 0; k < 6; k++) P.I[k] += Ic[b].I[k];
  }
#pragma unroll 1
  for (int j = 0; j < DG_NV; j++) {
    DgVec F;
    const int bj = M.dof_body[j];
    dg_inertia_apply(Ic[bj], K.S[j], F);
    for (int i = 0; i <= j; i++) {
      const bool on = M.anc[bj] >> M.dof_body[i] & 1u;
      const double v = on ? dg_sdot(K.S[i], F) : 0.0;
      K.L[dg_tri(j, i)] = v;  // (runtime indices: a computed LDS address)
      if (i < 2) K.Mlin[i][j] = v;
    }
  }
  K.Mlin[1][0] = 0;
}

__device__ inline void dg_mcross(const DgVec& v, const DgVec& s, DgVec& o) {
  double t1[3], t2[3];
  dg_cross(v.a, s.a, o.a);
  dg_cross(v.a, s.l, t1);
  dg_cross(v.l, s.a, t2);
  for (int k = 0; k < 3; k++) o.l[k] = t1[k] + t2[k];
}

// bias forces c(q, qdot) including gravity: recursive Newton-Euler with qacc = 0
__device__ __attribute__((noinline)) void dg_bias(const DgState& D, const DgWork& K, double* bias) {
  const DgModel& M = g_dg;
  DgVec v[DG_NB], a[DG_NB], F[DG_NB];
  double qd[DG_NV];
  dg_qd(D, qd);
  for (int k = 0; k < 3; k++) { v[0].l[k] = D.vlin[k]; v[0].a[k] = 0; a[0].a[k] = 0; }
  for (int k = 0; k < 3; k++)
    for (int c = 0; c < 3; c++) v[0].a[c] += K.S[3 + k].a[c] * D.wloc[k];
  dg_cross(D.vlin, v[0].a, a[0].l);
  a[0].l[2] += DG_GRAV;
  int dof = 6;
#pragma unroll 1
  for (int b = 1; b < DG_NB; b++) {
    DgVec u = v[M.parent[b]], acc = a[M.parent[b]];
    while (dof < DG_NV && M.dof_body[dof] == b) {
      DgVec sd;
      dg_mcross(u, K.S[dof], sd);
      for (int k = 0; k < 3; k++) {
        acc.a[k] += sd.a[k] * qd[dof]; acc.l[k] += sd.l[k] * qd[dof];
        u.a[k] += K.S[dof].a[k] * qd[dof]; u.l[k] += K.S[dof].l[k] * qd[dof];
      }
      dof++;
    }
    v[b] = u; a[b] = acc;
  }
#pragma unroll 1
  for (int b = 0; b < DG_NB; b++) {
    DgVec h;
    double t1[3], t2[3], t3[3];
    dg_inertia_apply(K.Ib[b], a[b], F[b]);
    dg_inertia_apply(K.Ib[b], v[b], h);
    dg_cross(v[b].a, h.a, t1); dg_cross(v[b].l, h.l, t2); dg_cross(v[b].a, h.l, t3);
    for (int k = 0; k < 3; k++) { F[b].a[k] += t1[k] + t2[k]; F[b].l[k] += t3[k]; }
  }
#pragma unroll 1
  for (int b = DG_NB - 1; b >= 0; b--) {
    for (int i = 0; i < DG_NV; i++)
      if (M.dof_body[i] == b) bias[i] = dg_sdot(K.S[i], F[b]);
    if (b > 0)
      for (int k = 0; k < 3; k++) { F[M.parent[b]].a[k] += F[b].a[k]; F[M.parent[b]].l[k] += F[b].l[k]; }
  }
}

// Cholesky of the 19x19 mass matrix in place.  Factor, inverse diagonal and the solve vectors live
// in LDS, where a run-time index is just an address: rolled loops, little code.  (Private-memory
// arrays made the row solves 76 % of the step; a fully unrolled register version was 60 KB of code
// that thrashed the instruction cache.)  LDS loads are issued in batches of 4 before they are used,
// otherwise every multiply-add waits a full LDS round trip.
__device__ __attribute__((noinline)) bool dg_cholesky(lds_f64* L, lds_f64* dinv) {
  bool ok = true;
#pragma unroll 1
  for (int j = 0; j < DG_NV; j++) {
    double d = L[dg_tri(j, j)];
    {
      int k = 0;
      for (; k + 4 <= j; k += 4) {
        double l[4];
#pragma unroll
        for (int u = 0; u < 4; u++) l[u] = L[dg_tri(j, k + u)];
#pragma unroll
        for (int u = 0; u < 4; u++) d -= l[u] * l[u];
      }
      for (; k < j; k++) { const double l = L[dg_tri(j, k)]; d -= l * l; }
    }
    ok = ok && d > 0;
    d = sqrt(d);
    L[dg_tri(j, j)] = d;
    const double inv = 1.0 / d;
    dinv[j * 64] = inv;
#pragma unroll 1
    for (int i = j + 1; i < DG_NV; i++) {
      double s = L[dg_tri(i, j)];
      int k = 0;
      for (; k + 4 <= j; k += 4) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = L[dg_tri(i, k + u)]; b[u] = L[dg_tri(j, k + u)]; }
#pragma unroll
        for (int u = 0; u < 4; u++) s -= a[u] * b[u];
      }
      for (; k < j; k++) s -= L[dg_tri(i, k)] * L[dg_tri(j, k)];
      L[dg_tri(i, j)] = s * inv;
    }
  }
  return ok;
}
// NR right-hand sides at once: x[(r * 19 + i) * 64] <- (L L^T)^-1 x.  The three rows of a contact
// (normal, two tangents) share every load of the factor.
template <int NR>
__device__ __attribute__((noinline)) void dg_solve_lds(const lds_f64* L, const lds_f64* dinv, lds_f64* x) {
#pragma unroll 1
  for (int i = 0; i < DG_NV; i++) {
    double s[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) s[r] = x[(r * DG_NV + i) * 64];
    int k = 0;
    for (; k + 4 <= i; k += 4) {
      double l[4], xv[NR][4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        l[u] = L[dg_tri(i, k + u)];
#pragma unroll
        for (int r = 0; r < NR; r++) xv[r][u] = x[(r * DG_NV + k + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < NR; r++) s[r] -= l[u] * xv[r][u];
    }
    for (; k < i; k++) {
      const double l = L[dg_tri(i, k)];
#pragma unroll
      for (int r = 0; r < NR; r++) s[r] -= l * x[(r * DG_NV + k) * 64];
    }
    const double inv = dinv[i * 64];
#pragma unroll
    for (int r = 0; r < NR; r++) x[(r * DG_NV + i) * 64] = s[r] * inv;
  }
#pragma unroll 1
  for (int i = DG_NV - 1; i >= 0; i--) {
    double s[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) s[r] = x[(r * DG_NV + i) * 64];
    int k = i + 1;
    for (; k + 4 <= DG_NV; k += 4) {
      double l[4], xv[NR][4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        l[u] = L[dg_tri(k + u, i)];
#pragma unroll
        for (int r = 0; r < NR; r++) xv[r][u] = x[(r * DG_NV + k + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < NR; r++) s[r] -= l[u] * xv[r][u];
    }
    for (; k < DG_NV; k++) {
      const double l = L[dg_tri(k, i)];
#pragma unroll
      for (int r = 0; r < NR; r++) s[r] -= l * x[(r * DG_NV + k) * 64];
    }
    const double inv = dinv[i * 64];
#pragma unroll
    for (int r = 0; r < NR; r++) x[(r * DG_NV + i) * 64] = s[r] * inv;
  }
}

// Jacobian row of point c (world) on body b along direction d
__device__ inline void dg_jac(const DgState& D, const DgWork& K, int b, const double* c, const double* d, double* J) {
  const DgModel& M = g_dg;
  const double r[3] = {c[0] - D.pos[0], c[1] - D.pos[1], c[2] - D.pos[2]};
#pragma unroll 1
  for (int i = 0; i < DG_NV; i++) {
    double v = 0;
    if (M.anc[b] >> M.dof_body[i] & 1u) {
      double t[3];
      dg_cross(K.S[i].a, r, t);
      v = d[0] * (K.S[i].l[0] + t[0]) + d[1] * (K.S[i].l[1] + t[1]) + d[2] * (K.S[i].l[2] + t[2]);
    }
    J[i] = v;
  }
}

__device__ inline double dg_impedance(double depth) {
  double x = depth / (double)SOL_WIDTH; if (x > 1) x = 1;
  const double y = x < 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);
  return (double)SOL_D0 + ((double)SOL_D1 - (double)SOL_D0) * y;
}

// planar free bodies as the Doggo code sees them: [x y yaw vx vy w ax ay aw] x NBODY (fp32, the
// arithmetic of the planar world) + inverse inertia
struct DgWorld {
  float fb[NBODY][9];
  float minv[NBODY][6];
  uint32_t valid;   // free bodies that exist
};

struct DgRows { DgRow row[DG_MAXROWS]; int n; int overflow; };  // overflow: a row did not fit (reported as record flag bit 2)

// Row from right-hand-side slot `slot` of K.xs: J (as stored, fp32) was written there and solved in
// place, so the slot now holds W = M^-1 J^T.  The other body (if any) moves along (dx, dy) at lever
// (px, py) - its centre.
__device__ __attribute__((noinline)) DgRow* dg_add_row(DgRows& R, const DgWork& K, const DgWorld& Wd, const float* Jf,
                                                       int slot, int other, double dx, double dy, double px, double py) {
  if (R.n >= DG_MAXROWS) { R.overflow = 1; return nullptr; }
  DgRow& r = R.row[R.n++];
  double A = 0;
  for (int i = 0; i < DG_NV; i++) {
    r.J[i] = Jf[i];
    r.W[i] = K.xs[(slot * DG_NV + i) * 64];
    A += (double)r.J[i] * r.W[i];
  }
  r.other = (short)other; r.parent = -1; r.touch = -1; r.f = 0; r.mu = 0;
  r.ou[0] = r.ou[1] = r.ou[2] = 0; r.od[0] = r.od[1] = 0; r.orx = r.ory = 0;
  if (other >= 0) {
    const float* m = Wd.minv[other];
    r.orx = (float)(px - (double)Wd.fb[other][0]); r.ory = (float)(py - (double)Wd.fb[other][1]);
    r.od[0] = (float)dx; r.od[1] = (float)dy;
    const float rxd = r.orx * r.od[1] - r.ory * r.od[0];
    r.ou[0] = m[0] * r.od[0] + m[1] * r.od[1] + m[2] * rxd;
    r.ou[1] = m[1] * r.od[0] + m[3] * r.od[1] + m[4] * rxd;
    r.ou[2] = m[2] * r.od[0] + m[4] * r.od[1] + m[5] * rxd;
    A += (double)(r.od[0] * r.ou[0] + r.od[1] * r.ou[1] + rxd * r.ou[2]);
  }
  r.A = A;
  return &r;
}
// one row: the fp64 Jacobian is rounded to fp32 for storage and W is computed from the stored
// row, so that A = J W stays consistent
__device__ inline DgRow* dg_add_row1(DgRows& R, const DgWork& K, const DgWorld& Wd, const double* J, double sign,
                                     int other, double dx, double dy, double px, double py) {
  float Jf[DG_NV];
  for (int i = 0; i < DG_NV; i++) { Jf[i] = (float)(sign * J[i]); K.xs[i * 64] = (double)Jf[i]; }
  dg_solve_lds<1>(K.L, K.dinv, K.xs);
  return dg_add_row(R, K, Wd, Jf, 0, other, dx, dy, px, py);
}
__device__ inline double dg_row_vel(const DgRow& r, const DgWorld& Wd, const double* qd) {
  double v = 0;
  for (int i = 0; i < DG_NV; i++) v += (double)r.J[i] * qd[i];
  if (r.other >= 0) {
    const float* B = Wd.fb[r.other];
    v += (double)((B[3] - B[5] * r.ory) * r.od[0] + (B[4] + B[5] * r.orx) * r.od[1]);
  }
  return v;
}
__device__ inline double dg_row_acc(const DgRow& r, const DgWorld& Wd, const double* qacc) {
  double a = 0;
  for (int i = 0; i < DG_NV; i++) a += (double)r.J[i] * qacc[i];
  if (r.other >= 0) {
    const float* B = Wd.fb[r.other];
    a += (double)((B[6] - B[8] * r.ory) * r.od[0] + (B[7] + B[8] * r.orx) * r.od[1]);
  }
  return a;
}

// contact of sphere s: normal n points INTO the robot, point c, penetration depth
__device__ __attribute__((noinline)) void dg_add_contact(DgRows& R, const DgState& D, const DgWork& K, const DgWorld& Wd,
                                                         const double* qd, int s, const double* n, const double* c,
                                                         double depth, int other, double bcoef, double kcoef, double mu) {
  if (R.n + 3 > DG_MAXROWS) { R.overflow = 1; return; }
  const int b = g_dg.sph_body[s];
  const int first = R.n;
  double dir[3][3];
  for (int k = 0; k < 3; k++) dir[0][k] = n[k];
  if (fabs(n[2]) > 0.5) { dir[1][0] = 1; dir[1][1] = 0; dir[1][2] = 0; }
  else { dir[1][0] = -n[1]; dir[1][1] = n[0]; dir[1][2] = 0; }
  dg_cross(dir[0], dir[1], dir[2]);
  float Jf[3][DG_NV];
#pragma unroll 1
  for (int k = 0; k < 3; k++) {
    double J[DG_NV];
    dg_jac(D, K, b, c, dir[k], J);
    for (int i = 0; i < DG_NV; i++) { Jf[k][i] = (float)J[i]; K.xs[(k * DG_NV + i) * 64] = (double)Jf[k][i]; }
  }
  dg_solve_lds<3>(K.L, K.dinv, K.xs);   // normal and both tangents share the factor loads
#pragma unroll 1
  for (int k = 0; k < 3; k++) {
    DgRow* r = dg_add_row(R, K, Wd, Jf[k], k, other, -dir[k][0], -dir[k][1], c[0], c[1]);
    if (k == 0) {
      r->imp = dg_impedance(depth);
      r->aref = -bcoef * dg_row_vel(*r, Wd, qd) + kcoef * depth;
      r->touch = (short)g_dg.sph_touch[s];
    } else {
      r->imp = R.row[first].imp;
      r->aref = -bcoef * dg_row_vel(*r, Wd, qd);
      r->parent = (short)first; r->mu = mu;
    }
  }
}

__device__ __attribute__((noinline)) void dg_pgs(DgRows& R, DgWork& K, DgWorld& Wd) {
  double qa[DG_NV];  // registers: every inner loop below is fully unrolled
#pragma unroll
  for (int i = 0; i < DG_NV; i++) qa[i] = K.qacc[i];
#pragma unroll 1
  for (int it = 0; it < DG_PGS_ITERS; it++)
#pragma unroll 1
    for (int k = 0; k < R.n; k++) {
      DgRow& r = R.row[k];
      if (!(r.A > 0)) continue;
      const double reg = r.A * (1 - r.imp) / r.imp;
      double acc = 0;
#pragma unroll
      for (int i = 0; i < DG_NV; i++) acc += (double)r.J[i] * qa[i];
      if (r.other >= 0) {
        const float* B = Wd.fb[r.other];
        acc += (double)((B[6] - B[8] * r.ory) * r.od[0] + (B[7] + B[8] * r.orx) * r.od[1]);
      }
      double fnew = r.f + (r.aref - acc - reg * r.f) / (r.A + reg);
      double lo = 0, hi = 1e30;
      if (r.parent >= 0) { const double fn = R.row[r.parent].f; lo = -r.mu * fn; hi = r.mu * fn; }
      if (fnew < lo) fnew = lo;
      if (fnew > hi) fnew = hi;
      const double df = fnew - r.f;
      if (df == 0) continue;
      r.f = fnew;
#pragma unroll
      for (int i = 0; i < DG_NV; i++) qa[i] += r.W[i] * df;
      if (r.other >= 0) {
        float* B = Wd.fb[r.other];
        B[6] += (float)((double)r.ou[0] * df); B[7] += (float)((double)r.ou[1] * df); B[8] += (float)((double)r.ou[2] * df);
      }
    }
#pragma unroll
  for (int i = 0; i < DG_NV; i++) K.qacc[i] = qa[i];
}

// circle (sphere footprint) vs the geoms of one planar body: contacts with normal from the circle
// (A) to the body (B); conventions of the planar narrowphase (cc_contact / cb_contact above)
struct DgHit { float nx, ny, px, py, depth; };
__device__ inline int dg_circle_geom(float ax, float ay, float ra, const Geom& g, float bx, float by, float cb,
                                     float sb, DgHit& h) {
  if (!g.box) {
    const float dx = bx - ax, dy = by - ay, d2 = dx * dx + dy * dy, rs = ra + g.a;
    if (d2 >= rs * rs) return 0;
    const float d = sqrtf(d2);
    float nx = 1, ny = 0;
    if (d > 1e-12f) { nx = dx / d; ny = dy / d; }
    h.nx = nx; h.ny = ny; h.depth = rs - d;
    h.px = ax + nx * (ra - 0.5f * h.depth); h.py = ay + ny * (ra - 0.5f * h.depth);
    return 1;
  }
  const float wx = ax - bx, wy = ay - by;
  const float lx = cb * wx + sb * wy, ly = -sb * wx + cb * wy;
  float qx = clampf(lx, -g.a, g.a), qy = clampf(ly, -g.b, g.b);
  const float ddx = lx - qx, ddy = ly - qy;
  float onx, ony, depth;
  if (ddx == 0 && ddy == 0) {
    const float px = g.a - fabsf(lx), py = g.b - fabsf(ly);
    if (px < py) { onx = lx >= 0 ? 1.f : -1.f; ony = 0; depth = ra + px; qx = onx * g.a; }
    else { onx = 0; ony = ly >= 0 ? 1.f : -1.f; depth = ra + py; qy = ony * g.b; }
  } else {
    const float d2 = ddx * ddx + ddy * ddy;
    if (d2 >= ra * ra) return 0;
    const float d = sqrtf(d2);
    onx = ddx / d; ony = ddy / d; depth = ra - d;
  }
  h.nx = -(cb * onx - sb * ony); h.ny = -(sb * onx + cb * ony);
  h.px = bx + cb * qx - sb * qy; h.py = by + sb * qx + cb * qy;
  h.depth = depth;
  return 1;
}

// robot spheres vs one planar body (free body index fbi >= 0, or a static circle); returns the
// number of contacts
__device__ __attribute__((noinline)) int dg_collide_body(DgRows& R, const DgState& D, const DgWork& K, const DgWorld& Wd,
                                                         const double* qd, int fbi, int shape, float bx, float by, float byaw,
                                                         float rbound, float vsz, float rstatic, double top, double bcoef,
                                                         double kcoef, double mu) {
  int total = 0;
  float cb = 1, sb = 0;
  if (fbi >= 0) sincosf(byaw, &sb, &cb);
#pragma unroll 1
  for (int s = 0; s < DG_NS; s++) {
    const double* c = K.sph[s];
    double r = g_dg.sph_r[s];
    if (c[2] - r >= top) continue;
    if (c[2] > top) r = sqrt(r * r - (c[2] - top) * (c[2] - top));
    const double dx = (double)bx - c[0], dy = (double)by - c[1], rs = r + (double)rbound;
    if (dx * dx + dy * dy > rs * rs) continue;
    const int ng = shape_ngeom(shape);
#pragma unroll 1
    for (int g = 0; g < ng; g++) {
      const Geom ge = shape_geom(shape, g, vsz, rstatic);
      const float gx = bx + cb * ge.ox - sb * ge.oy, gy = by + sb * ge.ox + cb * ge.oy;
      DgHit h;
      // the sphere centre enters the planar narrowphase in the arithmetic of the planar world
      if (dg_circle_geom((float)c[0], (float)c[1], (float)r, ge, gx, gy, cb, sb, h)) {
        const double nn[3] = {-(double)h.nx, -(double)h.ny, 0}, pc[3] = {(double)h.px, (double)h.py, c[2]};
        dg_add_contact(R, D, K, Wd, qd, s, nn, pc, (double)h.depth, fbi, bcoef, kcoef, mu);
        total++;
      }
    }
  }
  return total;
}

__device__ inline void dg_integrate(DgState& D, const double* qacc, double h) {
  for (int k = 0; k < 3; k++) { D.vlin[k] += h * qacc[k]; D.wloc[k] += h * qacc[3 + k]; }
  for (int j = 0; j < DG_NJ; j++) D.qd[j] += h * qacc[6 + j];
  for (int k = 0; k < 3; k++) D.pos[k] += h * D.vlin[k];
  for (int j = 0; j < DG_NJ; j++) D.q[j] += h * D.qd[j];
  const double wn = sqrt(dg_dot(D.wloc, D.wloc));
  if (wn > 0) {
    const double ang = 0.5 * h * wn, s = sin(ang) / wn, c = cos(ang);
    const double dq[4] = {c, s * D.wloc[0], s * D.wloc[1], s * D.wloc[2]};
    const double* q = D.quat;
    const double o[4] = {q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3],
                         q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2],
                         q[0] * dq[2] - q[1] * dq[3] + q[2] * dq[0] + q[3] * dq[1],
                         q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1] + q[3] * dq[0]};
    const double n = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int k = 0; k < 4; k++) D.quat[k] = o[k] / n;
  }
}

// subtree COM (xy) and its velocity (tasks/unsupervised.py:56-57)
__device__ inline void dg_com_vel(const DgState& D, const DgWork& K, double* out) {
  double m = 0, mc[2] = {0, 0}, qd[DG_NV];
  for (int b = 0; b < DG_NB; b++) { m += K.Ib[b].m; mc[0] += K.Ib[b].mc[0]; mc[1] += K.Ib[b].mc[1]; }
  dg_qd(D, qd);
  out[0] = D.pos[0] + mc[0] / m; out[1] = D.pos[1] + mc[1] / m;
  for (int k = 0; k < 2; k++) {
    double P = m * qd[k];
    for (int j = 3; j < DG_NV; j++) P += K.Mlin[k][j] * qd[j];
    out[2 + k] = P / m;
  }
}

}  // namespace sag
