// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of one Moby world that holds ONE RCArticulatedBody with 1-DOF joints and a fixed base (a floating one: six virtual joints of the model,
// mh_artic_model.floating_base; only conservative advancement reads the flag) (BASELINE config 5,
// example/ur10): TimeSteppingSimulator::step -> forward dynamics (CRB or articulated-body algorithm) -> joint-limit constraints ->
// the impact handler's no-slip path with NC = 0; and, when links carry sphere primitives (mh_artic_model.nspheres), the full step:
// conservative advancement, mini-steps, contact rows through calc_jacobian, the no-slip model or the Drumwright-Shell QP over
// contact AND limit rows (handle_impacts / do_mini_step below).
//
// PARITY UNPINNED for the dynamics: Ravelin::RCArticulatedBodyd (calc_fwd_dyn, get_generalized_inertia) is not in the
// reference tree and no reference artefact holds an articulated trajectory of a scene this build covers
// (regress/fixed-articulated-table.dat and contact-constrained-pendulum.dat need contact + bilateral rows).  The
// algorithm is Featherstone's, in world coordinates about the world origin, spatial vectors angular-first:
//   composite-rigid-body algorithm for H(q), recursive Newton-Euler with qdd = 0 for the bias C(q, qd) (gravity as a
//   base acceleration), Cholesky for H qdd = tau - C (LinAlgd::factor_chol / solve_chol_fast semantics, linalg.hpp).
// It is pinned by physics instead (tests/test_oracle_artic.py): H against the Jacobian form sum_i J_i' M_i J_i in numpy,
// energy conservation, the pendulum's period.  What IS restated from the reference, line by line: the limit
// constraints (include/Moby/ArticulatedBody.inl:9-43), compute_limit_components (src/ImpactConstraintHandler.cpp:
// 1755-1781 -- including its missing sign product between an upper and a lower limit), apply_no_slip_model with no
// contacts (ICH:1009-1417), update_from_stacked / update_constraint_velocities_from_impulses / apply_restitution
// (ICH:298-525), the stepping order of TimeSteppingSimulator::do_mini_step (TSS:114-222).
// sin / cos: an explicit fdlibm-style kernel (sincos below), because the HIP kernels must reproduce every bit.
#ifndef ORACLE_ARTIC_HPP
#define ORACLE_ARTIC_HPP
#include <cmath>
#include <cstring>
#include <vector>
#include "../include/moby_hip_artic.h"
#include "lcp.hpp"
#include "linalg.hpp"

namespace oracle {

static const double A_NEAR_ZERO = 1.4901161193847656e-08;

// sin and cos of x: Cody-Waite reduction by pi/2 (two terms: |x| < ~1e5 keeps full accuracy), then the fdlibm kernel
// polynomials on [-pi/4, pi/4]
static inline void sincos_kernel(double x, double& s, double& c)
{
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double kf = std::floor(x * invpio2 + 0.5);
  const double r = (x - kf * pio2_1) - kf * pio2_1t;
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double v = z * r;
  const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  const double ks = r + v * (S1 + z * rs);
  const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double kc = 1.0 - (0.5 * z - z * rc);
  long long k = (long long)kf;
  const int n = (int)(((k % 4) + 4) % 4);
  if (n == 0) { s = ks; c = kc; } else if (n == 1) { s = kc; c = -ks; } else if (n == 2) { s = -ks; c = -kc; } else { s = -kc; c = ks; }
}

namespace artic {
static inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void cross3(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
static inline void mat3mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3*i+j] = (A[3*i] * B[j] + A[3*i+1] * B[3+j]) + A[3*i+2] * B[6+j];
}
static inline void mat3vec(const double* A, const double* v, double* y) { for (int i = 0; i < 3; i++) y[i] = (A[3*i] * v[0] + A[3*i+1] * v[1]) + A[3*i+2] * v[2]; }
static inline double dot6(const double* a, const double* b) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + a[k] * b[k]; return acc; }
static inline void mat6vec(const double* A, const double* v, double* y) { for (int r = 0; r < 6; r++) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + A[6*r+k] * v[k]; y[r] = acc; } }
// spatial cross products, [angular; linear]
static inline void crm(const double* v, const double* m, double* o) {
  double a[3], b[3], c[3];
  cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
  for (int k = 0; k < 3; k++) { o[k] = a[k]; o[3+k] = b[k] + c[k]; }
}
static inline void crf(const double* v, const double* f, double* o) {
  double a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  for (int k = 0; k < 3; k++) { o[k] = a[k] + b[k]; o[3+k] = c[k]; }
}
}  // namespace artic

class Artic {
 public:
  static const int NJ = MH_ARTIC_MAX_JOINTS;
  const mh_artic_model* m; double* q; double* qd; mh_world_aux* aux;
  int nj;
  double R[NJ][9], x[NJ][3], S[NJ][6], Is[NJ][36], Ic[NJ][36], H[NJ * NJ], C[NJ];
  int32_t* trace = nullptr; int trace_cap = 0; int trace_len = 0;

  Artic(const mh_artic_model* model, double* q_, double* qd_, mh_world_aux* a) : m(model), q(q_), qd(qd_), aux(a), nj(model->nj) {}

  // link frames, motion subspaces and spatial inertias about the world origin
  void kinematics() {
    using namespace artic;
    for (int i = 0; i < nj; i++) {
      const int p = m->parent[i];
      const double* ax = m->axis[i];
      double Rl[9], tl[3];
      if (m->jtype[i] == MH_JOINT_REVOLUTE) {
        double s, c; sincos_kernel(q[i], s, c);
        const double t = 1.0 - c;
        const double K[9] = { 0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0 };
        double Rq[9];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rq[3*a+b] = (((a == b) ? c : 0.0) + (t * ax[a]) * ax[b]) + s * K[3*a+b];
        mat3mul(m->Rrel[i], Rq, Rl);
        for (int k = 0; k < 3; k++) tl[k] = m->trel[i][k];
      } else {
        for (int k = 0; k < 9; k++) Rl[k] = m->Rrel[i][k];
        double d[3], Rd[3];
        for (int k = 0; k < 3; k++) d[k] = ax[k] * q[i];
        mat3vec(m->Rrel[i], d, Rd);
        for (int k = 0; k < 3; k++) tl[k] = m->trel[i][k] + Rd[k];
      }
      if (p < 0) { for (int k = 0; k < 9; k++) R[i][k] = Rl[k]; for (int k = 0; k < 3; k++) x[i][k] = tl[k]; }
      else { mat3mul(R[p], Rl, R[i]); double Rt[3]; mat3vec(R[p], tl, Rt); for (int k = 0; k < 3; k++) x[i][k] = x[p][k] + Rt[k]; }
      double aw[3]; mat3vec(R[i], ax, aw);
      if (m->jtype[i] == MH_JOINT_REVOLUTE) { double xa[3]; cross3(x[i], aw, xa); for (int k = 0; k < 3; k++) { S[i][k] = aw[k]; S[i][3+k] = xa[k]; } }
      else for (int k = 0; k < 3; k++) { S[i][k] = 0.0; S[i][3+k] = aw[k]; }
      double rc[3], r[3]; mat3vec(R[i], m->com[i], rc);
      for (int k = 0; k < 3; k++) r[k] = x[i][k] + rc[k];
      double T[9], Iw[9];
      mat3mul(R[i], m->inertia[i], T);
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Iw[3*a+b] = (T[3*a] * R[i][3*b] + T[3*a+1] * R[i][3*b+1]) + T[3*a+2] * R[i][3*b+2];
      Iw[1] = Iw[3]; Iw[2] = Iw[6]; Iw[5] = Iw[7];
      const double mass = m->mass[i];
      const double rr = dot3(r, r);
      const double rx[9] = { 0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0 };
      double* I6 = Is[i];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        I6[6*a+b] = Iw[3*a+b] + mass * (((a == b) ? rr : 0.0) - r[a] * r[b]);
        I6[6*a+3+b] = mass * rx[3*a+b];
        I6[6*(3+a)+b] = mass * rx[3*b+a];
        I6[6*(3+a)+3+b] = (a == b) ? mass : 0.0;
      }
    }
  }
  // RCArticulatedBodyd::calc_jacobian at a point p (model frame) of link `link` (call kinematics() first): rows 0..2 the linear
  // velocity of the point, 3..5 the angular velocity; column j = S_j moved to p for j between the link and the base, else 0
  void jacobian(int link, const double p[3], double* J /* 6 x nj row-major */) const {
    for (int e = 0; e < 6 * nj; e++) J[e] = 0.0;
    for (int j = link; j >= 0; j = m->parent[j]) {
      const double* s = S[j];
      for (int r = 0; r < 3; r++) { const int k1 = (r + 1) % 3, k2 = (r + 2) % 3; J[r * nj + j] = s[3 + r] + (s[k1] * p[k2] - s[k2] * p[k1]); J[(3 + r) * nj + j] = s[r]; }
    }
  }
  // composite-rigid-body algorithm: H(i, j) = S_j' Ic_i S_i for j on the path from i to the base
  void crba() {
    using namespace artic;
    for (int i = 0; i < nj; i++) for (int e = 0; e < 36; e++) Ic[i][e] = Is[i][e];
    for (int i = nj - 1; i >= 0; i--) { const int p = m->parent[i]; if (p >= 0) for (int e = 0; e < 36; e++) Ic[p][e] = Ic[p][e] + Ic[i][e]; }
    for (int e = 0; e < nj * nj; e++) H[e] = 0.0;
    for (int i = 0; i < nj; i++) {
      double F[6]; mat6vec(Ic[i], S[i], F);
      H[i * nj + i] = dot6(S[i], F);
      for (int j = m->parent[i]; j >= 0; j = m->parent[j]) { const double h = dot6(S[j], F); H[i * nj + j] = h; H[j * nj + i] = h; }
    }
  }
  // recursive Newton-Euler with qdd = 0: C(q, qd) including gravity (base acceleration -g)
  void bias() {
    using namespace artic;
    double v[NJ][6], a[NJ][6], f[NJ][6];
    for (int i = 0; i < nj; i++) {
      const int p = m->parent[i];
      double vj[6], cv[6];
      for (int k = 0; k < 6; k++) vj[k] = S[i][k] * qd[i];
      for (int k = 0; k < 6; k++) v[i][k] = (p < 0) ? vj[k] : v[p][k] + vj[k];
      crm(v[i], vj, cv);
      for (int k = 0; k < 6; k++) {
        const double ap = (p < 0) ? ((k < 3) ? 0.0 : -m->gravity[k - 3]) : a[p][k];
        a[i][k] = ap + cv[k];
      }
      double Ia[6], Iv[6], cf[6];
      mat6vec(Is[i], a[i], Ia); mat6vec(Is[i], v[i], Iv); crf(v[i], Iv, cf);
      for (int k = 0; k < 6; k++) f[i][k] = Ia[k] + cf[k];
    }
    for (int i = nj - 1; i >= 0; i--) {
      C[i] = dot6(S[i], f[i]);
      const int p = m->parent[i];
      if (p >= 0) for (int k = 0; k < 6; k++) f[p][k] = f[p][k] + f[i][k];
    }
  }
  // calc_fwd_dyn: H qdd = tau - C by Cholesky; false if H is not positive definite
  bool fwd_dyn(const double* tau, double* qdd) {
    kinematics(); crba(); bias();
    std::vector<double> L(H, H + nj * nj);
    if (!chol_factor(nj, L.data(), nj)) return false;
    for (int i = 0; i < nj; i++) qdd[i] = (tau ? tau[i] : 0.0) - C[i];
    chol_solve(nj, L.data(), nj, qdd);
    return true;
  }
  // calc_fwd_dyn, eFeatherstone: the articulated-body algorithm (Featherstone, Rigid Body Dynamics Algorithms, table 7.1) with
  // every spatial quantity expressed at the world origin like the rest of this file (no link-to-link transforms).  Ravelin's
  // FSAB source is not in the tree: parity unpinned; the order below is the kernel's.  d_i = S' IA S > 0 for a physical body.
  bool fwd_dyn_aba(const double* tau, double* qdd) {
    using namespace artic;
    kinematics();
    double v[NJ][6], c[NJ][6], IA[NJ][36], pA[NJ][6], U[NJ][6], d[NJ], u[NJ], a[NJ][6];
    for (int i = 0; i < nj; i++) {                                   // pass 1, outward: velocities, bias accelerations and forces
      const int p = m->parent[i];
      double vj[6], Iv[6];
      for (int k = 0; k < 6; k++) vj[k] = S[i][k] * qd[i];
      for (int k = 0; k < 6; k++) v[i][k] = (p < 0) ? vj[k] : v[p][k] + vj[k];
      crm(v[i], vj, c[i]);
      for (int e = 0; e < 36; e++) IA[i][e] = Is[i][e];
      mat6vec(Is[i], v[i], Iv); crf(v[i], Iv, pA[i]);
    }
    for (int i = nj - 1; i >= 0; i--) {                              // pass 2, inward: articulated inertias and bias forces
      const int p = m->parent[i];
      mat6vec(IA[i], S[i], U[i]);
      d[i] = dot6(S[i], U[i]);
      if (!(d[i] > 0.0)) return false;
      u[i] = (tau ? tau[i] : 0.0) - dot6(S[i], pA[i]);
      if (p >= 0) {
        double Ia[36], Iac[6];
        for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) { double t = U[i][r] * U[i][cc]; t = t / d[i]; Ia[6*r+cc] = IA[i][6*r+cc] - t; }
        mat6vec(Ia, c[i], Iac);
        for (int r = 0; r < 6; r++) { double e = U[i][r] * u[i]; e = e / d[i]; const double pa = (pA[i][r] + Iac[r]) + e; pA[p][r] = pA[p][r] + pa; }
        for (int e = 0; e < 36; e++) IA[p][e] = IA[p][e] + Ia[e];
      }
    }
    for (int i = 0; i < nj; i++) {                                   // pass 3, outward: accelerations
      const int p = m->parent[i];
      double ap[6];
      for (int k = 0; k < 6; k++) { const double base = (p < 0) ? ((k < 3) ? 0.0 : -m->gravity[k - 3]) : a[p][k]; ap[k] = base + c[i][k]; }
      double t = u[i] - dot6(U[i], ap);
      qdd[i] = t / d[i];
      for (int k = 0; k < 6; k++) a[i][k] = ap[k] + S[i][k] * qdd[i];
    }
    return true;
  }
  void lcp_account(int n, unsigned pivots) { aux->lcp_solves++; aux->lcp_rows += (unsigned long long)n; aux->lcp_pivots += pivots; aux->lcp_alg_bytes += 8ull * ((unsigned long long)n * n + 2ull * n); }

  // find_limit_constraints + calc_impacting_unilateral_constraint_forces for the limits of this body
  void handle_limits() {
    int idx[2 * NJ]; bool upper[2 * NJ]; int nl = 0;
    for (int i = 0; i < nj; i++) {                                   // ArticulatedBody.inl:9-43 (q_tare = 0)
      if (q[i] >= m->hilimit[i]) { idx[nl] = i; upper[nl] = true; nl++; }
      if (q[i] <= m->lolimit[i]) { idx[nl] = i; upper[nl] = false; nl++; }
    }
    if (nl == 0) return;
    bool impacting = false;                                          // CSim:313-323
    for (int k = 0; k < nl; k++) { const double v = upper[k] ? -qd[idx[k]] : qd[idx[k]]; if (v < -A_NEAR_ZERO) impacting = true; }
    if (!impacting) return;
    if (nl > MH_NOSLIP_MAX) { aux->status |= MH_WORLD_UNSUPPORTED; return; }
    if (m->algorithm == MH_ARTIC_FSAB) crba();                       // get_generalized_inertia (ICH:1600-1607): CRB whatever the forward dynamics used
    // compute_X: X = inverse_SPD(H) (ICH:1607); compute_limit_components (ICH:1755-1781)
    std::vector<double> X(H, H + nj * nj);
    if (!inverse_spd(nj, X.data(), nj)) { aux->status |= MH_WORLD_LCP_FAILED; return; }
    std::vector<double> MM((size_t)nl * nl), Lv(nl), l(nl);
    for (int a = 0; a < nl; a++) for (int b = a; b < nl; b++) { const double e = X[idx[a] * nj + idx[b]]; MM[a + (size_t)nl * b] = e; MM[b + (size_t)nl * a] = e; }
    for (int k = 0; k < nl; k++) { Lv[k] = qd[idx[k]]; if (upper[k]) Lv[k] = -Lv[k]; }
    // apply_no_slip_model with no contacts: MM = L X L', qq = L v; lcp_fast on the persistent _v, then the Lemke ladder
    Vec z; z.d.assign(aux->vns, aux->vns + MH_NOSLIP_MAX); z.len = (unsigned)aux->vns_size;
    oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
    LCP lcp; lcp.rng = &rs;
    Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? ((trace_cap - trace_len > 0) ? trace_cap - trace_len : 0) : 0;
    lcp.trace = &tr;
    unsigned piv = 0;
    bool ok = lcp.lcp_fast(nl, MM.data(), nl, Lv.data(), z, -1.0);
    piv += lcp.pivots;
    if (!ok) { ok = lcp.lcp_lemke_regularized(nl, MM.data(), nl, Lv.data(), z); piv += lcp.pivots; }
    trace_len += tr.len;
    std::memcpy(aux->rng, &rs, sizeof(rs));
    lcp_account(nl, piv);
    if (!ok) { aux->status |= MH_WORLD_LCP_FAILED; return; }       // std::runtime_error("Unable to solve constraint LCP!")
    for (int k = 0; k < nl; k++) aux->vns[k] = z[k];
    aux->vns_size = nl;
    for (int k = 0; k < nl; k++) l[k] = z[k];
    auto apply = [&]() {                                             // update_from_stacked (ICH:298-397): dv = X_LT ls
      std::vector<double> dv(nj, 0.0);
      for (int k = 0; k < nl; k++) { const double ls = upper[k] ? -l[k] : l[k]; for (int r = 0; r < nj; r++) dv[r] = dv[r] + ls * X[idx[k] * nj + r]; }
      for (int r = 0; r < nj; r++) qd[r] = qd[r] + dv[r];
    };
    auto update_vels = [&]() {                                       // L_v += L_X_LT l (ICH:452)
      std::vector<double> t(nl, 0.0);
      for (int k = 0; k < nl; k++) for (int r = 0; r < nl; r++) t[r] = t[r] + l[k] * MM[r + (size_t)nl * k];
      for (int r = 0; r < nl; r++) Lv[r] = Lv[r] + t[r];
    };
    auto minv_of = [&]() { double mn = Lv[0]; for (int k = 1; k < nl; k++) mn = (Lv[k] < mn) ? Lv[k] : mn; return mn; };
    apply(); update_vels();
    const double minv = minv_of();
    bool changed = false;                                            // apply_restitution(q) (ICH:497-525)
    for (int k = 0; k < nl; k++) { l[k] = l[k] * m->limit_restitution[idx[k]]; if (!changed && l[k] > A_NEAR_ZERO) changed = true; }
    if (changed) {
      apply(); update_vels();
      const double minv_plus = minv_of();
      // ICH:284-291 would re-solve and then read the Drumwright-Shell solver's _z, which this path never sized
      if (minv_plus < 0.0 && minv_plus < minv - A_NEAR_ZERO) aux->status |= MH_WORLD_UNSUPPORTED;
    }
    for (int k = 0; k < nl; k++) { const double v = upper[k] ? -qd[idx[k]] : qd[idx[k]]; if (v < -A_NEAR_ZERO) aux->status |= MH_WORLD_IMPACT_TOL; }   // ICH:157-167
  }

  // ---- sphere primitives on links against the static plane (mh_artic_model.nspheres) -------------------------------------
  // Restated from the reference: find_contacts / signed distance of a sphere and a plane (CCD.inl:804-847, PlanePrimitive.cpp),
  // CCD::calc_CA_Euler_step_sphere / _generic / calc_next_CA_Euler_step_generic (CCD.cpp:138-405), the ARTICULATED
  // CCD::calc_max_dist (CCD.cpp:545-583), TimeSteppingSimulator::do_mini_step (TSS:114-222), the contact rows
  // [d, r x d] . calc_jacobian(link) of add_contact_dir_to_Jacobian (ICH:1847-1895), compute_problem_data's products
  // (ICH:2110-2160), apply_no_slip_model with contacts AND limits (ICH:1009-1417).  Link velocities / calc_jacobian are Ravelin's:
  // parity unpinned, spatial velocities about the world origin like the rest of this file.
  struct AContact { int s, link; double p[3], n[3], sv[3], tv[3], dist; };
  static constexpr double A_INF = 1.7976931348623157e308;
  void plane_n(double n[3]) const { n[0] = m->plane_R[1]; n[1] = m->plane_R[4]; n[2] = m->plane_R[7]; }
  void to_plane(const double p[3], double o[3]) const {
    const double* Rp = m->plane_R; const double d[3] = { p[0] - m->plane_o[0], p[1] - m->plane_o[1], p[2] - m->plane_o[2] };
    o[0] = (Rp[0]*d[0] + Rp[3]*d[1]) + Rp[6]*d[2]; o[1] = (Rp[1]*d[0] + Rp[4]*d[1]) + Rp[7]*d[2]; o[2] = (Rp[2]*d[0] + Rp[5]*d[1]) + Rp[8]*d[2];
  }
  void from_plane(double px, double py, double pz, double o[3]) const {
    const double* Rp = m->plane_R;
    o[0] = m->plane_o[0] + ((Rp[0]*px + Rp[1]*py) + Rp[2]*pz); o[1] = m->plane_o[1] + ((Rp[3]*px + Rp[4]*py) + Rp[5]*pz); o[2] = m->plane_o[2] + ((Rp[6]*px + Rp[7]*py) + Rp[8]*pz);
  }
  void sphere_center(int s, double c[3]) const {
    const int l = m->sphere_link[s]; double rc[3]; artic::mat3vec(R[l], m->sphere_center[s], rc);
    for (int k = 0; k < 3; k++) c[k] = x[l][k] + rc[k];
  }
  // spatial velocity of every link about the world origin, [angular; linear] (kinematics() first)
  void link_velocities(double V[][6]) const {
    for (int i = 0; i < nj; i++) { const int p = m->parent[i]; for (int k = 0; k < 6; k++) { const double vj = S[i][k] * qd[i]; V[i][k] = (p < 0) ? vj : V[p][k] + vj; } }
  }
  static double point_vel_dir(const double* V6, const double* p, const double* d) {
    double wxp[3]; artic::cross3(V6, p, wxp);
    const double v[3] = { V6[3] + wxp[0], V6[4] + wxp[1], V6[5] + wxp[2] };
    return artic::dot3(d, v);
  }
  static void orthonormal_basis(const double n[3], double s[3], double t[3]) {      // as world.hpp (Ravelin's: pinned choice)
    const double ax = std::fabs(n[0]), ay = std::fabs(n[1]), az = std::fabs(n[2]);
    double e[3] = { 0.0, 0.0, 0.0 };
    if (ax <= ay && ax <= az) e[0] = 1.0; else if (ay <= az) e[1] = 1.0; else e[2] = 1.0;
    artic::cross3(n, e, s);
    const double len = std::sqrt((s[0]*s[0] + s[1]*s[1]) + s[2]*s[2]);
    for (int k = 0; k < 3; k++) s[k] = s[k] / len;
    artic::cross3(n, s, t);
  }
  // find_contacts_sphere_plane (CCD.inl:804-847): at most one contact, midway between the closest points
  bool find_contact(int s, double TOL, AContact& c) const {
    double ctr[3], cp[3]; sphere_center(s, ctr); to_plane(ctr, cp);
    const double r = m->sphere_radius[s];
    const double dist = cp[1] - r;
    if (dist > TOL) return false;
    c.s = s; c.link = m->sphere_link[s]; c.dist = dist;
    from_plane(cp[0], 0.5 * (cp[1] - r), cp[2], c.p);
    plane_n(c.n); orthonormal_basis(c.n, c.sv, c.tv);
    return true;
  }
  // the articulated CCD::calc_max_dist (CCD.cpp:545-583): the base's linear velocity along n (0 for a fixed base; for mh_artic_model.floating_base the rates of the
  // three virtual sliders, which ARE the base link's COM velocity in global axes), then the chain of inner joints up to the base; a joint's pose is its link's frame origin
  double calc_max_dist(int link, const double n[3], double rmax) const {
    double mv = 0.0;
    if (m->floating_base) mv = (n[0] * qd[0] + n[1] * qd[1]) + n[2] * qd[2];
    int inner = link;
    mv = mv + (2.0 * rmax) * std::fabs(qd[inner]);
    while (m->parent[inner] >= 0) {
      const int nxt = m->parent[inner];
      const double d[3] = { x[nxt][0] - x[inner][0], x[nxt][1] - x[inner][1], x[nxt][2] - x[inner][2] };
      mv = mv + std::fabs(qd[nxt]) * std::sqrt((d[0]*d[0] + d[1]*d[1]) + d[2]*d[2]);
      inner = nxt;
    }
    return mv;
  }
  // CollisionGeometry::get_farthest_point_distance (CollisionGeometry.cpp:52-68): primitive radius + its offset from the link's
  // pose (the link COM frame, eLinkCOM)
  double rmax_of(int s) const {
    const int l = m->sphere_link[s];
    const double d[3] = { m->sphere_center[s][0] - m->com[l][0], m->sphere_center[s][1] - m->com[l][1], m->sphere_center[s][2] - m->com[l][2] };
    return m->sphere_radius[s] + std::sqrt((d[0]*d[0] + d[1]*d[1]) + d[2]*d[2]);
  }
  // CCD::calc_CA_Euler_step_sphere (CCD.cpp:138-166) -> _generic (:169-235) -> calc_next_CA_Euler_step_generic (:238-405)
  double CA_step(int s, const double V[][6]) const {
    double ctr[3], cp[3]; sphere_center(s, ctr); to_plane(ctr, cp);
    const double dist = cp[1] + (-1.0 * m->sphere_radius[s]);
    AContact c;
    if (!(dist > A_NEAR_ZERO)) {
      const bool has = find_contact(s, A_NEAR_ZERO, c);
      if (has && std::fabs(point_vel_dir(V[c.link], c.p, c.n)) < A_NEAR_ZERO * 10) return A_INF;
    }
    if (dist <= 0.0) {                                               // bodies in contact: the "next" step
      if (!find_contact(s, A_NEAR_ZERO, c)) return A_INF;
      if (point_vel_dir(V[c.link], c.p, c.n) < -A_NEAR_ZERO) return 0.0;
      return A_INF;
    }
    double pn[3]; plane_n(pn);                                       // n0 = from the plane to the sphere; body A moves toward B along -n0 (CCD.cpp:214)
    const double mn[3] = { -pn[0], -pn[1], -pn[2] };
    const double tA = calc_max_dist(m->sphere_link[s], mn, rmax_of(s));   // the plane's body is disabled: 0
    double total = tA + 0.0;
    if (total < 0.0) total = 0.0;
    const double cand = dist / total;
    return (cand < A_INF) ? cand : A_INF;
  }

  // calc_impacting_unilateral_constraint_forces (CSim:298-355) -> process_constraints: one island (every constraint touches the
  // articulated body) -> apply_no_slip_model_to_connected_constraints (ICH:236-295)
  void handle_impacts(const std::vector<AContact>& cs) {
    using namespace artic;
    const int nc = (int)cs.size();
    int idx[2 * NJ]; bool upper[2 * NJ]; int nl = 0;
    for (int i = 0; i < nj; i++) {                                   // ArticulatedBody.inl:9-43 (q_tare = 0)
      if (q[i] >= m->hilimit[i]) { idx[nl] = i; upper[nl] = true; nl++; }
      if (q[i] <= m->lolimit[i]) { idx[nl] = i; upper[nl] = false; nl++; }
    }
    if (nc + nl == 0) return;
    double V[NJ][6]; link_velocities(V);
    bool impacting = false;                                          // CSim:313-323
    for (int i = 0; i < nc; i++) if (point_vel_dir(V[cs[i].link], cs[i].p, cs[i].n) < -A_NEAR_ZERO) impacting = true;
    for (int k = 0; k < nl; k++) { const double v = upper[k] ? -qd[idx[k]] : qd[idx[k]]; if (v < -A_NEAR_ZERO) impacting = true; }
    if (!impacting) return;
    // ICH:123-146: the no-slip model when every CONTACT has mu_coulomb >= 100 (limits do not count: an island of limits alone
    // takes it too), otherwise the Drumwright-Shell QP
    const bool noslip = (nc == 0) || (m->cp_mu_coulomb >= 1e2);
    const int n = nc + nl;
    const int nk = (m->cp_nk > 0) ? m->cp_nk : 4, kh = nk / 2;
    const int nvars = 5 * nc + nl, N = nvars + nc + nl + nc * kh;    // ICH-QP:97-112
    // capacities of the build: the no-slip LCP's warm start _v holds MH_NOSLIP_MAX rows, the wave solver MH_LCP_MAX_N_WAVE, the limit tables MH_NOSLIP_MAX limits
    if (noslip ? (n > MH_NOSLIP_MAX) : (N > MH_LCP_MAX_N_WAVE || nl > MH_NOSLIP_MAX)) { aux->status |= MH_WORLD_UNSUPPORTED; return; }
    if (m->algorithm == MH_ARTIC_FSAB) crba();                       // get_generalized_inertia (ICH:1600-1607)
    std::vector<double> X(H, H + nj * nj);
    if (!inverse_spd(nj, X.data(), nj)) { aux->status |= MH_WORLD_LCP_FAILED; return; }
    // contact rows (ICH:1847-1895): wrench [d, r x d] about the link's COM times calc_jacobian at the COM (rows: linear, angular)
    std::vector<double> C[3], XC[3];                                 // C[d]: nc x nj; XC[d] = C[d] X (rows of X_CdT')
    for (int d = 0; d < 3; d++) { C[d].assign((size_t)nc * nj, 0.0); XC[d].assign((size_t)nc * nj, 0.0); }
    for (int i = 0; i < nc; i++) {
      const int l = cs[i].link;
      double rc[3], com[3], r[3], J[6 * NJ];
      mat3vec(R[l], m->com[l], rc);
      for (int k = 0; k < 3; k++) { com[k] = x[l][k] + rc[k]; r[k] = cs[i].p[k] - com[k]; }
      jacobian(l, com, J);
      const double* dirs[3] = { cs[i].n, cs[i].sv, cs[i].tv };
      for (int d = 0; d < 3; d++) {
        double w[6]; cross3(r, dirs[d], w + 3);
        for (int k = 0; k < 3; k++) w[k] = dirs[d][k];
        for (int j = 0; j < nj; j++) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + w[k] * J[k * nj + j]; C[d][(size_t)i * nj + j] = acc; }
      }
    }
    for (int d = 0; d < 3; d++) for (int i = 0; i < nc; i++) for (int c = 0; c < nj; c++) {
      double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[d][(size_t)i * nj + k] * X[k * nj + c];
      XC[d][(size_t)i * nj + c] = acc;
    }
    // the cross blocks (ICH:2127-2147) and vectors (:2153-2156); compute_limit_components (ICH:1755-1781), signs as there
    std::vector<double> G[3][3], CL[3], Cv[3];
    for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) {
      G[a][b].assign((size_t)nc * nc, 0.0);
      for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) { double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[a][(size_t)i * nj + k] * XC[b][(size_t)j * nj + k]; G[a][b][(size_t)i * nc + j] = acc; }
    }
    for (int d = 0; d < 3; d++) {
      CL[d].assign((size_t)nc * (nl > 0 ? nl : 1), 0.0); Cv[d].assign(nc, 0.0);
      for (int i = 0; i < nc; i++) {
        for (int k2 = 0; k2 < nl; k2++) { double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[d][(size_t)i * nj + k] * X[idx[k2] * nj + k]; CL[d][(size_t)i * nl + k2] = acc; }
        double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[d][(size_t)i * nj + k] * qd[k];
        Cv[d][i] = acc;
      }
    }
    std::vector<double> LL((size_t)nl * nl + 1), Lv(nl + 1);
    for (int a = 0; a < nl; a++) for (int b = a; b < nl; b++) { const double e = X[idx[a] * nj + idx[b]]; LL[a + (size_t)nl * b] = e; LL[b + (size_t)nl * a] = e; }
    for (int k = 0; k < nl; k++) { Lv[k] = qd[idx[k]]; if (upper[k]) Lv[k] = -Lv[k]; }

    std::vector<double> cn(nc, 0.0), csv(nc, 0.0), ctv(nc, 0.0), l(nl, 0.0);
    // dv = X_CnT cn + X_CsT cs + X_CtT ct + X_LT sl (ICH:1365-1373, 345-352): four products, added in this order
    auto apply = [&]() {
      std::vector<double> dv(nj, 0.0), t(nj);
      const std::vector<double>* imp[3] = { &cn, &csv, &ctv };
      for (int d = 0; d < 3; d++) {
        for (int r = 0; r < nj; r++) { double acc = 0.0; for (int i = 0; i < nc; i++) acc = acc + XC[d][(size_t)i * nj + r] * (*imp[d])[i]; t[r] = acc; }
        for (int r = 0; r < nj; r++) dv[r] = (d == 0) ? t[r] : dv[r] + t[r];
      }
      for (int r = 0; r < nj; r++) { double acc = 0.0; for (int k = 0; k < nl; k++) { const double ls = upper[k] ? -l[k] : l[k]; acc = acc + ls * X[idx[k] * nj + r]; } t[r] = acc; }
      for (int r = 0; r < nj; r++) dv[r] = dv[r] + t[r];
      for (int r = 0; r < nj; r++) qd[r] = qd[r] + dv[r];
    };
    // update_constraint_velocities_from_impulses (ICH:427-464)
    auto Gs = [&](int a, int b, int i, int j) -> double { return (a <= b) ? G[a][b][(size_t)i * nc + j] : G[b][a][(size_t)j * nc + i]; };
    auto update_vels = [&]() {
      const std::vector<double>* imp[3] = { &cn, &csv, &ctv };
      for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) {
          std::vector<double> t(nc, 0.0);
          for (int i = 0; i < nc; i++) { double acc = 0.0; for (int j = 0; j < nc; j++) acc = acc + Gs(a, b, i, j) * (*imp[b])[j]; t[i] = acc; }
          for (int i = 0; i < nc; i++) Cv[a][i] = Cv[a][i] + t[i];
        }
        for (int i = 0; i < nc; i++) { double acc = 0.0; for (int k = 0; k < nl; k++) acc = acc + CL[a][(size_t)i * nl + k] * l[k]; Cv[a][i] = Cv[a][i] + acc; }
      }
      const std::vector<double>* imp2[3] = { &cn, &csv, &ctv };
      for (int d = 0; d < 3; d++) for (int k = 0; k < nl; k++) { double acc = 0.0; for (int i = 0; i < nc; i++) acc = acc + CL[d][(size_t)i * nl + k] * (*imp2[d])[i]; Lv[k] = Lv[k] + acc; }
      std::vector<double> t(nl + 1, 0.0);
      for (int r = 0; r < nl; r++) { double acc = 0.0; for (int k = 0; k < nl; k++) acc = acc + LL[r + (size_t)nl * k] * l[k]; t[r] = acc; }
      for (int r = 0; r < nl; r++) Lv[r] = Lv[r] + t[r];
    };
    auto minv_of = [&]() {                                           // calc_min_constraint_velocity (ICH:413-424)
      double mn = A_INF;
      for (int i = 0; i < nc; i++) mn = (i == 0 || Cv[0][i] < mn) ? Cv[0][i] : mn;
      if (nl > 0) { double ml = Lv[0]; for (int k = 1; k < nl; k++) ml = (Lv[k] < ml) ? Lv[k] : ml; mn = (ml < mn) ? ml : mn; }
      return mn;
    };
    auto solve_noslip = [&]() -> bool {
    // apply_no_slip_model (ICH:1009-1417)
    std::vector<int> Sx, Tx; std::vector<double> Y;
    auto build_Y = [&](bool skew) -> int {
      const int ns = (int)Sx.size(), nt = (int)Tx.size(), mm = ns + nt;
      Y.assign((size_t)mm * mm + 1, 0.0);
      for (int a = 0; a < ns; a++) for (int b = 0; b < ns; b++) Y[a + (size_t)mm * b] = G[1][1][(size_t)Sx[a] * nc + Sx[b]];
      for (int a = 0; a < nt; a++) for (int b = 0; b < nt; b++) Y[(ns + a) + (size_t)mm * (ns + b)] = G[2][2][(size_t)Tx[a] * nc + Tx[b]];
      for (int a = 0; a < ns; a++) for (int b = 0; b < nt; b++) { const double g = G[1][2][(size_t)Sx[a] * nc + Tx[b]]; Y[a + (size_t)mm * (ns + b)] = g; Y[(ns + b) + (size_t)mm * a] = g; }
      if (skew) for (int j = 0; j < mm; j++) Y[j + (size_t)mm * j] = Y[j + (size_t)mm * j] - A_NEAR_ZERO;
      return mm;
    };
    for (int i = 0; i < nc; i++) {                                   // greedy largest non-singular tangent set (ICH:1087-1145)
      Sx.push_back(i); int mm = build_Y(true); if (!chol_factor(mm, Y.data(), mm)) Sx.pop_back();
      Tx.push_back(i); mm = build_Y(true);     if (!chol_factor(mm, Y.data(), mm)) Tx.pop_back();
    }
    const int ns = (int)Sx.size(), nt = (int)Tx.size();
    const int mm = build_Y(false);
    if (mm > 0 && !chol_factor(mm, Y.data(), mm)) { aux->status |= MH_WORLD_LCP_FAILED; return false; }   // assert(success)
    // Q X X' (n x mm): contact rows [Cn X Cs'(:,S)  Cn X Ct'(:,T)], limit rows [Cs X L'(S,:)'  Ct X L'(T,:)'] (ICH:1198-1207)
    std::vector<double> QX((size_t)n * mm + 1);
    for (int i = 0; i < nc; i++) {
      for (int a = 0; a < ns; a++) QX[(size_t)i * mm + a] = G[0][1][(size_t)i * nc + Sx[a]];
      for (int a = 0; a < nt; a++) QX[(size_t)i * mm + ns + a] = G[0][2][(size_t)i * nc + Tx[a]];
    }
    for (int k = 0; k < nl; k++) {
      for (int a = 0; a < ns; a++) QX[(size_t)(nc + k) * mm + a] = CL[1][(size_t)Sx[a] * nl + k];
      for (int a = 0; a < nt; a++) QX[(size_t)(nc + k) * mm + ns + a] = CL[2][(size_t)Tx[a] * nl + k];
    }
    std::vector<double> W((size_t)mm * n + 1), col(mm + 1);
    for (int j = 0; j < n; j++) {
      for (int a = 0; a < mm; a++) col[a] = QX[(size_t)j * mm + a];
      if (mm > 0) chol_solve(mm, Y.data(), mm, col.data());
      for (int a = 0; a < mm; a++) W[a + (size_t)mm * j] = col[a];
    }
    std::vector<double> MM((size_t)n * n), qq(n);
    auto QMQ = [&](int i, int j) -> double {                         // Q inv(M) Q' (ICH:1190-1196)
      if (i < nc && j < nc) return G[0][0][(size_t)i * nc + j];
      if (i < nc) return CL[0][(size_t)i * nl + (j - nc)];
      if (j < nc) return CL[0][(size_t)j * nl + (i - nc)];
      return LL[(i - nc) + (size_t)nl * (j - nc)];
    };
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
      double acc = 0.0; for (int a = 0; a < mm; a++) acc = acc + QX[(size_t)i * mm + a] * W[a + (size_t)mm * j];
      MM[i + (size_t)n * j] = QMQ(i, j) - acc;
    }
    std::vector<double> YXv(mm + 1);
    for (int a = 0; a < ns; a++) YXv[a] = Cv[1][Sx[a]];
    for (int a = 0; a < nt; a++) YXv[ns + a] = Cv[2][Tx[a]];
    if (mm > 0) chol_solve(mm, Y.data(), mm, YXv.data());
    for (int i = 0; i < n; i++) {
      double acc = 0.0; for (int a = 0; a < mm; a++) acc = acc + QX[(size_t)i * mm + a] * YXv[a];
      qq[i] = ((i < nc) ? Cv[0][i] : Lv[i - nc]) - acc;
    }
    Vec z; z.d.assign(aux->vns, aux->vns + MH_NOSLIP_MAX); z.len = (unsigned)aux->vns_size;
    oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
    LCP lcp; lcp.rng = &rs;
    Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? ((trace_cap - trace_len > 0) ? trace_cap - trace_len : 0) : 0;
    lcp.trace = &tr;
    unsigned piv = 0;
    bool ok = lcp.lcp_fast(n, MM.data(), n, qq.data(), z, -1.0);
    piv += lcp.pivots;
    if (!ok) { ok = lcp.lcp_lemke_regularized(n, MM.data(), n, qq.data(), z); piv += lcp.pivots; }
    trace_len += tr.len;
    std::memcpy(aux->rng, &rs, sizeof(rs));
    lcp_account(n, piv);
    if (!ok) { aux->status |= MH_WORLD_LCP_FAILED; return false; }
    for (int k = 0; k < n; k++) aux->vns[k] = z[k];
    aux->vns_size = n;
    std::vector<double> t2(mm + 1);                                  // [cs; ct] = -(Y^-1 X v + Y^-1 (QX)' z) (ICH:1293-1298)
    for (int a = 0; a < mm; a++) { double acc = 0.0; for (int i = 0; i < n; i++) acc = acc + QX[(size_t)i * mm + a] * z[i]; t2[a] = acc; }
    if (mm > 0) chol_solve(mm, Y.data(), mm, t2.data());
    for (int i = 0; i < nc; i++) cn[i] = z[i];
    for (int k = 0; k < nl; k++) l[k] = z[nc + k];
    for (int a = 0; a < ns; a++) csv[Sx[a]] = -(YXv[a] + t2[a]);
    for (int a = 0; a < nt; a++) ctv[Tx[a]] = -(YXv[ns + a] + t2[ns + a]);
      return true;
    };
    // ---- Drumwright-Shell QP -> LCP with contact and limit variables (ICH-QP:94-497) on the persistent _z / _zlast -------
    // variables [cn cs ct ncs nct l], inequality rows [Cn v+ >= 0 (NC); L v+ >= 0 (NL); friction polygons (NC nk/2)]
    auto solve_qp = [&]() -> bool {
      std::vector<double> MM((size_t)N * N, 0.0), qq(N, 0.0);
      auto at = [&](int r, int c2) -> double& { return MM[(size_t)r + (size_t)N * c2]; };
      const int dirs[5] = { 0, 1, 2, 1, 2 }; const double sgn[5] = { 1, 1, 1, -1, -1 };
      for (int a2 = 0; a2 < 5; a2++) {
        for (int b2 = 0; b2 < 5; b2++) for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) {
          double g = Gs(dirs[a2], dirs[b2], i, j);
          if (sgn[a2] * sgn[b2] < 0) g = -g;
          at(a2 * nc + i, b2 * nc + j) = g;
        }
        for (int i = 0; i < nc; i++) for (int k = 0; k < nl; k++) {      // Cd X L' and its transpose (ICH-QP:411-434)
          double g = CL[dirs[a2]][(size_t)i * nl + k];
          if (sgn[a2] < 0) g = -g;
          at(a2 * nc + i, 5 * nc + k) = g; at(5 * nc + k, a2 * nc + i) = g;
        }
      }
      for (int a2 = 0; a2 < nl; a2++) for (int b2 = 0; b2 < nl; b2++) at(5 * nc + a2, 5 * nc + b2) = LL[a2 + (size_t)nl * b2];
      for (int i = 0; i < nc; i++) at(i, i) = at(i, i) + m->cp_compliance;                          // ICH-QP:438-440
      for (int i = 0; i < nc; i++) { qq[i] = Cv[0][i]; qq[nc + i] = Cv[1][i]; qq[2*nc + i] = Cv[2][i]; qq[3*nc + i] = -Cv[1][i]; qq[4*nc + i] = -Cv[2][i]; }
      for (int k = 0; k < nl; k++) qq[5 * nc + k] = Lv[k];
      for (int i = 0; i < nc; i++) { for (int c2 = 0; c2 < nvars; c2++) at(nvars + i, c2) = at(i, c2); qq[nvars + i] = Cv[0][i]; }
      for (int k = 0; k < nl; k++) { for (int c2 = 0; c2 < nvars; c2++) at(nvars + nc + k, c2) = at(5 * nc + k, c2); qq[nvars + nc + k] = Lv[k]; }
      int row = nvars + nc + nl;
      for (int i = 0; i < nc; i++) {
        const double vel = std::sqrt(Cv[1][i] * Cv[1][i] + Cv[2][i] * Cv[2][i]);
        for (int j = 0; j < kh; j++) {
          const double theta = (double)j / (kh - 1) * M_PI_2;
          const double ct = std::cos(theta), st_ = std::sin(theta);
          at(row, i) = m->cp_mu_coulomb;
          at(row, nc + i) = -ct; at(row, 3*nc + i) = -ct;
          at(row, 2*nc + i) = -st_; at(row, 4*nc + i) = -st_;
          qq[row] = m->cp_mu_viscous * vel;
          row++;
        }
      }
      for (int r = nvars; r < N; r++) for (int c2 = 0; c2 < nvars; c2++) at(c2, r) = -at(r, c2);
      // solve_qp_work's chain (ICH-QP:157-233)
      Vec z; z.d.assign(aux->zbuf, aux->zbuf + aux->zbuf_cap); z.len = (unsigned)aux->zbuf_size;
      z.resize((unsigned)N);
      if ((int)z.size() == aux->zlast_size) for (int i = 0; i < N; i++) z[i] = aux->zlast[i];
      oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
      LCP lcp; lcp.rng = &rs;
      Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? ((trace_cap - trace_len > 0) ? trace_cap - trace_len : 0) : 0;
      lcp.trace = &tr;
      unsigned piv = 0;
      std::vector<double> z_in(z.d.begin(), z.d.begin() + N); const oracle_rand_t rs_in = rs;
      bool ok = lcp.lcp_fast_regularized(N, MM.data(), N, qq.data(), z, -20, 4, -8);
      piv += lcp.pivots;
      const unsigned piv_fast = lcp.pivots; const bool ok_fast = ok; unsigned piv_lemke = 0;
      if (!ok) { z.set_zero(); ok = lcp.lcp_lemke_regularized(N, MM.data(), N, qq.data(), z); piv += lcp.pivots; piv_lemke = lcp.pivots; }
      if (g_lcp_dump) {                                                // diagnostic (oracle_dbg_lcp_dump), same record as world.hpp's
        const int hdr[5] = { N, ok_fast ? 1 : 0, (int)piv_fast, (int)piv_lemke, ok ? 1 : 0 };
        std::fwrite(hdr, sizeof(int), 5, g_lcp_dump); std::fwrite(&rs_in, sizeof(rs_in), 1, g_lcp_dump);
        std::fwrite(MM.data(), 8, (size_t)N * N, g_lcp_dump); std::fwrite(qq.data(), 8, N, g_lcp_dump); std::fwrite(z_in.data(), 8, N, g_lcp_dump);
        std::fflush(g_lcp_dump);
      }
      trace_len += tr.len;
      std::memcpy(aux->rng, &rs, sizeof(rs));
      lcp_account(N, piv);
      if (!ok) { aux->status |= MH_WORLD_LCP_FAILED; return false; }   // LCPSolverException
      aux->zlast_size = N;
      for (int i = 0; i < N; i++) { aux->zlast[i] = z[i]; aux->zbuf[i] = z[i]; }
      if (aux->zbuf_cap < N) aux->zbuf_cap = N;
      aux->zbuf_size = nvars;                                          // z repacked to the epd layout = its first N_VARS entries (ICH-QP:236-250)
      return true;
    };
    auto from_stacked = [&]() {                                        // update_from_stacked(q, z) (UCPD:218-228)
      for (int i = 0; i < nc; i++) {
        cn[i] = aux->zbuf[i];
        double sv2 = aux->zbuf[nc + i];   sv2 = sv2 - aux->zbuf[3*nc + i]; csv[i] = sv2;
        double tv2 = aux->zbuf[2*nc + i]; tv2 = tv2 - aux->zbuf[4*nc + i]; ctv[i] = tv2;
      }
      for (int k = 0; k < nl; k++) l[k] = aux->zbuf[5 * nc + k];
    };
    if (noslip) {
      if (!solve_noslip()) return;
      apply(); update_vels();
      const double minv = minv_of();
      bool changed = false;                                            // apply_restitution(q) (ICH:497-525)
      for (int i = 0; i < nc; i++) { cn[i] = cn[i] * m->cp_epsilon; if (!changed && cn[i] > A_NEAR_ZERO) changed = true; }
      for (int k = 0; k < nl; k++) { l[k] = l[k] * m->limit_restitution[idx[k]]; if (!changed && l[k] > A_NEAR_ZERO) changed = true; }
      if (changed) {
        for (int i = 0; i < nc; i++) { csv[i] = 0.0; ctv[i] = 0.0; }
        apply(); update_vels();
        const double minv_plus = minv_of();
        // ICH:284-291 would re-solve and then read the Drumwright-Shell solver's _z, which this path never sized
        if (minv_plus < 0.0 && minv_plus < minv - A_NEAR_ZERO) aux->status |= MH_WORLD_UNSUPPORTED;
      }
    } else {                                                           // apply_model_to_connected_constraints (ICH:530-626)
      if (!solve_qp()) return;
      from_stacked(); apply(); update_vels();
      const double minv = minv_of();
      bool changed = false;                                            // apply_restitution(q, z) (ICH:470-491): cn and l entries of z only
      for (int i = 0; i < nc; i++) { aux->zbuf[i] = aux->zbuf[i] * m->cp_epsilon; if (!changed && aux->zbuf[i] > A_NEAR_ZERO) changed = true; }
      for (int k = 0; k < nl; k++) { double& zl = aux->zbuf[5 * nc + k]; zl = zl * m->limit_restitution[idx[k]]; if (!changed && zl > A_NEAR_ZERO) changed = true; }
      if (changed) {
        from_stacked(); apply(); update_vels();                        // the tangential impulses are applied again in full, as the reference does
        const double minv_plus = minv_of();
        if (minv_plus < 0.0 && minv_plus < minv - A_NEAR_ZERO) {       // ICH:591-600: second solve on the updated C v vectors
          if (!solve_qp()) return;
          from_stacked(); apply();
        }
      }
    }
    link_velocities(V);                                              // ICH:157-167
    for (int i = 0; i < nc; i++) if (point_vel_dir(V[cs[i].link], cs[i].p, cs[i].n) < -A_NEAR_ZERO) aux->status |= MH_WORLD_IMPACT_TOL;
    for (int k = 0; k < nl; k++) { const double v = upper[k] ? -qd[idx[k]] : qd[idx[k]]; if (v < -A_NEAR_ZERO) aux->status |= MH_WORLD_IMPACT_TOL; }
  }

  // TimeSteppingSimulator::do_mini_step (TSS:114-222) for a body with sphere primitives
  double do_mini_step(double dt) {
    double qsave[NJ], V[NJ][6];
    for (int i = 0; i < nj; i++) qsave[i] = q[i];
    double h = 0.0;
    unsigned long guard = 0;
    while (h < dt) {
      if (++guard > MH_CA_HARD_CAP) { aux->status |= MH_WORLD_STALLED; break; }
      kinematics(); link_velocities(V);
      double CA = A_INF;                                             // calc_next_CA_Euler_step (TSS:272-331): every (sphere, plane) pair -- the plane's DummyBV is infinite
      for (int s = 0; s < m->nspheres; s++) { const double e = CA_step(s, V); CA = (e < CA) ? e : CA; }
      if (CA <= 0.0) break;
      double tc = (m->min_step_size > CA) ? m->min_step_size : CA;
      tc = ((dt - h) < tc) ? (dt - h) : tc;
      for (int i = 0; i < nj; i++) { double qn = qd[i] * (h + tc); qn = qn + qsave[i]; q[i] = qn; }
      h += tc;
    }
    double qdd[NJ];
    // (an exception -- here: a generalized inertia that is not positive definite -- ends the run: nothing below happens)
    if (!((m->algorithm == MH_ARTIC_FSAB) ? fwd_dyn_aba(nullptr, qdd) : fwd_dyn(nullptr, qdd))) { aux->status |= MH_WORLD_LCP_FAILED; return h; }
    for (int i = 0; i < nj; i++) qd[i] = qd[i] + qdd[i] * h;                                       // TSS:182-192
    std::vector<AContact> cs;                                        // find_unilateral_constraints (CSim:488-537)
    for (int s = 0; s < m->nspheres; s++) {
      double ctr[3], cp[3]; sphere_center(s, ctr); to_plane(ctr, cp);
      const double dist = cp[1] + (-1.0 * m->sphere_radius[s]);
      AContact c;
      if (dist < m->contact_dist_thresh && find_contact(s, m->contact_dist_thresh, c)) cs.push_back(c);
    }
    handle_impacts(cs);
    if (aux->status & MH_WORLD_LCP_FAILED) return h;                 // thrown out of the impact handler: current_time += h (TSS:215) is not reached
    aux->time += h; aux->mini_steps++;
    return h;
  }

  // ---- ConstraintStabilization::stabilize for this body: contact rows of the link spheres and joint-limit rows (CStab:167-254, 257-304,
  // 306-345, 431-441, 705-904, 932-970, 1056-1216) ----
  // evaluate_unilateral_constraints (CStab:88-131): the pairwise distances (none without collision geometry), then for every
  // joint j of body i the limit slacks of joints[i] -- the body's index in the simulator's list, not j (CStab:117, kept): with one
  // articulated body per world that is joint 0, nj times
  // With sphere primitives on the links the pairwise distances come first (sim->calc_pairwise_distances over the simulator's pair list:
  // every (sphere, plane) pair, the plane's bounding volume being infinite), each the sphere's signed distance to the plane at the
  // CURRENT q (SpherePrimitive.cpp:104-136 / PlanePrimitive.cpp:385-411, as CA_step reads it).
  double cstab_eval(std::vector<double>& uC) {
    double vio = A_INF;
    uC.clear();
    if (m->nspheres > 0) {
      kinematics();
      for (int s = 0; s < m->nspheres; s++) {
        double ctr[3], cp[3]; sphere_center(s, ctr); to_plane(ctr, cp);
        uC.push_back(cp[1] + (-1.0 * m->sphere_radius[s])); vio = (uC.back() < vio) ? uC.back() : vio;
      }
    }
    for (int j = 0; j < nj; j++) {
      uC.push_back((m->hilimit[0] - q[0]) - 0.0); vio = (uC.back() < vio) ? uC.back() : vio;      // hilimit - q - tare, tare = 0
      uC.push_back((q[0] + 0.0) - m->lolimit[0]); vio = (uC.back() < vio) ? uC.back() : vio;
    }
    return vio;
  }
  double cstab_eval_at(double t, unsigned i, const double* dq, const double* qv) {                // CStab:1281-1298
    std::vector<double> uC;
    for (int k = 0; k < nj; k++) { double v = dq[k] * t; v = v + qv[k]; q[k] = v; }
    cstab_eval(uC);
    return uC[i];
  }
  static double sign2(double x, double y) { return (y > 0.0) ? std::fabs(x) : -std::fabs(x); }
  double cstab_ridders(double x1, double x2, double fl, double fh, unsigned idx, const double* dq, const double* qv) {   // CStab:1322-1379
    const double TOL = 1e-4;
    double ans = A_INF, fm, fnew, s2, xh, xl, xm, xnew;
    if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
      xl = x1; xh = x2;
      for (unsigned j = 0; j < 25; j++) {
        xm = 0.5 * (xl + xh);
        fm = cstab_eval_at(xm, idx, dq, qv);
        s2 = std::sqrt(fm * fm - fl * fh);
        if (s2 == 0.0) return ans;
        xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s2);
        ans = xnew;
        fnew = cstab_eval_at(ans, idx, dq, qv);
        if (std::fabs(fnew) < TOL && fnew >= 0.0) return xnew;
        if (sign2(fm, fnew) != fm) { xl = xm; fl = fm; xh = ans; fh = fnew; }
        else if (sign2(fl, fnew) != fl) { xh = ans; fh = fnew; }
        else if (sign2(fh, fnew) != fh) { xl = ans; fl = fnew; }
      }
    } else {
      if (fl == 0.0) return x1;
      if (fh == 0.0) return x2;
    }
    return 0.0;
  }
  bool cstab_update_q(const double* dq, double* qv) {                                            // CStab:1056-1216 (no implicit joints: C is empty)
    std::vector<double> uC, uC_old;
    cstab_eval(uC_old);
    for (int k = 0; k < nj; k++) { double v = dq[k]; v = v + qv[k]; q[k] = v; }
    cstab_eval(uC);
    std::vector<char> br(uC.size(), 0);
    for (size_t i = 0; i < uC.size(); i++) br[i] = ((uC_old[i] < 0.0 && uC[i] > 0.0) || (uC_old[i] > 0.0 && uC[i] < 0.0)) ? 1 : 0;
    double t = 1.0;
    for (size_t i = 0; i < br.size(); i++) {
      if (!br[i]) continue;
      const double root = cstab_ridders(0, t, uC_old[i], uC[i], (unsigned)i, dq, qv);
      if (root > 0.0 && root < 1.0) t = (root < t) ? root : t;
    }
    for (int k = 0; k < nj; k++) { double v = dq[k] * t; v = v + qv[k]; q[k] = v; }
    cstab_eval(uC);
    const double BETA = 0.6;
    while (true) {
      bool stop = true;
      for (size_t i = 0; i < br.size(); i++) if (!br[i] && uC[i] < 0.0 && uC_old[i] > uC[i]) { stop = false; break; }
      if (stop) break;                                             // bilateral_cvio = 0 < bilateral_eps
      t *= BETA;
      if (t < A_NEAR_ZERO) return false;
      for (int k = 0; k < nj; k++) { double v = dq[k] * t; v = v + qv[k]; q[k] = v; }
      cstab_eval(uC);
    }
    for (int k = 0; k < nj; k++) qv[k] = q[k];
    return true;
  }
  void stabilize() {
    if (m->cstab_max_iterations == 0) return;
    double qd_save[NJ], qv[NJ], dq[NJ];
    for (int i = 0; i < nj; i++) { qd_save[i] = qd[i]; qv[i] = q[i]; }
    std::vector<double> uC;
    double max_uvio = cstab_eval(uC);
    unsigned iterations = 0;
    while (max_uvio < m->cstab_eps) {
      if (iterations == (unsigned)m->cstab_max_iterations) break;
      if (iterations == MH_CSTAB_HARD_CAP) { aux->status |= MH_WORLD_STALLED; break; }
      for (int i = 0; i < nj; i++) { qd[i] = 0.0; dq[i] = 0.0; }
      // compute_problem_data (CStab:347-492): one contact per (sphere, plane) pair of the broad phase -- a synthetic one "between
      // separated bodies" when the signed distance is at least NEAR_ZERO (add_contact_constraints, CStab:306-331: point = the closest
      // point on the sphere, normal = from it towards the closest point on the plane, as the reference computes p2 - p1), the contact
      // of find_contacts otherwise (:337); then a row for every FINITE limit (CStab:257-304), upper before lower per joint; one
      // island (every constraint touches the one body)
      std::vector<AContact> cs;
      if (m->nspheres > 0) {
        kinematics();
        for (int s = 0; s < m->nspheres; s++) {
          double ctr[3], cp[3]; sphere_center(s, ctr); to_plane(ctr, cp);
          const double low = cp[1] + (-1.0 * m->sphere_radius[s]);
          AContact c;
          if (low >= A_NEAR_ZERO) {
            double on_plane[3]; from_plane(cp[0], 0.0, cp[2], on_plane); from_plane(cp[0], low, cp[2], c.p);
            const double d[3] = { on_plane[0] - c.p[0], on_plane[1] - c.p[1], on_plane[2] - c.p[2] };
            const double len = std::sqrt((d[0]*d[0] + d[1]*d[1]) + d[2]*d[2]);
            for (int k = 0; k < 3; k++) c.n[k] = d[k] / len;
            c.s = s; c.link = m->sphere_link[s]; c.dist = low;
            orthonormal_basis(c.n, c.sv, c.tv);
            cs.push_back(c);
          } else if (find_contact(s, A_NEAR_ZERO, c)) cs.push_back(c);
        }
      }
      const int nc = (int)cs.size();
      int idx[2 * NJ]; bool upper[2 * NJ]; double viol[2 * NJ]; int nl = 0;
      for (int i = 0; i < nj; i++) {
        if (m->hilimit[i] < A_INF) { idx[nl] = i; upper[nl] = true; viol[nl] = (m->hilimit[i] - q[i]) - 0.0; nl++; }
        if (m->lolimit[i] > -A_INF) { idx[nl] = i; upper[nl] = false; viol[nl] = (q[i] + 0.0) - m->lolimit[i]; nl++; }
      }
      if (nc + nl > 0) {
        const int n = nc + nl;
        if (n > MH_LCP_MAX_N_WAVE) { aux->status |= MH_WORLD_UNSUPPORTED; break; }
        kinematics(); crba();                                        // compute_X at the CURRENT configuration (ICH:1600-1607)
        std::vector<double> X(H, H + nj * nj);
        if (!inverse_spd(nj, X.data(), nj)) { aux->status |= MH_WORLD_LCP_FAILED; break; }
        // set_unilateral_constraint_data (CStab:705-904): Cn rows [n, r x n] . calc_jacobian(link) at the link's COM (add_contact_to_Jacobian,
        // CStab:906-929: the plane's body is disabled and adds nothing), X_CnT = (Cn X)', Cn_X_CnT = Cn X_CnT, Cn_X_LT = Cn X_LT with
        // X_LT's columns the UNSIGNED rows of X (compute_limit_components, ICH:1755-1781) -- the products of handle_impacts' normal rows
        std::vector<double> C((size_t)nc * nj, 0.0), XC((size_t)nc * nj, 0.0);
        for (int i = 0; i < nc; i++) {
          const int l = cs[i].link;
          double rc[3], com[3], r[3], J[6 * NJ], w[6];
          artic::mat3vec(R[l], m->com[l], rc);
          for (int k = 0; k < 3; k++) { com[k] = x[l][k] + rc[k]; r[k] = cs[i].p[k] - com[k]; }
          jacobian(l, com, J);
          artic::cross3(r, cs[i].n, w + 3);
          for (int k = 0; k < 3; k++) w[k] = cs[i].n[k];
          for (int j = 0; j < nj; j++) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + w[k] * J[k * nj + j]; C[(size_t)i * nj + j] = acc; }
        }
        for (int i = 0; i < nc; i++) for (int c = 0; c < nj; c++) {
          double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[(size_t)i * nj + k] * X[k * nj + c];
          XC[(size_t)i * nj + c] = acc;
        }
        // determine_dq (CStab:932-970): MM = [Cn_X_CnT  Cn_X_LT; Cn_X_LT'  L_X_LT], qq = [Cn_v; L_v]
        std::vector<double> MM((size_t)n * n), Lv(n);
        for (int i = 0; i < nc; i++) for (int j = 0; j < nc; j++) { double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[(size_t)i * nj + k] * XC[(size_t)j * nj + k]; MM[i + (size_t)n * j] = acc; }
        for (int i = 0; i < nc; i++) for (int k2 = 0; k2 < nl; k2++) {
          double acc = 0.0; for (int k = 0; k < nj; k++) acc = acc + C[(size_t)i * nj + k] * X[idx[k2] * nj + k];
          MM[i + (size_t)n * (nc + k2)] = acc; MM[(nc + k2) + (size_t)n * i] = acc;
        }
        for (int a = 0; a < nl; a++) for (int b = a; b < nl; b++) { const double e = X[idx[a] * nj + idx[b]]; MM[(nc + a) + (size_t)n * (nc + b)] = e; MM[(nc + b) + (size_t)n * (nc + a)] = e; }   // ICH:1763-1771 (no signs)
        for (int i = 0; i < nc; i++) Lv[i] = (cs[i].dist - std::fabs(m->cstab_eps)) - A_NEAR_ZERO;      // CStab:431
        for (int k = 0; k < nl; k++) Lv[nc + k] = (viol[k] - std::fabs(m->cstab_eps)) - A_NEAR_ZERO;  // CStab:434-441
        const int nl_only = nl; (void)nl_only;
        Vec z;                                                       // determine_dq's local z: size 0 -> cold lcp_fast (CStab:954)
        oracle_rand_t rs; std::memcpy(&rs, aux->rng, sizeof(rs));
        LCP lcp; lcp.rng = &rs;
        Trace tr; tr.buf = trace ? trace + trace_len : nullptr; tr.cap = trace ? ((trace_cap - trace_len > 0) ? trace_cap - trace_len : 0) : 0;
        lcp.trace = &tr;
        unsigned piv = 0;
        bool ok = lcp.lcp_fast(n, MM.data(), n, Lv.data(), z, -1.0);
        piv += lcp.pivots;
        if (!ok) { ok = lcp.lcp_lemke_regularized(n, MM.data(), n, Lv.data(), z); piv += lcp.pivots; }
        trace_len += tr.len;
        std::memcpy(aux->rng, &rs, sizeof(rs));
        lcp_account(n, piv); aux->stab_rows += (unsigned long long)n;
        // update_from_stacked(pd, z): cn, l = z (whatever it holds), dv = X_CnT cn + X_LT ls, v += dv (ICH:298-397; the tangential
        // products have no columns here); dq = the Euler velocities
        std::vector<double> dv(nj, 0.0);
        if (nc > 0) for (int r = 0; r < nj; r++) { double acc = 0.0; for (int i = 0; i < nc; i++) { const double ci = (i < (int)z.size()) ? z[i] : 0.0; acc = acc + XC[(size_t)i * nj + r] * ci; } dv[r] = acc; }
        { std::vector<double> t2(nj, 0.0);
          for (int k = 0; k < nl; k++) { const double lk = (nc + k < (int)z.size()) ? z[nc + k] : 0.0; const double ls = upper[k] ? -lk : lk; for (int r = 0; r < nj; r++) t2[r] = t2[r] + ls * X[idx[k] * nj + r]; }
          for (int r = 0; r < nj; r++) dv[r] = (nc > 0) ? dv[r] + t2[r] : t2[r]; }
        for (int r = 0; r < nj; r++) { qd[r] = qd[r] + dv[r]; dq[r] = qd[r]; }
      }
      if (!cstab_update_q(dq, qv)) { aux->status |= MH_WORLD_STAB_FAILED; break; }
      max_uvio = cstab_eval(uC);
      iterations++;
      aux->stab_iters++;
    }
    for (int i = 0; i < nj; i++) { qd[i] = qd_save[i]; q[i] = qv[i]; }
  }

  // TimeSteppingSimulator::step (TSS:52-111).  Without collision geometry: one mini-step of dt.
  // An exception of the impact handler, of calc_fwd_dyn or of compute_X (LCPSolverException, "Unable to solve constraint LCP!", a factorisation that fails) is caught
  // nowhere up to main(): the run of this simulator is over (as in world.hpp).  The state stays where the throw left it -- positions integrated, velocities
  // without the impulses, time and counters without this mini-step / step -- and every later step() returns at once.
  void step(double dt) {
    if (aux->status & MH_WORLD_LCP_FAILED) return;
    if (m->nspheres > 0 || force_general) {
      // a world whose contacts need a model this build does not have (the reference would throw out of step()), or whose step ran
      // into the mini-step cap (the reference would never return from step()), is frozen; without this it would burn the cap on
      // every following step, its impacting contact never resolved
      const int FROZEN = MH_WORLD_UNSUPPORTED | MH_WORLD_STALLED;
      if (m->nspheres > 0 && (aux->status & FROZEN)) return;
      double h = 0.0; unsigned guard = 0;
      while (h < dt) {
        h += do_mini_step(dt - h);
        if (aux->status & MH_WORLD_LCP_FAILED) return;
        if (m->nspheres > 0 && (aux->status & FROZEN)) break;
        if (++guard > 100000u) { aux->status |= MH_WORLD_STALLED; break; }
      }
      stabilize();                                                   // TSS:97
      if (aux->status & MH_WORLD_LCP_FAILED) return;
      aux->steps++;
      return;
    }
    for (int i = 0; i < nj; i++) { double qn = qd[i] * dt; qn = qn + q[i]; q[i] = qn; }        // positions with the OLD velocity (TSS:156-164)
    double qdd[NJ];
    if (!((m->algorithm == MH_ARTIC_FSAB) ? fwd_dyn_aba(nullptr, qdd) : fwd_dyn(nullptr, qdd))) { aux->status |= MH_WORLD_LCP_FAILED; return; }
    for (int i = 0; i < nj; i++) qd[i] = qd[i] + qdd[i] * dt;                                      // TSS:182-192
    handle_limits();
    if (aux->status & MH_WORLD_LCP_FAILED) return;
    aux->time += dt; aux->mini_steps++;
    stabilize();                                                   // TSS:97
    if (aux->status & MH_WORLD_LCP_FAILED) return;
    aux->steps++;
  }
  bool force_general = false;     // tests: run a body without spheres through do_mini_step / handle_impacts (must agree with the path above)
};

}  // namespace oracle
#endif
