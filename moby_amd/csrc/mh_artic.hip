// mh_artic.hip -- many-worlds stepper for fixed-base articulated bodies (include/moby_hip_artic.h; BASELINE config 5).
//
// One wavefront per world.  Per step (TimeSteppingSimulator::do_mini_step with no collision geometry, TSS:114-222):
//   q += dt qd (old velocity)  ->  link frames (chain walk, 12 lanes per link), motion subspaces and spatial inertias
//   about the world origin (lane = link)  ->  bias C(q, qd) by recursive Newton-Euler (6 lanes per link, gravity as a
//   base acceleration)  ->  composite inertias (36 lanes) and H = S' Ic S (lane = matrix entry)  ->  Cholesky
//   (lane = row) and  qdd = H^-1 (tau - C)  ->  qd += dt qdd  ->  joint limits (ballot masks -> the reference's
//   constraint order, ArticulatedBody.inl:9-43)  ->  X = H^-1 (lane = column), the limit LCP  L X L' l + L v >= 0  through
//   the wave solver of mh_lcp_wave.h (lcp_fast on the persistent _v, then the Lemke ladder: ICH:1239-1283)  ->  impulses,
//   restitution (ICH:298-525).
// H, its factor, H^-1, the LCP and all link quantities live in LDS (dynamic, sized by the joint count; the link quantities
// and the limit LCP share one region: 10 KB at 10 joints => 16 worlds per CU); HBM sees q, qd and the aux record once per launch.
// The dynamics algorithm is Featherstone's (Ravelin's source is not in the reference tree: SURVEY F2); operation
// order = oracle/artic.hpp, checked bit for bit.  sin / cos: the same explicit kernel as the oracle (no libm call).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>
#include "../../include/moby_hip_artic.h"
#include "mh_host.h"
#include "mh_lcp_wave.h"

namespace mh { namespace artic {

constexpr int NJ = MH_ARTIC_MAX_JOINTS;
constexpr int NLMAX = MH_NOSLIP_MAX;
constexpr double NEAR_ZERO_ = 1.4901161193847656e-08;

struct Model {            // mh_artic_model + what the kernel wants precomputed: ancestor masks
  mh_artic_model m;
  unsigned anc[NJ];       // bit j: joint j lies on the path from joint i to the base (i itself included)
  double fcos[32], fsin[32];   // friction polygon of the Drumwright-Shell model: cos / sin(j / (nk/2 - 1) pi/2) by the HOST's libm (ICH-QP:462-470)
};

__constant__ Pow10Table c_pow10a;

MH_DEV void sincos_kernel(double x, double& s, double& c)
{
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double kf = floor(x * invpio2 + 0.5);
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
  const long long k = (long long)kf;
  const int n = (int)(((k % 4) + 4) % 4);
  if (n == 0) { s = ks; c = kc; } else if (n == 1) { s = kc; c = -ks; } else if (n == 2) { s = -ks; c = -kc; } else { s = -kc; c = ks; }
}

MH_DEV double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
MH_DEV void cross3(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
MH_DEV void mat3mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3*i+j] = (A[3*i] * B[j] + A[3*i+1] * B[3+j]) + A[3*i+2] * B[6+j];
}
MH_DEV void mat3vec(const double* A, const double* v, double* y) { for (int i = 0; i < 3; i++) y[i] = (A[3*i] * v[0] + A[3*i+1] * v[1]) + A[3*i+2] * v[2]; }
MH_DEV double dot6(const double* a, const double* b) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + a[k] * b[k]; return acc; }
// component k of the spatial cross products ([angular; linear]), operands in LDS
MH_DEV double cross_c(const double* a, const double* b, int k) { const int k1 = (k + 1) % 3, k2 = (k + 2) % 3; return a[k1] * b[k2] - a[k2] * b[k1]; }
MH_DEV double crm_c(const double* v, const double* m, int k) { return (k < 3) ? cross_c(v, m, k) : cross_c(v, m + 3, k - 3) + cross_c(v + 3, m, k - 3); }
MH_DEV double crf_c(const double* v, const double* f, int k) { return (k < 3) ? cross_c(v, f, k) + cross_c(v + 3, f + 3, k) : cross_c(v, f + 3, k - 3); }

// the LDS image of one world (doubles), nj = number of joints
struct Lay {
  int nj;
  int q, qd, qdd, C, R, x, Rl, tl, S, I6, v, a, f, F, Iv, H, L, X, MM, A, art, Lv, l, idx, total;
  // q, qd, qdd, C and H, L, X live for the whole step; the link quantities of dynamics() (R .. Iv) and the limit LCP's
  // storage (MM .. idx) are never alive together -- handle_limits reads q, qd, L, X only and the next dynamics() call
  // rebuilds everything from q, qd -- so they share one region: 10 KB per world at 10 joints = 16 worlds per CU
  MH_DEV Lay(int n, int nlcap = NLMAX) : nj(n) {     // nlcap: rows the limit LCP's storage takes (NLSTAB in the stabilising kernel)
    int o = 0;
    q = o; o += n; qd = o; o += n; qdd = o; o += n; C = o; o += n;
    H = o; o += n * n; L = o; o += n * n; X = o; o += n * n;
    const int u = o;
    R = o; o += 9 * n; x = o; o += 3 * n;
    S = o; o += 6 * n; I6 = o; o += 36 * n; v = o; o += 6 * n; a = o; o += 6 * n; f = o; o += 6 * n; F = o; o += 6 * n; Iv = o; o += 12;
    // the local transforms live only from kin_inertia's first loop to the end of its chain walk, before anything writes v or a (RNEA / the articulated-body
    // recursion come after): they share those 12 n doubles (round 5: 1252 -> 1132 doubles at 10 joints = 18 images per CU instead of 16)
    Rl = v; tl = v + 9 * n;
    const int end_dyn = o;
    o = u;
    MM = o; o += nlcap * nlcap; A = o; o += nlcap * nlcap; art = o; o += nlcap; Lv = o; o += nlcap; l = o; o += nlcap;
    idx = o; o += nlcap;           // ints stored as doubles' slots (one int each, low half)
    total = (o > end_dyn) ? o : end_dyn;
  }
};
constexpr int NLSTAB = 2 * NJ;      // the stabiliser's LCP has a row for every finite limit (CStab:257-304): up to two per joint
static size_t lds_bytes(int nj, int nlcap = NLMAX) {
  const int n = nj;
  const int dyn = 78 * n + 12, lim = 2 * nlcap * nlcap + 4 * nlcap;
  return sizeof(double) * (size_t)(4 * n + 3 * n * n + (dyn > lim ? dyn : lim));
}

// link frames, motion subspaces and spatial inertias about the world origin for the q in LDS (oracle Artic::kinematics)
// PACK worlds per wavefront: PACK = 1 -- the whole wave works on the image at g; PACK = 2 -- lanes 0-31 and 32-63 each work on THEIR
// world's image (g is then a per-lane pointer, `lane` the lane's index within its half): the same instruction stream steps two worlds
template <int PACK = 1>
MH_DEV void kin_inertia(const Model& M, const Lay& Y, double* g)
{
  const mh_artic_model& m = M.m;
  const int nj = Y.nj, lane = (PACK == 1) ? lane_id() : (lane_id() & (64 / PACK - 1));
  // local transforms: lane = link
  if (lane < nj) {
    const int i = lane;
    const double* ax = m.axis[i];
    double Rl[9], tl[3];
    const double qi = g[Y.q + i];
    if (m.jtype[i] == MH_JOINT_REVOLUTE) {
      double s, c; sincos_kernel(qi, s, c);
      const double t = 1.0 - c;
      const double K[9] = { 0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0 };
      double Rq[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rq[3*a+b] = (((a == b) ? c : 0.0) + (t * ax[a]) * ax[b]) + s * K[3*a+b];
      mat3mul(m.Rrel[i], Rq, Rl);
      for (int k = 0; k < 3; k++) tl[k] = m.trel[i][k];
    } else {
      for (int k = 0; k < 9; k++) Rl[k] = m.Rrel[i][k];
      double d[3], Rd[3];
      for (int k = 0; k < 3; k++) d[k] = ax[k] * qi;
      mat3vec(m.Rrel[i], d, Rd);
      for (int k = 0; k < 3; k++) tl[k] = m.trel[i][k] + Rd[k];
    }
    for (int k = 0; k < 9; k++) g[Y.Rl + 9 * i + k] = Rl[k];
    for (int k = 0; k < 3; k++) g[Y.tl + 3 * i + k] = tl[k];
  }
  wave_sync();
  // chain walk: lanes 0..8 = entries of R_i, lanes 9..11 = x_i
  for (int i = 0; i < nj; i++) {
    const int p = m.parent[i];
    if (lane < 9) {
      const int a = lane / 3, b = lane - 3 * a;
      double e;
      if (p < 0) e = g[Y.Rl + 9 * i + lane];
      else { const double* Rp = g + Y.R + 9 * p; const double* Rl = g + Y.Rl + 9 * i; e = (Rp[3*a] * Rl[b] + Rp[3*a+1] * Rl[3+b]) + Rp[3*a+2] * Rl[6+b]; }
      g[Y.R + 9 * i + lane] = e;
    } else if (lane < 12) {
      const int k = lane - 9;
      double e;
      if (p < 0) e = g[Y.tl + 3 * i + k];
      else { const double* Rp = g + Y.R + 9 * p; const double* tl = g + Y.tl + 3 * i; e = g[Y.x + 3 * p + k] + ((Rp[3*k] * tl[0] + Rp[3*k+1] * tl[1]) + Rp[3*k+2] * tl[2]); }
      g[Y.x + 3 * i + k] = e;
    }
    wave_sync();
  }
  // motion subspace and spatial inertia about the world origin: lane = link
  if (lane < nj) {
    const int i = lane;
    double R[9], x[3];
    for (int k = 0; k < 9; k++) R[k] = g[Y.R + 9 * i + k];
    for (int k = 0; k < 3; k++) x[k] = g[Y.x + 3 * i + k];
    double aw[3]; mat3vec(R, m.axis[i], aw);
    double* S = g + Y.S + 6 * i;
    if (m.jtype[i] == MH_JOINT_REVOLUTE) { double xa[3]; cross3(x, aw, xa); for (int k = 0; k < 3; k++) { S[k] = aw[k]; S[3+k] = xa[k]; } }
    else for (int k = 0; k < 3; k++) { S[k] = 0.0; S[3+k] = aw[k]; }
    double rc[3], r[3]; mat3vec(R, m.com[i], rc);
    for (int k = 0; k < 3; k++) r[k] = x[k] + rc[k];
    double T[9], Iw[9];
    mat3mul(R, m.inertia[i], T);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Iw[3*a+b] = (T[3*a] * R[3*b] + T[3*a+1] * R[3*b+1]) + T[3*a+2] * R[3*b+2];
    Iw[1] = Iw[3]; Iw[2] = Iw[6]; Iw[5] = Iw[7];
    const double mass = m.mass[i];
    const double rr = dot3(r, r);
    const double rx[9] = { 0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0 };
    double* I6 = g + Y.I6 + 36 * i;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      I6[6*a+b] = Iw[3*a+b] + mass * (((a == b) ? rr : 0.0) - r[a] * r[b]);
      I6[6*a+3+b] = mass * rx[3*a+b];
      I6[6*(3+a)+b] = mass * rx[3*b+a];
      I6[6*(3+a)+3+b] = (a == b) ? mass : 0.0;
    }
  }
  wave_sync();
}

// kinematics + spatial inertias + bias + H + Cholesky + qdd for the q / qd in LDS.  Returns false if H is not PD.
template <int PACK = 1>
MH_DEV bool dynamics(const Model& M, const Lay& Y, double* g, const double* tau_w)
{
  const mh_artic_model& m = M.m;
  constexpr int STR = 64 / PACK;
  const int nj = Y.nj, lane = (PACK == 1) ? lane_id() : (lane_id() & (STR - 1));
  kin_inertia<PACK>(M, Y, g);
  // recursive Newton-Euler with qdd = 0 (the links' OWN inertias): lanes 0..5 = spatial components
  for (int i = 0; i < nj; i++) {
    const int p = m.parent[i];
    const double* S = g + Y.S + 6 * i;
    const double qdi = g[Y.qd + i];
    double* vj = g + Y.Iv;                        // S_i qd_i (6), then I v (6)
    if (lane < 6) { const double e = S[lane] * qdi; vj[lane] = e; g[Y.v + 6 * i + lane] = (p < 0) ? e : g[Y.v + 6 * p + lane] + e; }
    wave_sync();
    if (lane < 6) {
      const double cv = crm_c(g + Y.v + 6 * i, vj, lane);
      const double ap = (p < 0) ? ((lane < 3) ? 0.0 : -m.gravity[lane - 3]) : g[Y.a + 6 * p + lane];
      g[Y.a + 6 * i + lane] = ap + cv;
    }
    wave_sync();
    double Ia = 0.0;
    if (lane < 6) {
      const double* I6 = g + Y.I6 + 36 * i + 6 * lane;
      const double* a = g + Y.a + 6 * i; const double* v = g + Y.v + 6 * i;
      double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + I6[k] * a[k];
      Ia = acc;
      acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + I6[k] * v[k];
      g[Y.Iv + 6 + lane] = acc;
    }
    wave_sync();
    if (lane < 6) g[Y.f + 6 * i + lane] = Ia + crf_c(g + Y.v + 6 * i, g + Y.Iv + 6, lane);
    wave_sync();
  }
  for (int i = nj - 1; i >= 0; i--) {
    const int p = m.parent[i];
    if (lane == 6) g[Y.C + i] = dot6(g + Y.S + 6 * i, g + Y.f + 6 * i);
    if (p >= 0 && lane < 6) g[Y.f + 6 * p + lane] = g[Y.f + 6 * p + lane] + g[Y.f + 6 * i + lane];
    wave_sync();
  }
  // composite inertias, in place: lanes 0..35 = entries
  for (int i = nj - 1; i >= 0; i--) {
    const int p = m.parent[i];
    if (p >= 0) for (int e = lane; e < 36; e += STR) g[Y.I6 + 36 * p + e] = g[Y.I6 + 36 * p + e] + g[Y.I6 + 36 * i + e];
    wave_sync();
  }
  // F_i = Ic_i S_i: lanes (i, r)
  for (int e = lane; e < 6 * nj; e += STR) {
    const int i = e / 6, r = e - 6 * i;
    const double* I6 = g + Y.I6 + 36 * i + 6 * r; const double* S = g + Y.S + 6 * i;
    double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + I6[k] * S[k];
    g[Y.F + e] = acc;
  }
  wave_sync();
  // H(i, j) = S_j' F_i for j on i's path to the base (and its mirror); 0 elsewhere: lanes = entries
  for (int e = lane; e < nj * nj; e += STR) {
    const int i = e / nj, j = e - nj * i;
    double h = 0.0;
    if ((M.anc[i] >> j) & 1u) h = dot6(g + Y.S + 6 * j, g + Y.F + 6 * i);
    else if ((M.anc[j] >> i) & 1u) h = dot6(g + Y.S + 6 * i, g + Y.F + 6 * j);
    g[Y.H + e] = h;
  }
  wave_sync();
  // dpotf2('L') on a copy: lane = row
  for (int e = lane; e < nj * nj; e += STR) g[Y.L + e] = g[Y.H + e];
  wave_sync();
  double* L = g + Y.L;                             // symmetric: row-major == column-major; L(i, k) at L[i + nj k]
  bool pd = true;
  for (int j = 0; j < nj; j++) {
    if (lane == j) {
      double ajj = L[j + nj * j];
      for (int k = 0; k < j; k++) ajj = ajj - L[j + nj * k] * L[j + nj * k];
      g[Y.Iv] = ajj;
    }
    wave_sync();
    const double ajj0 = g[Y.Iv];
    if (!(ajj0 > 0.0)) { pd = false; if (PACK == 1) break; }      // (packed: the other world of the wave goes on; this one computes on, its result is dropped)
    const double ajj = sqrt(ajj0);
    if (lane == j) L[j + nj * j] = ajj;
    if (lane > j && lane < nj) {
      double s = L[lane + nj * j];
      for (int k = 0; k < j; k++) s = s - L[lane + nj * k] * L[j + nj * k];
      L[lane + nj * j] = s / ajj;
    }
    wave_sync();
  }
  if (PACK == 1 && !pd) return false;
  // dpotrs: lane = row for the column sweeps
  double* b = g + Y.qdd;
  if (lane < nj) b[lane] = (tau_w ? tau_w[lane] : 0.0) - g[Y.C + lane];
  wave_sync();
  for (int k = 0; k < nj; k++) {
    if (lane == k) b[k] = b[k] / L[k + nj * k];
    wave_sync();
    const double bk = b[k];
    if (lane > k && lane < nj) b[lane] = b[lane] - bk * L[lane + nj * k];
    wave_sync();
  }
  for (int k = nj - 1; k >= 0; k--) {
    if (lane == k) { double s = b[k]; for (int i = k + 1; i < nj; i++) s = s - L[i + nj * k] * b[i]; b[k] = s / L[k + nj * k]; }
    wave_sync();
  }
  return pd;
}

// calc_fwd_dyn, eFeatherstone (RCArticulatedBody::algorithm_type): the articulated-body recursion, every spatial quantity at the
// world origin (oracle Artic::fwd_dyn_aba, same operation order).  In place: I6 becomes the articulated inertias, `a` holds the
// bias accelerations c_i and then the accelerations, `f` the bias forces, `F` the U_i; d_i and u_i sit in the (unused) H region.
// Returns false when some d_i = S' IA S is not positive.
MH_DEV bool dynamics_aba(const Model& M, const Lay& Y, double* g, const double* tau_w)
{
  const mh_artic_model& m = M.m;
  const int nj = Y.nj, lane = lane_id();
  kin_inertia(M, Y, g);
  double* dd = g + Y.H; double* uu = g + Y.H + nj;
  for (int i = 0; i < nj; i++) {                          // pass 1, outward
    const int p = m.parent[i];
    const double* S = g + Y.S + 6 * i;
    const double qdi = g[Y.qd + i];
    double* vj = g + Y.Iv;                                // S_i qd_i (6), then I v (6)
    if (lane < 6) { const double e = S[lane] * qdi; vj[lane] = e; g[Y.v + 6 * i + lane] = (p < 0) ? e : g[Y.v + 6 * p + lane] + e; }
    wave_sync();
    if (lane < 6) {
      g[Y.a + 6 * i + lane] = crm_c(g + Y.v + 6 * i, vj, lane);
      const double* I6 = g + Y.I6 + 36 * i + 6 * lane; const double* v = g + Y.v + 6 * i;
      double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + I6[k] * v[k];
      g[Y.Iv + 6 + lane] = acc;
    }
    wave_sync();
    if (lane < 6) g[Y.f + 6 * i + lane] = crf_c(g + Y.v + 6 * i, g + Y.Iv + 6, lane);
    wave_sync();
  }
  bool pd = true;
  for (int i = nj - 1; i >= 0; i--) {                     // pass 2, inward
    const int p = m.parent[i];
    const double* S = g + Y.S + 6 * i;
    double* IA = g + Y.I6 + 36 * i;
    double* U = g + Y.F + 6 * i;
    if (lane < 6) { double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + IA[6 * lane + k] * S[k]; U[lane] = acc; }
    wave_sync();
    if (lane == 0) { dd[i] = dot6(S, U); uu[i] = (tau_w ? tau_w[i] : 0.0) - dot6(S, g + Y.f + 6 * i); }
    wave_sync();
    const double d = dd[i], u = uu[i];
    if (!(d > 0.0)) { pd = false; break; }
    if (p >= 0) {
      if (lane < 36) { const int r = lane / 6, c = lane - 6 * r; double t = U[r] * U[c]; t = t / d; IA[lane] = IA[lane] - t; }   // Ia, in place
      wave_sync();
      if (lane < 36) g[Y.I6 + 36 * p + lane] = g[Y.I6 + 36 * p + lane] + IA[lane];
      else if (lane < 42) {
        const int r = lane - 36;
        const double* c = g + Y.a + 6 * i;
        double acc = 0.0; for (int k = 0; k < 6; k++) acc = acc + IA[6 * r + k] * c[k];
        double e = U[r] * u; e = e / d;
        const double pa = (g[Y.f + 6 * i + r] + acc) + e;
        g[Y.f + 6 * p + r] = g[Y.f + 6 * p + r] + pa;
      }
      wave_sync();
    }
  }
  if (!pd) return false;
  for (int i = 0; i < nj; i++) {                          // pass 3, outward
    const int p = m.parent[i];
    double* ap = g + Y.Iv;
    if (lane < 6) { const double base = (p < 0) ? ((lane < 3) ? 0.0 : -m.gravity[lane - 3]) : g[Y.a + 6 * p + lane]; ap[lane] = base + g[Y.a + 6 * i + lane]; }
    wave_sync();
    if (lane == 0) { const double t = uu[i] - dot6(g + Y.F + 6 * i, ap); g[Y.qdd + i] = t / dd[i]; }
    wave_sync();
    if (lane < 6) g[Y.a + 6 * i + lane] = ap[lane] + g[Y.S + 6 * i + lane] * g[Y.qdd + i];
    wave_sync();
  }
  return true;
}

// X = inverse_SPD(H) from the factor in Y.L (linalg.hpp inverse_spd): lane = column
MH_DEV void inverse_from_factor(const Lay& Y, double* g)
{
  const int nj = Y.nj, lane = lane_id();
  const double* L = g + Y.L;
  if (lane < nj) {
    double e[NJ];
    for (int i = 0; i < nj; i++) e[i] = (i == lane) ? 1.0 : 0.0;
    for (int k = 0; k < nj; k++) { e[k] = e[k] / L[k + nj * k]; const double bk = e[k]; for (int i = k + 1; i < nj; i++) e[i] = e[i] - bk * L[i + nj * k]; }
    for (int k = nj - 1; k >= 0; k--) { double s = e[k]; for (int i = k + 1; i < nj; i++) s = s - L[i + nj * k] * e[i]; e[k] = s / L[k + nj * k]; }
    for (int i = 0; i < nj; i++) g[Y.X + i + nj * lane] = e[i];          // column `lane`
  }
  wave_sync();
  // mirror the lower triangle into the upper one: A(c, i) = A(i, c), i > c
  for (int e2 = lane; e2 < nj * nj; e2 += 64) { const int r = e2 % nj, c = e2 / nj; if (r < c) g[Y.X + r + nj * c] = g[Y.X + c + nj * r]; }
  wave_sync();
}

// find_limit_constraints + the impact handler's no-slip path with NC = 0 (oracle Artic::handle_limits)
MH_DEV void handle_limits(const Model& M, const Lay& Y, double* g, mh_world_aux* aux, WaveRand& rng, int& status,
                          unsigned long long& solves, unsigned long long& rows, unsigned long long& pivs, unsigned long long& bytes)
{
  const mh_artic_model& m = M.m;
  const int nj = Y.nj, lane = lane_id();
  const double qi = (lane < nj) ? g[Y.q + lane] : 0.0;
  const bool up = lane < nj && qi >= m.hilimit[lane], lo = lane < nj && qi <= m.lolimit[lane];
  const uint64_t mu = ballot(up), ml = ballot(lo);
  const int nl = popc(mu) + popc(ml);
  if (nl == 0) return;
  const double qdi = (lane < nj) ? g[Y.qd + lane] : 0.0;
  const bool impacting = ballot((up && -qdi < -NEAR_ZERO_) || (lo && qdi < -NEAR_ZERO_)) != 0ull;   // CSim:313-323
  if (!impacting) return;
  if (nl > NLMAX) { status |= MH_WORLD_UNSUPPORTED; return; }
  // eFeatherstone bodies: the handler's X is still the inverse of the generalized inertia (get_generalized_inertia + inverse_SPD,
  // ICH:1600-1607): H and its factor by the CRB path, at the current q (before the limit storage below reuses the link arrays)
  if (m.algorithm == MH_ARTIC_FSAB) { wave_sync(); if (!dynamics(M, Y, g, nullptr)) { status |= MH_WORLD_LCP_FAILED; return; } }
  int* idx = reinterpret_cast<int*>(g + Y.idx);                  // idx[k] = joint | (upper << 8)
  if (lane < nj) {
    const int base = popc(mu & lanes_below(lane)) + popc(ml & lanes_below(lane));
    if (up) idx[base] = lane | 256;
    if (lo) idx[base + (up ? 1 : 0)] = lane;
  }
  wave_sync();
  inverse_from_factor(Y, g);                                     // compute_X (ICH:1607)
  const double* X = g + Y.X;
  // compute_limit_components (ICH:1755-1781): L_X_LT(a, b) = X(idx_a, idx_b) for b >= a, mirrored; L_v = +-qd
  double* MM = g + Y.MM;
  for (int e = lane; e < nl * nl; e += 64) {
    const int a = e % nl, b2 = e / nl;
    const int ia = idx[a] & 255, ib = idx[b2] & 255;
    MM[e] = (b2 >= a) ? X[ia * nj + ib] : X[ib * nj + ia];
  }
  const bool valid = lane < nl;
  const int my = valid ? idx[lane] : 0;
  const int myj = my & 255; const bool myup = (my & 256) != 0;
  double Lv = 0.0;
  if (valid) { Lv = g[Y.qd + myj]; if (myup) Lv = -Lv; }
  wave_sync();
  // lcp_fast on the persistent _v, then the Lemke ladder (ICH:1239, 1281)
  double nrm0 = 0.0;
  for (int e = lane; e < nl * nl; e += 64) { const double a = fabs(MM[e]); nrm0 = (a > nrm0) ? a : nrm0; }
  nrm0 = wave_max(nrm0);
  const double dii = valid ? MM[lane + nl * lane] : 0.0;
  int zsize = uni(aux->vns_size);
  double zi = (valid && zsize == nl) ? aux->vns[lane] : 0.0;
  DenseLds Md; Md.M = MM; Md.n = nl;
  LuScratch S; S.small = g + Y.A; S.ka = nl; S.big = g + Y.A;
  Trace tr; tr.buf = nullptr; tr.cap = 0; tr.len = 0;
  LcpParams P; P.kind = MH_LCP_FAST; P.min_exp = -20; P.step_exp = 1u; P.max_exp = 1; P.piv_tol = -1.0; P.zero_tol = -1.0;
  unsigned piv = 0, total = 0;
  bool ok = lcp_solve_wave(P, c_pow10a, nl, Md, S, g + Y.art, nrm0, dii, Lv, zi, zsize, rng, piv, tr);
  total += piv;
  if (!ok) {
    P.kind = MH_LCP_LEMKE_REG;
    ok = lcp_solve_wave(P, c_pow10a, nl, Md, S, g + Y.art, nrm0, dii, Lv, zi, zsize, rng, piv, tr);
    total += piv;
  }
  solves += 1ull; rows += (unsigned long long)nl; pivs += total; bytes += 8ull * ((unsigned long long)nl * nl + 2ull * nl);
  if (!ok) { status |= MH_WORLD_LCP_FAILED; return; }           // std::runtime_error("Unable to solve constraint LCP!")
  if (valid) aux->vns[lane] = zi;
  if (lane == 0) aux->vns_size = nl;
  double li = zi;
  double* lv = g + Y.Lv; double* ll = g + Y.l;
  auto apply = [&]() {                                           // update_from_stacked (ICH:298-397) + ICH:452
    if (valid) ll[lane] = li;
    wave_sync();
    if (lane < nj) {
      double dv = 0.0;
      for (int k = 0; k < nl; k++) { const int c = idx[k]; const double ls = (c & 256) ? -ll[k] : ll[k]; dv = dv + ls * X[(c & 255) * nj + lane]; }
      g[Y.qd + lane] = g[Y.qd + lane] + dv;
    }
    if (valid) { double t = 0.0; for (int k = 0; k < nl; k++) t = t + ll[k] * MM[lane + nl * k]; Lv = Lv + t; }
    wave_sync();
  };
  auto minv_of = [&]() -> double {                               // first-minimum over rows 0 .. nl-1, like the oracle's scan
    if (valid) lv[lane] = Lv;
    wave_sync();
    double mn = lv[0];
    for (int k = 1; k < nl; k++) mn = (lv[k] < mn) ? lv[k] : mn;
    wave_sync();
    return mn;
  };
  apply();
  const double minv = minv_of();
  if (valid) li = li * m.limit_restitution[myj];
  const bool changed = ballot(valid && li > NEAR_ZERO_) != 0ull;   // apply_restitution(q) (ICH:497-525)
  if (changed) {
    apply();
    const double minv_plus = minv_of();
    if (minv_plus < 0.0 && minv_plus < minv - NEAR_ZERO_) status |= MH_WORLD_UNSUPPORTED;   // ICH:284-291 reads an unsized _z
  }
  const double qd2 = valid ? g[Y.qd + myj] : 0.0;
  if (ballot(valid && ((myup ? -qd2 : qd2) < -NEAR_ZERO_)) != 0ull) status |= MH_WORLD_IMPACT_TOL;   // ICH:157-167
}

// ConstraintStabilization::stabilize for this body, joint-limit rows (oracle Artic::stabilize; CStab:167-254, 257-304, 434-441, 932-970,
// 1056-1216, 1322-1379).  evaluate_unilateral_constraints reads joints[i] with i the BODY's index (CStab:117) -- joint 0 here, once per
// joint: every entry of uC is one of joint 0's two slacks, so the line search of update_q is wave-uniform scalar code on
// (q0, dq0); the LCP has a row for every finite limit of every joint and needs H^-1 at the current configuration.
struct StabSlack { double hi0, lo0;
  MH_DEV double at(double q0, unsigned i) const { return (i & 1u) ? (q0 + 0.0) - lo0 : (hi0 - q0) - 0.0; }     // hilimit - q - tare / q + tare - lolimit, tare = 0
  MH_DEV double vio(double q0) const { const double a = (hi0 - q0) - 0.0, b = (q0 + 0.0) - lo0; return (b < a) ? b : a; } };
MH_DEV double stab_sign2(double x, double y) { return (y > 0.0) ? fabs(x) : -fabs(x); }
MH_DEV double stab_q0_at(double t, double dq0, double qv0) { double v = dq0 * t; v = v + qv0; return v; }
MH_DEV double stab_ridders(const StabSlack& K, double x1, double x2, double fl, double fh, unsigned idx, double dq0, double qv0) {   // CStab:1322-1379
  const double TOL = 1e-4, INF_ = 1.7976931348623157e308;
  double ans = INF_, fm, fnew, s2, xh, xl, xm, xnew;
  if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
    xl = x1; xh = x2;
    for (unsigned j = 0; j < 25; j++) {
      xm = 0.5 * (xl + xh);
      fm = K.at(stab_q0_at(xm, dq0, qv0), idx);
      s2 = sqrt(fm * fm - fl * fh);
      if (s2 == 0.0) return ans;
      xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s2);
      ans = xnew;
      fnew = K.at(stab_q0_at(ans, dq0, qv0), idx);
      if (fabs(fnew) < TOL && fnew >= 0.0) return xnew;
      if (stab_sign2(fm, fnew) != fm) { xl = xm; fl = fm; xh = ans; fh = fnew; }
      else if (stab_sign2(fl, fnew) != fl) { xh = ans; fh = fnew; }
      else if (stab_sign2(fh, fnew) != fh) { xl = ans; fl = fnew; }
    }
  } else {
    if (fl == 0.0) return x1;
    if (fh == 0.0) return x2;
  }
  return 0.0;
}
__device__ __noinline__ void stabilize_limits(const Model& M, const Lay& Y, double* g, WaveRand& rng, int& status,
                                              unsigned long long& solves, unsigned long long& rows, unsigned long long& pivs, unsigned long long& bytes,
                                              unsigned long long& stab_iters, unsigned long long& stab_rows)
{
  const mh_artic_model& m = M.m;
  const unsigned maxit = (unsigned)m.cstab_max_iterations;
  if (maxit == 0) return;
  const int nj = Y.nj, lane = lane_id();
  const double INF_ = 1.7976931348623157e308;
  StabSlack K; K.hi0 = m.hilimit[0]; K.lo0 = m.lolimit[0];
  const double qd_save = (lane < nj) ? g[Y.qd + lane] : 0.0;
  double qv = (lane < nj) ? g[Y.q + lane] : 0.0;                   // the stabiliser's q (CStab:183)
  double max_uvio = K.vio(uni(g[Y.q]));
  unsigned iterations = 0;
  const bool hfin = lane < nj && m.hilimit[lane] < INF_, lfin = lane < nj && m.lolimit[lane] > -INF_;
  const uint64_t mu = ballot(hfin), ml = ballot(lfin);
  const int nl = popc(mu) + popc(ml);                              // a row for every finite limit (CStab:257-304), upper before lower per joint
  while (max_uvio < m.cstab_eps) {
    if (iterations == maxit) break;
    if (iterations == MH_CSTAB_HARD_CAP) { status |= MH_WORLD_STALLED; break; }
    wave_sync();
    if (lane < nj) g[Y.qd + lane] = 0.0;
    double dq = 0.0;
    wave_sync();
    if (nl > 0) {
      if (nl > NLSTAB) { status |= MH_WORLD_UNSUPPORTED; break; }
      // compute_X at the CURRENT configuration: H by the CRB path, its factor, the inverse (ICH:1600-1607)
      if (!dynamics(M, Y, g, nullptr)) { status |= MH_WORLD_LCP_FAILED; break; }
      int* idx = reinterpret_cast<int*>(g + Y.idx);              // idx[k] = joint | (upper << 8)
      if (lane < nj) {
        const int base = popc(mu & lanes_below(lane)) + popc(ml & lanes_below(lane));
        if (hfin) idx[base] = lane | 256;
        if (lfin) idx[base + (hfin ? 1 : 0)] = lane;
      }
      wave_sync();
      inverse_from_factor(Y, g);
      const double* X = g + Y.X;
      double* MM = g + Y.MM;
      for (int e = lane; e < nl * nl; e += 64) {                   // L X L' without the limits' signs (ICH:1763-1771)
        const int a = e % nl, b2 = e / nl;
        const int ia = idx[a] & 255, ib = idx[b2] & 255;
        MM[e] = (b2 >= a) ? X[ia * nj + ib] : X[ib * nj + ia];
      }
      const bool valid = lane < nl;
      const int my = valid ? idx[lane] : 0;
      const int myj = my & 255; const bool myup = (my & 256) != 0;
      double Lv = 0.0;
      if (valid) { const double qj = g[Y.q + myj]; const double viol = myup ? (m.hilimit[myj] - qj) - 0.0 : (qj + 0.0) - m.lolimit[myj];
                   Lv = (viol - fabs(m.cstab_eps)) - NEAR_ZERO_; }     // CStab:434-441
      wave_sync();
      double nrm0 = 0.0;
      for (int e = lane; e < nl * nl; e += 64) { const double a = fabs(MM[e]); nrm0 = (a > nrm0) ? a : nrm0; }
      nrm0 = wave_max(nrm0);
      const double dii = valid ? MM[lane + nl * lane] : 0.0;
      int zsize = 0;                                               // determine_dq's local z: cold lcp_fast, then the Lemke ladder (CStab:954-955)
      double zi = 0.0;
      DenseLds Md; Md.M = MM; Md.n = nl;
      LuScratch S; S.small = g + Y.A; S.ka = nl; S.big = g + Y.A;
      Trace tr; tr.buf = nullptr; tr.cap = 0; tr.len = 0;
      LcpParams P; P.kind = MH_LCP_FAST; P.min_exp = -20; P.step_exp = 1u; P.max_exp = 1; P.piv_tol = -1.0; P.zero_tol = -1.0;
      unsigned piv = 0, total = 0;
      bool ok = lcp_solve_wave(P, c_pow10a, nl, Md, S, g + Y.art, nrm0, dii, Lv, zi, zsize, rng, piv, tr);
      total += piv;
      if (!ok) {
        P.kind = MH_LCP_LEMKE_REG;
        ok = lcp_solve_wave(P, c_pow10a, nl, Md, S, g + Y.art, nrm0, dii, Lv, zi, zsize, rng, piv, tr);
        total += piv;
      }
      solves += 1ull; rows += (unsigned long long)nl; pivs += total; bytes += 8ull * ((unsigned long long)nl * nl + 2ull * nl);
      stab_rows += (unsigned long long)nl;
      // update_from_stacked(pd, z): l = z whatever it holds; dv = X_LT ls; v += dv; dq = the joint velocities
      double* ll = g + Y.l;
      wave_sync();
      if (valid) ll[lane] = (lane < uni(zsize)) ? zi : 0.0;
      wave_sync();
      if (lane < nj) {
        double dv = 0.0;
        for (int k = 0; k < nl; k++) { const int c = idx[k]; const double ls = (c & 256) ? -ll[k] : ll[k]; dv = dv + ls * X[(c & 255) * nj + lane]; }
        const double v = g[Y.qd + lane] + dv;
        g[Y.qd + lane] = v; dq = v;
      }
      wave_sync();
    }
    // update_q (CStab:1056-1216): the line search lives on joint 0's slacks alone
    { const double dq0 = read_lane(dq, 0), qv0 = read_lane(qv, 0);
      double t = 1.0;
      const double q1 = stab_q0_at(1.0, dq0, qv0);                 // qstar = dq + q: dq * 1.0 is dq
      for (unsigned i = 0; i < 2u * (unsigned)nj; i++) {
        const double fo = K.at(qv0, i), fn = K.at(q1, i);
        if (!((fo < 0.0 && fn > 0.0) || (fo > 0.0 && fn < 0.0))) continue;
        const double root = stab_ridders(K, 0.0, t, fo, fn, i, dq0, qv0);
        if (root > 0.0 && root < 1.0) t = (root < t) ? root : t;
      }
      bool failed = false;
      while (true) {
        const double qt = stab_q0_at(t, dq0, qv0);
        bool stop = true;
        for (unsigned i = 0; i < 2u * (unsigned)nj; i++) {
          const double fo = K.at(qv0, i), fn1 = K.at(q1, i), fc = K.at(qt, i);
          const bool br = (fo < 0.0 && fn1 > 0.0) || (fo > 0.0 && fn1 < 0.0);
          if (!br && fc < 0.0 && fo > fc) { stop = false; break; }
        }
        if (stop) break;
        t *= 0.6;
        if (t < NEAR_ZERO_) { failed = true; break; }
      }
      if (failed) { status |= MH_WORLD_STAB_FAILED; break; }
      if (lane < nj) { double v = dq * t; v = v + qv; qv = v; g[Y.q + lane] = v; }
      wave_sync();
    }
    max_uvio = K.vio(uni(g[Y.q]));
    iterations++;
    stab_iters += 1ull;
  }
  wave_sync();
  if (lane < nj) { g[Y.qd + lane] = qd_save; g[Y.q + lane] = qv; }
  wave_sync();
}

template <bool STAB>
MH_DEV void artic_step_body(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                            mh_world_aux* __restrict__ auxg)
{
  extern __shared__ double g[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const Model& M = *Mg;
  const int nj = M.m.nj, lane = lane_id();
  const Lay Y(nj, STAB ? NLSTAB : NLMAX);
  mh_world_aux* aux = auxg + b;
  if (lane < nj) { g[Y.q + lane] = qg[(size_t)b * nj + lane]; g[Y.qd + lane] = qdg[(size_t)b * nj + lane]; }
  WaveRand rng; rng.load(aux->rng);
  if (lane == 0) g_lcp_prof_on = 0;
  int status = uni(aux->status);
  unsigned long long solves = 0, rows = 0, pivs = 0, bytes = 0, stab_iters = 0, stab_rows = 0;
  // An exception of calc_fwd_dyn, the impact handler or compute_X ends the run (oracle Artic::step): the state stays where the throw left it (positions
  // integrated, velocities without the impulses), time and counters without that step, and a world that carries MH_WORLD_LCP_FAILED is not stepped again.
  int minis = 0, steps = 0;                                          // mini-steps whose time was added / steps that ran to their end
  wave_sync();
  for (int s = 0; s < nsteps; s++) {
    if (status & MH_WORLD_LCP_FAILED) break;
    // positions with the OLD velocity (TSS:156-164)
    if (lane < nj) { double qn = g[Y.qd + lane] * dt; qn = qn + g[Y.q + lane]; g[Y.q + lane] = qn; }
    wave_sync();
    const bool ok = (M.m.algorithm == MH_ARTIC_FSAB) ? dynamics_aba(M, Y, g, nullptr) : dynamics(M, Y, g, nullptr);
    if (!ok) { status |= MH_WORLD_LCP_FAILED; break; }
    if (lane < nj) g[Y.qd + lane] = g[Y.qd + lane] + g[Y.qdd + lane] * dt;   // TSS:182-192
    wave_sync();
    handle_limits(M, Y, g, aux, rng, status, solves, rows, pivs, bytes);
    wave_sync();
    if (status & MH_WORLD_LCP_FAILED) break;
    minis++;
    if (STAB) { stabilize_limits(M, Y, g, rng, status, solves, rows, pivs, bytes, stab_iters, stab_rows); if (status & MH_WORLD_LCP_FAILED) break; }   // TSS:97
    steps++;
  }
  wave_sync();
  if (lane < nj) { qg[(size_t)b * nj + lane] = g[Y.q + lane]; qdg[(size_t)b * nj + lane] = g[Y.qd + lane]; }
  rng.store(aux->rng);
  if (lane == 0) {
    double tm = aux->time; for (int s = 0; s < minis; s++) tm += dt;
    aux->time = tm; aux->status = status;
    aux->steps += (unsigned long long)steps; aux->mini_steps += (unsigned long long)minis;
    aux->lcp_solves += solves; aux->lcp_rows += rows; aux->lcp_pivots += pivs; aux->lcp_alg_bytes += bytes;
    aux->stab_iters += stab_iters; aux->stab_rows += stab_rows;
  }
}

// The same step at three register budgets: 128 VGPRs (4 waves per SIMD = the 16 worlds per CU the 10 KB LDS image allows; 95 spilled
// VGPRs), 168 (3 per SIMD, 14 spilled) and 193 (2 per SIMD, none).  The kernel waits on ~250 LDS round trips per step, so
// resident waves win over spills: ur10 x 8192, 200 steps: 25.6 / 31.2 / 39.0 ms (profiles/r02_c_artic_occupancy.jsonl).
// Default 4; MH_ARTIC_WAVES=2|3|5 selects the others (experiments).  Round 5: with the local transforms sharing v / a the image is 9 KB = 18 worlds per CU, and the
// five-waves build (96 VGPRs, 132 spilled) was measured on them: 29.2 ms against 25.5 -- at this point the SPILLS cost more than the fifth wave hides, so a smaller
// image alone buys nothing: the routine needs fewer live registers first (profiles/r05_d_artic_occupancy.txt).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_artic_step_w3(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                     mh_world_aux* __restrict__ auxg) { artic_step_body<false>(Mg, B, dt, nsteps, qg, qdg, auxg); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_artic_step_w4(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                     mh_world_aux* __restrict__ auxg) { artic_step_body<false>(Mg, B, dt, nsteps, qg, qdg, auxg); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5)))
void k_artic_step_w5(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                     mh_world_aux* __restrict__ auxg) { artic_step_body<false>(Mg, B, dt, nsteps, qg, qdg, auxg); }
__global__ __launch_bounds__(64)
void k_artic_step_w2(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                     mh_world_aux* __restrict__ auxg) { artic_step_body<false>(Mg, B, dt, nsteps, qg, qdg, auxg); }

// TWO worlds per wavefront (round 4): lanes 0-31 step world 2b, lanes 32-63 world 2b + 1, out of two LDS images, through ONE instruction
// stream -- the kernel is bound by instruction issue and LDS round trips with 6-36 of 64 lanes at work, so the second world rides on
// instructions the first one pays for.  The forward dynamics (kinematics, RNEA, CRBA, Cholesky: no data-dependent control flow) run
// packed; the joint-limit handler, whose pivoting loops are wave-uniform per WORLD, runs on each world in turn with the whole wave (it
// leaves at once when no limit is hit).  16 worlds per CU as before (8 waves x 2 images of 10 KB) at 256 registers per lane: no spills.
// CRB bodies without spheres and without the stabiliser (config 5).  Selected by MH_ARTIC_PACK=1 (see mh_artic_batch_step for what it measured).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_artic_step_p2(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                     mh_world_aux* __restrict__ auxg)
{
  extern __shared__ double g[];
  const int b0 = 2 * (int)blockIdx.x;
  if (b0 >= B) return;
  const Model& M = *Mg;
  const int nj = M.m.nj, lane = lane_id(), wl = lane >> 5, hl = lane & 31;
  const Lay Y(nj);
  const bool two = b0 + 1 < B;                                       // (an odd batch: the last wave's second half idles on an image nobody reads)
  const int b = b0 + wl;
  double* gw = g + (size_t)wl * Y.total;                             // this lane's world
  const bool mine = hl < nj && (wl == 0 || two);
  if (hl < nj) { gw[Y.q + hl] = mine ? qg[(size_t)b * nj + hl] : 0.0; gw[Y.qd + hl] = mine ? qdg[(size_t)b * nj + hl] : 0.0; }
  WaveRand rng0, rng1; rng0.load(auxg[b0].rng); rng1.load(auxg[two ? b0 + 1 : b0].rng);
  if (lane == 0) g_lcp_prof_on = 0;
  int status0 = uni(auxg[b0].status), status1 = two ? uni(auxg[b0 + 1].status) : 0;
  unsigned long long solves0 = 0, rows0 = 0, pivs0 = 0, bytes0 = 0, solves1 = 0, rows1 = 0, pivs1 = 0, bytes1 = 0;
  int n0 = 0, n1 = 0;                                                // steps each world ran to their end (an exception ends a world's run: artic_step_body)
  wave_sync();
  for (int s = 0; s < nsteps; s++) {
    const bool a0 = !(status0 & MH_WORLD_LCP_FAILED), a1 = two && !(status1 & MH_WORLD_LCP_FAILED);    // still running (uniform)
    if (!a0 && !a1) break;
    const bool alive = (wl == 0) ? a0 : a1;                           // this lane's world: a dead world's image is computed on and never written back
    if (hl < nj && alive) { double qn = gw[Y.qd + hl] * dt; qn = qn + gw[Y.q + hl]; gw[Y.q + hl] = qn; }      // positions with the OLD velocity (TSS:156-164)
    wave_sync();
    const bool okl = dynamics<2>(M, Y, gw, nullptr);
    const bool ok0 = (ballot(okl) & 1ull) != 0ull, ok1 = ((ballot(okl) >> 32) & 1ull) != 0ull;
    if (a0 && !ok0) status0 |= MH_WORLD_LCP_FAILED;
    if (a1 && !ok1) status1 |= MH_WORLD_LCP_FAILED;
    if (hl < nj && alive && okl) gw[Y.qd + hl] = gw[Y.qd + hl] + gw[Y.qdd + hl] * dt;   // TSS:182-192
    wave_sync();
    if (a0 && ok0) handle_limits(M, Y, g, auxg + b0, rng0, status0, solves0, rows0, pivs0, bytes0);
    wave_sync();
    if (a1 && ok1) handle_limits(M, Y, g + Y.total, auxg + b0 + 1, rng1, status1, solves1, rows1, pivs1, bytes1);
    wave_sync();
    if (a0 && !(status0 & MH_WORLD_LCP_FAILED)) n0++;
    if (a1 && !(status1 & MH_WORLD_LCP_FAILED)) n1++;
  }
  if (mine) { qg[(size_t)b * nj + hl] = gw[Y.q + hl]; qdg[(size_t)b * nj + hl] = gw[Y.qd + hl]; }
  rng0.store(auxg[b0].rng);
  if (two) rng1.store(auxg[b0 + 1].rng);
  if (lane == 0) {
    for (int w = 0; w < (two ? 2 : 1); w++) {
      mh_world_aux* aux = auxg + b0 + w;
      const int nw = w ? n1 : n0;
      double tm = aux->time; for (int s = 0; s < nw; s++) tm += dt;
      aux->time = tm; aux->status = w ? status1 : status0;
      aux->steps += (unsigned long long)nw; aux->mini_steps += (unsigned long long)nw;
      aux->lcp_solves += w ? solves1 : solves0; aux->lcp_rows += w ? rows1 : rows0; aux->lcp_pivots += w ? pivs1 : pivs0; aux->lcp_alg_bytes += w ? bytes1 : bytes0;
    }
  }
}

// the same step followed by ConstraintStabilization::stabilize (joint-limit rows): its own kernel, so that bodies stepped with
// stabilisation off (ur10.xml:11) carry neither its registers nor its 20 KB LDS image (a row for every finite limit: 2 nj)
__global__ __launch_bounds__(64)
void k_artic_step_stab(const Model* __restrict__ Mg, int B, double dt, int nsteps, double* __restrict__ qg, double* __restrict__ qdg,
                       mh_world_aux* __restrict__ auxg) { artic_step_body<true>(Mg, B, dt, nsteps, qg, qdg, auxg); }

// seam B4: qdd = H^-1 (tau - C), H, link poses of the resident states
__global__ __launch_bounds__(64)
void k_artic_fwd_dyn(const Model* __restrict__ Mg, int B, const double* __restrict__ qg, const double* __restrict__ qdg,
                     const double* __restrict__ tau, double* __restrict__ qdd_out, double* __restrict__ H_out, double* __restrict__ poses,
                     int* __restrict__ okflag)
{
  extern __shared__ double g[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const Model& M = *Mg;
  const int nj = M.m.nj, lane = lane_id();
  const Lay Y(nj);
  if (lane < nj) { g[Y.q + lane] = qg[(size_t)b * nj + lane]; g[Y.qd + lane] = qdg[(size_t)b * nj + lane]; }
  wave_sync();
  const double* tw = tau ? tau + (size_t)b * nj : nullptr;
  bool ok;
  if (M.m.algorithm == MH_ARTIC_FSAB) {
    ok = dynamics_aba(M, Y, g, tw);
    if (qdd_out && lane < nj) qdd_out[(size_t)b * nj + lane] = ok ? g[Y.qdd + lane] : 0.0;
    wave_sync();
    if (H_out) (void)dynamics(M, Y, g, tw);                // the generalized inertia is CRB's whatever the algorithm
  } else {
    ok = dynamics(M, Y, g, tw);
    if (qdd_out && lane < nj) qdd_out[(size_t)b * nj + lane] = ok ? g[Y.qdd + lane] : 0.0;
  }
  if (H_out) for (int e = lane; e < nj * nj; e += 64) H_out[(size_t)b * nj * nj + e] = g[Y.H + e];
  if (poses) for (int e = lane; e < 12 * nj; e += 64) { const int i = e / 12, k = e - 12 * i; poses[(size_t)b * 12 * nj + e] = (k < 9) ? g[Y.R + 9 * i + k] : g[Y.x + 3 * i + k - 9]; }
  if (okflag && lane == 0) okflag[b] = ok ? 1 : 0;
}

// calc_jacobian: column j = the twist of joint j (S_j, about the world origin) moved to the point, for j on the link's path
__global__ __launch_bounds__(64)
void k_artic_jacobian(const Model* __restrict__ Mg, int B, const double* __restrict__ qg, int link, const double* __restrict__ points,
                      double* __restrict__ J_out)
{
  extern __shared__ double g[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const Model& M = *Mg;
  const int nj = M.m.nj, lane = lane_id();
  const Lay Y(nj);
  if (lane < nj) { g[Y.q + lane] = qg[(size_t)b * nj + lane]; g[Y.qd + lane] = 0.0; }
  wave_sync();
  kin_inertia(M, Y, g);
  const double* p = points + (size_t)b * 3;
  for (int e = lane; e < 6 * nj; e += 64) {
    const int r = e / nj, j = e - r * nj;
    double v = 0.0;
    if ((M.anc[link] >> j) & 1u) {
      const double* S = g + Y.S + 6 * j;                  // [angular; linear at the origin]
      if (r >= 3) v = S[r - 3];
      else { const int k1 = (r + 1) % 3, k2 = (r + 2) % 3; v = S[3 + r] + (S[k1] * p[k2] - S[k2] * p[k1]); }   // v_o + w x p
    }
    J_out[(size_t)b * 6 * nj + e] = v;
  }
}

#include "mh_artic_contacts.inc"

}} // namespace mh::artic

struct mh_artic_batch {
  int device;                // the HIP device the batch lives on (current at create); every entry point runs there (MH_ON_DEVICE)
  int B, nj, nspheres, cstab, algorithm;
  mh::artic::Model* d_model;
  double* d_q; double* d_qd; mh_world_aux* d_aux;
  double* d_ws;           // link contacts with the Drumwright-Shell model: _MM + LU workspace, 2 x 64 x 64 doubles per world
};

extern "C" {

int mh_artic_batch_device(const mh_artic_batch* ab) { return ab ? ab->device : fail(MH_ERR_INVALID_ARG, "null batch"); }

int mh_artic_batch_destroy(mh_artic_batch* ab)
{
  if (!ab) return MH_OK;
  MH_ON_DEVICE(ab);
  (void)hipDeviceSynchronize();
  void* ps[] = { ab->d_model, ab->d_q, ab->d_qd, ab->d_aux, ab->d_ws };
  for (void* p : ps) if (p) (void)hipFree(p);
  delete ab;
  return MH_OK;
}

int mh_artic_batch_create(const mh_artic_model* model, int B, mh_artic_batch** out)
{
  namespace ar = mh::artic;
  if (!out) return fail(MH_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (!model) return fail(MH_ERR_INVALID_ARG, "null model");
  if (B <= 0) return fail(MH_ERR_INVALID_ARG, "batch must be > 0");
  const int nj = model->nj;
  if (nj < 1 || nj > MH_ARTIC_MAX_JOINTS) return fail(MH_ERR_INVALID_ARG, "nj = %d outside [1, %d]", nj, MH_ARTIC_MAX_JOINTS);
  ar::Model hm; std::memset(&hm, 0, sizeof(hm)); hm.m = *model;
  for (int i = 0; i < nj; i++) {
    const int p = model->parent[i];
    if (p >= i || p < -1) return fail(MH_ERR_INVALID_ARG, "joint %d: parent %d must come before it (-1 = base)", i, p);
    if (model->jtype[i] != MH_JOINT_REVOLUTE && model->jtype[i] != MH_JOINT_PRISMATIC) return fail(MH_ERR_INVALID_ARG, "joint %d: type %d is not built (revolute, prismatic)", i, model->jtype[i]);
    if (!(model->mass[i] >= 0.0)) return fail(MH_ERR_INVALID_ARG, "link %d: mass must be >= 0", i);
    const double* a = model->axis[i]; const double nn = a[0]*a[0] + a[1]*a[1] + a[2]*a[2];
    if (!(nn > 0.999999 && nn < 1.000001)) return fail(MH_ERR_INVALID_ARG, "joint %d: axis is not a unit vector", i);
    if (!(model->lolimit[i] <= model->hilimit[i])) return fail(MH_ERR_INVALID_ARG, "joint %d: lower limit above the upper one", i);
    hm.anc[i] = (1u << i) | (p >= 0 ? hm.anc[p] : 0u);
  }
  if (model->floating_base != 0 && model->floating_base != 1) return fail(MH_ERR_INVALID_ARG, "floating_base = %d: 0 or 1", model->floating_base);
  if (model->floating_base) {                                        // the layout include/moby_hip_artic.h states (conservative advancement reads joints 0..2 as the base's velocity)
    if (nj < 6) return fail(MH_ERR_INVALID_ARG, "floating_base needs the six virtual joints in front (nj = %d)", nj);
    for (int v = 0; v < 6; v++) {
      bool ok = model->parent[v] == v - 1 && model->jtype[v] == ((v < 3) ? MH_JOINT_PRISMATIC : MH_JOINT_REVOLUTE);
      for (int k = 0; k < 3; k++) ok = ok && model->axis[v][k] == ((k == v % 3) ? 1.0 : 0.0) && model->com[v][k] == 0.0 && (v == 0 || model->trel[v][k] == 0.0);
      for (int k = 0; k < 9; k++) ok = ok && (v == 3 || model->Rrel[v][k] == ((k % 4 == 0) ? 1.0 : 0.0));
      if (!ok) return fail(MH_ERR_INVALID_ARG, "floating_base: joint %d is not the virtual joint the layout states (sliders along global x, y, z, then hinges about the base link's x, y, z through its COM)", v);
    }
  }
  // a link may be massless (the virtual links under a floating base, mh_io_load_xml_artic) as long as every joint moves mass: the composite inertia outboard of it
  { double sub[MH_ARTIC_MAX_JOINTS];
    for (int i = 0; i < nj; i++) sub[i] = model->mass[i];
    for (int i = nj - 1; i >= 0; i--) { if (!(sub[i] > 0.0)) return fail(MH_ERR_INVALID_ARG, "joint %d carries no mass (its link and everything outboard of it are massless)", i); if (model->parent[i] >= 0) sub[model->parent[i]] += sub[i]; } }
  if (model->nspheres < 0 || model->nspheres > MH_ARTIC_MAX_SPHERES) return fail(MH_ERR_INVALID_ARG, "nspheres = %d outside [0, %d]", model->nspheres, MH_ARTIC_MAX_SPHERES);
  for (int s = 0; s < model->nspheres; s++) {
    if (model->sphere_link[s] < 0 || model->sphere_link[s] >= nj) return fail(MH_ERR_INVALID_ARG, "sphere %d: link %d outside [0, %d)", s, model->sphere_link[s], nj);
    if (!(model->sphere_radius[s] > 0.0)) return fail(MH_ERR_INVALID_ARG, "sphere %d: radius must be > 0", s);
  }
  if (model->cstab_max_iterations < 0) return fail(MH_ERR_INVALID_ARG, "cstab_max_iterations = %d < 0", model->cstab_max_iterations);
  if (model->nspheres > 0) {
    const double* Rp = model->plane_R; const double nn = Rp[1]*Rp[1] + Rp[4]*Rp[4] + Rp[7]*Rp[7];
    if (!(nn > 0.999999 && nn < 1.000001)) return fail(MH_ERR_INVALID_ARG, "plane_R is not a rotation (its +Y column is the plane normal)");
    if (!(model->min_step_size > 0.0) || !(model->contact_dist_thresh > 0.0)) return fail(MH_ERR_INVALID_ARG, "min_step_size and contact_dist_thresh must be > 0 when spheres are present");
    if (!(model->cp_epsilon >= 0.0) || !(model->cp_mu_coulomb >= 0.0)) return fail(MH_ERR_INVALID_ARG, "contact parameters must be >= 0");
    if (!(model->cp_mu_coulomb >= 1e2)) {                         // the Drumwright-Shell model's parameters
      const int nk = model->cp_nk > 0 ? model->cp_nk : 4;
      if (nk < 4 || nk > 64 || (nk & 1)) return fail(MH_ERR_INVALID_ARG, "cp_nk = %d: friction-cone-edges must be even, in [4, 64]", nk);
      if (!(model->cp_mu_viscous >= 0.0) || !(model->cp_compliance >= 0.0)) return fail(MH_ERR_INVALID_ARG, "contact parameters must be >= 0");
      const int kh = nk / 2;
      for (int j = 0; j < kh; j++) { const double theta = (double)j / (kh - 1) * M_PI_2; hm.fcos[j] = std::cos(theta); hm.fsin[j] = std::sin(theta); }
    }
  }
  if (mh_device_count() <= 0) return fail(MH_ERR_NO_DEVICE, "no HIP device visible");
  {
    // the regularisation ladder's powers of ten, once per DEVICE (a second GPU of the process has its own copy of the symbol), under a lock
    static std::mutex mu; static std::vector<char> done;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0; MH_HIP(hipGetDevice(&dev));
    if ((int)done.size() <= dev) done.resize(dev + 1, 0);
    if (!done[dev]) {
      mh::Pow10Table p10; for (int i = 0; i < 64; i++) p10.v[i] = std::pow(10.0, (double)(i - 32));   // LCP.cpp:285
      MH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ar::c_pow10a), &p10, sizeof(p10)));
      done[dev] = 1;
    }
  }
  mh_artic_batch* ab = new mh_artic_batch();
  if (hipGetDevice(&ab->device) != hipSuccess) { delete ab; return fail(MH_ERR_HIP, "hipGetDevice failed"); }
  ab->B = B; ab->nj = nj; ab->algorithm = model->algorithm; ab->nspheres = model->nspheres; ab->cstab = model->cstab_max_iterations != 0 ? 1 : 0; ab->d_model = nullptr; ab->d_q = nullptr; ab->d_qd = nullptr; ab->d_aux = nullptr; ab->d_ws = nullptr;
  const size_t sB = (size_t)B;
  bool ok = hipMalloc((void**)&ab->d_model, sizeof(ar::Model)) == hipSuccess && hipMalloc((void**)&ab->d_q, sB * nj * 8) == hipSuccess
         && hipMalloc((void**)&ab->d_qd, sB * nj * 8) == hipSuccess && hipMalloc((void**)&ab->d_aux, sB * sizeof(mh_world_aux)) == hipSuccess;
  if (ok && model->nspheres > 0 && (!(model->cp_mu_coulomb >= 1e2) || model->cstab_max_iterations != 0))   // (the stabiliser's LCP with contact AND limit rows lives there too)
    ok = hipMalloc((void**)&ab->d_ws, sB * 2 * MH_LCP_MAX_N_WAVE * MH_LCP_MAX_N_WAVE * sizeof(double)) == hipSuccess;
  if (ok) {
    std::vector<mh_world_aux> a(sB);
    mh_world_aux_init(&a[0], 1);
    for (int b = 1; b < B; b++) a[b] = a[0];
    ok = hipMemcpy(ab->d_model, &hm, sizeof(hm), hipMemcpyHostToDevice) == hipSuccess
      && hipMemset(ab->d_q, 0, sB * nj * 8) == hipSuccess && hipMemset(ab->d_qd, 0, sB * nj * 8) == hipSuccess
      && hipMemcpy(ab->d_aux, a.data(), sB * sizeof(mh_world_aux), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok) { mh_artic_batch_destroy(ab); return fail(MH_ERR_HIP, "device allocation / upload failed"); }
  *out = ab;
  return MH_OK;
}

int mh_artic_batch_upload(mh_artic_batch* ab, const double* q, const double* qd, const mh_world_aux* aux)
{
  if (!ab) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ab);
  MH_HIP(hipDeviceSynchronize());                              // a step may be in flight on a caller's non-blocking stream
  const size_t n = (size_t)ab->B * ab->nj * 8;
  if (q) MH_HIP(hipMemcpy(ab->d_q, q, n, hipMemcpyHostToDevice));
  if (qd) MH_HIP(hipMemcpy(ab->d_qd, qd, n, hipMemcpyHostToDevice));
  if (aux) MH_HIP(hipMemcpy(ab->d_aux, aux, (size_t)ab->B * sizeof(mh_world_aux), hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_artic_batch_step(mh_artic_batch* ab, void* stream, double dt, int nsteps)
{
  namespace ar = mh::artic;
  if (!ab) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ab);
  if (nsteps < 0) return fail(MH_ERR_INVALID_ARG, "negative step count");
  if (nsteps == 0) return MH_OK;
  if (!(dt > 0.0)) return fail(MH_ERR_INVALID_ARG, "dt must be > 0");
  if (ab->nspheres > 0) {                                     // bodies with collision geometry: the full step with mini-steps and contact rows
    hipLaunchKernelGGL(ab->cstab ? ar::k_artic_step_contacts_stab : ar::k_artic_step_contacts, dim3(ab->B), dim3(64), ar::lds_bytes_contacts(ab->nj), (hipStream_t)stream,
                       (const ar::Model*)ab->d_model, ab->B, dt, nsteps, ab->d_q, ab->d_qd, ab->d_aux, ab->d_ws);
    MH_HIP(hipGetLastError());
    return MH_OK;
  }
  if (ab->cstab) {
    hipLaunchKernelGGL(ar::k_artic_step_stab, dim3(ab->B), dim3(64), ar::lds_bytes(ab->nj, ar::NLSTAB), (hipStream_t)stream,
                       (const ar::Model*)ab->d_model, ab->B, dt, nsteps, ab->d_q, ab->d_qd, ab->d_aux);
    MH_HIP(hipGetLastError());
    return MH_OK;
  }
  // (off by default: measured on ur10 x 8192 x 200 steps it retires a world-step in 27 % fewer vector and 40 % fewer LDS instructions and takes
  //  27.4 ms against 26.2 -- two images per wave halve the resident waves, and with them what hides the LDS round trips; profiles/r04_a_artic_issue.json)
  static const int pack_env = [] { const char* e = std::getenv("MH_ARTIC_PACK"); return e ? std::atoi(e) : 0; }();
  const int pack = pack_env | mh_g_debug_artic_pack;                 // mh_debug_set(9, 1)
  if (pack != 0 && ab->algorithm == MH_ARTIC_CRB && !std::getenv("MH_ARTIC_WAVES")) {          // two worlds per wavefront (k_artic_step_p2)
    hipLaunchKernelGGL(ar::k_artic_step_p2, dim3((ab->B + 1) / 2), dim3(64), 2 * ar::lds_bytes(ab->nj), (hipStream_t)stream,
                       (const ar::Model*)ab->d_model, ab->B, dt, nsteps, ab->d_q, ab->d_qd, ab->d_aux);
    MH_HIP(hipGetLastError());
    return MH_OK;
  }
  static const int waves = [] { const char* e = std::getenv("MH_ARTIC_WAVES"); const int w = e ? std::atoi(e) : 4; return (w == 2 || w == 3 || w == 5) ? w : 4; }();
  hipLaunchKernelGGL(waves == 5 ? ar::k_artic_step_w5 : waves == 4 ? ar::k_artic_step_w4 : (waves == 2 ? ar::k_artic_step_w2 : ar::k_artic_step_w3), dim3(ab->B), dim3(64), ar::lds_bytes(ab->nj), (hipStream_t)stream,
                     (const ar::Model*)ab->d_model, ab->B, dt, nsteps, ab->d_q, ab->d_qd, ab->d_aux);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

int mh_artic_batch_fwd_dyn(mh_artic_batch* ab, const double* tau, double* qdd_out, double* H_out)
{
  namespace ar = mh::artic;
  if (!ab) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ab);
  MH_HIP(hipDeviceSynchronize());                              // a step may be in flight on a caller's non-blocking stream
  const size_t B = (size_t)ab->B, nj = (size_t)ab->nj;
  double *d_tau = nullptr, *d_qdd = nullptr, *d_H = nullptr; int* d_ok = nullptr;
  auto cleanup = [&]() { void* ps[] = { d_tau, d_qdd, d_H, d_ok }; for (void* p : ps) if (p) (void)hipFree(p); };
  bool ok = hipMalloc((void**)&d_qdd, B * nj * 8) == hipSuccess && hipMalloc((void**)&d_ok, B * 4) == hipSuccess;
  if (ok && tau) ok = hipMalloc((void**)&d_tau, B * nj * 8) == hipSuccess && hipMemcpy(d_tau, tau, B * nj * 8, hipMemcpyHostToDevice) == hipSuccess;
  if (ok && H_out) ok = hipMalloc((void**)&d_H, B * nj * nj * 8) == hipSuccess;
  if (!ok) { cleanup(); return fail(MH_ERR_HIP, "device allocation failed"); }
  hipLaunchKernelGGL(ar::k_artic_fwd_dyn, dim3(ab->B), dim3(64), ar::lds_bytes(ab->nj), (hipStream_t)nullptr,
                     (const ar::Model*)ab->d_model, ab->B, (const double*)ab->d_q, (const double*)ab->d_qd, (const double*)d_tau, d_qdd, d_H, (double*)nullptr, d_ok);
  hipError_t e = hipDeviceSynchronize();
  std::vector<int> hok(B);
  if (e == hipSuccess && qdd_out) e = hipMemcpy(qdd_out, d_qdd, B * nj * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && H_out) e = hipMemcpy(H_out, d_H, B * nj * nj * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(hok.data(), d_ok, B * 4, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(MH_ERR_HIP, "forward dynamics launch failed: %s", hipGetErrorString(e));
  for (size_t b = 0; b < B; b++) if (!hok[b]) return fail(MH_ERR_INVALID_ARG, "world %zu: the generalized inertia is not positive definite", b);
  return MH_OK;
}

int mh_artic_batch_link_poses(mh_artic_batch* ab, double* poses)
{
  namespace ar = mh::artic;
  if (!ab || !poses) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(ab);
  MH_HIP(hipDeviceSynchronize());                              // a step may be in flight on a caller's non-blocking stream
  const size_t bytes = (size_t)ab->B * ab->nj * 12 * 8;
  double* d_p = nullptr;
  MH_HIP(hipMalloc((void**)&d_p, bytes));
  hipLaunchKernelGGL(ar::k_artic_fwd_dyn, dim3(ab->B), dim3(64), ar::lds_bytes(ab->nj), (hipStream_t)nullptr,
                     (const ar::Model*)ab->d_model, ab->B, (const double*)ab->d_q, (const double*)ab->d_qd, (const double*)nullptr, (double*)nullptr, (double*)nullptr, d_p, (int*)nullptr);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(poses, d_p, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(d_p);
  if (e != hipSuccess) return fail(MH_ERR_HIP, "link pose launch failed: %s", hipGetErrorString(e));
  return MH_OK;
}

int mh_artic_batch_jacobian(mh_artic_batch* ab, int link, const double* points, double* J_out)
{
  namespace ar = mh::artic;
  if (!ab || !points || !J_out) return fail(MH_ERR_INVALID_ARG, "null batch / buffer");
  MH_ON_DEVICE(ab);
  MH_HIP(hipDeviceSynchronize());                              // a step may be in flight on a caller's non-blocking stream
  if (link < 0 || link >= ab->nj) return fail(MH_ERR_INVALID_ARG, "link %d outside [0, %d)", link, ab->nj);
  const size_t pb = (size_t)ab->B * 3 * 8, jb = (size_t)ab->B * 6 * ab->nj * 8;
  double* d_p = nullptr; double* d_J = nullptr;
  MH_HIP(hipMalloc((void**)&d_p, pb));
  hipError_t e = hipMalloc((void**)&d_J, jb);
  if (e == hipSuccess) e = hipMemcpy(d_p, points, pb, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ar::k_artic_jacobian, dim3(ab->B), dim3(64), ar::lds_bytes(ab->nj), (hipStream_t)nullptr,
                       (const ar::Model*)ab->d_model, ab->B, (const double*)ab->d_q, link, (const double*)d_p, d_J);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(J_out, d_J, jb, hipMemcpyDeviceToHost);
  (void)hipFree(d_p); if (d_J) (void)hipFree(d_J);
  if (e != hipSuccess) return fail(MH_ERR_HIP, "Jacobian launch failed: %s", hipGetErrorString(e));
  return MH_OK;
}

int mh_artic_batch_download(mh_artic_batch* ab, double* q, double* qd, mh_world_aux* aux)
{
  if (!ab) return fail(MH_ERR_INVALID_ARG, "null batch");
  MH_ON_DEVICE(ab);
  MH_HIP(hipDeviceSynchronize());
  const size_t n = (size_t)ab->B * ab->nj * 8;
  if (q) MH_HIP(hipMemcpy(q, ab->d_q, n, hipMemcpyDeviceToHost));
  if (qd) MH_HIP(hipMemcpy(qd, ab->d_qd, n, hipMemcpyDeviceToHost));
  if (aux) MH_HIP(hipMemcpy(aux, ab->d_aux, (size_t)ab->B * sizeof(mh_world_aux), hipMemcpyDeviceToHost));
  return MH_OK;
}

} // extern "C"
