// One Moby world per wavefront: TimeSteppingSimulator::step and everything it
// drives, for scenes of free rigid spheres + one static plane.
//
// Replaces (per world; same results as the CPU oracle oracle/world.hpp, which
// restates these reference functions statement by statement):
//   TimeSteppingSimulator::step / do_mini_step / calc_next_CA_Euler_step
//                                   src/TimeSteppingSimulator.cpp:52-111,114-222,272-331
//   CCD::broad_phase, calc_CA_Euler_step*, calc_max_dist, find_contacts_sphere_*
//                                   src/CCD.cpp:122-235,238-405,585-609,702-876; include/Moby/CCD.inl:804-847,1164-1207
//   ConstraintSimulator::calc_pairwise_distances / find_unilateral_constraints /
//   calc_impacting_unilateral_constraint_forces   src/ConstraintSimulator.cpp:298-355,450-537
//   UnilateralConstraint::determine_connected_constraints / remove_inactive_groups
//                                   src/UnilateralConstraint.cpp:940-1225
//   ImpactConstraintHandler::apply_model_to_connected_constraints, compute_problem_data,
//   update_from_stacked, update_constraint_velocities_from_impulses, apply_restitution
//                                   src/ImpactConstraintHandler.cpp:298-626,1590-2166
//   ImpactConstraintHandler::solve_qp_work / setup_QP   src/ImpactConstraintHandlerQP.cpp:94-497
//   ConstraintStabilization::stabilize / compute_problem_data / determine_dq /
//   update_q / ridders_unilateral   src/ConstraintStabilization.cpp:167-254,347-492,932-970,1056-1216,1322-1379
//
// MI355X mapping: the wave's 64 lanes are, phase by phase, BODY lanes (state
// integration, inverse inertias), PAIR lanes (swept-bounds overlap, signed
// distance, conservative-advancement time, contact generation -- compacted with
// ballot/popcount), CONTACT lanes, JACOBIAN-ROW lanes (rows [d, r x d], X J^T,
// the C X C^T blocks) and LCP-VARIABLE lanes (mh_lcp_wave.h).  Per-world control
// flow (mini-step loop, restitution, Ridders, backtracking) is wave-uniform.
// World state, contacts, Jacobian rows and the LCP matrix live in LDS; HBM is
// touched only to load the state once and to store it (and the optional
// trajectory) -- all `nsteps` steps run inside one launch.
#pragma once
#include "mh_lcp_wave.h"

namespace mh {

struct V3 { double x, y, z; };
MH_DEV V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MH_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
MH_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MH_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
MH_DEV V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
MH_DEV V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
MH_DEV double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MH_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MH_DEV double norm(V3 a) { return sqrt(dot(a, a)); }

#define MHW_INF 1.7976931348623157e+308
#define MHW_MAX_CONTACTS 12
#define MHW_MAX_ROWS 36
#define MHW_MAX_GROWS 24      /* rows of the C X C^T matrix: 3*nc (impact, nc<=8) or nc (stabilisation) */

// friction polygon directions cos/sin(j/(kh-1) * pi/2), filled by the host's libm
// (ImpactConstraintHandlerQP.cpp:466-468) so that device and oracle agree bit for bit
struct FricTable { double c[33][32]; double s[33][32]; };
__constant__ FricTable c_fric;
__constant__ Pow10Table c_pow10;

// LDS carve-up: every array sits at a compile-time offset of two static __shared__
// arrays, so no register holds an LDS pointer.  The LU scratch in LDS is MHW_KA x MHW_KA
// (the usual nonbasic set of a warm-started contact LCP is 1..6 wide); larger systems
// and Lemke's n x n basis use the per-world HBM workspace.
#define MHW_KA 8
#define MHW_LDS_DOUBLES (13 * MH_MAX_BODIES + 7 * MH_MAX_BODIES + 6 * MH_MAX_BODIES + 7 * MH_MAX_BODIES + 7 * MH_MAX_BODIES + 10 * MH_MAX_BODIES \
                         + MHW_MAX_CONTACTS * (3 * 4 + 5) + MHW_MAX_ROWS * 12 * 2 + MHW_MAX_GROWS * MHW_MAX_GROWS + MHW_MAX_ROWS * 2 + 64 + 64 \
                         + MHW_KA * MHW_KA + 64)
#define MHW_LDS_INTS (MH_MAX_PAIRS + MHW_MAX_CONTACTS * 4 + MHW_MAX_ROWS * 2 + MHW_MAX_CONTACTS + MH_MAX_BODIES * 4 + MHW_MAX_CONTACTS + 1 + 64)
// the one LDS image of a world (file scope so that out-of-line device functions see it)
__shared__ double g_lds_d[MHW_LDS_DOUBLES];
__shared__ int g_lds_i[MHW_LDS_INTS];

struct WorldLds {
  double *st, *qsave, *vsave, *qv, *dqv, *xinv;
  double *cpt, *cnr, *cs1, *cs2, *cdist, *cmu, *cmuv, *ceps, *ccomp;
  double *J, *XJ, *G, *Cv, *imp, *zsol, *qq, *A, *art;
  int *pairs, *cg1, *cg2, *cpair, *cnk, *rowoff, *islc, *islb, *koff, *scr;
  MH_DEV void carve() {
    double* p = g_lds_d; int* ibase = g_lds_i;
    st = p; p += 13 * MH_MAX_BODIES; qsave = p; p += 7 * MH_MAX_BODIES; vsave = p; p += 6 * MH_MAX_BODIES;
    qv = p; p += 7 * MH_MAX_BODIES; dqv = p; p += 7 * MH_MAX_BODIES; xinv = p; p += 10 * MH_MAX_BODIES;
    cpt = p; p += 3 * MHW_MAX_CONTACTS; cnr = p; p += 3 * MHW_MAX_CONTACTS; cs1 = p; p += 3 * MHW_MAX_CONTACTS; cs2 = p; p += 3 * MHW_MAX_CONTACTS;
    cdist = p; p += MHW_MAX_CONTACTS; cmu = p; p += MHW_MAX_CONTACTS; cmuv = p; p += MHW_MAX_CONTACTS; ceps = p; p += MHW_MAX_CONTACTS; ccomp = p; p += MHW_MAX_CONTACTS;
    J = p; p += MHW_MAX_ROWS * 12; XJ = p; p += MHW_MAX_ROWS * 12; G = p; p += MHW_MAX_GROWS * MHW_MAX_GROWS;
    Cv = p; p += MHW_MAX_ROWS; imp = p; p += MHW_MAX_ROWS; zsol = p; p += 64; qq = p; p += 64;
    A = p; p += MHW_KA * MHW_KA; art = p; p += 64;
    int* q = ibase;
    pairs = q; q += MH_MAX_PAIRS; cg1 = q; q += MHW_MAX_CONTACTS; cg2 = q; q += MHW_MAX_CONTACTS; cpair = q; q += MHW_MAX_CONTACTS; cnk = q; q += MHW_MAX_CONTACTS;
    rowoff = q; q += MHW_MAX_ROWS * 2; islc = q; q += MHW_MAX_CONTACTS; islb = q; q += MH_MAX_BODIES * 4; koff = q; q += MHW_MAX_CONTACTS + 1; scr = q; q += 64;
  }
};

// ---- implicit LCP matrices -------------------------------------------------------
// mode 0: _MM of solve_qp_work (ICH-QP:129-148 + setup_QP :271-497), never materialised:
//   [ H  -N' ]   H = [n s t -s -t] blocks of C X C' (+ compliance), N = [Cn rows of H ; friction polygons]
//   [ N   0  ]
// Every entry is one value of G = [Cn;Cs;Ct] X [Cn;Cs;Ct]' (upper blocks stored, lower
// ones read transposed exactly as setup_QP does), mu_i, or -cos/-sin of a polygon edge.
// mode 1: the stabilisation matrix Cn X Cn' (CStab:932-947), read straight from G.
// Lane r caches the description of ITS row; the column is wave-uniform.  One type for
// both so the solvers are instantiated once.
struct WorldMat {
  int mode;
  const double* G; int nic, R, nvars, n;
  const int* koff; const int* islc; const double* cmu; const double* ccomp; const int* cnk;
  int rkind;      // 0 variable row (block ra, contact ri), 1 Cn v+ >= 0 row, 2 friction polygon row, 3 none
  int ra, ri, rj;
  double rmu, rcos, rsin, rcomp;
  MH_DEV static int dir_of(int a) { return (a == 0) ? 0 : ((a == 1 || a == 3) ? 1 : 2); }
  MH_DEV double Gab(int a, int b, int i, int j) const {
    return (a <= b) ? G[(a * nic + i) * R + (b * nic + j)] : G[(b * nic + j) * R + (a * nic + i)];
  }
  MH_DEV void init(int mode_, const WorldLds& L, int nic_, int n_) {
    mode = mode_; G = L.G; nic = nic_; n = n_; koff = L.koff; islc = L.islc; cmu = L.cmu; ccomp = L.ccomp; cnk = L.cnk;
    R = (mode == 0) ? 3 * nic_ : nic_; nvars = 5 * nic_;
    const int r = lane_id();
    rkind = 3; ra = 0; ri = 0; rj = 0; rmu = 0.0; rcos = 0.0; rsin = 0.0; rcomp = 0.0;
    if (mode != 0) return;
    if (r < nvars) { rkind = 0; ra = r / nic; ri = r - ra * nic; rcomp = ccomp[islc[ri]]; }
    else if (r < nvars + nic) { rkind = 1; ri = r - nvars; rcomp = ccomp[islc[ri]]; }
    else if (r < n) {
      rkind = 2;
      const int rr = r - nvars - nic;
      int i = 0; while (rr >= koff[i + 1]) i++;
      ri = i; rj = rr - koff[i];
      const int ci = islc[i]; const int kh = cnk[ci] / 2;
      rmu = cmu[ci]; rcos = c_fric.c[kh][rj]; rsin = c_fric.s[kh][rj];
    }
  }
  MH_DEV double H(int a, int i, int b, int j, double comp) const {
    double g = Gab(dir_of(a), dir_of(b), i, j);
    if ((a >= 3) != (b >= 3)) g = -g;
    if (a == 0 && b == 0 && i == j) g = g + comp;
    return g;
  }
  MH_DEV double at_row(int c) const {
    if (mode != 0) { const int l = lane_id(); return (l < n) ? G[l * n + c] : 0.0; }
    if (c < nvars) {
      const int b = c / nic, j = c - b * nic;
      if (rkind == 0) return H(ra, ri, b, j, rcomp);
      if (rkind == 1) return H(0, ri, b, j, rcomp);
      if (rkind == 2) {
        if (j != ri) return 0.0;
        return (b == 0) ? rmu : ((b == 1 || b == 3) ? -rcos : -rsin);
      }
      return 0.0;
    }
    if (rkind != 0) return 0.0;
    const int cc = c - nvars;
    if (cc < nic) {
      double g = Gab(0, dir_of(ra), cc, ri);
      if (ra >= 3) g = -g;
      if (ra == 0 && ri == cc) g = g + rcomp;
      return -g;
    }
    int i = 0; const int rr = cc - nic; while (rr >= koff[i + 1]) i++;
    if (i != ri) return -0.0;
    const int jj = rr - koff[i]; const int ci = islc[i]; const int kh = cnk[ci] / 2;
    const double v = (ra == 0) ? cmu[ci] : ((ra == 1 || ra == 3) ? -c_fric.c[kh][jj] : -c_fric.s[kh][jj]);
    return -v;
  }
  MH_DEV double diag() const {
    if (mode != 0) { const int l = lane_id(); return (l < n) ? G[l * n + l] : 0.0; }
    return (rkind == 0) ? H(ra, ri, ra, ri, rcomp) : 0.0;
  }
  // norm_inf(M) = max |entry|.  mode 0: H entries, mu, and the polygon edges (cos 0 = sin pi/2 = 1)
  MH_DEV double norm_all() const {
    const int lane = lane_id();
    double m = 0.0;
    if (mode != 0) {
      if (lane < n) for (int c = 0; c < n; c++) { const double a = fabs(G[lane * n + c]); m = (a > m) ? a : m; }
      return wave_max(m);
    }
    for (int e = lane; e < R * R; e += 64) {
      const int r = e / R, c = e - r * R;
      if (c < r && (c / nic) != (r / nic)) continue;       // lower direction blocks are never read
      double g = G[e];
      if (r == c && r < nic) g = g + ccomp[islc[r]];
      const double a = fabs(g); m = (a > m) ? a : m;
    }
    if (lane < nic) { const double a = fabs(cmu[islc[lane]]); m = (a > m) ? a : m; }
    m = wave_max(m);
    return (m > 1.0) ? m : 1.0;
  }
  MH_DEV double offdiag_max() const {
    if (mode != 0) {
      const int lane = lane_id(); double m = 0.0;
      if (lane < n) for (int c = 0; c < n; c++) { const double a = fabs(G[lane * n + c]); if (c != lane && a > m) m = a; }
      return wave_max(m);
    }
    // mode 0: the off-diagonal part contains every polygon edge, every mu, every off-diagonal
    // H entry and the H DIAGONAL values again through the Cn v+ rows / -N' columns
    return norm_all();
  }
};

// The solver chains of the two handlers as ONE out-of-line function (a single copy of
// lcp_fast / lcp_lemke / verify / ladder in the code object):
//   mode 0  ICH-QP:219-225  lcp_fast_regularized(-20,4,-8); on failure z.set_zero(), lcp_lemke_regularized
//   mode 1  CStab:954-955   lcp_fast; on failure lcp_lemke_regularized
struct ChainOut { int ok; double zi; int zsize; unsigned rng_r; int rng_idx; unsigned piv; };
__device__ __noinline__ ChainOut world_lcp_chain(int mode, int nic, int n, double qi, double zi, int zsize,
                                                 unsigned rng_r, int rng_idx, double* lu_ws, int ka)
{
  mode = uni(mode); nic = uni(nic); n = uni(n); zsize = uni(zsize); rng_idx = uni(rng_idx); ka = uni(ka);
  lu_ws = reinterpret_cast<double*>(uni((uint64_t)reinterpret_cast<uintptr_t>(lu_ws)));
  WorldLds L; L.carve();
  WorldMat M; M.init(mode, L, nic, n);
  const double nrm0 = M.norm_all(), dii = M.diag();
  WaveRand rng; rng.r = rng_r; rng.idx = rng_idx;
  LuScratch S; S.small = L.A; S.ka = ka; S.big = lu_ws;
  Trace tr; tr.buf = nullptr; tr.cap = 0; tr.len = 0;
  unsigned ptot = 0;
  bool ok = false;
  for (int stage = 0; stage < 2; stage++) {
    LcpParams P; P.piv_tol = -1.0; P.zero_tol = -1.0; P.min_exp = -20;
    if (stage == 0) {
      if (mode == 0) { P.kind = MH_LCP_FAST_REG; P.step_exp = 4; P.max_exp = -8; }
      else { P.kind = MH_LCP_FAST; P.step_exp = 1; P.max_exp = 1; }
    } else {
      if (mode == 0) zi = 0.0;                              // z.set_zero() keeps the size (ICH-QP:222)
      P.kind = MH_LCP_LEMKE_REG; P.step_exp = 1; P.max_exp = 1;
    }
    unsigned piv = 0;
    ok = lcp_solve_wave(P, c_pow10, n, M, S, L.art, nrm0, dii, qi, zi, zsize, rng, piv, tr);
    ptot += piv;
    if (ok) break;
  }
  ChainOut o; o.ok = ok ? 1 : 0; o.zi = zi; o.zsize = zsize; o.rng_r = rng.r; o.rng_idx = rng.idx; o.piv = ptot;
  return o;
}

// phase ids of the optional in-kernel profile (diagnostic launches only)
enum { PH_BROAD_CA = 0, PH_INTEGRATE, PH_FWDDYN, PH_CONTACTS, PH_ISLANDS, PH_PDATA, PH_MBUILD, PH_LCP, PH_APPLY, PH_STAB, PH_COUNT };

struct WorldWave {
  const mh_scene& sc;
  unsigned long long* prof = nullptr;   // PH_COUNT accumulators (cycles) or null
  unsigned long long pacc[PH_COUNT];
  MH_DEV unsigned long long tick() const { return prof ? __builtin_amdgcn_s_memtime() : 0ull; }
  MH_DEV void tock(int ph, unsigned long long t0) { if (prof) pacc[ph] += __builtin_amdgcn_s_memtime() - t0; }
  WorldLds L;
  int lane, nb, ntot, npt;       // npt: number of (i<j) pairs of the scene
  int nmax;
  double* lu_ws;                 // this world's HBM LU workspace (nmax x nmax doubles)
  int ka;                        // LDS LU block edge (MHW_KA; 0 forces the workspace: test hook)
  // persistent solver state
  WaveRand rng;
  double zlast_l, zbuf_l;        // lane i holds _zlast[i] / storage of _z [i]
  int zlast_size, zbuf_size, zbuf_cap;
  int status;
  double time;
  unsigned long long n_steps, n_mini, n_lcp, n_rows, n_piv, n_stab, n_bytes;
  int npairs;                    // ConstraintSimulator::_pairs_to_check (L.pairs)
  int nc;                        // current constraint list size (L.c*)

  MH_DEV WorldWave(const mh_scene& s) : sc(s) {}

  // ---- state helpers (any lane, any body) -----------------------------------
  MH_DEV bool enabled(int b) const { return b < nb; }
  MH_DEV V3 X(int b) const { return v3(L.st[13*b], L.st[13*b+1], L.st[13*b+2]); }
  MH_DEV V3 Vl(int b) const { return v3(L.st[13*b+7], L.st[13*b+8], L.st[13*b+9]); }
  MH_DEV V3 Wa(int b) const { return v3(L.st[13*b+10], L.st[13*b+11], L.st[13*b+12]); }
  MH_DEV V3 point_vel(int b, V3 p) const {
    if (!enabled(b)) return v3(0.0, 0.0, 0.0);
    return Vl(b) + cross(Wa(b), p - X(b));
  }
  MH_DEV void pair_bodies(int p, int& a, int& b) const {
    int i = 0, rem = p;
    while (rem >= ntot - 1 - i) { rem -= ntot - 1 - i; i++; }
    a = i; b = i + 1 + rem;
  }
  MH_DEV V3 plane_n() const { return v3(sc.plane_R[1], sc.plane_R[4], sc.plane_R[7]); }
  MH_DEV V3 to_plane(V3 p) const {
    const double* R = sc.plane_R; V3 d = p - v3(sc.plane_o[0], sc.plane_o[1], sc.plane_o[2]);
    return v3((R[0]*d.x + R[3]*d.y) + R[6]*d.z, (R[1]*d.x + R[4]*d.y) + R[7]*d.z, (R[2]*d.x + R[5]*d.y) + R[8]*d.z);
  }
  MH_DEV V3 from_plane(V3 p) const {
    const double* R = sc.plane_R;
    return v3(sc.plane_o[0] + ((R[0]*p.x + R[1]*p.y) + R[2]*p.z),
              sc.plane_o[1] + ((R[3]*p.x + R[4]*p.y) + R[5]*p.z),
              sc.plane_o[2] + ((R[6]*p.x + R[7]*p.y) + R[8]*p.z));
  }
  MH_DEV static void orthonormal_basis(V3 n, V3& s, V3& t) {
    const double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
    V3 e;
    if (ax <= ay && ax <= az) e = v3(1.0, 0.0, 0.0); else if (ay <= az) e = v3(0.0, 1.0, 0.0); else e = v3(0.0, 0.0, 1.0);
    s = cross(n, e); s = s / norm(s);
    t = cross(n, s);
  }

  // ---- pair-lane geometry ---------------------------------------------------------
  // swept bounds overlap of pair p (CCD.cpp:702-876 with SSL.cpp:550-579 bounds)
  MH_DEV void bounds(int b, double dt, V3& lo, V3& hi) const {
    if (!enabled(b)) { lo = v3(-MHW_INF, -MHW_INF, -MHW_INF); hi = v3(MHW_INF, MHW_INF, MHW_INF); return; }
    const V3 c = X(b);
    const V3 vdt = Vl(b) * dt, wdt = Wa(b) * dt;
    const V3 lin = vdt + cross(c, wdt);
    const V3 p2 = c + lin;
    const double r = sc.geom_dim[b][0];
    lo = v3(((c.x < p2.x) ? c.x : p2.x) - r, ((c.y < p2.y) ? c.y : p2.y) - r, ((c.z < p2.z) ? c.z : p2.z) - r);
    hi = v3(((c.x > p2.x) ? c.x : p2.x) + r, ((c.y > p2.y) ? c.y : p2.y) + r, ((c.z > p2.z) ? c.z : p2.z) + r);
  }
  // writes the compacted pair list to `out` (LDS ints); returns its length
  MH_DEV int broad_phase(double dt, int* out) {
    bool keep = false;
    if (lane < npt) {
      int a, b; pair_bodies(lane, a, b);
      V3 loa, hia, lob, hib; bounds(a, dt, loa, hia); bounds(b, dt, lob, hib);
      keep = (loa.x <= hib.x && lob.x <= hia.x) && (loa.y <= hib.y && lob.y <= hia.y) && (loa.z <= hib.z && lob.z <= hia.z);
      keep = keep && (sc.pair_enabled[lane] != 0) && (enabled(a) || enabled(b));
    }
    const uint64_t m = ballot(keep);
    wave_sync();
    if (keep) out[popc(m & lanes_below(lane))] = lane;
    wave_sync();
    return popc(m);
  }
  // signed distance + closest points of pair p (SpherePrimitive.cpp:104-136, PlanePrimitive.cpp:385-411)
  MH_DEV double signed_dist(int p, V3& pa, V3& pb, int& a, int& b) const {
    pair_bodies(p, a, b);
    if (enabled(a) && enabled(b)) {
      const V3 ca = X(a), cb = X(b);
      const double ra = sc.geom_dim[a][0], rb = sc.geom_dim[b][0];
      const V3 ab = cb - ca;
      const double len = norm(ab);
      const double d = len - ra - rb;
      const V3 u = ab / len;
      const double sa = (d > 0.0) ? ra : ra + d, sb = (d > 0.0) ? rb : rb + d;
      pa = ca + u * sa; pb = cb - u * sb;
      return d;
    }
    const int s = enabled(a) ? a : b;
    const V3 cp = to_plane(X(s));
    const double r = sc.geom_dim[s][0];
    const double low = cp.y + (-1.0 * r);
    const V3 on_plane = from_plane(v3(cp.x, 0.0, cp.z));
    const V3 on_sphere = from_plane(v3(cp.x, low, cp.z));
    if (s == a) { pa = on_sphere; pb = on_plane; } else { pa = on_plane; pb = on_sphere; }
    return low;
  }
  // contact of pair p if within TOL (CCD.inl:804-847, 1164-1207)
  MH_DEV bool make_contact(int p, double TOL, int& g1, int& g2, V3& pt, V3& n, double& dist) const {
    int a, b; pair_bodies(p, a, b);
    if (enabled(a) && enabled(b)) {
      const V3 cA = X(a), cB = X(b);
      const double rA = sc.geom_dim[a][0], rB = sc.geom_dim[b][0];
      const V3 d = cA - cB;
      const double len = norm(d);
      dist = len - rA - rB;
      if (dist > TOL) return false;
      n = d / len;
      const V3 closest_A = cA - n * rA, closest_B = cB + n * rB;
      pt = (closest_A + closest_B) * 0.5;
      g1 = a; g2 = b;
      return true;
    }
    const int s = enabled(a) ? a : b, pl = enabled(a) ? b : a;
    const V3 cp = to_plane(X(s));
    const double r = sc.geom_dim[s][0];
    dist = cp.y - r;
    if (dist > TOL) return false;
    pt = from_plane(v3(cp.x, 0.5 * (cp.y - r), cp.z));
    n = plane_n(); g1 = s; g2 = pl;
    return true;
  }
  MH_DEV double rel_vel(int g1, int g2, V3 pt, V3 dir) const { return dot(dir, point_vel(g1, pt) - point_vel(g2, pt)); }
  MH_DEV double calc_max_dist(int b, V3 n, double rmax) const {      // CCD.cpp:585-609
    if (!enabled(b)) return 0.0;
    const V3 xd0 = Vl(b) + cross(X(b), Wa(b));
    return dot(n, xd0) + norm(cross(Wa(b), n)) * rmax;
  }
  MH_DEV double next_CA_generic(int p) const {                     // CCD.cpp:238-405 (sphere pairs)
    int g1, g2; V3 pt, n; double dist;
    if (!make_contact(p, MH_NEAR_ZERO, g1, g2, pt, n, dist)) return MHW_INF;
    if (rel_vel(g1, g2, pt, n) < -MH_NEAR_ZERO) return 0.0;
    return MHW_INF;
  }
  MH_DEV double CA_generic(int p, double dist, V3 pa, V3 pb, int a, int b) const {   // CCD.cpp:169-235
    if (dist <= 0.0) return next_CA_generic(p);
    const V3 d0 = pa - pb;
    const V3 n0 = d0 / norm(d0);
    const double tA = calc_max_dist(a, -n0, enabled(a) ? sc.geom_dim[a][0] : 0.0);
    const double tB = calc_max_dist(b, n0, enabled(b) ? sc.geom_dim[b][0] : 0.0);
    double total = tA + tB;
    if (total < 0.0) total = 0.0;
    const double cand = dist / total;
    return (cand < MHW_INF) ? cand : MHW_INF;
  }
  MH_DEV double CA_step(int p) const {                             // CCD.cpp:138-166
    V3 pa, pb; int a, b;
    const double dist = signed_dist(p, pa, pb, a, b);
    if (dist > MH_NEAR_ZERO) return CA_generic(p, dist, pa, pb, a, b);
    int g1, g2; V3 pt, n; double cd;
    if (make_contact(p, MH_NEAR_ZERO, g1, g2, pt, n, cd) && fabs(rel_vel(g1, g2, pt, n)) < MH_NEAR_ZERO * 10) return MHW_INF;
    return CA_generic(p, dist, pa, pb, a, b);
  }

  // ---- body-lane dynamics ---------------------------------------------------------
  MH_DEV void rot(int b, double R[9]) const {
    const double x = L.st[13*b+3], y = L.st[13*b+4], z = L.st[13*b+5], w = L.st[13*b+6];
    R[0] = 1.0 - 2.0 * (y*y + z*z); R[1] = 2.0 * (x*y - z*w);       R[2] = 2.0 * (x*z + y*w);
    R[3] = 2.0 * (x*y + z*w);       R[4] = 1.0 - 2.0 * (x*x + z*z); R[5] = 2.0 * (y*z - x*w);
    R[6] = 2.0 * (x*z - y*w);       R[7] = 2.0 * (y*z + x*w);       R[8] = 1.0 - 2.0 * (x*x + y*y);
  }
  MH_DEV void inertia_world(int b, double Jw[9]) const {
    double R[9]; rot(b, R);
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) T[3*i+j] = R[3*i+j] * sc.inertia[b][j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Jw[3*i+j] = (T[3*i] * R[3*j] + T[3*i+1] * R[3*j+1]) + T[3*i+2] * R[3*j+2];
    Jw[1] = Jw[3]; Jw[2] = Jw[6]; Jw[5] = Jw[7];
  }
  // inverse_SPD of the 1x1 mass block and the 3x3 world inertia, same operation
  // order as oracle/linalg.hpp chol_factor + chol_solve (left-looking dpotf2)
  MH_DEV void inv_inertia(int b, double& im, double Ji[9]) const {
    { const double l = sqrt(sc.mass[b]); double e = 1.0 / l; im = e / l; }
    double Jw[9]; inertia_world(b, Jw);
    // lower Cholesky of A (col-major A[i+3j] = Jw[3i+j], symmetric)
    double l00, l10, l20, l11, l21, l22;
    { double ajj = Jw[0]; ajj = sqrt(ajj); l00 = ajj; l10 = Jw[3] / ajj; l20 = Jw[6] / ajj; }
    { double ajj = Jw[4]; ajj = ajj - l10 * l10; ajj = sqrt(ajj); l11 = ajj; double s = Jw[7]; s = s - l20 * l10; l21 = s / ajj; }
    { double ajj = Jw[8]; ajj = ajj - l20 * l20; ajj = ajj - l21 * l21; ajj = sqrt(ajj); l22 = ajj; }
    double col[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double b0 = (c == 0) ? 1.0 : 0.0, b1 = (c == 1) ? 1.0 : 0.0, b2 = (c == 2) ? 1.0 : 0.0;
      // forward: L y = e
      b0 = b0 / l00; b1 = b1 - b0 * l10; b2 = b2 - b0 * l20;
      b1 = b1 / l11; b2 = b2 - b1 * l21;
      b2 = b2 / l22;
      // backward: L^T x = y
      { double s = b2; b2 = s / l22; }
      { double s = b1; s = s - l21 * b2; b1 = s / l11; }
      { double s = b0; s = s - l10 * b1; s = s - l20 * b2; b0 = s / l00; }
      col[c][0] = b0; col[c][1] = b1; col[c][2] = b2;
    }
    // A^-1 (col-major): column c = col[c]; mirror lower -> upper; then Ji[3r+c] = Ainv[r + 3c]
    double Ai[9];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int i = 0; i < 3; i++) Ai[i + 3*c] = col[c][i];
    Ai[0 + 3*1] = Ai[1 + 3*0]; Ai[0 + 3*2] = Ai[2 + 3*0]; Ai[1 + 3*2] = Ai[2 + 3*1];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Ji[3*r+c] = Ai[r + 3*c];
  }
  // all bodies: X blocks into LDS (lane b)
  MH_DEV void compute_xinv() {
    wave_sync();
    if (lane < nb) {
      double im, Ji[9]; inv_inertia(lane, im, Ji);
      L.xinv[10*lane] = im;
#pragma unroll
      for (int k = 0; k < 9; k++) L.xinv[10*lane + 1 + k] = Ji[k];
    }
    wave_sync();
  }
  MH_DEV void euler_vel(int b, double qd[7]) const {
    const V3 v = Vl(b), w = Wa(b);
    const double x = L.st[13*b+3], y = L.st[13*b+4], z = L.st[13*b+5], ww = L.st[13*b+6];
    qd[0] = v.x; qd[1] = v.y; qd[2] = v.z;
    qd[3] = 0.5 * ((ww * w.x + z * w.y) - y * w.z);
    qd[4] = 0.5 * ((ww * w.y + x * w.z) - z * w.x);
    qd[5] = 0.5 * ((ww * w.z + y * w.x) - x * w.y);
    qd[6] = 0.5 * (((-x * w.x) - y * w.y) - z * w.z);
  }
  MH_DEV void set_coords(int b, const double q[7]) {
    L.st[13*b] = q[0]; L.st[13*b+1] = q[1]; L.st[13*b+2] = q[2];
    const double nrm = sqrt(((q[3]*q[3] + q[4]*q[4]) + q[5]*q[5]) + q[6]*q[6]);
    L.st[13*b+3] = q[3] / nrm; L.st[13*b+4] = q[4] / nrm; L.st[13*b+5] = q[5] / nrm; L.st[13*b+6] = q[6] / nrm;
  }
  // q <- base + t * dq for all bodies (stabilisation), then set (normalising)
  MH_DEV void set_q_from(const double* base, const double* dq, double t, bool scale) {
    wave_sync();
    if (lane < nb) {
      double q[7];
#pragma unroll
      for (int k = 0; k < 7; k++) { double v = dq[7*lane + k]; if (scale) v = v * t; q[k] = v + base[7*lane + k]; }
      set_coords(lane, q);
    }
    wave_sync();
  }

  // ---- constraint list (contact lanes) -----------------------------------------------
  MH_DEV void store_contact(int slot, int p, int g1, int g2, V3 pt, V3 n, double dist) {
    V3 s, t; orthonormal_basis(n, s, t);
    L.cg1[slot] = g1; L.cg2[slot] = g2; L.cpair[slot] = p; L.cnk[slot] = sc.cp_nk[p];
    L.cpt[3*slot] = pt.x; L.cpt[3*slot+1] = pt.y; L.cpt[3*slot+2] = pt.z;
    L.cnr[3*slot] = n.x; L.cnr[3*slot+1] = n.y; L.cnr[3*slot+2] = n.z;
    L.cs1[3*slot] = s.x; L.cs1[3*slot+1] = s.y; L.cs1[3*slot+2] = s.z;
    L.cs2[3*slot] = t.x; L.cs2[3*slot+1] = t.y; L.cs2[3*slot+2] = t.z;
    L.cdist[slot] = dist; L.cmu[slot] = sc.cp_mu_coulomb[p]; L.cmuv[slot] = sc.cp_mu_viscous[p];
    L.ceps[slot] = sc.cp_epsilon[p]; L.ccomp[slot] = sc.cp_compliance[p];
  }
  MH_DEV V3 Cpt(int i) const { return v3(L.cpt[3*i], L.cpt[3*i+1], L.cpt[3*i+2]); }
  MH_DEV V3 Cdir(int i, int d) const {
    const double* s = (d == 0) ? L.cnr : (d == 1 ? L.cs1 : L.cs2);
    return v3(s[3*i], s[3*i+1], s[3*i+2]);
  }
  MH_DEV double contact_vn(int i) const { return rel_vel(L.cg1[i], L.cg2[i], Cpt(i), Cdir(i, 0)); }

  // ---- islands (UC:940-1194), uniform scalar BFS over <= MH_MAX_BODIES nodes --------
  // Output: list of islands as (contact list in L.islc[off..], sorted unique bodies).
  // To keep LDS small the caller processes one island at a time through a callback-
  // style loop: next_island() pops the next island into L.islc / L.islb.
  uint32_t isl_node_mask;   // bodies still to visit
  uint32_t isl_done_mask;   // contacts already assigned
  MH_DEV void islands_begin() {
    uint32_t nodes = 0;
    for (int i = 0; i < nc; i++) {
      const int a = L.cg1[i], b = L.cg2[i];
      if (enabled(a)) nodes |= 1u << a;
      if (enabled(b)) nodes |= 1u << b;
    }
    isl_node_mask = nodes; isl_done_mask = 0;
  }
  // returns false when no island is left; else fills L.islc[0..nic), L.islb[0..nib) (sorted unique)
  MH_DEV bool next_island(int& nic, int& nib) {
    while (isl_node_mask) {
      const int start = __ffs((int)isl_node_mask) - 1;
      uint32_t processed = 0, bodies = 0;
      int qh = 0, qt = 0;
      int* queue = L.scr;           // <= 64 entries: each edge can enqueue a body once more
      nic = 0;
      wave_sync();
      if (lane == 0) queue[0] = start;
      qt = 1;
      wave_sync();
      while (qh < qt) {
        const int nd = queue[qh]; qh++;
        isl_node_mask &= ~(1u << nd);
        bodies |= 1u << nd;
        processed |= 1u << nd;
        // neighbours in insertion order: edges were inserted per contact as (a,b),(b,a)
        for (int i = 0; i < nc; i++) {
          const int a = L.cg1[i], b = L.cg2[i];
          if (!(enabled(a) && enabled(b))) continue;
          int nbr = -1;
          if (a == nd) nbr = b; else if (b == nd) nbr = a;
          if (nbr >= 0 && !((processed >> nbr) & 1u) && qt < 64) { wave_sync(); if (lane == 0) queue[qt] = nbr; qt++; wave_sync(); }
        }
        for (int i = 0; i < nc; i++)
          if (!((isl_done_mask >> i) & 1u) && (L.cg1[i] == nd || L.cg2[i] == nd)) {
            wave_sync(); if (lane == 0) L.islc[nic] = i; nic++; isl_done_mask |= 1u << i; wave_sync();
          }
      }
      if (nic == 0) continue;
      nib = 0;
      wave_sync();
      for (int b = 0; b < nb; b++) if ((bodies >> b) & 1u) { if (lane == 0) L.islb[nib] = b; nib++; }
      wave_sync();
      return true;
    }
    return false;
  }

  // ---- problem data (Jacobian-row lanes) ---------------------------------------------
  // rows r = d*nic + i (direction-major); ndir = 3 (impact) or 1 (stabilisation)
  MH_DEV int gc_of(int body, int nib) const { for (int k = 0; k < nib; k++) if (L.islb[k] == body) return 6 * k; return -1; }
  MH_DEV void compute_problem_data(int nic, int nib, int ndir) {
    const int R = ndir * nic;
    compute_xinv();
    if (lane < R) {
      const int d = lane / nic, i = lane - d * nic;
      const int ci = L.islc[i];
      const V3 dir = Cdir(ci, d), pt = Cpt(ci);
      const int bodies2[2] = { L.cg1[ci], L.cg2[ci] };
      double cv = 0.0;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int b = bodies2[k];
        double* Jr = L.J + 12 * lane + 6 * k; double* XJr = L.XJ + 12 * lane + 6 * k;
        if (!enabled(b)) { L.rowoff[2*lane + k] = -1; continue; }
        const V3 dd = (k == 0) ? dir : -dir;
        const V3 r = pt - X(b);
        const V3 rxd = cross(r, dd);
        L.rowoff[2*lane + k] = gc_of(b, nib);
        Jr[0] = dd.x; Jr[1] = dd.y; Jr[2] = dd.z; Jr[3] = rxd.x; Jr[4] = rxd.y; Jr[5] = rxd.z;
        const double im = L.xinv[10*b]; const double* Ji = L.xinv + 10*b + 1;
        XJr[0] = dd.x * im; XJr[1] = dd.y * im; XJr[2] = dd.z * im;
#pragma unroll
        for (int c = 0; c < 3; c++) XJr[3 + c] = (rxd.x * Ji[c] + rxd.y * Ji[3 + c]) + rxd.z * Ji[6 + c];
        const V3 vl = Vl(b), wa = Wa(b);
        double tmp = 0.0;
        tmp = tmp + dd.x * vl.x; tmp = tmp + dd.y * vl.y; tmp = tmp + dd.z * vl.z;
        tmp = tmp + rxd.x * wa.x; tmp = tmp + rxd.y * wa.y; tmp = tmp + rxd.z * wa.z;
        cv = cv + tmp;
      }
      L.Cv[lane] = cv;
    }
    wave_sync();
    // G(r,c) for direction blocks a <= b:  sum over blocks of row r of dot6(J_r[blk], XJ_c[same body])
    if (lane < R) {
      const int a = lane / nic;
      for (int c = 0; c < R; c++) {
        const int b = c / nic;
        if (b < a) continue;
        double res = 0.0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int off = L.rowoff[2*lane + k];
          if (off < 0) continue;
          double tmp = 0.0;
          int kk = -1;
          if (L.rowoff[2*c] == off) kk = 0; else if (L.rowoff[2*c + 1] == off) kk = 1;
          if (kk >= 0) {
            const double* Jr = L.J + 12 * lane + 6 * k; const double* XJc = L.XJ + 12 * c + 6 * kk;
#pragma unroll
            for (int q = 0; q < 6; q++) tmp = tmp + Jr[q] * XJc[q];
          }
          res = res + tmp;
        }
        L.G[lane * R + c] = res;
      }
    }
    wave_sync();
  }
  // block element with a<=b stored, transposes otherwise (setup_QP :392-401)
  MH_DEV double Gab(int R, int nic, int a, int b, int i, int j) const {
    return (a <= b) ? L.G[(a * nic + i) * R + (b * nic + j)] : L.G[(b * nic + j) * R + (a * nic + i)];
  }
  MH_DEV void account(int n, unsigned piv) { n_lcp++; n_rows += (unsigned long long)n; n_piv += piv; n_bytes += 8ull * ((unsigned long long)n * n + 2ull * n); }

  // solve_qp_work's chain on the persistent _z/_zlast (ICH-QP:157-233); on success the
  // solution is left in L.zsol[0..n) and in zlast/zbuf
  MH_DEV bool solve_impact_lcp(int nic) {
    int nk_total = 0;
    wave_sync();
    if (lane == 0) { int acc = 0; for (int i = 0; i < nic; i++) { L.koff[i] = acc; acc += L.cnk[L.islc[i]] / 2; } L.koff[nic] = acc; }
    wave_sync();
    nk_total = L.koff[nic];
    const int nvars = 5 * nic, n = nvars + nic + nk_total;
    if (n > nmax || n > MH_LCP_MAX_N_WAVE) { status |= MH_WORLD_UNSUPPORTED; return false; }
    // _MM stays implicit (WorldMat mode 0); _qq per lane
    unsigned long long tm = tick();
    double qi = 0.0;
    if (lane < n) {
      const int r = lane;
      if (r < nvars) { const int a = r / nic, i = r - a * nic; const double v = L.Cv[WorldMat::dir_of(a) * nic + i]; qi = (a >= 3) ? -v : v; }
      else if (r < nvars + nic) qi = L.Cv[r - nvars];
      else {
        int i = 0; const int rr = r - nvars - nic; while (rr >= L.koff[i + 1]) i++;
        const double cs = L.Cv[nic + i], ct = L.Cv[2 * nic + i];
        const double vel = sqrt(cs * cs + ct * ct);
        qi = L.cmuv[L.islc[i]] * vel;
      }
    }
    tock(PH_MBUILD, tm);
    tm = tick();
    // z.resize(n); warm start from _zlast when sizes match (ICH-QP:158-162)
    double zi;
    if (n > zbuf_cap) zi = 0.0; else zi = zbuf_l;
    if (lane >= n) zi = 0.0;
    if (n == zlast_size) zi = (lane < n) ? zlast_l : 0.0;
    const ChainOut co = world_lcp_chain(0, nic, n, qi, zi, n, rng.r, rng.idx, lu_ws, ka);
    rng.r = co.rng_r; rng.idx = uni(co.rng_idx);
    zi = co.zi;
    const bool ok = uni(co.ok) != 0;
    const unsigned ptot = uni(co.piv);
    account(n, ptot);
    tock(PH_LCP, tm);
    if (!ok) { status |= MH_WORLD_LCP_FAILED; return false; }
    zlast_size = n; if (lane < n) zlast_l = zi;
    if (lane < n) zbuf_l = zi;
    if (zbuf_cap < n) zbuf_cap = n;
    zbuf_size = n;
    wave_sync();
    if (lane < n) L.zsol[lane] = zi;
    wave_sync();
    return true;
  }

  // dv = X_CnT cn + X_CsT cs + X_CtT ct ; v += dv   (ICH:343-348, 1784-1798); gc lanes
  MH_DEV void apply_impulses(int nic, int nib, int ndir) {
    wave_sync();
    const int ngc = 6 * nib;
    if (lane < ngc) {
      const int bi = lane / 6, k = lane - 6 * bi;
      double dv = 0.0;
      for (int d = 0; d < ndir; d++) {
        double tmp = 0.0;
        for (int j = 0; j < nic; j++) {
          const int r = d * nic + j;
          const double t = L.imp[r];
          int kk = -1;
          if (L.rowoff[2*r] == 6 * bi) kk = 0; else if (L.rowoff[2*r + 1] == 6 * bi) kk = 1;
          if (kk >= 0) tmp = tmp + t * L.XJ[12 * r + 6 * kk + k];
        }
        if (d == 0) dv = tmp; else dv = dv + tmp;
      }
      const int b = L.islb[bi];
      L.st[13*b + 7 + k] = L.st[13*b + 7 + k] + dv;
    }
    wave_sync();
  }
  // update_constraint_velocities_from_impulses (ICH:427-464); row lanes
  MH_DEV void update_constraint_vels(int nic) {
    const int R = 3 * nic;
    double cv = 0.0;
    if (lane < R) {
      const int a = lane / nic, i = lane - a * nic;
      cv = L.Cv[lane];
      for (int b = 0; b < 3; b++) {
        double t = 0.0;
        for (int j = 0; j < nic; j++) t = t + L.imp[b * nic + j] * Gab(R, nic, a, b, i, j);
        cv = cv + t;
      }
    }
    wave_sync();
    if (lane < R) L.Cv[lane] = cv;
    wave_sync();
  }
  // q.update_from_stacked_qp(z) (UCPD:218-228): imp rows from the epd-layout z in L.zsol
  MH_DEV void impulses_from_z(int nic) {
    wave_sync();
    if (lane < nic) {
      const int i = lane;
      L.imp[i] = L.zsol[i];
      double s = L.zsol[nic + i];   s = s - L.zsol[3 * nic + i]; L.imp[nic + i] = s;
      double t = L.zsol[2 * nic + i]; t = t - L.zsol[4 * nic + i]; L.imp[2 * nic + i] = t;
    }
    wave_sync();
  }
  MH_DEV double min_cnv(int nic) const { return wave_min((lane < nic) ? L.Cv[lane] : MHW_INF); }

  // apply_model_to_connected_constraints (ICH:530-626)
  MH_DEV void apply_model(int nic, int nib) {
    unsigned long long t0 = tick();
    compute_problem_data(nic, nib, 3);
    tock(PH_PDATA, t0);
    if (!solve_impact_lcp(nic)) return;
    t0 = tick();
    // repack: the epd layout equals the first 5 nic entries; _z shrinks to N_VARS (ICH-QP:244)
    zbuf_size = 5 * nic;
    impulses_from_z(nic); apply_impulses(nic, nib, 3);
    update_constraint_vels(nic);
    const double minv = min_cnv(nic);
    // apply_restitution(_epd, _z) (ICH:470-491)
    bool ch = false;
    double zr = 0.0;
    if (lane < nic) { zr = L.zsol[lane] * L.ceps[L.islc[lane]]; ch = zr > MH_NEAR_ZERO; }
    const bool changed = ballot(ch) != 0ull;
    wave_sync();
    if (lane < nic) { L.zsol[lane] = zr; zbuf_l = zr; }
    wave_sync();
    if (changed) {
      impulses_from_z(nic); apply_impulses(nic, nib, 3);
      update_constraint_vels(nic);
      const double minv_plus = min_cnv(nic);
      if (minv_plus < 0.0 && minv_plus < minv - MH_NEAR_ZERO) {
        if (!solve_impact_lcp(nic)) return;
        zbuf_size = 5 * nic;
        impulses_from_z(nic); apply_impulses(nic, nib, 3);
      }
    }
    tock(PH_APPLY, t0);
  }

  // calc_impacting_unilateral_constraint_forces + apply_model (CSim:298-355, ICH:96-168)
  MH_DEV void handle_impacts() {
    if (nc == 0) return;
    if (ballot(lane < nc && contact_vn(lane) < -MH_NEAR_ZERO) == 0ull) return;
    islands_begin();
    // the reference first removes inactive groups using the PRE-impact velocities, then
    // applies the model island by island; later islands never change earlier activity
    // decisions because islands share no enabled body
    uint32_t active_contacts = 0;
    int nic, nib;
    while (true) {
      const unsigned long long ti = tick();
      const bool more = next_island(nic, nib);
      tock(PH_ISLANDS, ti);
      if (!more) break;
      const uint64_t act = ballot(lane < nic && contact_vn(L.islc[lane < nic ? lane : 0]) < -MH_NEAR_ZERO);
      if (act == 0ull) continue;                                    // remove_inactive_groups
      const bool all_inf = ballot(lane < nic && L.cmu[L.islc[lane < nic ? lane : 0]] < 1e2) == 0ull;
      for (int i = 0; i < nic; i++) active_contacts |= 1u << L.islc[i];
      if (all_inf) { status |= MH_WORLD_UNSUPPORTED; continue; }  // no-slip model: not built yet
      apply_model(nic, nib);
    }
    if (ballot(lane < nc && ((active_contacts >> lane) & 1u) && contact_vn(lane) < -MH_NEAR_ZERO) != 0ull) status |= MH_WORLD_IMPACT_TOL;
  }

  // ---- time stepping ---------------------------------------------------------------
  MH_DEV double next_CA_step() const {
    const double e = (lane < npairs) ? CA_step(L.pairs[lane]) : MHW_INF;
    return wave_min(e);
  }
  // fills the constraint list from the sim pair list (CSim:488-537)
  MH_DEV void find_unilateral_constraints() {
    bool has = false; int g1 = 0, g2 = 0, p = 0; V3 pt = v3(0, 0, 0), n = v3(0, 0, 0); double dist = 0.0;
    if (lane < npairs) {
      p = L.pairs[lane];
      V3 pa, pb; int a, b;
      const double d = signed_dist(p, pa, pb, a, b);
      if (d < sc.contact_dist_thresh) has = make_contact(p, sc.contact_dist_thresh, g1, g2, pt, n, dist);
    }
    const uint64_t m = ballot(has);
    nc = popc(m);
    wave_sync();
    if (nc > MHW_MAX_CONTACTS) { status |= MH_WORLD_UNSUPPORTED; nc = 0; return; }
    if (has) store_contact(popc(m & lanes_below(lane)), p, g1, g2, pt, n, dist);
    wave_sync();
  }
  MH_DEV double do_mini_step(double dt) {                          // TSS:114-222
    wave_sync();
    if (lane < nb) for (int k = 0; k < 7; k++) L.qsave[7*lane + k] = L.st[13*lane + k];
    wave_sync();
    double h = 0.0;
    while (h < dt) {
      unsigned long long t0 = tick();
      npairs = broad_phase(dt - h, L.pairs);
      const double CA = next_CA_step();
      tock(PH_BROAD_CA, t0);
      if (CA <= 0.0) break;
      t0 = tick();
      double tc = (sc.min_step_size > CA) ? sc.min_step_size : CA;
      tc = ((dt - h) < tc) ? (dt - h) : tc;
      wave_sync();
      if (lane < nb) {
        set_coords(lane, L.qsave + 7 * lane);
        double qd[7]; euler_vel(lane, qd);
        double q[7];
#pragma unroll
        for (int i = 0; i < 7; i++) { q[i] = qd[i] * (h + tc); q[i] = q[i] + L.qsave[7*lane + i]; }
        set_coords(lane, q);
      }
      wave_sync();
      h += tc;
      tock(PH_INTEGRATE, t0);
    }
    // forward dynamics + velocity integration (TSS:173-192; GravityForce.cpp:33-69)
    unsigned long long t1 = tick();
    if (lane < nb) {
      const int b = lane;
      const double m = sc.mass[b];
      const V3 f = v3(sc.gravity[0] * m, sc.gravity[1] * m, sc.gravity[2] * m);
      const V3 xdd = f / m;
      double Jw[9]; inertia_world(b, Jw);
      const V3 w = Wa(b);
      const V3 Jww = v3((Jw[0]*w.x + Jw[1]*w.y) + Jw[2]*w.z, (Jw[3]*w.x + Jw[4]*w.y) + Jw[5]*w.z, (Jw[6]*w.x + Jw[7]*w.y) + Jw[8]*w.z);
      const V3 tau = -cross(w, Jww);
      double im, Ji[9]; inv_inertia(b, im, Ji);
      const V3 wd = v3((Ji[0]*tau.x + Ji[1]*tau.y) + Ji[2]*tau.z, (Ji[3]*tau.x + Ji[4]*tau.y) + Ji[5]*tau.z, (Ji[6]*tau.x + Ji[7]*tau.y) + Ji[8]*tau.z);
      const V3 vn = Vl(b) + xdd * h, wn = w + wd * h;
      L.st[13*b+7] = vn.x; L.st[13*b+8] = vn.y; L.st[13*b+9] = vn.z;
      L.st[13*b+10] = wn.x; L.st[13*b+11] = wn.y; L.st[13*b+12] = wn.z;
    }
    wave_sync();
    tock(PH_FWDDYN, t1);
    t1 = tick();
    find_unilateral_constraints();
    tock(PH_CONTACTS, t1);
    handle_impacts();
    time += h;
    n_mini++;
    return h;
  }

  // ---- constraint stabilisation -----------------------------------------------------
  // pairwise distances over the SIM pair list: lane k <-> k-th pair (CStab:88-131)
  MH_DEV double eval_unilateral(double& uC) const {
    uC = MHW_INF;
    if (lane < npairs) { V3 pa, pb; int a, b; uC = signed_dist(L.pairs[lane], pa, pb, a, b); }
    return wave_min(uC);
  }
  MH_DEV double eval_at(double t, int idx) {                       // CStab:1281-1298
    set_q_from(L.qv, L.dqv, t, true);
    double uC; eval_unilateral(uC);
    return read_lane(uC, idx);
  }
  MH_DEV static double sign2(double x, double y) { return (y > 0.0) ? fabs(x) : -fabs(x); }
  MH_DEV double ridders(double x1, double x2, double fl, double fh, int idx) {   // CStab:1322-1379
    const double TOL = 1e-4;
    double ans = MHW_INF, fm, fnew, s, xh, xl, xm, xnew;
    if ((fl > 0.0 && fh < 0.0) || (fl < 0.0 && fh > 0.0)) {
      xl = x1; xh = x2;
      for (unsigned j = 0; j < 25; j++) {
        xm = 0.5 * (xl + xh);
        fm = eval_at(xm, idx);
        s = sqrt(fm * fm - fl * fh);
        if (s == 0.0) return ans;
        xnew = xm + (xm - xl) * ((fl >= fh ? 1.0 : -1.0) * fm / s);
        ans = xnew;
        fnew = eval_at(ans, idx);
        if (fabs(fnew) < TOL && fnew >= 0.0) return xnew;
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
  MH_DEV bool update_q() {                                         // CStab:1056-1216 (unilateral part)
    double uC_old, uC;
    eval_unilateral(uC_old);
    set_q_from(L.qv, L.dqv, 1.0, false);
    eval_unilateral(uC);
    const bool br = (lane < npairs) && ((uC_old < 0.0 && uC > 0.0) || (uC_old > 0.0 && uC < 0.0));
    uint64_t brm = ballot(br);
    double t = 1.0;
    { uint64_t m = brm;
      while (m) {
        const int i = ctz(m); m &= m - 1;
        const double root = ridders(0.0, t, read_lane(uC_old, i), read_lane(uC, i), i);
        if (root > 0.0 && root < 1.0) t = (root < t) ? root : t;
      } }
    set_q_from(L.qv, L.dqv, t, true);
    eval_unilateral(uC);
    const double BETA = 0.6;
    while (true) {
      const bool bad = (lane < npairs) && !br && uC < 0.0 && uC_old > uC;
      if (ballot(bad) == 0ull) break;
      t *= BETA;
      if (t < MH_NEAR_ZERO) return false;
      set_q_from(L.qv, L.dqv, t, true);
      eval_unilateral(uC);
    }
    // q = qstar: the bodies already sit at qstar; refresh the q vector from the
    // un-normalised qstar = dq*t + q as the reference does
    wave_sync();
    if (lane < nb) for (int k = 0; k < 7; k++) { double v = L.dqv[7*lane + k] * t; L.qv[7*lane + k] = v + L.qv[7*lane + k]; }
    wave_sync();
    return true;
  }
  MH_DEV void stabilize() {                                        // CStab:167-254
    if (sc.cstab_max_iterations == 0) return;
    wave_sync();
    if (lane < nb) {
      for (int k = 0; k < 6; k++) L.vsave[6*lane + k] = L.st[13*lane + 7 + k];
      for (int k = 0; k < 7; k++) L.qv[7*lane + k] = L.st[13*lane + k];
    }
    wave_sync();
    double uC;
    double max_uvio = eval_unilateral(uC);
    unsigned iterations = 0;
    while (max_uvio < sc.cstab_eps) {
      if (iterations == sc.cstab_max_iterations) break;
      wave_sync();
      if (lane < nb) { for (int k = 0; k < 6; k++) L.st[13*lane + 7 + k] = 0.0; for (int k = 0; k < 7; k++) L.dqv[7*lane + k] = 0.0; }
      wave_sync();
      // compute_problem_data (CStab:347-492): own broad phase (dt = 0), one contact per pair
      int* cpairs = L.scr;
      const int ncp = broad_phase(0.0, cpairs);
      bool has = false; int g1 = 0, g2 = 0, p = 0; V3 pt = v3(0, 0, 0), n = v3(0, 0, 0); double dist = 0.0;
      if (lane < ncp) {
        p = cpairs[lane];
        V3 pa, pb; int a, b;
        const double d = signed_dist(p, pa, pb, a, b);
        if (d >= MH_NEAR_ZERO) {                                   // separated: synthetic contact (CStab:316-331)
          const V3 nn = pb - pa;
          n = nn / norm(nn); pt = pa; g1 = a; g2 = b; dist = d; has = true;
        } else has = make_contact(p, MH_NEAR_ZERO, g1, g2, pt, n, dist);
      }
      const uint64_t m = ballot(has);
      nc = popc(m);
      wave_sync();
      if (nc > MHW_MAX_CONTACTS) { status |= MH_WORLD_UNSUPPORTED; nc = 0; break; }
      if (has) store_contact(popc(m & lanes_below(lane)), p, g1, g2, pt, n, dist);
      wave_sync();
      islands_begin();
      int nic, nib;
      while (next_island(nic, nib)) {
        compute_problem_data(nic, nib, 1);
        const int n_ = nic;
        if (n_ > nmax) { status |= MH_WORLD_UNSUPPORTED; continue; }
        // determine_dq (CStab:932-970): MM = Cn X Cn', qq = dist - |eps| - NEAR_ZERO
        double qi = 0.0;
        if (lane < n_) qi = L.cdist[L.islc[lane]] - fabs(sc.cstab_eps) - MH_NEAR_ZERO;
        wave_sync();
        const ChainOut co = world_lcp_chain(1, nic, n_, qi, 0.0, 0, rng.r, rng.idx, lu_ws, ka);   // fresh local z: cold start
        rng.r = co.rng_r; rng.idx = uni(co.rng_idx);
        const double zi = co.zi; const int zsize = uni(co.zsize);
        const unsigned ptot = uni(co.piv);
        account(n_, ptot);
        // update_from_stacked(pd, z): cn = z[0..nc) (zeros where z is shorter)
        wave_sync();
        if (lane < n_) L.imp[lane] = (lane < zsize) ? zi : 0.0;
        wave_sync();
        apply_impulses(nic, nib, 1);
        // dq of the island's bodies = their Euler-form velocities (CStab:962-969)
        if (lane < nib) { const int b = L.islb[lane]; double qd[7]; euler_vel(b, qd); for (int k = 0; k < 7; k++) L.dqv[7*b + k] = qd[k]; }
        wave_sync();
      }
      if (!update_q()) { status |= MH_WORLD_STAB_FAILED; break; }
      max_uvio = eval_unilateral(uC);
      iterations++;
      n_stab++;
    }
    wave_sync();
    if (lane < nb) for (int k = 0; k < 6; k++) L.st[13*lane + 7 + k] = L.vsave[6*lane + k];
    wave_sync();
  }

  MH_DEV void step(double dt) {                                    // TSS:52-111
    npairs = broad_phase(dt, L.pairs);
    double h = 0.0;
    unsigned guard = 0;
    while (h < dt) {
      h += do_mini_step(dt - h);
      if (++guard > 100000u) { status |= MH_WORLD_STALLED; break; }
    }
    const unsigned long long ts = tick();
    stabilize();
    tock(PH_STAB, ts);
    n_steps++;
  }
};

} // namespace mh
