// Dense LCP solvers for n > 64: one 256-thread workgroup per problem, M read in
// place from HBM, vectors / index sets / the LU scratch in a per-problem HBM
// workspace (L2-resident for the sizes the handlers produce), LDS only for the
// block reductions.
//
// Same algorithms and the same floating-point operation order as the wave
// solvers in mh_lcp_wave.h (and so as the CPU oracle oracle/lcp.hpp):
//   LCP::lcp_fast               /root/reference/src/LCP.cpp:41-196
//   LCP::rand_min               src/LCP.cpp:199-209
//   LCP::lcp_fast_regularized   src/LCP.cpp:212-350
//   LCP::lcp_lemke_regularized  src/LCP.cpp:353-487
//   LCP::lcp_lemke (dense)      src/LCP.cpp:545-1003
//   LinAlgd::solve_fast (dgesv = dgetf2 + dgetrs) call sites src/LCP.cpp:120,838
//
// Mapping: thread t owns rows t, t+256, ...; the sorted index vectors
// _bas/_nonbas are a flag array + a list rebuilt by a block scan; every control
// decision is block-uniform (broadcast through LDS), the libc rand() ring lives
// in LDS and is advanced by thread 0.  The LU is right-looking in panels of 8
// columns (trailing matrix touched once per panel); what bounds it is the number
// of dependent global-memory phases per column (pivot search, swap, scale:
// ~12 barriers each), not flops -- measured: n = 256 x 256 problems, 220 pivots,
// 333 ms.  Keeping the panel in LDS is the next step.
#pragma once
#include "mh_lcp_wave.h"

namespace mh { namespace blk {

constexpr int T = 256;

struct Ws {
  double* A;      // n x n LU scratch (col-major, ld = k)
  double* b;      // rhs / solution by position
  double* w;      // w by variable
  double* x;      // Lemke: basic values by position
  double* d;      // Lemke: direction by position
  double* art;    // Lemke: artificial column
  int* list;      // sorted nonbasic variables
  int* flag;      // 1 = nonbasic
  int* pos;       // position of variable i in list (-1 if basic)
  int* bv;        // Lemke: basis by position
};
MH_DEV size_t ws_doubles(int n) { return (size_t)n * n + 5 * (size_t)n; }
MH_DEV size_t ws_ints(int n) { return 4 * (size_t)n; }

__shared__ double s_rd[T];
__shared__ int s_ri[T];
__shared__ double s_bd[4];
__shared__ int s_bi[4];
__shared__ unsigned s_rng[32];

MH_DEV int tid() { return (int)threadIdx.x; }
MH_DEV void sync() { __syncthreads(); }
MH_DEV double inf() { return __longlong_as_double(0x7ff0000000000000ll); }

// lexicographic (value ascending, index ascending) minimum over the block
MH_DEV void red_min_first(double v, int idx, double& vout, int& iout) {
  const int t = tid();
  s_rd[t] = v; s_ri[t] = idx; sync();
  for (int s = T / 2; s > 0; s >>= 1) {
    if (t < s) { const double v2 = s_rd[t + s]; const int i2 = s_ri[t + s];
      if (v2 < s_rd[t] || (v2 == s_rd[t] && i2 < s_ri[t])) { s_rd[t] = v2; s_ri[t] = i2; } }
    sync();
  }
  vout = s_rd[0]; iout = s_ri[0]; sync();
}
MH_DEV void red_max_first(double v, int idx, double& vout, int& iout) {
  const int t = tid();
  s_rd[t] = v; s_ri[t] = idx; sync();
  for (int s = T / 2; s > 0; s >>= 1) {
    if (t < s) { const double v2 = s_rd[t + s]; const int i2 = s_ri[t + s];
      if (v2 > s_rd[t] || (v2 == s_rd[t] && i2 < s_ri[t])) { s_rd[t] = v2; s_ri[t] = i2; } }
    sync();
  }
  vout = s_rd[0]; iout = s_ri[0]; sync();
}
MH_DEV double red_max(double v) { double o; int i; red_max_first(v, 0, o, i); return o; }
MH_DEV double red_min(double v) { double o; int i; red_min_first(v, 0, o, i); return o; }
MH_DEV int red_sum_int(int v) {
  const int t = tid();
  s_ri[t] = v; sync();
  for (int s = T / 2; s > 0; s >>= 1) { if (t < s) s_ri[t] += s_ri[t + s]; sync(); }
  const int o = s_ri[0]; sync();
  return o;
}
MH_DEV int bcast_i(int v) { if (tid() == 0) s_bi[0] = v; sync(); const int o = s_bi[0]; sync(); return o; }
MH_DEV int rand_next() {           // block-uniform; glibc TYPE_3 ring in LDS (oracle/glibc_rand.h layout)
  if (tid() == 0) {
    const unsigned idx = s_rng[31];
    const unsigned v = s_rng[idx] + s_rng[(idx + 28) % 31];
    s_rng[idx] = v; s_rng[31] = (idx + 1) % 31;
    s_bi[1] = (int)(v >> 1);
  }
  sync(); const int o = s_bi[1]; sync();
  return o;
}

struct Trace2 { int32_t* buf; int cap; int len;
  MH_DEV void push(int32_t v) { if (buf && len < cap && tid() == 0) buf[len] = v; len++; } };

struct Mat { const double* M; int ld; int n;
  MH_DEV double at(int r, int c, double lam) const { const double m = M[r + (size_t)ld * c]; return (r == c) ? m + lam : m; } };

// sorted list of the flagged variables + their positions; returns the count
MH_DEV int build_list(int n, const Ws& W) {
  const int t = tid();
  const int per = (n + T - 1) / T, lo = t * per, hi = (lo + per < n) ? lo + per : n;
  int c = 0;
  for (int i = lo; i < hi; i++) c += W.flag[i] ? 1 : 0;
  s_ri[t] = c; sync();
  if (t == 0) { int acc = 0; for (int k = 0; k < T; k++) { const int v = s_ri[k]; s_ri[k] = acc; acc += v; } s_bi[0] = acc; }
  sync();
  int o = s_ri[t]; const int total = s_bi[0];
  for (int i = lo; i < hi; i++) { if (W.flag[i]) { W.list[o] = i; W.pos[i] = o; o++; } else W.pos[i] = -1; }
  sync();
  return total;
}

// dgesv, one rhs: A k x k col-major (ld = k) and b in the workspace.  Returns LAPACK info (uniform).
//
// Right-looking LU in panels of NB columns.  Every element receives exactly the updates
// a <- a - l*u of dgetf2, in the same order (column steps ascending, each product rounded on its own),
// so the factors are bit-identical to the unblocked routine; what changes is the traffic: the
// trailing matrix is read and written once per PANEL (with the NB multipliers of its row in registers
// and the NB pivot rows staged through LDS) instead of once per column.
constexpr int NB = 8;
constexpr int UCH = 256;                 // columns of the pivot-row block staged in LDS at a time
__shared__ double s_u[NB][UCH];

MH_DEV int lu_solve(int k, double* A, double* b) {
  const int t = tid();
  for (int j0 = 0; j0 < k; j0 += NB) {
    const int nbk = (k - j0 < NB) ? k - j0 : NB;
    // ---- panel: columns j0 .. j0+nbk-1, unblocked, updates confined to the panel ----
    for (int j = j0; j < j0 + nbk; j++) {
      double best = -1.0; int bi = 0x7fffffff;
      for (int r = j + t; r < k; r += T) { const double a = fabs(A[r + (size_t)k * j]); if (a > best) { best = a; bi = r; } }
      double amax; int jp; red_max_first(best, bi, amax, jp);
      if (!(amax != 0.0)) return j + 1;
      if (jp != j) {                                       // full-row swap, as dgetf2 (dlaswp on both sides)
        for (int c = t; c < k; c += T) { const double t0 = A[j + (size_t)k * c], t1 = A[jp + (size_t)k * c]; A[j + (size_t)k * c] = t1; A[jp + (size_t)k * c] = t0; }
        if (t == 0) { const double t0 = b[j]; b[j] = b[jp]; b[jp] = t0; }
        sync();
      }
      if (j < k - 1) {
        const double piv = A[j + (size_t)k * j];
        const bool big = fabs(piv) >= MH_SFMIN;
        const double rcp = 1.0 / piv;
        for (int r = j + 1 + t; r < k; r += T) { double l = A[r + (size_t)k * j]; l = big ? l * rcp : l / piv; A[r + (size_t)k * j] = l; }
        sync();
        const int m = k - j - 1, pc = j0 + nbk - j - 1;    // rows below, panel columns to the right
        const long tot = (long)m * pc;
        for (long e = t; e < tot; e += T) {
          const int r = j + 1 + (int)(e % m), c = j + 1 + (int)(e / m);
          A[r + (size_t)k * c] = A[r + (size_t)k * c] - A[r + (size_t)k * j] * A[j + (size_t)k * c];
        }
        sync();
      }
    }
    const int c0 = j0 + nbk;                               // first trailing column
    if (c0 >= k) break;
    // ---- trailing columns, UCH at a time ----
    for (int cb = c0; cb < k; cb += UCH) {
      const int ncb = (k - cb < UCH) ? k - cb : UCH;
      // pivot-row block U12[s][c] = A[j0+s][c] - sum_{s' < s} L[j0+s][j0+s'] * U12[s'][c]  (updates of steps j0.. in order)
      for (int c = t; c < ncb; c += T) {
        double u[NB];
#pragma unroll
        for (int s2 = 0; s2 < NB; s2++) {
          if (s2 < nbk) {
            double a = A[(j0 + s2) + (size_t)k * (cb + c)];
#pragma unroll
            for (int s1 = 0; s1 < NB; s1++) if (s1 < s2) a = a - A[(j0 + s2) + (size_t)k * (j0 + s1)] * u[s1];
            u[s2] = a;
            A[(j0 + s2) + (size_t)k * (cb + c)] = a;
            s_u[s2][c] = a;
          }
        }
      }
      sync();
      // A22[r][c] -= L[r][j0+s] * U12[s][c], s ascending; one row per thread, its multipliers in registers
      for (int r = c0 + t; r < k; r += T) {
        double l[NB];
#pragma unroll
        for (int s2 = 0; s2 < NB; s2++) l[s2] = (s2 < nbk) ? A[r + (size_t)k * (j0 + s2)] : 0.0;
        for (int c = 0; c < ncb; c++) {
          double a = A[r + (size_t)k * (cb + c)];
#pragma unroll
          for (int s2 = 0; s2 < NB; s2++) if (s2 < nbk) a = a - l[s2] * s_u[s2][c];
          A[r + (size_t)k * (cb + c)] = a;
        }
      }
      sync();
    }
  }
  for (int kk = 0; kk < k; kk++) {                       // unit lower
    const double bk = b[kk];
    sync();
    for (int i = kk + 1 + t; i < k; i += T) b[i] = b[i] - bk * A[i + (size_t)k * kk];
    sync();
  }
  for (int kk = k - 1; kk >= 0; kk--) {                  // upper
    if (t == 0) b[kk] = b[kk] / A[kk + (size_t)k * kk];
    sync();
    const double bk = b[kk];
    for (int i = t; i < kk; i += T) b[i] = b[i] - bk * A[i + (size_t)k * kk];
    sync();
  }
  return 0;
}

// LCP.cpp:199-209 over the variables i with member(i) (list order = index order); val(i) reads
// the candidate.  Consumes exactly one rand().  Returns the chosen variable (uniform).
template <class Val, class Mem>
MH_DEV int rand_min(int n, Val val, Mem member, double tol, double& vsel) {
  const int t = tid();
  double best = inf(); int bi = 0x7fffffff;
  for (int i = t; i < n; i += T) if (member(i)) { const double v = val(i); if (v < best) { best = v; bi = i; } }
  double vmin; int imin; red_min_first(best, bi, vmin, imin);
  int c = 0;
  for (int i = t; i < n; i += T) if (member(i) && i != imin && val(i) < vmin + tol) c++;
  const int cnt = 1 + red_sum_int(c);
  const int r = rand_next() % cnt;
  int chosen = imin;
  if (r != 0) {
    if (t == 0) { int seen = 0, pick = imin; for (int i = 0; i < n; i++) if (member(i) && i != imin && val(i) < vmin + tol) { if (seen == r - 1) { pick = i; break; } seen++; } s_bi[2] = pick; }
    sync(); chosen = s_bi[2]; sync();
  }
  vsel = val(chosen);
  return chosen;
}

// LCP.cpp:41-196.  z (n, in/out) and zsize as in lcp_fast_wave.
MH_DEV bool lcp_fast(const Mat& M, double lam, const Ws& W, const double* q, double* z, int& zsize, double zero_tol,
                     double nrm_lam, unsigned& pivots, Trace2& tr)
{
  const int n = M.n, t = tid();
  if (zero_tol < 0.0) zero_tol = (double)n * nrm_lam * MH_DBL_EPS;
  if (zsize == n) {
    for (int i = t; i < n; i += T) W.flag[i] = !(fabs(z[i]) < zero_tol) ? 1 : 0;
  } else {
    double best = inf(); int bi = 0x7fffffff;
    for (int i = t; i < n; i += T) { const double v = q[i]; if (v < best) { best = v; bi = i; } }
    double qmin; int minw; red_min_first(best, bi, qmin, minw);
    if (qmin > -zero_tol) { for (int i = t; i < n; i += T) z[i] = 0.0; sync(); zsize = n; pivots = 0; return true; }
    for (int i = t; i < n; i += T) W.flag[i] = (i == minw) ? 1 : 0;
  }
  sync();
  const unsigned MAX_PIV = 2u * (unsigned)n;
  for (pivots = 0; pivots < MAX_PIV; pivots++) {
    const int k = build_list(n, W);
    if (k > 0) {
      const long kk2 = (long)k * k;
      for (long e = t; e < kk2; e += T) { const int r = (int)(e % k), c = (int)(e / k); W.A[e] = M.at(W.list[r], W.list[c], lam); }
      for (int r = t; r < k; r += T) W.b[r] = -q[W.list[r]];
      sync();
      if (lu_solve(k, W.A, W.b) != 0) return false;
    }
    // w = Mmix z + qbas on the basic variables (dgemv column order)
    for (int i = t; i < n; i += T) if (!W.flag[i]) {
      double w = 0.0;
      for (int c = 0; c < k; c++) w = w + W.b[c] * M.at(i, W.list[c], lam);
      W.w[i] = w + q[i];
    }
    sync();
    auto wval = [&](int i) { return W.w[i]; };
    auto isb = [&](int i) { return W.flag[i] == 0; };
    auto zval = [&](int i) { return W.b[W.pos[i]]; };
    auto isnb = [&](int i) { return W.flag[i] != 0; };
    double wsel = 0.0; int minw = -1;
    if (k < n) minw = rand_min(n, wval, isb, zero_tol, wsel);
    if (minw < 0 || wsel > -zero_tol) {
      double zsel = 0.0; int minz = -1;
      if (k > 0) minz = rand_min(n, zval, isnb, zero_tol, zsel);
      if (minz >= 0 && zsel < -zero_tol) {
        if (t == 0) W.flag[minz] = 0;
        tr.push(-(int32_t)(minz + 1));
        sync();
      } else {
        for (int i = t; i < n; i += T) z[i] = W.flag[i] ? W.b[W.pos[i]] : 0.0;
        sync();
        zsize = n;
        return true;
      }
    } else {
      tr.push((int32_t)(minw + 1));
      double zsel = 0.0; int minzv = -1;
      if (k > 0) minzv = rand_min(n, zval, isnb, zero_tol, zsel);
      int idx2 = -1;
      if (minzv >= 0 && zsel < -zero_tol) {
        // LCP.cpp:176-187: the POSITION found in the old _z indexes the NEW, re-sorted _nonbas
        if (t == 0) {
          const int posz = W.pos[minzv];
          int lo = 0, hi = k;                               // insertion point of minw in the old list
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (W.list[mid] < minw) lo = mid + 1; else hi = mid; }
          s_bi[3] = (posz < lo) ? W.list[posz] : ((posz == lo) ? minw : W.list[posz - 1]);
        }
        sync(); idx2 = s_bi[3]; sync();
      }
      if (t == 0) { W.flag[minw] = 1; if (idx2 >= 0) W.flag[idx2] = 0; }
      if (idx2 >= 0) tr.push(-(int32_t)(idx2 + 1));
      sync();
    }
  }
  return false;
}

// LCP.cpp:240-249 (strict = false) / :303-312 (strict = true, against M + lam I)
MH_DEV bool verify(const Mat& M, double lam, const Ws& W, const double* q, const double* z, double ZERO_TOL, bool strict)
{
  const int n = M.n, t = tid();
  const double nT = -ZERO_TOL;
  int bad = 0;
  for (int i = t; i < n; i += T) {
    const double zi = z[i];
    if (!(strict ? (zi > nT) : (zi >= nT))) bad = 1;
  }
  if (red_sum_int(bad) != 0) return false;
  int bad_w = 0, bad_zw = 0, bad_hi = 0;
  for (int i = t; i < n; i += T) {
    double w = 0.0;
    for (int c = 0; c < n; c++) { const double zc = z[c]; if (zc != 0.0) w = w + zc * M.at(i, c, lam); }
    w = w + q[i];
    if (!(strict ? (w > nT) : (w >= nT))) bad_w = 1;
    const double zw = z[i] * w;
    if (!(strict ? (zw > nT) : (zw >= nT))) bad_zw = 1;
    if (!(zw < ZERO_TOL)) bad_hi = 1;
  }
  if (red_sum_int(bad_w) != 0) return false;
  if (red_sum_int(bad_zw) != 0) return false;
  return red_sum_int(bad_hi) == 0;
}

// LCP.cpp:545-1003 (dense)
MH_DEV bool lcp_lemke(const Mat& M, double lam, const Ws& W, const double* q, double* z, int& zsize, double piv_tol, double zero_tol,
                      double nrm_lam, unsigned& pivots, Trace2& tr)
{
  const int n = M.n, t = tid();
  const unsigned MAXITER = (50u * (unsigned)n < 1000u) ? 50u * (unsigned)n : 1000u;
  pivots = 0;
  for (int i = t; i < n; i += T) z[i] = 0.0;                        // z.set_zero() (:564)
  sync();
  const int z0size = zsize;
  if (zero_tol <= 0.0) zero_tol = MH_DBL_EPS * nrm_lam * (double)n;
  double best = inf(); int bi = 0x7fffffff;
  for (int i = t; i < n; i += T) { const double v = q[i]; if (v < best) { best = v; bi = i; } }
  double xmin; int lvindex; red_min_first(best, bi, xmin, lvindex);
  if (xmin > -zero_tol) { zsize = n; return true; }
  zsize = 2 * n;                                                    // z.set_zero(2n) (:596)
  const int tt = 2 * n;
  if (z0size != n) for (int i = 0; i < n; i++) (void)rand_next();   // _restart_z0 (:618-620)
  for (int p = t; p < n; p += T) { W.bv[p] = n + p; W.x[p] = q[p]; }
  sync();
  if (!(xmin < 0.0)) { zsize = n; return true; }                    // no negative entry (:737)
  const double PIV_TOL = (piv_tol > 0.0) ? piv_tol : MH_DBL_EPS * (double)n * ((nrm_lam > 1.0) ? nrm_lam : 1.0);
  const double tval = -xmin;
  int leaving = n + lvindex;
  int entering = tt;
  for (int p = t; p < n; p += T) {
    const double xp = W.x[p];
    double u = (xp < 0.0) ? 1.0 : 0.0;
    W.art[p] = u;
    u = u * tval;
    W.x[p] = xp + u;
  }
  sync();
  if (t == 0) { W.x[lvindex] = tval; W.bv[lvindex] = tt; }
  sync();
  for (pivots = 0; pivots < MAXITER; pivots++) {
    if (leaving == tt) {
      for (int p = t; p < n; p += T) { const int id = W.bv[p]; if (id < n) z[id] = W.x[p]; }   // (:804-806)
      sync();
      zsize = n;
      return true;
    }
    if (leaving < n) { entering = n + leaving; for (int p = t; p < n; p += T) W.d[p] = (p == leaving) ? -1.0 : 0.0; }
    else { entering = leaving - n; for (int p = t; p < n; p += T) W.d[p] = M.at(p, entering, lam); }
    // Al = Bl from the basis description; solve Al d = Be
    const long nn = (long)n * n;
    for (long e = t; e < nn; e += T) {
      const int r = (int)(e % n), p = (int)(e / n);
      const int id = W.bv[p];
      double a;
      if (id == tt) a = W.art[r];
      else if (id >= n) a = (r == id - n) ? -1.0 : 0.0;
      else a = M.at(r, id, lam);
      W.A[e] = a;
    }
    sync();
    if (lu_solve(n, W.A, W.d) != 0) return false;                   // singular basis (:840-850), size stays 2n
    double th = inf(); int any = 0;
    for (int p = t; p < n; p += T) { const double dp = W.d[p]; if (dp > PIV_TOL) { any = 1; const double r = (W.x[p] + zero_tol) / dp; th = (r < th) ? r : th; } }
    if (red_sum_int(any) == 0) return false;                        // ray termination (:892-903)
    const double theta = red_min(th);
    int first_keep = 0x7fffffff, tkeep = 0x7fffffff;
    for (int p = t; p < n; p += T) {
      const double dp = W.d[p];
      if (dp > PIV_TOL && (W.x[p] / dp <= theta)) { if (p < first_keep) first_keep = p; if (W.bv[p] == tt && p < tkeep) tkeep = p; }
    }
    double dummy; int fk, tk;
    red_min_first(0.0, first_keep, dummy, fk);
    red_min_first(0.0, tkeep, dummy, tk);
    if (fk == 0x7fffffff) { zsize = n; return false; }              // (:946-958)
    lvindex = (tk != 0x7fffffff) ? tk : fk;
    leaving = W.bv[lvindex];
    const double ratio = W.x[lvindex] / W.d[lvindex];
    sync();
    for (int p = t; p < n; p += T) { const double dp = W.d[p] * ratio; W.x[p] = W.x[p] - dp; }
    sync();
    if (t == 0) { W.x[lvindex] = ratio; W.bv[lvindex] = entering; }
    sync();
    tr.push((int32_t)entering + 1); tr.push((int32_t)leaving + 1);
  }
  zsize = n;
  return false;
}

// the four public solvers (lcp_solve_wave's attempt loop)
MH_DEV bool lcp_solve(const LcpParams& P, const Pow10Table& p10, const Mat& M, const Ws& W, const double* q, double* z, int& zsize,
                      unsigned& pivots, Trace2& tr)
{
  const int n = M.n, t = tid();
  const bool reg = (P.kind == MH_LCP_FAST_REG) || (P.kind == MH_LCP_LEMKE_REG);
  const bool fast = (P.kind == MH_LCP_FAST) || (P.kind == MH_LCP_FAST_REG);
  double m0 = 0.0;
  const long nn = (long)n * n;
  for (long e = t; e < nn; e += T) { const double a = fabs(M.M[(e % n) + (size_t)M.ld * (e / n)]); m0 = (a > m0) ? a : m0; }
  const double nrm0 = red_max(m0);
  const double ZERO_TOL = (P.zero_tol > 0.0) ? P.zero_tol : (double)n * nrm0 * MH_NEAR_ZERO;
  unsigned total = 0;
  double offmax = 0.0;
  int rf = P.min_exp;
  for (int attempt = 0; ; attempt++) {
    double lam = 0.0, nrm = nrm0;
    if (attempt > 0) {
      if (!reg || !(rf < P.max_exp)) break;
      if (attempt == 1) {
        double mo = 0.0;
        for (long e = t; e < nn; e += T) { const int r = (int)(e % n), c = (int)(e / n); if (r != c) { const double a = fabs(M.M[r + (size_t)M.ld * c]); mo = (a > mo) ? a : mo; } }
        offmax = red_max(mo);
      }
      lam = p10.v[rf + 32];
      double md = 0.0;
      for (int i = t; i < n; i += T) { const double a = fabs(M.M[i + (size_t)M.ld * i] + lam); md = (a > md) ? a : md; }
      md = red_max(md);
      nrm = (md > offmax) ? md : offmax;
    }
    if (reg) tr.push(0x40000000 | attempt);
    const bool ok = fast ? lcp_fast(M, lam, W, q, z, zsize, P.zero_tol, nrm, pivots, tr)
                         : lcp_lemke(M, lam, W, q, z, zsize, P.piv_tol, P.zero_tol, nrm, pivots, tr);
    if (!reg) return ok;
    const bool good = ok && verify(M, lam, W, q, z, ZERO_TOL, attempt > 0);
    if (attempt == 0) { if (good) return true; total += pivots; }
    else { total += pivots; if (good) { pivots = total; return true; } rf += (int)P.step_exp; }
  }
  pivots = total;
  return false;
}

} } // namespace mh::blk
