// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of Moby's dense LCP solvers, following /root/reference
// src/LCP.cpp statement by statement:
//
//   lcp_fast              src/LCP.cpp:41-196
//   rand_min              src/LCP.cpp:199-209
//   lcp_fast_regularized  src/LCP.cpp:212-350
//   lcp_lemke_regularized src/LCP.cpp:353-487   (dense)
//   lcp_lemke             src/LCP.cpp:545-1003  (dense)
//   insertion_sort        include/Moby/insertion_sort:18-39
//
// The sparse Lemke overloads (LCP.cpp:1006-1381) and fast_pivoting
// (LCP.cpp:1401-1513) have no caller in the handlers and are out
// of scope.
//
// PARITY STATUS: the reference cannot be built here (Ravelin/Boost/LAPACK are
// absent) and stores no (M,q,z) vectors, so this file
// is pinned by: hand-derived KATs, the analytic 3-sphere-stack impact
// (tests/test_oracle_lcp.py), SciPy cross-checks of the LU, the glibc rand()
// stream checked against libc itself, and -- through the world stepper -- the
// reference's regress/sphere-stack.dat.  Semantics that live in Ravelin are
// restated from its published behaviour and are "parity unpinned":
//   * MatrixNd::norm_inf() = max |a_ij|                       (LCP.cpp:58)
//   * VectorNd::resize(n) keeps the leading contents when shrinking and
//     VectorNd::set_zero() keeps the size (LCP.cpp:564 relies on this)
//   * LinAlgd::solve_fast = dgesv (oracle/linalg.hpp)
//   * MatrixNd::mult(v) = reference-BLAS dgemv: y=0; for j: y += x_j * A(:,j)
//
// A "vector with a size" is modelled explicitly (Vec) because two reference
// behaviours depend on z.size(): lcp_fast warm-starts iff z.size()==n
// (LCP.cpp:65) and lcp_lemke draws n rand() values iff z.size()!=n after
// z.set_zero() (LCP.cpp:564-567,611-621).
#ifndef ORACLE_LCP_HPP
#define ORACLE_LCP_HPP
#include <vector>
#include <cmath>
#include <limits>
#include <algorithm>
#include <cstdint>
#include "glibc_rand.h"
#include "linalg.hpp"
#include "compact_lu.hpp"

namespace oracle {
// diagnostic (oracle_dbg_fast_repeats): how lcp_fast spends its iterations.  [0] iterations run, [1] of them with the index set of
// the iteration before (LCP.cpp:176-187 moved the entering variable straight out again: the loop spins on one basis), [2] with the set
// of one of the 2..8 iterations before that, [3] draws that decided something (rand_min with more than one minimum), [4] calls that ran
// into MAX_PIV.  Off (g_fast_diag = 0) it costs one test per iteration.
static int g_fast_diag = 0;
static unsigned long long g_fast_stats[13] = {0};   // [5 + h]: iterations on the set of h + 1 iterations before (h = 0..7)
static unsigned long long g_lu_hist[130] = {0};   // diagnostic: LU sizes (lcp_fast: [k], Lemke: [65 + n])


// Ravelin::VectorNd stand-in: contents survive a shrinking resize.
// diagnostic (oracle_dbg_compact_check): every basis lcp_lemke factorises is ALSO solved by the model of the device's
// structure-exploiting LU (compact_lu.hpp); [0] factorisations compared, [1] that differ in info or in any bit of the solution
// other than the sign of a zero, [2] fallbacks requested, [3] dense steps, [4] fill-ins, [5] panels, [6] truncated panels
// diagnostic (oracle_dbg_lemke_compact): lcp_lemke solves its bases with the structure-exploiting model ALONE (the dense routine only
// when the model asks for it) -- bit-equal by tests/test_oracle_compact_lu.py, and what makes n = 2048 bases affordable on the CPU
static int g_lemke_compact = 0;
static int g_compact_check = 0;
static unsigned long long g_compact_stats[8] = {0};
static int g_lemke_exit = 0;   // diagnostic: why the last failing lcp_lemke gave up (1000 + LAPACK info: singular basis; 2: ray; 3: empty ratio set)
struct Vec {
  std::vector<double> d; // capacity storage
  unsigned len = 0;
  unsigned size() const { return len; }
  double& operator[](unsigned i) { return d[i]; }
  const double& operator[](unsigned i) const { return d[i]; }
  void resize(unsigned n, bool preserve = false) {
    if (n > d.size()) {
      std::vector<double> nd(n, 0.0); // fresh storage: defined as zeros here
      if (preserve) std::copy(d.begin(), d.begin() + len, nd.begin());
      d.swap(nd);
    }
    len = n;
  }
  void set_zero() { std::fill(d.begin(), d.begin() + len, 0.0); }
  void set_zero(unsigned n) { resize(n); set_zero(); }
};

// pivot trace events (shared encoding with the HIP kernels, include/moby_hip.h)
//   lcp_fast : +(i+1)  index i moved basic -> nonbasic   (LCP.cpp:170-173)
//              -(i+1)  index i moved nonbasic -> basic   (LCP.cpp:144-149,182-187)
//   lcp_lemke: entering variable id, then leaving variable id, per pivot
//              (LCP.cpp:823-833, 977-988), offset by +1
//   0x40000000|k : start of solver attempt k (0 = unregularised, 1.. = ladder)
struct Trace {
  int32_t* buf = nullptr; int cap = 0; int len = 0;
  void push(int32_t v) { if (buf && len < cap) buf[len] = v; len++; }
};

class LCP {
 public:
  oracle_rand_t* rng = nullptr;   // the per-world libc stream
  unsigned pivots = 0;            // LCP.h:30
  Trace* trace = nullptr;

  static constexpr double EPS = std::numeric_limits<double>::epsilon();
  static double near_zero() { return std::sqrt(EPS); } // Constants.h:21

  static double norm_inf(int n, const double* M, int ld) {
    double nrm = 0.0;
    for (int c = 0; c < n; c++)
      for (int r = 0; r < n; r++) nrm = std::max(nrm, std::fabs(M[r + ld*c]));
    return nrm;
  }

  // include/Moby/insertion_sort:18-39 (result: ascending order)
  static void insertion_sort(std::vector<unsigned>& v) { std::sort(v.begin(), v.end()); }

  // LCP.cpp:199-209.  Always consumes exactly one rand().
  unsigned rand_min(const std::vector<double>& v, double zero_tol) {
    unsigned minv = std::min_element(v.begin(), v.end()) - v.begin();
    _minima.clear();
    _minima.push_back(minv);
    for (unsigned i = 0; i < v.size(); i++)
      if (i != minv && v[i] < v[minv] + zero_tol) _minima.push_back(i);
    if (g_fast_diag && _minima.size() > 1) g_fast_stats[3]++;
    return _minima[(unsigned)oracle_rand(rng) % _minima.size()];
  }

  // LCP.cpp:41-196.  `diag_add` is the regularisation term the caller has
  // added to the diagonal of its copy _MM (LCP.cpp:290-292); applying it on
  // access gives the same rounded values.
  bool lcp_fast(int n, const double* M, int ld, const double* q, Vec& z,
                double zero_tol, double diag_add = 0.0)
  {
    const unsigned N = n;
    const unsigned UINF = std::numeric_limits<unsigned>::max();
    auto Mat = [&](unsigned r, unsigned c) { return (r == c) ? M[r + ld*c] + diag_add : M[r + ld*c]; };
    g_lu_fma_now = (g_lu_fma && n > 64) ? 1 : 0;                   // (EXPERIMENT switch of linalg.hpp; 0 unless oracle_dbg_lu_fma was called)
    if (N == 0) { z.set_zero(0); return true; }
    if (zero_tol < 0.0) {
      double nrm = 0.0; // norm_inf of the (regularised) matrix
      for (unsigned c = 0; c < N; c++) for (unsigned r = 0; r < N; r++) nrm = std::max(nrm, std::fabs(Mat(r, c)));
      zero_tol = N * nrm * EPS;
    }
    _nonbas.clear(); _bas.clear();
    if (z.size() == N) {
      for (unsigned i = 0; i < N; i++)
        if (std::fabs(z[i]) < zero_tol) _bas.push_back(i); else _nonbas.push_back(i);
    } else {
      unsigned minw = std::min_element(q, q + N) - q;
      if (q[minw] > -zero_tol) { z.set_zero(N); return true; }
      _nonbas.push_back(minw);
      for (unsigned i = 0; i < N; i++) if (i != minw) _bas.push_back(i);
    }
    const unsigned MAX_PIV = 2*N;
    std::vector<std::vector<unsigned>> seen;                                   // diagnostic: the index sets of the last 8 iterations
    for (pivots = 0; pivots < MAX_PIV; pivots++) {
      if (g_fast_diag) {
        g_fast_stats[0]++;
        for (size_t h = 0; h < seen.size(); h++) if (seen[seen.size() - 1 - h] == _nonbas) { g_fast_stats[h == 0 ? 1 : 2]++; g_fast_stats[5 + h]++; break; }
        if (seen.size() == 8) seen.erase(seen.begin());
        seen.push_back(_nonbas);
        if (pivots + 1 == MAX_PIV) g_fast_stats[4]++;
      }
      const unsigned k = _nonbas.size(), nb = _bas.size();
      _Msub.assign((size_t)k*k, 0.0);
      for (unsigned c = 0; c < k; c++) for (unsigned r = 0; r < k; r++) _Msub[r + k*c] = Mat(_nonbas[r], _nonbas[c]);
      _z.resize(k);
      for (unsigned r = 0; r < k; r++) _z[r] = -q[_nonbas[r]];
      g_lu_hist[k < 65 ? k : 64]++;                                            // diagnostic
      if (k > 0 && lu_solve(k, _Msub.data(), k, _z.data()) != 0) return false; // SingularException
      // _Mmix.mult(_z,_w) += _qbas : dgemv then add
      _w.assign(nb, 0.0);
      for (unsigned c = 0; c < k; c++) {
        const double t = _z[c];
        for (unsigned r = 0; r < nb; r++) _w[r] = _w[r] + t * Mat(_bas[r], _nonbas[c]);
      }
      for (unsigned r = 0; r < nb; r++) _w[r] = _w[r] + q[_bas[r]];
      unsigned minw = (nb > 0) ? rand_min(_w, zero_tol) : UINF;
      if (minw == UINF || _w[minw] > -zero_tol) {
        unsigned minz = (k > 0) ? rand_min(_z, zero_tol) : UINF;
        if (minz < UINF && _z[minz] < -zero_tol) {
          unsigned idx = _nonbas[minz];
          _nonbas.erase(_nonbas.begin() + minz);
          _bas.push_back(idx); insertion_sort(_bas);
          if (trace) trace->push(-(int32_t)(idx + 1));
        } else {
          z.set_zero(N);
          for (unsigned j = 0; j < _nonbas.size(); j++) z[_nonbas[j]] = _z[j];
          return true;
        }
      } else {
        unsigned idx = _bas[minw];
        _bas.erase(_bas.begin() + minw);
        _nonbas.push_back(idx); insertion_sort(_nonbas);
        if (trace) trace->push((int32_t)(idx + 1));
        // NB: minz indexes the PRE-insertion _z / the POST-insertion _nonbas,
        // exactly as the reference does (LCP.cpp:176-187)
        unsigned minz = (k > 0) ? rand_min(_z, zero_tol) : UINF;
        if (minz < UINF && _z[minz] < -zero_tol) {
          unsigned idx2 = _nonbas[minz];
          _nonbas.erase(_nonbas.begin() + minz);
          _bas.push_back(idx2); insertion_sort(_bas);
          if (trace) trace->push(-(int32_t)(idx2 + 1));
        }
      }
    }
    return false;
  }

  // shared solution check of the two regularised wrappers
  // (LCP.cpp:240-249 strict=false, :303-312 strict=true)
  bool verify(int n, const double* M, int ld, double diag_add, const double* q, const Vec& z, double ZERO_TOL, bool strict) {
    auto ge = [&](double a, double b) { return strict ? (a > b) : (a >= b); };
    double minz = *std::min_element(z.d.begin(), z.d.begin() + n);
    if (!ge(minz, -ZERO_TOL)) return false;
    _wx.assign(n, 0.0);
    for (int c = 0; c < n; c++) {
      const double t = z[c];
      for (int r = 0; r < n; r++) {
        double m = M[r + ld*c]; if (r == c) m = m + diag_add;
        _wx[r] = _wx[r] + t * m;
      }
    }
    for (int r = 0; r < n; r++) _wx[r] = _wx[r] + q[r];
    if (!ge(*std::min_element(_wx.begin(), _wx.end()), -ZERO_TOL)) return false;
    for (int r = 0; r < n; r++) _wx[r] = z[r] * _wx[r];
    double mn = *std::min_element(_wx.begin(), _wx.end());
    double mx = *std::max_element(_wx.begin(), _wx.end());
    return ge(mn, -ZERO_TOL) && mx < ZERO_TOL;
  }

  // LCP.cpp:212-350
  bool lcp_fast_regularized(int n, const double* M, int ld, const double* q, Vec& z,
                            int min_exp, unsigned step_exp, int max_exp,
                            double piv_tol = -1.0, double zero_tol = -1.0)
  {
    (void)piv_tol;
    if (n == 0) { z.resize(0); return true; }
    const double ZERO_TOL = (zero_tol > 0.0) ? zero_tol : n * norm_inf(n, M, ld) * near_zero();
    unsigned total_piv = 0;
    if (trace) trace->push(0x40000000);
    bool result = lcp_fast(n, M, ld, q, z, zero_tol, 0.0);
    if (result && verify(n, M, ld, 0.0, q, z, ZERO_TOL, false)) return true;
    total_piv += pivots;
    int rf = min_exp, attempt = 1;
    while (rf < max_exp) {
      const double lambda = std::pow(10.0, (double)rf);
      if (trace) trace->push(0x40000000 | attempt);
      result = lcp_fast(n, M, ld, q, z, zero_tol, lambda);
      total_piv += pivots;
      if (result && verify(n, M, ld, lambda, q, z, ZERO_TOL, true)) { pivots = total_piv; return true; }
      rf += step_exp; attempt++;
    }
    pivots = total_piv;
    return false;
  }

  // LCP.cpp:353-487
  bool lcp_lemke_regularized(int n, const double* M, int ld, const double* q, Vec& z,
                             int min_exp = -20, unsigned step_exp = 1, int max_exp = 1,
                             double piv_tol = -1.0, double zero_tol = -1.0)
  {
    if (n == 0) { z.resize(0); return true; }
    const double ZERO_TOL = (zero_tol > 0.0) ? zero_tol : n * norm_inf(n, M, ld) * near_zero();
    unsigned total_piv = 0;
    if (trace) trace->push(0x40000000);
    bool result = lcp_lemke(n, M, ld, q, z, piv_tol, zero_tol, 0.0);
    if (result && verify(n, M, ld, 0.0, q, z, ZERO_TOL, false)) return true;
    total_piv += pivots;
    int rf = min_exp, attempt = 1;
    while (rf < max_exp) {
      const double lambda = std::pow(10.0, (double)rf);
      if (trace) trace->push(0x40000000 | attempt);
      result = lcp_lemke(n, M, ld, q, z, piv_tol, zero_tol, lambda);
      total_piv += pivots;
      if (result && verify(n, M, ld, lambda, q, z, ZERO_TOL, true)) { pivots = total_piv; return true; }
      rf += step_exp; attempt++;
    }
    pivots = total_piv;
    return false;
  }

  // LCP.cpp:545-1003 (dense).  The "restart" label is unreachable (its only
  // goto is commented out, LCP.cpp:851-867), so `restarted` is always false.
  bool lcp_lemke(int nn, const double* M, int ld, const double* q, Vec& z,
                 double piv_tol, double zero_tol, double diag_add = 0.0)
  {
    const unsigned n = nn;
    const unsigned MAXITER = std::min((unsigned)1000, 50*n);
    auto Mat = [&](unsigned r, unsigned c) { return (r == c) ? M[r + ld*c] + diag_add : M[r + ld*c]; };
    g_lu_fma_now = (g_lu_fma && nn > 64) ? 1 : 0;                  // (EXPERIMENT switch of linalg.hpp)
    pivots = 0;
    if (n == 0) { z.resize(0); return true; }
    z.set_zero();                 // :564 keeps z.size()
    const unsigned z0_size = z.size(); // _z0 = z  (:567): all zeros, size kept
    if (zero_tol <= 0.0) {
      double nrm = 0.0;
      for (unsigned c = 0; c < n; c++) for (unsigned r = 0; r < n; r++) nrm = std::max(nrm, std::fabs(Mat(r, c)));
      _norm = nrm;
      zero_tol = EPS * nrm * n;
    } else {
      double nrm = 0.0;
      for (unsigned c = 0; c < n; c++) for (unsigned r = 0; r < n; r++) nrm = std::max(nrm, std::fabs(Mat(r, c)));
      _norm = nrm;
    }
    if (*std::min_element(q, q + n) > -zero_tol) { z.set_zero(n); return true; }
    // restart: (:586)
    z.set_zero(n*2);
    const unsigned t = 2*n;
    unsigned entering = t, leaving = 0, lvindex;
    _bas.clear(); _nonbas.clear();
    if (z0_size != n) {
      for (unsigned i = 0; i < n; i++) _nonbas.push_back(i);
      for (unsigned i = 0; i < n; i++) (void)oracle_rand(rng); // _restart_z0 (:618-620), value unused
    } else {
      // _z0 is all zeros: nothing is > 0, every index is nonbasic (:625-629)
      for (unsigned i = 0; i < n; i++) _nonbas.push_back(i);
      // !restarted -> _restart_z0.set_zero(n), no rand()
    }
    // _bas is empty -> standard initial basis B = -I, x = q (:691-699)
    _Bl.assign((size_t)n*n, 0.0);
    for (unsigned i = 0; i < n; i++) _Bl[i + n*i] = -1.0;
    _x.assign(q, q + n);
    // initial basis solves it? (:737) -- cannot happen here since min q < 0, kept for fidelity
    bool anyneg = false;
    for (unsigned i = 0; i < n; i++) if (_x[i] < 0.0) { anyneg = true; break; }
    if (!anyneg) { z.resize(n, true); return true; }
    const double PIV_TOL = (piv_tol > 0.0) ? piv_tol : EPS * n * std::max(1.0, _norm);
    // initial leaving variable (:764-771)
    lvindex = std::min_element(_x.begin(), _x.begin() + n) - _x.begin();
    double tval = -_x[lvindex];
    for (unsigned i = 0; i < _nonbas.size(); i++) _bas.push_back(_nonbas[i] + n);
    leaving = _bas[lvindex];
    _bas[lvindex] = t;
    // pivot in the artificial variable (:776-785)
    _u.assign(n, 0.0);
    for (unsigned i = 0; i < n; i++) _u[i] = (_x[i] < 0.0) ? 1.0 : 0.0;
    _Be.assign(n, 0.0);
    for (unsigned c = 0; c < n; c++) { const double tt = _u[c]; for (unsigned r = 0; r < n; r++) _Be[r] = _Be[r] + tt * _Bl[r + n*c]; }
    for (unsigned r = 0; r < n; r++) _Be[r] = -_Be[r];
    for (unsigned i = 0; i < n; i++) _u[i] = _u[i] * tval;
    for (unsigned i = 0; i < n; i++) _x[i] = _x[i] + _u[i];
    _x[lvindex] = tval;
    for (unsigned r = 0; r < n; r++) _Bl[r + n*lvindex] = _Be[r];
    for (pivots = 0; pivots < MAXITER; pivots++) {
      if (leaving == t) {
        for (unsigned idx = 0; idx < _bas.size(); idx++) z[_bas[idx]] = _x[idx];
        z.resize(n, true);
        return true;
      } else if (leaving < n) {
        entering = n + leaving;
        _Be.assign(n, 0.0); _Be[leaving] = -1.0;
      } else {
        entering = leaving - n;
        for (unsigned r = 0; r < n; r++) _Be[r] = Mat(r, entering);
      }
      _dl = _Be;
      _Al = _Bl;
      g_lu_hist[65 + (n < 65 ? n : 64)]++;                                     // diagnostic
      std::vector<double> chk_b; std::vector<int> chk_kind, chk_idx;
      if (g_compact_check) {
        chk_b = _Be; chk_kind.resize(n); chk_idx.resize(n);
        for (unsigned p = 0; p < n; p++) { const unsigned id = _bas[p]; if (id >= n && id != t) { chk_kind[p] = CL_UNIT; chk_idx[p] = (int)(id - n); } else { chk_kind[p] = CL_DENSE; chk_idx[p] = (int)p; } }
      }
      { int info = CL_FALLBACK;
        if (g_lemke_compact) {
          std::vector<int> kd(n), ix(n);
          for (unsigned p = 0; p < n; p++) { const unsigned id = _bas[p]; if (id >= n && id != t) { kd[p] = CL_UNIT; ix[p] = (int)(id - n); } else { kd[p] = CL_DENSE; ix[p] = (int)p; } }
          if (g_lemke_compact & 0x100) {               // the model WITH reuse across pivots: the basis differs from the last one at lvindex only
            if (pivots == 0) _keep.valid = false;
            info = lu_solve_compact_keep(_keep, (int)n, kd.data(), ix.data(), _Bl.data(), (int)n, _dl.data(), g_lemke_compact & 0xff, (int)lvindex);
            g_compact_stats[7] += (unsigned long long)_keep.reused_steps * (_keep.valid ? 1 : 0);
          } else info = lu_solve_compact((int)n, kd.data(), ix.data(), _Bl.data(), (int)n, _dl.data(), g_lemke_compact);
          if (info == CL_FALLBACK) { _dl = _Be; _keep.valid = false; }
        }
        if (info == CL_FALLBACK) info = lu_solve(n, _Al.data(), n, _dl.data());
        if (g_compact_check) {
          CompactLuStats cs;
          const int ci = lu_solve_compact((int)n, chk_kind.data(), chk_idx.data(), _Bl.data(), (int)n, chk_b.data(), g_compact_check, nullptr, &cs);
          g_compact_stats[0]++;
          if (ci == CL_FALLBACK) g_compact_stats[2]++;
          else {
            bool same = (ci == info);
            if (same && info == 0) for (unsigned i = 0; i < n; i++) if (!(chk_b[i] == _dl[i])) same = false;
            if (!same) g_compact_stats[1]++;
            g_compact_stats[3] += cs.dense_steps; g_compact_stats[4] += cs.fill_ins; g_compact_stats[5] += cs.panels; g_compact_stats[6] += cs.truncated_panels;
          }
        }
        if (info != 0) { g_lemke_exit = 1000 + info; return false; } } // z keeps size 2n (:840-850)
      _j.clear();
      for (unsigned i = 0; i < n; i++) if (_dl[i] > PIV_TOL) _j.push_back(i);
      if (_j.empty()) { g_lemke_exit = 2; return false; }             // ray termination, size 2n (:892-903)
      // min ratio with zero_tol slack (:915-924)
      double theta = std::numeric_limits<double>::max();
      bool first = true;
      for (unsigned jj : _j) { double r = (_x[jj] + zero_tol) / _dl[jj]; if (first || r < theta) { theta = r; first = false; } }
      // keep those with x/d <= theta (:930-935)
      _jkeep.clear();
      for (unsigned jj : _j) if (_x[jj] / _dl[jj] <= theta) _jkeep.push_back(jj);
      _j.swap(_jkeep);
      if (_j.empty()) { g_lemke_exit = 3; z.resize(n, true); return false; }   // (:946-958)
      // artificial variable among candidates? (:961-975)
      bool has_t = false;
      for (unsigned jj : _j) if (_bas[jj] == t) has_t = true;
      if (has_t) lvindex = std::find(_bas.begin(), _bas.end(), t) - _bas.begin();
      else lvindex = _j[0];
      leaving = _bas[lvindex];
      // pivot (:983-988)
      const double ratio = _x[lvindex] / _dl[lvindex];
      for (unsigned i = 0; i < n; i++) _dl[i] = _dl[i] * ratio;
      for (unsigned i = 0; i < n; i++) _x[i] = _x[i] - _dl[i];
      _x[lvindex] = ratio;
      for (unsigned r = 0; r < n; r++) _Bl[r + n*lvindex] = _Be[r];
      _bas[lvindex] = entering;
      if (trace) { trace->push((int32_t)entering + 1); trace->push((int32_t)leaving + 1); }
    }
    z.resize(n, true);
    return false; // MAXITER (:992-1002)
  }

 private:
  std::vector<unsigned> _bas, _nonbas, _minima, _j, _jkeep;
  CompactLuKeep _keep;                                  // (g_lemke_compact & 0x100: the reuse model of compact_lu.hpp)
  std::vector<double> _Msub, _z, _w, _wx, _Bl, _Al, _x, _u, _Be, _dl;
  double _norm = 0.0;
};

} // namespace oracle
#endif
