// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Dense linear algebra with the semantics of the LAPACK routines that
// Ravelin::LinAlgd forwards to.  Ravelin and LAPACK are NOT in
// /root/reference, so this restates the *published reference-LAPACK 3.x
// algorithms* (unblocked variants), which also fixes the floating-point
// operation order the HIP kernels must reproduce:
//
//   solve_fast(A,b)   = dgesv  -> dgetf2 (right-looking LU, partial pivoting,
//                       pivot = FIRST max |a| (idamax), column scaled by the
//                       RECIPROCAL of the pivot) + dgetrs (dlaswp, unit-lower
//                       dtrsm, upper dtrsm with a division per row).
//                       call sites: src/LCP.cpp:120,670,838
//   factor_chol(A)    = dpotf2 upper? -- see chol_factor below
//                       call sites: src/ImpactConstraintHandler.cpp:366,1116,...
//   inverse_SPD(A)    = dpotrf + dpotri  (ICH:1607,1667)
//
// All matrices are column-major with an explicit leading dimension.
// Compile with -ffp-contract=off: no FMA contraction anywhere.
#ifndef ORACLE_LINALG_HPP
#define ORACLE_LINALG_HPP
#include <cmath>
#include <cfloat>
#include <vector>

namespace oracle {

// dgesv semantics for one right-hand side.  A (n x n, ld) is overwritten by
// its LU factors, b by the solution.  Returns LAPACK `info`: 0 = ok,
// j>0 = U(j,j) is exactly zero (Ravelin throws SingularException; the
// solution is then NOT computed, b is left untouched).
static unsigned long long g_lu_nan_pivots = 0;
// EXPERIMENT (oracle_dbg_lu_fma; DESIGN 8, "the oracle's dgesv above n = 64"): every multiply-subtract of dgesv -- dgetf2's rank-1 updates, both triangular
// solves -- as ONE fused operation while an LCP of more than 64 rows is being solved (g_lu_fma_now, set by lcp.hpp).  Off by default: the oracle's definition,
// and the device's, is the unfused one at every size.  No .dat pin can see the switch (every recorded scene has n <= 48).
static int g_lu_fma = 0, g_lu_fma_now = 0;
inline double lu_upd(double a, double l, double u) { return g_lu_fma_now ? std::fma(-l, u, a) : a - l * u; }   // diagnostic: columns whose diagonal entry was NaN at its pivot search
inline int lu_solve(int n, double* A, int ld, double* b, int* ipiv_out = nullptr)
{
  std::vector<int> ipiv_local;
  int* ipiv = ipiv_out;
  if (!ipiv) { ipiv_local.resize(n > 0 ? n : 1); ipiv = ipiv_local.data(); }
  int info = 0;
  const double sfmin = DBL_MIN; // dlamch('S')
  for (int j = 0; j < n; j++) {
    // idamax: first index of max |A(i,j)|, i>=j
    int jp = j; double amax = std::fabs(A[j + ld*j]);
    if (amax != amax) g_lu_nan_pivots++;     // idamax keeps index j then: no |a| compares greater than NaN
    for (int i = j+1; i < n; i++) {
      double v = std::fabs(A[i + ld*j]);
      if (v > amax) { amax = v; jp = i; }
    }
    ipiv[j] = jp;
    if (A[jp + ld*j] != 0.0) {
      if (jp != j)
        for (int c = 0; c < n; c++) { double t = A[j + ld*c]; A[j + ld*c] = A[jp + ld*c]; A[jp + ld*c] = t; }
      if (j < n-1) {
        if (std::fabs(A[j + ld*j]) >= sfmin) {
          const double r = 1.0 / A[j + ld*j];
          for (int i = j+1; i < n; i++) A[i + ld*j] *= r;
        } else {
          for (int i = j+1; i < n; i++) A[i + ld*j] /= A[j + ld*j];
        }
      }
    } else if (info == 0) {
      info = j + 1;
    }
    // dger: trailing update A22 -= l * u'
    for (int c = j+1; c < n; c++) {
      const double u = A[j + ld*c];
      for (int i = j+1; i < n; i++)
        A[i + ld*c] = lu_upd(A[i + ld*c], A[i + ld*j], u);
    }
  }
  if (info != 0) return info;
  // dlaswp
  for (int j = 0; j < n; j++)
    if (ipiv[j] != j) { double t = b[j]; b[j] = b[ipiv[j]]; b[ipiv[j]] = t; }
  // L y = b (unit lower), column oriented
  for (int k = 0; k < n; k++) {
    const double bk = b[k];
    for (int i = k+1; i < n; i++) b[i] = lu_upd(b[i], bk, A[i + ld*k]);
  }
  // U x = y
  for (int k = n-1; k >= 0; k--) {
    b[k] = b[k] / A[k + ld*k];
    const double bk = b[k];
    for (int i = 0; i < k; i++) b[i] = lu_upd(b[i], bk, A[i + ld*k]);
  }
  return 0;
}

// dpotf2('L'): lower Cholesky, unblocked, left-looking as reference LAPACK.
// Returns false (factor_chol's "not PD") when a diagonal term is <= 0 or NaN.
// Only the lower triangle is referenced/written.
inline bool chol_factor(int n, double* A, int ld)
{
  for (int j = 0; j < n; j++) {
    double ajj = A[j + ld*j];
    for (int k = 0; k < j; k++) ajj = ajj - A[j + ld*k] * A[j + ld*k];
    if (!(ajj > 0.0)) return false;
    ajj = std::sqrt(ajj);
    A[j + ld*j] = ajj;
    for (int i = j+1; i < n; i++) {
      double s = A[i + ld*j];
      for (int k = 0; k < j; k++) s = s - A[i + ld*k] * A[j + ld*k];
      A[i + ld*j] = s / ajj;
    }
  }
  return true;
}

// dpotrs with the lower factor: solves A x = b in place.
inline void chol_solve(int n, const double* L, int ld, double* b)
{
  for (int k = 0; k < n; k++) {
    b[k] = b[k] / L[k + ld*k];
    const double bk = b[k];
    for (int i = k+1; i < n; i++) b[i] = b[i] - bk * L[i + ld*k];
  }
  for (int k = n-1; k >= 0; k--) {
    double s = b[k];
    for (int i = k+1; i < n; i++) s = s - L[i + ld*k] * b[i];
    b[k] = s / L[k + ld*k];
  }
}

// inverse_SPD: A <- A^{-1} (full symmetric storage).  Column-by-column solve
// against the Cholesky factor; returns false if A is not PD.
inline bool inverse_spd(int n, double* A, int ld)
{
  std::vector<double> L(A, A + (size_t)ld*n);
  if (!chol_factor(n, L.data(), ld)) return false;
  std::vector<double> e(n);
  for (int c = 0; c < n; c++) {
    for (int i = 0; i < n; i++) e[i] = (i == c) ? 1.0 : 0.0;
    chol_solve(n, L.data(), ld, e.data());
    for (int i = 0; i < n; i++) A[i + ld*c] = e[i];
  }
  // symmetrise exactly (dpotri returns one triangle; Ravelin mirrors it)
  for (int c = 0; c < n; c++)
    for (int i = c+1; i < n; i++) A[c + ld*i] = A[i + ld*c];
  return true;
}

} // namespace oracle
#endif
