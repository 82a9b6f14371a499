// MobyHipLCP.h -- reference-side adapter: a class with the interface of Moby::LCP
// (/root/reference/include/Moby/LCP.h:17-58) that forwards to libmoby_hip.so.
//
// A Moby maintainer replaces `#include <Moby/LCP.h>` by this header (or aliases
// `namespace Moby { typedef MobyHip::LCP<Ravelin::MatrixNd, Ravelin::VectorNd> LCP; }`)
// and links -lmoby_hip; ImpactConstraintHandler (`_lcp.lcp_fast_regularized(_MM,_qq,z,-20,4,-8)`,
// src/ImpactConstraintHandlerQP.cpp:219) and ConstraintStabilization (`_lcp.lcp_fast(MM,qq,z)`,
// src/ConstraintStabilization.cpp:954) compile unchanged.
//
// The class is a template over the matrix / vector types so that it builds without
// Ravelin: any types with Ravelin's accessors work --
//   Matrix: rows(), columns(), leading_dim(), data() (column-major doubles)
//   Vector: size(), data(), resize(n) (contents kept when shrinking), set_zero(n)
// Conventions kept from the reference: returns true/false, never throws for a solver
// failure (only for a broken library / device), `z` is warm-start in and solution out,
// `pivots` counts pivots of the last call.  rand(): the reference draws from the PROCESS-global libc stream
// (src/LCP.cpp:208,620), so ImpactConstraintHandler::_lcp and ConstraintStabilization::_lcp interleave their
// draws.  By default every MobyHip::LCP object therefore shares ONE stream per process (process_rng(), srand(1)
// state on first use), which is what a Moby build with two adapters needs to pivot like the fused kernel
// (one stream per world).  A batched caller that steps several worlds from one process gives each world's
// solver objects their own stream with use_rng(state).
#ifndef MOBY_HIP_LCP_ADAPTER_H
#define MOBY_HIP_LCP_ADAPTER_H
#include <stdexcept>
#include <string>
#include <vector>
#include <stdint.h>
#include "../../include/moby_hip.h"

namespace MobyHip {

template <class Matrix, class Vector>
class LCP {
 public:
  LCP() : pivots(0), _rng(process_rng()) {}

  // the stream every adapter object of this process shares unless told otherwise (libc rand() after srand(1))
  static uint32_t* process_rng() {
    static uint32_t* st = 0;
    static uint32_t storage[MH_RAND_WORDS];
    if (!st) { mh_rand_seed(storage, 1); st = storage; }
    return st;
  }
  // draw from a caller-owned stream instead (MH_RAND_WORDS words, mh_rand_seed); one per simulated world
  void use_rng(uint32_t* state) { _rng = state ? state : process_rng(); }

  bool lcp_fast(const Matrix& M, const Vector& q, Vector& z, double zero_tol = -1.0) {
    return solve(MH_LCP_FAST, M, q, z, -20, 1, 1, -1.0, zero_tol);
  }
  bool lcp_fast_regularized(const Matrix& M, const Vector& q, Vector& z, int min_exp = -20, unsigned step_exp = 4,
                            int max_exp = 20, double piv_tol = -1.0, double zero_tol = -1.0) {
    return solve(MH_LCP_FAST_REG, M, q, z, min_exp, step_exp, max_exp, piv_tol, zero_tol);
  }
  bool lcp_lemke(const Matrix& M, const Vector& q, Vector& z, double piv_tol = -1.0, double zero_tol = -1.0) {
    return solve(MH_LCP_LEMKE, M, q, z, -20, 1, 1, piv_tol, zero_tol);
  }
  bool lcp_lemke_regularized(const Matrix& M, const Vector& q, Vector& z, int min_exp = -20, unsigned step_exp = 1,
                             int max_exp = 1, double piv_tol = -1.0, double zero_tol = -1.0) {
    return solve(MH_LCP_LEMKE_REG, M, q, z, min_exp, step_exp, max_exp, piv_tol, zero_tol);
  }

  unsigned pivots;   // LCP.h:30 (private there; exposed for diagnostics)

 private:
  uint32_t* _rng;
  std::vector<double> _zbuf;

  bool solve(int kind, const Matrix& M, const Vector& q, Vector& z, int min_exp, unsigned step_exp, int max_exp,
             double piv_tol, double zero_tol)
  {
    const int n = (int)q.size();
    if (n == 0) { z.resize(0); return true; }                 // LCP.cpp:49-54,218-222,557-561
    if ((int)M.rows() != n || (int)M.columns() != n) throw std::invalid_argument("MobyHip::LCP: M must be n x n");
    // the library works on an n-vector plus z.size(): lcp_fast warm-starts iff z.size()==n (LCP.cpp:65),
    // lcp_lemke draws n rand() values iff it differs (LCP.cpp:611-621)
    int zsize_in = (int)z.size(), zsize_out = 0, status = 0;
    _zbuf.assign((size_t)n, 0.0);
    if (zsize_in == n) for (int i = 0; i < n; i++) _zbuf[i] = z.data()[i];
    mh_lcp_opts o; o.min_exp = min_exp; o.step_exp = step_exp; o.max_exp = max_exp; o.piv_tol = piv_tol; o.zero_tol = zero_tol;
    unsigned piv = 0;
    const int rc = mh_lcp_solve_batch(kind, 1, n, M.data(), (int)M.leading_dim(), (long)M.leading_dim() * n,
                                      q.data(), _zbuf.data(), &zsize_in, &zsize_out, _rng, &status, &piv,
                                      (int32_t*)0, 0, (int*)0, &o);
    if (rc != MH_OK) throw std::runtime_error(std::string("libmoby_hip: ") + mh_last_error());
    pivots = piv;
    if (status) {                                             // success: z has size n
      z.resize((unsigned)n);
      for (int i = 0; i < n; i++) z.data()[i] = _zbuf[i];
    } else if (zsize_out != zsize_in) {
      // failure paths leave z zeroed with size n or 2n (LCP.cpp:596,840-903,946-958)
      z.set_zero((unsigned)zsize_out);
    }
    return status != 0;
  }
};

} // namespace MobyHip
#endif
