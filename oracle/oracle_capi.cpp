// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// extern "C" surface of the CPU oracle so tests/ and bench.py's cpu_baseline
// leg can drive it through ctypes.  Nothing under moby_amd/ may link this.
#include <cstring>
#include <ctime>
#include "lcp.hpp"

using namespace oracle;

extern "C" {

void oracle_srand_state(uint32_t* state32, uint32_t seed)
{
  oracle_rand_t s; oracle_srand(&s, seed);
  std::memcpy(state32, &s, sizeof(s));
}

int oracle_rand_next(uint32_t* state32)
{
  oracle_rand_t s; std::memcpy(&s, state32, sizeof(s));
  int v = oracle_rand(&s);
  std::memcpy(state32, &s, sizeof(s));
  return v;
}

// dgesv-semantics solve (oracle/linalg.hpp); returns LAPACK info
int oracle_lu_solve(int n, double* A, int ld, double* b) { return lu_solve(n, A, ld, b); }

// kind: 0 lcp_fast, 1 lcp_fast_regularized, 2 lcp_lemke, 3 lcp_lemke_regularized
// z: buffer of at least max(2n, *z_size) doubles; *z_size is z.size() on entry
// (== n requests lcp_fast's warm start) and on exit.
// rng: 32 uint32 words (oracle_rand_t), in/out.
// returns 1 = solver returned true, 0 = false
int oracle_lcp_solve(int kind, int n, const double* M, int ld, const double* q,
                     double* z, int* z_size,
                     int min_exp, unsigned step_exp, int max_exp,
                     double piv_tol, double zero_tol,
                     uint32_t* rng, unsigned* pivots,
                     int32_t* trace, int trace_cap, int* trace_len)
{
  oracle_rand_t rs; std::memcpy(&rs, rng, sizeof(rs));
  LCP lcp; lcp.rng = &rs;
  Trace tr; tr.buf = trace; tr.cap = trace_cap;
  lcp.trace = &tr;
  Vec zz;
  const unsigned cap = std::max<unsigned>(2u * (unsigned)n, (unsigned)*z_size);
  zz.d.assign(z, z + cap); zz.len = (unsigned)*z_size;
  bool ok = false;
  switch (kind) {
    case 0: ok = lcp.lcp_fast(n, M, ld, q, zz, zero_tol); break;
    case 1: ok = lcp.lcp_fast_regularized(n, M, ld, q, zz, min_exp, step_exp, max_exp, piv_tol, zero_tol); break;
    case 2: ok = lcp.lcp_lemke(n, M, ld, q, zz, piv_tol, zero_tol); break;
    case 3: ok = lcp.lcp_lemke_regularized(n, M, ld, q, zz, min_exp, step_exp, max_exp, piv_tol, zero_tol); break;
    default: return -1;
  }
  const unsigned ncopy = std::min<unsigned>(cap, (unsigned)zz.d.size());
  std::memcpy(z, zz.d.data(), sizeof(double) * ncopy);
  *z_size = (int)zz.len;
  if (pivots) *pivots = lcp.pivots;
  if (trace_len) *trace_len = tr.len;
  std::memcpy(rng, &rs, sizeof(rs));
  return ok ? 1 : 0;
}

// Batch driver used for the CPU baseline: solves B independent problems
// (problem b at M + b*strideM, q + b*n, z + b*2n, rng + b*32) sequentially on
// one thread and returns elapsed CPU-wall seconds.
double oracle_lcp_solve_batch(int kind, int B, int n, const double* M, int ld, long strideM,
                              const double* q, double* z, int* z_size,
                              int min_exp, unsigned step_exp, int max_exp,
                              double piv_tol, double zero_tol, uint32_t* rng,
                              int* status, unsigned* pivots)
{
  timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int b = 0; b < B; b++) {
    int zs = z_size ? z_size[b] : n;
    unsigned piv = 0;
    int st = oracle_lcp_solve(kind, n, M + (size_t)b*strideM, ld, q + (size_t)b*n, z + (size_t)b*2*n, &zs,
                              min_exp, step_exp, max_exp, piv_tol, zero_tol, rng + (size_t)b*32, &piv,
                              nullptr, 0, nullptr);
    if (z_size) z_size[b] = zs;
    if (status) status[b] = st;
    if (pivots) pivots[b] = piv;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

} // extern "C"
