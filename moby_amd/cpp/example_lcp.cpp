// Builds without Ravelin: minimal stand-ins with Ravelin's accessors, then the
// 3-sphere-stack normal LCP (SURVEY 8c KAT) through the Moby::LCP-shaped adapter.
//   g++ -std=c++11 example_lcp.cpp -L.. -lmoby_hip -Wl,-rpath,.. -o example_lcp
#include <cstdio>
#include <vector>
#include "MobyHipLCP.h"

struct Mat {
  unsigned r, c; std::vector<double> d;
  Mat(unsigned r_, unsigned c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
  unsigned rows() const { return r; } unsigned columns() const { return c; } unsigned leading_dim() const { return r; }
  const double* data() const { return d.data(); }
  double& operator()(unsigned i, unsigned j) { return d[i + (size_t)r * j]; }
};
struct Vec {
  std::vector<double> d; unsigned n;
  Vec() : n(0) {}
  unsigned size() const { return n; }
  double* data() { return d.data(); } const double* data() const { return d.data(); }
  void resize(unsigned m) { if (m > d.size()) d.assign(m, 0.0); n = m; }
  void set_zero(unsigned m) { resize(m); for (unsigned i = 0; i < m; i++) d[i] = 0.0; }
};

int main()
{
  const double g = 9.81, dt = 1e-3;
  Mat M(3, 3);
  const double m[3][3] = { {1, -1, 0}, {-1, 2, -1}, {0, -1, 2} };
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = m[i][j];
  Vec q; q.resize(3); q.data()[0] = -g * dt; q.data()[1] = 0; q.data()[2] = 0;
  Vec z;
  MobyHip::LCP<Mat, Vec> lcp;
  try {
    const bool ok = lcp.lcp_fast(M, q, z);
    std::printf("ok=%d pivots=%u z= %.17g %.17g %.17g\n", (int)ok, lcp.pivots, z.data()[0], z.data()[1], z.data()[2]);
    Vec z2;
    const bool ok2 = lcp.lcp_lemke_regularized(M, q, z2);
    std::printf("lemke ok=%d z= %.17g %.17g %.17g\n", (int)ok2, z2.data()[0], z2.data()[1], z2.data()[2]);
  } catch (const std::exception& e) {
    std::printf("error: %s\n", e.what());
    return 2;
  }
  return 0;
}
