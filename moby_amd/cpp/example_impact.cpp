// The batched impact handler adapter on a stack of boxes (4 corner contacts per interface):
//   g++ -std=c++11 example_impact.cpp -L.. -lmoby_hip -Wl,-rpath,.. -o example_impact && ./example_impact 3
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "MobyHipImpactHandler.h"

int main(int argc, char** argv)
{
  const int nbx = (argc > 1) ? std::atoi(argv[1]) : 3, B = 2, nc = 4 * nbx;
  std::vector<double> mass(nbx), J(3 * nbx), state((size_t)B * nbx * MH_BODY_STATE, 0.0);
  std::vector<mh_contact> cs((size_t)B * nc);
  std::memset(cs.data(), 0, cs.size() * sizeof(mh_contact));
  double weight = 0.0;
  for (int k = 0; k < nbx; k++) {
    const double x = 1.0 - 0.005 * k, m = 10.0 * x * x;                  // box k: x by 1 by x, density 10
    mass[k] = m; J[3*k] = m / 12.0 * (1.0 + x * x); J[3*k+1] = m / 12.0 * (2.0 * x * x); J[3*k+2] = J[3*k];
    weight += m;
    for (int w = 0; w < B; w++) {
      double* s = &state[((size_t)w * nbx + k) * MH_BODY_STATE];
      s[1] = 0.5 + k; s[6] = 1.0; s[8] = -9.81e-3;                       // one free-fall step of 1 ms
      for (int c = 0; c < 4; c++) {
        mh_contact& ct = cs[(size_t)w * nc + 4 * k + c];
        ct.point[0] = ((c & 2) ? -0.5 : 0.5) * x; ct.point[1] = k; ct.point[2] = ((c & 1) ? -0.5 : 0.5) * x;
        ct.normal[1] = 1.0; ct.body1 = k; ct.body2 = k ? k - 1 : -1; ct.nk = 4;
      }
    }
  }
  try {
    MobyHip::BatchedImpactHandler h(B, nbx, nc, 4, mass.data(), J.data());
    h.process_constraints(state.data(), cs.data());
    double vmax = 0.0, ground = 0.0;
    for (int k = 0; k < nbx; k++) for (int q = 7; q < 13; q++) vmax = std::fmax(vmax, std::fabs(state[(size_t)k * MH_BODY_STATE + q]));
    for (int c = 0; c < 4; c++) { double f[3]; h.contact_impulse(0, c, f); ground += f[0]; }
    const bool same = std::memcmp(&state[0], &state[(size_t)nbx * MH_BODY_STATE], (size_t)nbx * MH_BODY_STATE * sizeof(double)) == 0;
    std::printf("n=%d status=%d solves=%d pivots=%u vmax=%.3g ground_impulse/weight_dt=%.12f same=%d\n", h.lcp_size(), h.status(0), h.solves(0),
                h.pivots(0), vmax, ground / (weight * 9.81e-3), (int)same);
    return (h.status(0) == 0 && same) ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
