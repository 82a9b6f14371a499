// The large-world adapter on a stack of 3 boxes (BASELINE config 4 in the small): steps, then stabilize() alone on an
// interpenetrating copy.
//   g++ -std=c++11 example_stack.cpp -L.. -lmoby_hip -Wl,-rpath,.. -o example_stack
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "MobyHipStackSimulator.h"

int main()
{
  const int N = 3, B = 2;
  std::vector<int> gt(N, MH_GEOM_BOX), pa, pb, pm;
  std::vector<double> dim, mass, inertia;
  for (int k = 0; k < N; k++) {
    const double s = 1.0 - 0.005 * k, m = 10.0 * s * s;
    dim.push_back(s); dim.push_back(1.0); dim.push_back(s); mass.push_back(m);
    inertia.push_back(m / 12.0 * (1.0 + s * s)); inertia.push_back(m / 12.0 * (2.0 * s * s)); inertia.push_back(m / 12.0 * (1.0 + s * s));
  }
  for (int k = 0; k < N; k++) { if (k + 1 < N) { pa.push_back(k); pb.push_back(k + 1); pm.push_back(MH_PAIR_VERTEX_FACE); } pa.push_back(k); pb.push_back(N); pm.push_back(MH_PAIR_CLOSED_FORM); }
  const int np = (int)pa.size();
  std::vector<double> eps(np, 0.0), mu(np, 1e-4), muv(np, 0.0), comp(np, 0.0);
  mh_big_scene sc; std::memset(&sc, 0, sizeof(sc));
  sc.nb = N; sc.has_ground = 1; sc.geom_type = gt.data(); sc.geom_dim = dim.data(); sc.mass = mass.data(); sc.inertia = inertia.data();
  sc.plane_R[0] = sc.plane_R[4] = sc.plane_R[8] = 1.0; sc.gravity[1] = -9.81;
  sc.npairs = np; sc.pair_a = pa.data(); sc.pair_b = pb.data(); sc.pair_model = pm.data();
  sc.cp_epsilon = eps.data(); sc.cp_mu_coulomb = mu.data(); sc.cp_mu_viscous = muv.data(); sc.cp_compliance = comp.data(); sc.nk = 4;
  sc.min_step_size = std::sqrt(2.220446049250313e-16); sc.contact_dist_thresh = 1e-6; sc.cstab_eps = sc.min_step_size;
  sc.cstab_max_iterations = 50; sc.lcp_n_max = 32 * N;
  std::vector<double> st((size_t)B * N * MH_BODY_STATE, 0.0);
  for (int w = 0; w < B; w++) for (int k = 0; k < N; k++) { double* s = &st[((size_t)w * N + k) * MH_BODY_STATE]; s[1] = 0.5 + k - (w == 1 ? 1e-4 * (k + 1) : 0.0); s[6] = 1.0; }
  try {
    MobyHip::BatchedStackSimulator sim(sc, B, st.data());
    sim.stabilize();                                             // world 1 starts interpenetrating by 1e-4 per interface
    double q[7]; sim.get_generalized_coordinates_euler(1, N - 1, q);
    const double top_after_stab = q[1];
    for (int s = 0; s < 3; s++) sim.step(1e-3);
    sim.get_generalized_coordinates_euler(0, N - 1, q);
    std::printf("time=%.3f status=%d/%d top(world 0)=%.9f top(world 1 after stabilize)=%.9f rows=%llu\n", sim.current_time, sim.status(0), sim.status(1),
                q[1], top_after_stab, (unsigned long long)sim.counters(0).lcp_rows);
    return (std::fabs(q[1] - (0.5 + N - 1)) < 1e-5 && top_after_stab > 0.5 + N - 1 - 1e-7) ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
