// Implicit joints through the large-world adapter: a two-link pendulum (revolute joint to the world, universal joint between
// the links) next to a box kept on a plane by a planar joint, 200 steps with the stabiliser on.
//   g++ -std=c++11 example_joints.cpp -L.. -lmoby_hip -Wl,-rpath,.. -o example_joints
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "MobyHipStackSimulator.h"

int main()
{
  const int N = 3, B = 2;
  std::vector<int> gt; gt.push_back(MH_GEOM_SPHERE); gt.push_back(MH_GEOM_SPHERE); gt.push_back(MH_GEOM_BOX);
  const double dim[9] = { 0.2, 0, 0,  0.2, 0, 0,  1, 1, 1 };
  const double mass[3] = { 1.0, 1.0, 1.0 };
  const double inertia[9] = { 0.016, 0.016, 0.016,  0.016, 0.016, 0.016,  1.0 / 6, 1.0 / 6, 1.0 / 6 };
  // reference poses: link 0 at (0.5, 2, 0), link 1 at (1.5, 2, 0), the box at (5, 0.5, 0)
  std::vector<double> st((size_t)B * N * MH_BODY_STATE, 0.0);
  for (int w = 0; w < B; w++) {
    double* s = &st[(size_t)w * N * MH_BODY_STATE];
    s[0] = 0.5; s[1] = 2.0; s[6] = 1.0;
    s[13] = 1.5; s[14] = 2.0; s[19] = 1.0;
    s[26] = 5.0; s[27] = 0.5; s[32] = 1.0; s[26 + 10] = 3.0 * (w + 1); s[26 + 11] = 1.0;     // the box: spin about x (forbidden) and y
  }
  MobyHip::ImplicitJoints joints;
  const double z[3] = { 0, 0, 1 }, y[3] = { 0, 1, 0 }, x[3] = { 1, 0, 0 };
  const double p0[3] = { 0, 2, 0 }, p1[3] = { 1, 2, 0 }, p2[3] = { 5, 0, 0 };
  joints.add(MH_IJOINT_REVOLUTE, N, NULL, 0, &st[0], p0, z);
  joints.add(MH_IJOINT_UNIVERSAL, 0, &st[0], 1, &st[13], p1, z, x);
  joints.add(MH_IJOINT_PLANAR, N, NULL, 2, &st[26], p2, y);
  mh_big_scene sc; std::memset(&sc, 0, sizeof(sc));
  sc.nb = N; sc.has_ground = 1; sc.geom_type = gt.data(); sc.geom_dim = dim; sc.mass = mass; sc.inertia = inertia;
  sc.plane_R[0] = sc.plane_R[4] = sc.plane_R[8] = 1.0; sc.gravity[1] = -9.81;
  sc.npairs = 0; sc.nk = 4;
  sc.min_step_size = std::sqrt(2.220446049250313e-16); sc.contact_dist_thresh = 1e-6; sc.cstab_eps = sc.min_step_size;
  sc.cstab_max_iterations = 20; sc.lcp_n_max = 64;
  joints.attach(sc);
  try {
    MobyHip::BatchedStackSimulator sim(sc, B, st.data());
    for (int s = 0; s < 200; s++) sim.step(1e-3);
    double q0[7], q1[7], qb[7], vb[6];
    sim.get_generalized_coordinates_euler(0, 0, q0); sim.get_generalized_coordinates_euler(0, 1, q1);
    sim.get_generalized_coordinates_euler(1, 2, qb); sim.get_generalized_velocity(1, 2, vb);
    const double l0 = std::sqrt(q0[0] * q0[0] + (q0[1] - 2.0) * (q0[1] - 2.0) + q0[2] * q0[2]);          // link 0 stays 0.5 from the hinge
    std::printf("time=%.3f status=%d/%d link0 r=%.9f y=%.6f link1 y=%.6f box y=%.9f wx=%.2e wy=%.6f stab=%llu\n", sim.current_time, sim.status(0), sim.status(1),
                l0, q0[1], q1[1], qb[1], vb[3], vb[4], (unsigned long long)sim.counters(0).stab_iters);
    const bool ok = sim.status(0) == 0 && sim.status(1) == 0 && std::fabs(l0 - 0.5) < 1e-5 && q0[1] < 1.99 && q1[1] < 1.9 &&
                    std::fabs(qb[1] - 0.5) < 1e-6 && std::fabs(vb[3]) < 1e-5 && std::fabs(vb[4] - 1.0) < 1e-2;
    return ok ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
