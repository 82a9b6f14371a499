// The articulated-body adapter on an SDF model: forward dynamics of B copies, then 100 steps one by one against one
// call of step(dt, 100).
//   g++ -std=c++11 example_articulated.cpp -L.. -lmoby_hip -lmoby_hip_io -Wl,-rpath,.. -o example_articulated
//   ./example_articulated ../../tests/scenes/ten_joint_arm.sdf
#include <cstdio>
#include <cstring>
#include "MobyHipArticulatedBody.h"
#include "../../include/moby_hip_io.h"

int main(int argc, char** argv)
{
  if (argc < 2) { std::printf("usage: example_articulated <model.sdf>\n"); return 2; }
  const double g[3] = { 0.0, 0.0, -9.81 };
  mh_io_artic io;
  if (mh_io_load_sdf(argv[1], g, &io) != 0) { std::printf("error: %s\n", mh_io_last_error()); return 1; }
  try {
    const int B = 4, nj = io.model.nj;
    std::vector<double> q((size_t)B * nj, 0.0), qd((size_t)B * nj, 0.0);
    for (int w = 0; w < B; w++) { q[(size_t)w * nj + 2] = -0.3 * (w + 1); qd[(size_t)w * nj + 1] = 0.5; }   // shoulder_lift bent, shoulder_pan turning
    MobyHip::BatchedArticulatedBody a(io.model, B, q.data(), qd.data()), b(io.model, B, q.data(), qd.data());
    std::vector<double> qdd((size_t)B * nj), H((size_t)B * nj * nj);
    a.calc_fwd_dyn(NULL, qdd.data());
    a.get_generalized_inertia(H.data());
    for (int s = 0; s < 100; s++) a.step(5e-4);
    b.step(5e-4, 100);
    const bool same = std::memcmp(a.q().data(), b.q().data(), a.q().size() * sizeof(double)) == 0 &&
                      std::memcmp(a.qd().data(), b.qd().data(), a.qd().size() * sizeof(double)) == 0;
    std::printf("joints=%d %s..%s same=%d status=%d H00=%.9g qdd[2]=%.9g q[2]=%.9g\n", nj, io.joint_id[0], io.joint_id[nj - 1], (int)same, a.status(0),
                H[0], qdd[2], a.q()[2]);
    return same ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
