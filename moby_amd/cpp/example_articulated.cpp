// The articulated-body adapter on an SDF model: forward dynamics of B copies, then 100 steps one by one against one
// call of step(dt, 100); then the same arm with a sphere on each of its last two links over a table (contacts on links).
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
    // collision geometry: spheres on the last two links, a table 5 cm below the lower one at the start pose
    mh_artic_model mc = io.model;
    const double origin[3] = { 0.0, 0.0, 0.0 };
    MobyHip::add_link_sphere(mc, nj - 2, origin, 0.03);
    MobyHip::add_link_sphere(mc, nj - 1, origin, 0.03);
    std::vector<double> poses((size_t)B * nj * 12);
    {
      mh_artic_batch* tmp = NULL;
      if (mh_artic_batch_create(&io.model, B, &tmp) != MH_OK || mh_artic_batch_upload(tmp, q.data(), qd.data(), NULL) != MH_OK ||
          mh_artic_batch_link_poses(tmp, poses.data()) != MH_OK) throw std::runtime_error(mh_last_error());
      mh_artic_batch_destroy(tmp);
    }
    double zmin = 1e300;
    for (int w = 0; w < B; w++) for (int l = nj - 2; l < nj; l++) { const double z = poses[((size_t)w * nj + l) * 12 + 11]; if (z < zmin) zmin = z; }
    const double up[3] = { 0.0, 0.0, 1.0 }, pt[3] = { 0.0, 0.0, zmin - 0.03 - 0.05 };
    MobyHip::set_ground_plane(mc, up, pt, 0.0, 100.0);
    MobyHip::BatchedArticulatedBody c(mc, B, q.data(), qd.data()), d(mc, B, q.data(), qd.data());
    for (int s2 = 0; s2 < 300; s2++) c.step(5e-4);
    d.step(5e-4, 300);
    const bool same2 = std::memcmp(c.q().data(), d.q().data(), c.q().size() * sizeof(double)) == 0 &&
                       std::memcmp(c.qd().data(), d.qd().data(), c.qd().size() * sizeof(double)) == 0;
    std::printf("contacts: same=%d status=%d mini_steps=%llu lcp_solves=%llu\n", (int)same2, c.status(0), c.aux(0).mini_steps, c.aux(0).lcp_solves);
    return (same && same2) ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
