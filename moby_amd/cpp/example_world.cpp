// The batched simulator adapter on a scene file: 100 calls of step(dt) against one call of step(dt, 100).
//   g++ -std=c++11 example_world.cpp -L.. -lmoby_hip -lmoby_hip_io -Wl,-rpath,.. -o example_world
//   ./example_world ../../tests/scenes/three_spheres_on_a_plane.xml
#include <cstdio>
#include <cstring>
#include "MobyHipSimulator.h"
#include "../../include/moby_hip_io.h"

int main(int argc, char** argv)
{
  if (argc < 2) { std::printf("usage: example_world <scene.xml>\n"); return 2; }
  mh_io_scene io;
  if (mh_io_load_xml(argv[1], &io) != 0) { std::printf("error: %s\n", mh_io_last_error()); return 1; }
  try {
    const int B = 4;
    MobyHip::BatchedTimeSteppingSimulator a(io.scene, B, io.state, /*replicate=*/true), b(io.scene, B, io.state, true);
    for (int s = 0; s < 100; s++) a.step(1e-3);
    b.step(1e-3, 100);
    const bool same = std::memcmp(a.state().data(), b.state().data(), a.state().size() * sizeof(double)) == 0;
    double q[7]; a.get_generalized_coordinates_euler(0, io.scene.nb - 1, q);
    std::printf("worlds=%d bodies=%d time=%.3f same=%d status=%d top body: %.9g %.9g %.9g\n", a.num_worlds(), a.num_bodies(), a.current_time, (int)same,
                a.status(0), q[0], q[1], q[2]);
    return same ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
