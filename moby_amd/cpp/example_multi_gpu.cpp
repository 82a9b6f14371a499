// One process, every GPU of the node: a batch of worlds cut into one mh_world_batch + one stream per device, the per-interval
// counters reduced with RCCL (ncclAllReduce, SUM and MAX of the MH_COUNTERS-element vector).  The same worlds are
// then stepped as ONE batch on device 0 (created while ANOTHER device is current where there is one: the library switches
// to the batch's device by itself) -- states, solver records and the reduced counters must agree bit for bit.
//   hipcc -std=c++17 example_multi_gpu.cpp -L.. -lmoby_hip -lmoby_hip_io -lrccl -Wl,-rpath,.. -o example_multi_gpu
//   ./example_multi_gpu ../../tests/scenes/three_spheres_on_a_plane.xml [worlds] [steps]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "MobyHipMultiGpu.h"
#include "MobyHipSimulator.h"
#include "../../include/moby_hip_io.h"

int main(int argc, char** argv)
{
  if (argc < 2) { std::printf("usage: example_multi_gpu <scene.xml> [worlds] [steps]\n"); return 2; }
  const int B = (argc > 2) ? std::atoi(argv[2]) : 64, nsteps = (argc > 3) ? std::atoi(argv[3]) : 200;
  mh_io_scene io;
  if (mh_io_load_xml(argv[1], &io) != 0) { std::printf("error: %s\n", mh_io_last_error()); return 1; }
  try {
    const size_t nst = (size_t)io.scene.nb * MH_BODY_STATE;
    std::vector<double> st0((size_t)B * nst);
    for (int w = 0; w < B; w++) {                                   // world w: the scene's state, the top body a little lower per world
      std::memcpy(&st0[(size_t)w * nst], io.state, nst * sizeof(double));
      st0[(size_t)w * nst + (size_t)(io.scene.nb - 1) * MH_BODY_STATE + 9] = -1e-3 * w;
    }
    MobyHip::MultiDeviceTimeSteppingSimulator node(io.scene, B, st0.data());
    const int G = node.num_devices();
    node.step(1e-3, nsteps);
    const MobyHip::Interval iv = node.reduce();
    std::vector<double> st; std::vector<mh_world_aux> aux;
    node.download(st, aux);
    bool placed = true;
    for (int g = 0; g < G; g++) placed = placed && node.device_of_batch(g) == g;

    // the same worlds as one batch of device 0, driven from a thread whose current device is the LAST one
    mh_world_batch* one = NULL;
    if (mh_device_set(0) != MH_OK || mh_world_batch_create(&io.scene, B, &one) != MH_OK) throw std::runtime_error(mh_last_error());
    if (mh_device_set(G - 1) != MH_OK) throw std::runtime_error(mh_last_error());
    std::vector<double> st1((size_t)B * nst); std::vector<mh_world_aux> aux1((size_t)B);
    if (mh_world_batch_upload(one, st0.data(), NULL) != MH_OK || mh_world_batch_step(one, NULL, 1e-3, nsteps, NULL) != MH_OK
        || mh_world_batch_download(one, st1.data(), aux1.data()) != MH_OK) throw std::runtime_error(mh_last_error());
    const bool affinity = mh_world_batch_device(one) == 0 && mh_device_get() == G - 1;      // the caller's device was restored
    mh_world_batch_destroy(one);

    const bool same = std::memcmp(st.data(), st1.data(), st.size() * sizeof(double)) == 0 && std::memcmp(aux.data(), aux1.data(), aux.size() * sizeof(mh_world_aux)) == 0;
    unsigned long long rows = 0, steps = 0, pivmax = 0;
    for (int w = 0; w < B; w++) { rows += aux1[(size_t)w].lcp_rows; steps += aux1[(size_t)w].steps; if (aux1[(size_t)w].lcp_pivots > pivmax) pivmax = aux1[(size_t)w].lcp_pivots; }
    const bool reduced = iv.sums[0] == steps && iv.sums[1] == rows && iv.maxs[2] == pivmax && iv.sums[3] == 0;
    std::printf("devices=%d worlds=%d steps=%d placed=%d affinity=%d same=%d reduced=%d rows=%llu world_steps=%llu max_pivots_of_a_world=%llu\n",
                G, B, nsteps, (int)placed, (int)affinity, (int)same, (int)reduced, iv.sums[1], iv.sums[0], iv.maxs[2]);
    return (placed && affinity && same && reduced) ? 0 : 1;
  } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
}
