// MobyHipMultiGpu.h -- one process driving every GPU of a node through libmoby_hip.so.
//
// Worlds are independent: a batch of B worlds is cut into contiguous ranges [g B/G, (g+1) B/G), one mh_world_batch and one
// HIP stream per device, and nothing crosses devices but the per-interval counters -- a SUM and a MAX all-reduce of an
// MH_COUNTERS-element vector through RCCL's C API (ncclAllReduce over xGMI), never anything per step.  The reference has no
// counterpart (it is single-threaded); the class keeps the calling conventions of Moby::TimeSteppingSimulator
// (/root/reference/include/Moby/TimeSteppingSimulator.h:36: step(dt) returns dt, current_time advances).
//
//   MobyHip::MultiDeviceTimeSteppingSimulator sim(scene, B, state0, /*replicate=*/true);   // all devices mh_device_count() reports
//   sim.step(1e-3, 200);                      // every device advances its share, 200 steps in one launch each, concurrently
//   MobyHip::Interval iv = sim.reduce();      // device-side counters -> ncclAllReduce (SUM, MAX) -> host
//   iv.sums[1]  // LCP rows solved by the whole node since t = 0;   iv.maxs[2]  // pivots of the node's slowest world
//
// Build: hipcc (or g++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include) ... -lmoby_hip -lrccl -lamdhip64
#ifndef MOBY_HIP_MULTI_GPU_H
#define MOBY_HIP_MULTI_GPU_H
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/moby_hip.h"

namespace MobyHip {

struct Interval { unsigned long long sums[MH_COUNTERS], maxs[MH_COUNTERS]; };

class MultiDeviceTimeSteppingSimulator {
 public:
  double current_time;

  /// state: B x nb x 13 doubles (or nb x 13, replicated); ndev <= 0: every visible device
  MultiDeviceTimeSteppingSimulator(const mh_scene& scene, int B, const double* state, bool replicate = false, int ndev = 0)
      : current_time(0.0), _scene(scene), _B(B)
  {
    const int have = mh_device_count();
    if (have <= 0) throw std::runtime_error("no HIP device visible");
    const int G = (ndev > 0 && ndev < have) ? ndev : have;
    if (B < G) throw std::runtime_error("fewer worlds than devices");
    _dev.resize((size_t)G);
    const size_t nst = (size_t)scene.nb * MH_BODY_STATE;
    std::vector<double> part;
    for (int g = 0; g < G; g++) {
      Dev& d = _dev[(size_t)g];
      d.first = (int)((long long)g * B / G); d.count = (int)((long long)(g + 1) * B / G) - d.first;     // [g B/G, (g+1) B/G)
      check(mh_device_set(g));
      hip(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking), "hipStreamCreate");
      hip(hipMalloc((void**)&d.red, 2 * MH_COUNTERS * sizeof(unsigned long long)), "hipMalloc");
      check(mh_world_batch_create(&_scene, d.count, &d.wb));                                             // lives on device g from here on
      part.resize((size_t)d.count * nst);
      for (int w = 0; w < d.count; w++) for (size_t k = 0; k < nst; k++) part[(size_t)w * nst + k] = state[replicate ? k : ((size_t)(d.first + w)) * nst + k];
      check(mh_world_batch_upload(d.wb, part.data(), NULL));
    }
    std::vector<int> ids((size_t)G); for (int g = 0; g < G; g++) ids[(size_t)g] = g;
    _comms.resize((size_t)G);
    nccl(ncclCommInitAll(_comms.data(), G, ids.data()), "ncclCommInitAll");      // one communicator per device, all in this process
    check(mh_device_set(0));
  }
  ~MultiDeviceTimeSteppingSimulator() {
    for (size_t g = 0; g < _comms.size(); g++) if (_comms[g]) (void)ncclCommDestroy(_comms[g]);
    for (size_t g = 0; g < _dev.size(); g++) {
      Dev& d = _dev[g];
      if (d.wb) (void)mh_world_batch_destroy(d.wb);               // (switches to the batch's device by itself)
      if (mh_device_set((int)g) == MH_OK) { if (d.red) (void)hipFree(d.red); if (d.stream) (void)hipStreamDestroy(d.stream); }
    }
    (void)mh_device_set(0);
  }

  /// Simulator::step(dt) x nsteps for every world of the node: one launch per device, all in flight together
  double step(double dt, int nsteps = 1) {
    for (size_t g = 0; g < _dev.size(); g++) check(mh_world_batch_step(_dev[g].wb, _dev[g].stream, dt, nsteps, NULL));   // no device switch by the caller
    current_time += dt * nsteps;
    return dt;
  }
  /// waits for every device's stream
  void synchronize() { for (size_t g = 0; g < _dev.size(); g++) { check(mh_device_set((int)g)); hip(hipStreamSynchronize(_dev[g].stream), "hipStreamSynchronize"); } check(mh_device_set(0)); }

  /// the node's counters since t = 0: per-device sums / maxima on the device, then one SUM and one MAX all-reduce over the devices
  Interval reduce() {
    for (size_t g = 0; g < _dev.size(); g++) check(mh_world_batch_counters_dev(_dev[g].wb, _dev[g].stream, _dev[g].red, _dev[g].red + MH_COUNTERS));
    nccl(ncclGroupStart(), "ncclGroupStart");
    for (size_t g = 0; g < _dev.size(); g++) {
      nccl(ncclAllReduce(_dev[g].red, _dev[g].red, MH_COUNTERS, ncclUint64, ncclSum, _comms[g], _dev[g].stream), "ncclAllReduce(sum)");
      nccl(ncclAllReduce(_dev[g].red + MH_COUNTERS, _dev[g].red + MH_COUNTERS, MH_COUNTERS, ncclUint64, ncclMax, _comms[g], _dev[g].stream), "ncclAllReduce(max)");
    }
    nccl(ncclGroupEnd(), "ncclGroupEnd");
    synchronize();
    Interval iv, other;
    for (size_t g = 0; g < _dev.size(); g++) {                  // every device holds the same reduced vectors: read them all, check it
      check(mh_device_set((int)g));
      Interval& dst = (g == 0) ? iv : other;
      hip(hipMemcpy(dst.sums, _dev[g].red, MH_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost), "hipMemcpy");
      hip(hipMemcpy(dst.maxs, _dev[g].red + MH_COUNTERS, MH_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost), "hipMemcpy");
      if (g > 0) for (int k = 0; k < MH_COUNTERS; k++) if (other.sums[k] != iv.sums[k] || other.maxs[k] != iv.maxs[k]) throw std::runtime_error("all-reduce results differ between devices");
    }
    check(mh_device_set(0));
    return iv;
  }

  int num_devices() const { return (int)_dev.size(); }
  int num_worlds() const { return _B; }
  int first_world(int g) const { return _dev[(size_t)g].first; }
  int world_count(int g) const { return _dev[(size_t)g].count; }
  int device_of_batch(int g) const { return mh_world_batch_device(_dev[(size_t)g].wb); }
  /// body states and solver records of the whole node, worlds in their global order
  void download(std::vector<double>& state, std::vector<mh_world_aux>& aux) {
    const size_t nst = (size_t)_scene.nb * MH_BODY_STATE;
    state.resize((size_t)_B * nst); aux.resize((size_t)_B);
    synchronize();
    for (size_t g = 0; g < _dev.size(); g++) check(mh_world_batch_download(_dev[g].wb, state.data() + (size_t)_dev[g].first * nst, aux.data() + _dev[g].first));
  }

 private:
  struct Dev { int first, count; hipStream_t stream; unsigned long long* red; mh_world_batch* wb; Dev() : first(0), count(0), stream(NULL), red(NULL), wb(NULL) {} };
  MultiDeviceTimeSteppingSimulator(const MultiDeviceTimeSteppingSimulator&);
  MultiDeviceTimeSteppingSimulator& operator=(const MultiDeviceTimeSteppingSimulator&);
  static void check(int rc) { if (rc != MH_OK) throw std::runtime_error(mh_last_error()); }
  static void hip(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
  static void nccl(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
  mh_scene _scene; int _B;
  std::vector<Dev> _dev; std::vector<ncclComm_t> _comms;
};

} // namespace MobyHip
#endif
