// MobyHipSimulator.h -- reference-side adapter for the whole-step seam:
// a class with the calling conventions of Moby::TimeSteppingSimulator
// (/root/reference/include/Moby/TimeSteppingSimulator.h:36, include/Moby/Simulator.h:50) that advances
// B copies of one scene on the GPU through libmoby_hip.so.
//
//   MobyHip::BatchedTimeSteppingSimulator sim(scene, B, state);     // or ::from_xml(path, B) with libmoby_hip_io
//   sim.step(1e-3);                 // Simulator::step(dt): every world advances by dt; returns dt
//   sim.step(1e-3, 1000);           // the same step 1000 times inside ONE launch
//   sim.current_time;               // Simulator::current_time of world 0
//   sim.get_generalized_coordinates_euler(w, b, q);   // x y z qx qy qz qw of body b of world w
//
// Conventions kept from the reference: step() returns the step size; bodies are addressed in id order
// (programs/regress.cpp:66-69); exceptions of the reference become sticky per-world status bits
// (status(w) & MH_WORLD_LCP_FAILED <=> LCPSolverException, MH_WORLD_IMPACT_TOL <=> the warned
// ImpactToleranceException); the rand() stream of every world starts at srand(1) like a fresh process.
// Throws std::runtime_error only for a broken library / device / unsupported scene.
#ifndef MOBY_HIP_SIMULATOR_ADAPTER_H
#define MOBY_HIP_SIMULATOR_ADAPTER_H
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/moby_hip.h"

namespace MobyHip {

class BatchedTimeSteppingSimulator {
 public:
  double current_time;

  BatchedTimeSteppingSimulator(const mh_scene& scene, int B, const double* state /* B x nb x 13, or nb x 13 replicated if replicate */,
                               bool replicate = false)
      : current_time(0.0), _scene(scene), _B(B), _wb(NULL), _dirty(false)
  {
    if (mh_world_batch_create(&_scene, B, &_wb) != MH_OK) throw std::runtime_error(mh_last_error());
    const size_t nst = (size_t)_scene.nb * MH_BODY_STATE;
    _state.resize((size_t)B * nst);
    for (int w = 0; w < B; w++) for (size_t k = 0; k < nst; k++) _state[(size_t)w * nst + k] = state[replicate ? k : (size_t)w * nst + k];
    _aux.resize((size_t)B);
    if (mh_world_batch_upload(_wb, _state.data(), NULL) != MH_OK) throw std::runtime_error(mh_last_error());
  }
  ~BatchedTimeSteppingSimulator() { if (_wb) mh_world_batch_destroy(_wb); }

  /// Simulator::step(dt), for every world; nsteps > 1 repeats it inside one launch
  double step(double dt, int nsteps = 1) {
    if (mh_world_batch_step(_wb, /*stream=*/NULL, dt, nsteps, /*traj=*/NULL) != MH_OK) throw std::runtime_error(mh_last_error());
    _dirty = true;
    current_time += dt * nsteps;
    return dt;
  }
  int num_worlds() const { return _B; }
  int num_bodies() const { return _scene.nb; }
  /// DynamicBodyd::get_generalized_coordinates_euler of body b of world w
  void get_generalized_coordinates_euler(int w, int b, double q[7]) { sync(); for (int k = 0; k < 7; k++) q[k] = _state[((size_t)w * _scene.nb + b) * MH_BODY_STATE + k]; }
  /// get_generalized_velocity(eSpatial): linear (COM, world axes) then angular
  void get_generalized_velocity(int w, int b, double v[6]) { sync(); for (int k = 0; k < 6; k++) v[k] = _state[((size_t)w * _scene.nb + b) * MH_BODY_STATE + 7 + k]; }
  int status(int w) { sync(); return _aux[(size_t)w].status; }
  const mh_world_aux& solver_state(int w) { sync(); return _aux[(size_t)w]; }   // what a checkpoint must keep besides the body state
  const std::vector<double>& state() { sync(); return _state; }

 private:
  BatchedTimeSteppingSimulator(const BatchedTimeSteppingSimulator&);
  BatchedTimeSteppingSimulator& operator=(const BatchedTimeSteppingSimulator&);
  void sync() {
    if (!_dirty) return;
    if (mh_world_batch_download(_wb, _state.data(), _aux.data()) != MH_OK) throw std::runtime_error(mh_last_error());
    _dirty = false;
  }
  mh_scene _scene; int _B; mh_world_batch* _wb; bool _dirty;
  std::vector<double> _state; std::vector<mh_world_aux> _aux;
};

} // namespace MobyHip
#endif
