// MobyHipStackSimulator.h -- reference-side adapter for LARGE worlds (include/moby_hip_stack.h): the whole-step seam B5
// (Moby::TimeSteppingSimulator::step, include/Moby/TimeSteppingSimulator.h:36) and the stabilisation seam B3
// (Moby::ConstraintStabilization::stabilize, include/Moby/ConstraintStabilization.h:24; call site
// src/TimeSteppingSimulator.cpp:97) for B copies of one scene of up to 128 bodies.
//
//   MobyHip::BatchedStackSimulator sim(scene, B, state);
//   sim.step(1e-3);               // Simulator::step(dt) on every world; returns dt
//   sim.stabilize();              // cstab.stabilize(sim) alone: configurations change, velocities are restored
//
// Conventions as in MobyHipSimulator.h: bodies in id order, x y z qx qy qz qw + eSpatial velocity per body, exceptions
// of the reference as sticky status bits, rand() of every world at srand(1).
#ifndef MOBY_HIP_STACK_SIMULATOR_ADAPTER_H
#define MOBY_HIP_STACK_SIMULATOR_ADAPTER_H
#include <stdexcept>
#include <vector>
#include "../../include/moby_hip_stack.h"

namespace MobyHip {

class BatchedStackSimulator {
 public:
  double current_time;
  BatchedStackSimulator(const mh_big_scene& scene, int B, const double* state /* B x nb x 13 */)
      : current_time(0.0), _nb(scene.nb), _B(B), _bb(NULL), _dirty(false)
  {
    if (mh_big_batch_create(&scene, B, &_bb) != MH_OK) throw std::runtime_error(mh_last_error());
    _state.assign(state, state + (size_t)B * _nb * MH_BODY_STATE);
    _aux.resize((size_t)B);
    if (mh_big_batch_upload(_bb, _state.data(), NULL) != MH_OK) throw std::runtime_error(mh_last_error());
  }
  ~BatchedStackSimulator() { if (_bb) mh_big_batch_destroy(_bb); }
  double step(double dt, int nsteps = 1) {
    if (mh_big_batch_step(_bb, NULL, dt, nsteps) != MH_OK) throw std::runtime_error(mh_last_error());
    _dirty = true; current_time += dt * nsteps;
    return dt;
  }
  void stabilize() { if (mh_big_batch_stabilize(_bb, NULL) != MH_OK) throw std::runtime_error(mh_last_error()); _dirty = true; }
  void get_generalized_coordinates_euler(int w, int b, double q[7]) { sync(); for (int k = 0; k < 7; k++) q[k] = _state[((size_t)w * _nb + b) * MH_BODY_STATE + k]; }
  void get_generalized_velocity(int w, int b, double v[6]) { sync(); for (int k = 0; k < 6; k++) v[k] = _state[((size_t)w * _nb + b) * MH_BODY_STATE + 7 + k]; }
  int status(int w) { sync(); return _aux[(size_t)w].status; }
  const mh_world_aux& counters(int w) { sync(); return _aux[(size_t)w]; }
 private:
  BatchedStackSimulator(const BatchedStackSimulator&);
  BatchedStackSimulator& operator=(const BatchedStackSimulator&);
  void sync() { if (!_dirty) return; if (mh_big_batch_download(_bb, _state.data(), _aux.data()) != MH_OK) throw std::runtime_error(mh_last_error()); _dirty = false; }
  int _nb, _B; mh_big_batch* _bb; bool _dirty;
  std::vector<double> _state; std::vector<mh_world_aux> _aux;
};

} // namespace MobyHip
#endif
