// MobyHipImpactHandler.h -- reference-side adapter for the impact-handler seam:
// a class with the calling convention of Moby::ImpactConstraintHandler::process_constraints
// (/root/reference/include/Moby/ImpactConstraintHandler.h:47; caller ConstraintSimulator::
// calc_impacting_unilateral_constraint_forces, src/ConstraintSimulator.cpp:335-340) for B worlds at once,
// through libmoby_hip.so (include/moby_hip_impact.h).
//
//   MobyHip::BatchedImpactHandler h(B, nb, nc, nk, mass, inertia);   // one per batch shape, like the simulator's handler
//   h.process_constraints(state, contacts);       // state: B x nb x 13 in/out (velocities change), contacts: B x nc
//   h.contact_impulse(w, i, cn_cs_ct);            // what update_from_stacked accumulated for contact i of world w
//   if (h.status(w) & MH_WORLD_LCP_FAILED) ...    // the reference would have thrown LCPSolverException
//
// Kept from the reference: the handler object owns _zlast and the rand() stream between calls (warm starts,
// ImpactConstraintHandlerQP.cpp:158-162,233); exceptions become sticky status bits; contacts are the caller's
// (geometry pairs the GPU stepper does not generate -- box-box, box-sphere -- come from Moby's own CCD).
#ifndef MOBY_HIP_IMPACT_HANDLER_ADAPTER_H
#define MOBY_HIP_IMPACT_HANDLER_ADAPTER_H
#include <stdexcept>
#include <vector>
#include "../../include/moby_hip_impact.h"

namespace MobyHip {

class BatchedImpactHandler {
 public:
  BatchedImpactHandler(int B, int nb, int nc, int nk, const double* mass, const double* inertia /* nb x 3 */)
      : _B(B), _nb(nb), _nc(nc), _ib(NULL), _imp((size_t)B * nc * 3), _status((size_t)B), _pivots((size_t)B), _solves((size_t)B)
  {
    if (mh_impact_batch_create(B, nb, nc, nk, mass, inertia, &_ib) != MH_OK) throw std::runtime_error(mh_last_error());
  }
  ~BatchedImpactHandler() { if (_ib) mh_impact_batch_destroy(_ib); }

  /// ImpactConstraintHandler::process_constraints for every world; `state` receives the post-impact velocities
  void process_constraints(double* state, const mh_contact* contacts) {
    if (mh_impact_batch_upload(_ib, state, contacts) != MH_OK) throw std::runtime_error(mh_last_error());
    if (mh_impact_batch_process(_ib, /*stream=*/NULL) != MH_OK) throw std::runtime_error(mh_last_error());
    if (mh_impact_batch_download(_ib, state, _imp.data(), _status.data(), _pivots.data(), _solves.data()) != MH_OK)
      throw std::runtime_error(mh_last_error());
  }
  // what the reference fixes at build time with -DUSE_AP (CMakeLists.txt:19, ImpactConstraintHandler.cpp:139-146):
  // MH_IMPACT_MODEL_DS (Drumwright-Shell, default) or MH_IMPACT_MODEL_AP (Anitescu-Potra) for islands with finite friction
  void set_model(int model) { if (mh_impact_batch_set_model(_ib, model) != MH_OK) throw std::runtime_error(mh_last_error()); }
  int lcp_size() const { return mh_impact_batch_lcp_size(_ib); }
  int status(int w) const { return _status[(size_t)w]; }
  unsigned pivots(int w) const { return _pivots[(size_t)w]; }
  int solves(int w) const { return _solves[(size_t)w]; }
  void contact_impulse(int w, int i, double out[3]) const { for (int d = 0; d < 3; d++) out[d] = _imp[((size_t)w * _nc + i) * 3 + d]; }

 private:
  BatchedImpactHandler(const BatchedImpactHandler&);
  BatchedImpactHandler& operator=(const BatchedImpactHandler&);
  int _B, _nb, _nc; mh_impact_batch* _ib;
  std::vector<double> _imp; std::vector<int> _status; std::vector<unsigned> _pivots; std::vector<int> _solves;
};

} // namespace MobyHip
#endif
