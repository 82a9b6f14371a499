// The "small" variant of the many-worlds kernel (mh_world_wave.inc): <= 4 bodies, <= 6 pairs, <= 6 contacts, <= 12 Jacobian rows, spheres + Drumwright-Shell model only.
// One translation unit per variant: the LDS image, occupancy and feature set differ, and the three compile in parallel.
#include <hip/hip_runtime.h>
#include "../../include/moby_hip.h"
#include "mh_host.h"
#ifdef MH_PROFILE_BUILD      /* mh_world_small_prof.hip: the same kernel with the s_memtime stamps compiled in (mh_world_batch_profile) */
#define MHW_NS small_prof
#define MHW_VARIANT_GETTER mh_world_variant_small_prof
#else
#define MHW_NS small
#define MHW_VARIANT_GETTER mh_world_variant_small
#endif
#define MHW_NOSLIP 0
#define MHW_BOX 0
#define MHW_NB 4
#define MHW_MAX_PAIRS 6
#define MHW_MAX_CONTACTS 6
#define MHW_MAX_ROWS 12
#define MHW_MAX_GROWS 12
#define MHW_WAVES_PER_SIMD 4
#include "mh_world_wave.inc"

static hipError_t upload_tables(const void* fric, size_t fric_bytes, const void* pow10, size_t pow10_bytes)
{
  if (fric_bytes != sizeof(mh::FricTable) || pow10_bytes != sizeof(mh::Pow10Table)) return hipErrorInvalidValue;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(mh::c_fric), fric, fric_bytes);
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(mh::c_pow10), pow10, pow10_bytes);
  return e;
}

const mh_world_variant* MHW_VARIANT_GETTER()
{
  static const mh_world_variant v = { mh::MHW_NS::mh_k_world_step, mh::MHW_NS::PH_COUNT, upload_tables };
  return &v;
}
