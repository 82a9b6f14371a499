// The "large" variant of the many-worlds kernel with the in-kernel phase profiler compiled in: what mh_world_batch_profile launches (tools/world_profile.py,
// bench.py's long_horizon idle figures).  The production kernel of mh_world_large.hip carries no stamp code.
#define MH_PROFILE_BUILD 1
#include "mh_world_large.hip"
