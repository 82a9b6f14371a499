// The workgroup-per-problem LCP solver, 256-thread geometry for the lcp_lemke kinds with 512 < n <= 1024 (box stacks of 17-32 boxes): the THROUGHPUT form
// for batches that offer more ladder tasks than the chip has CUs.  Four rows per lane, panels of 12 columns and rounds of 8 steps in 256 registers, two
// problems per CU -- where mh_lcp_blkw.hip (1024 threads, one row per lane) holds one.  The solver is bound by the instruction stream of each wave between
// barriers: fewer, fatter waves per problem retire more pivots per second on a full chip; one problem alone is faster in the wide geometry.
// Only the lcp_lemke kinds are instantiated.
#include <hip/hip_runtime.h>
#define MH_BLK_NS blky
#define MH_BLK_T 256
#define MH_BLK_UCH 128
#define MH_BLK_PANEL_CAP 2944
#define MH_BLK_RHS_CAP 1024
#define MH_BLK_LIST_CAP 1024
#define MH_BLK_CN 1024
#define MH_BLK_LEMKE_ONLY 1
#ifndef MH_LL_W
#define MH_LL_W 12
#endif
#ifndef MH_LL_SPARE
#define MH_LL_SPARE 2
#endif
#ifndef MH_LL_G
#define MH_LL_G 8
#endif
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#define MH_BLK_LAUNCHER mh_launch_lcp_blky
#include "mh_lcp_block.h"
