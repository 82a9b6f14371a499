// The workgroup-per-problem LCP solver, TWO-WAVEFRONT geometry (128 threads, four rows of a 512-row problem per lane): the throughput
// form of the lcp_lemke kinds.  The solver is bound by instruction issue -- every wave of a problem executes the same "uniform" code
// (pivot bookkeeping, the per-round triangle of the left-looking LU, barriers) besides its share of the arithmetic -- so fewer, fatter
// waves per problem retire more pivots per second on a full chip, as long as two waves fit a SIMD: four problems per CU here, against
// two (256 threads, 256 registers) or three (168 registers).  One problem alone is slower than in the 256-thread geometry.
#include <hip/hip_runtime.h>
#define MH_BLK_NS blk2
#define MH_BLK_T 128
#define MH_BLK_UCH 64
#define MH_BLK_PANEL_CAP 1472
#define MH_BLK_RHS_CAP 512
#define MH_BLK_LIST_CAP 512
#define MH_BLK_CN 512
#ifndef MH_BLK2_WAVES
#define MH_BLK2_WAVES 2
#endif
#ifndef MH_LL_W
#define MH_LL_W 12
#endif
#ifndef MH_LL_SPARE
#define MH_LL_SPARE 2
#endif
#ifndef MH_LL_G
#define MH_LL_G 8
#endif
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(MH_BLK2_WAVES, MH_BLK2_WAVES)))
#define MH_BLK_LAUNCHER mh_launch_lcp_blk2
#include "mh_lcp_block.h"
