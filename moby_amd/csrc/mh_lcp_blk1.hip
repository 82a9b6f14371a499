// The workgroup-per-problem LCP solver, ONE-WAVEFRONT geometry: the throughput form of the lcp_lemke kinds for batches that offer several
// times more problems (ladder tasks) than the chip has CUs.  A wavefront owns a problem (lane l holds rows l, l + 64, ...: eight rows at
// n = 512), so nothing in it waits for another wave -- the barriers of mh_lcp_block.h / mh_lu_left.inc cost nothing in a one-wave
// workgroup, the per-round triangle of the left-looking LU is solved once instead of once per wave, every "uniform" instruction is issued
// once per problem instead of four times -- and six to eight problems share a CU instead of three.  One problem alone is slower than in
// the 256-thread geometry (a quarter of the lanes); a full chip retires more pivots per second.  Panels of 8 columns (64 registers of
// a lane's 8 rows), rounds of 4 steps.
#include <hip/hip_runtime.h>
#define MH_BLK_NS blk1
#define MH_BLK_T 64
#define MH_BLK_UCH 64
#define MH_BLK_PANEL_CAP 1472
#define MH_BLK_RHS_CAP 512
#define MH_BLK_LIST_CAP 512
#define MH_BLK_CN 512
#ifndef MH_BLK1_WAVES
#define MH_BLK1_WAVES 2
#endif
#ifndef MH_LL_W
#define MH_LL_W 8
#endif
#ifndef MH_LL_SPARE
#define MH_LL_SPARE 2
#endif
#ifndef MH_LL_G
#define MH_LL_G 4
#endif
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(MH_BLK1_WAVES, MH_BLK1_WAVES)))
#define MH_BLK_LAUNCHER mh_launch_lcp_blk1
#include "mh_lcp_block.h"
