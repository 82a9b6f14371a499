// The workgroup-per-problem LCP solver, 256-thread geometry: the throughput one (batches larger than the chip).  Half the LDS
// staging of the wide geometry (panel 14 KB, pivot-row chunk 128 columns).  TWO problems share a CU since round 4 (256 registers: the left-looking LU keeps a
// 20-column panel and a 16-step round of multipliers in registers; at 170 registers it spilled what it saved).  Round 3: three problems per CU: with the
// state the factor reuse of mh_lu_compact.inc keeps alive, four problems at 128 VGPRs spilled 161 of them and measured 5-7 % slower
// (16 boxes x 256 worlds 4.42 -> 4.10 s, 16 x 1024 full step 30.6 -> 28.8 s; it was the other way round before, 5.5 vs 5.2 s at four).
#include <hip/hip_runtime.h>
#define MH_BLK_NS blk
#define MH_BLK_T 256
#ifndef MH_BLK_UCH
#define MH_BLK_UCH 128
#endif
#define MH_BLK_PANEL_CAP 1792
#define MH_BLK_CN 512
#ifndef MH_BLK_WAVES
#define MH_BLK_WAVES 2
#endif
#ifndef MH_LL_W
#define MH_LL_W 20            /* left-looking LU (mh_lu_left.inc): panels of 16 columns + 4 slots for the fill-ins that join, rounds of 16 steps */
#endif
#ifndef MH_LL_SPARE
#define MH_LL_SPARE 4
#endif
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(MH_BLK_WAVES, MH_BLK_WAVES)))
#define MH_BLK_LAUNCHER mh_launch_lcp_blk
#include "mh_lcp_block.h"
