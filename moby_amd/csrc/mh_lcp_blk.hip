// The workgroup-per-problem LCP solver, 256-thread geometry: the throughput one (batches larger than the chip).  Half the LDS
// staging of the wide geometry (panel 14 KB, pivot-row chunk 128 columns).  THREE problems share a CU (a 170-VGPR budget): with the
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
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(MH_BLK_WAVES, MH_BLK_WAVES)))
#define MH_BLK_LAUNCHER mh_launch_lcp_blk
#include "mh_lcp_block.h"
