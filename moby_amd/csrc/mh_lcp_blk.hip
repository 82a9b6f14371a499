// The workgroup-per-problem LCP solver, 256-thread geometry: the throughput one (batches larger than the chip).  Half the LDS
// staging of the wide geometry (panel 14 KB, pivot-row chunk 128 columns) and a 128-VGPR budget let FOUR problems share a CU.
#include <hip/hip_runtime.h>
#define MH_BLK_NS blk
#define MH_BLK_T 256
#define MH_BLK_UCH 128
#define MH_BLK_PANEL_CAP 1792
#define MH_BLK_CN 512
#ifndef MH_BLK_WAVES
#define MH_BLK_WAVES 4
#endif
#define MH_BLK_KATTR __attribute__((amdgpu_waves_per_eu(MH_BLK_WAVES, MH_BLK_WAVES)))
#define MH_BLK_LAUNCHER mh_launch_lcp_blk
#include "mh_lcp_block.h"
