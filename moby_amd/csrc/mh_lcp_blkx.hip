// The workgroup-per-problem LCP solver, 1024-thread geometry for the lcp_lemke kinds with 1024 < n <= 2048 (BASELINE config 4 at the size it
// states: stacks of 64 boxes, impact LCP n = 2048; /root/reference/example/stacks/stack.xml:36-96, src/LCP.cpp:837-838).  The structure-exploiting
// LU of mh_lu_left.inc with TWO rows per lane: its index arrays (23 bytes per row) take 46 KB of LDS, the right-hand side 16 KB; a 128-register
// budget (sixteen waves on one CU) holds panels of 12 columns and rounds of 8 steps.  Before this geometry existed every pivot above 1024 rows
// was a dense dgesv of the assembled basis (one impact-handler call of 8 such worlds: 65 s, profiles/r04_e_config4_64_boxes_x8_impact_call.json).
// Only the lcp_lemke kinds are instantiated: the lcp_fast kinds keep mh_lcp_blkw.hip at every size.
#include <hip/hip_runtime.h>
#define MH_BLK_NS blkx
#define MH_BLK_T 1024
#define MH_BLK_UCH 256
#define MH_BLK_PANEL_CAP 5888
#define MH_BLK_RHS_CAP 2048
#define MH_BLK_CN 2048
#define MH_BLK_LEMKE_ONLY 1
#define MH_BLK_NO_REGLU 1
#ifndef MH_LL_W
#define MH_LL_W 12
#endif
#ifndef MH_LL_SPARE
#define MH_LL_SPARE 2
#endif
#ifndef MH_LL_G
#define MH_LL_G 8
#endif
#define MH_BLK_KATTR
#define MH_BLK_LAUNCHER mh_launch_lcp_blkx
#include "mh_lcp_block.h"
