// The workgroup-per-problem LCP solver, 1024-thread geometry: one problem per CU with 16 waves to hide its round trips
// (faster per problem from n = 192 up while the batch does not fill the chip twice over; always from n = 384 up).
#include <hip/hip_runtime.h>
#define MH_BLK_NS blkw
#define MH_BLK_T 1024
#define MH_BLK_UCH 256
#define MH_BLK_PANEL_CAP 3584
#define MH_BLK_CN 1024
#define MH_BLK_KATTR
#define MH_BLK_LAUNCHER mh_launch_lcp_blkw
#include "mh_lcp_block.h"
