// 192x128x64 tiles, 8 waves (wave tile 96x32), two blocks per CU (2 x 80 KiB of LDS): instantiations of gemm_kernel.h for the
// launches whose row count makes 128-row tiles spill into a second round (packed decoder rows 2049..3072 and the ViT's 3200 rows
// against N = 3072 / 4096: 600..672 tiles of 128x128 on 512 slots, 408..448 tiles of 192x128).  bf16, k-contiguous A only.
#include "gemm_kernel.h"
void launch_gemm_t192(const LaunchTable& tab, int bkm, hipStream_t s) {
  const bool plain = table_is_plain(tab);
  if (plain) launch_cfg_p<96, 32, 4, 64, 1, true>(tab, 0, bkm, s);
  else launch_cfg_p<96, 32, 4, 64, 1, false>(tab, 0, bkm, s);
}
