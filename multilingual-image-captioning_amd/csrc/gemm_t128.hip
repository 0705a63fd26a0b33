// 128x128x64 tiles, 8 waves per K-group (wave tile 64x32), one or two K-groups: instantiations of gemm_kernel.h
#include "gemm_kernel.h"
void launch_gemm_t128(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8, int kgroups) {
  if (kgroups == 2) launch_cfg<64, 32, 4, 64, 2>(tab, akm, bkm, s, f8);
  else launch_cfg<64, 32, 4, 64>(tab, akm, bkm, s, f8);
}
