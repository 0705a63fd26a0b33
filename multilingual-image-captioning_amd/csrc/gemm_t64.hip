// 64x64x64 tiles, 4 waves per K-group (wave tile 32x32), one, two or four K-groups: instantiations of gemm_kernel.h
#include "gemm_kernel.h"
void launch_gemm_t64(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8, int kgroups) {
  if (kgroups == 4) launch_cfg<32, 32, 2, 64, 4>(tab, akm, bkm, s, f8);
  else if (kgroups == 2) launch_cfg<32, 32, 2, 64, 2>(tab, akm, bkm, s, f8);
  else launch_cfg<32, 32, 2, 64>(tab, akm, bkm, s, f8);
}
