// 256x256x64 tiles, 8 waves (wave tile 128x64): instantiations of gemm_kernel.h
#include "gemm_kernel.h"
void launch_gemm_t256(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8) { launch_cfg<128, 64, 4, 64>(tab, akm, bkm, s, f8); }
