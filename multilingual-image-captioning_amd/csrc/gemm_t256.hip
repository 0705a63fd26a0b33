// 256x256x64 tiles, 8 waves (wave tile 128x64): instantiations of gemm_kernel.h
#include "gemm_kernel.h"
void launch_gemm_t256(const LaunchTable& tab, int akm, int bkm, hipStream_t s, int f8) { launch_cfg<128, 64, 4, 64>(tab, akm, bkm, s, f8); }
void launch_gemm_t256_ce(const LaunchTable& tab, int akm, hipStream_t s) {
  const bool plain = table_is_plain(tab);
  if (akm) {  // dE: bare fp32 epilogue + row sums only (the feature-rich epilogue does not fit the register file beside the transform)
    if (plain) launch_cfg_ce<true, true>(tab, s);
    else mic_set_error("mic_gemm: fused cross-entropy backward with A k-major needs the bare epilogue (no activation / Z / accumulate / split)");
  } else {
    if (plain) launch_cfg_ce<false, true>(tab, s); else launch_cfg_ce<false, false>(tab, s);
  }
}
