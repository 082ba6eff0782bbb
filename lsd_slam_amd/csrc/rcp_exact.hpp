// Part of liblsdhip (gfx950); included by lsdhip_internal.hpp and by tools/rcp_exhaustive.hip, which checks exactly this function.
#pragma once
#include <hip/hip_runtime.h>
// 1.0f / x, bit for bit, in 4 instructions and a never-taken branch instead of the IEEE division sequence (two v_div_scale, v_rcp, five
// fused multiply-adds, v_div_fmas, v_div_fixup): v_rcp_f32 (1 ulp) and one Newton-Raphson correction with fused multiply-adds is the
// correctly rounded reciprocal of every float whose reciprocal is a normal number.  Where the hardware's estimate is not a normal number
// (x = 0 / inf / NaN; |x| > 2^126 and denormal x, whose denormal / huge reciprocals v_rcp_f32 flushes) the division itself runs.
// Checked on the device for all 2^32 bit patterns against `1.0f / x` by tools/rcp_exhaustive.hip (profiles/r05_rcp_exhaustive.json:
// 0 mismatches).  For the kernels bound by instruction issue — the regulariser holds one reciprocal per neighbour.
// -DLSD_RCP_IEEE builds the division back in (A/B builds).
// `normalMask`: the class mask of +-normal numbers (LSD_RCP_NORMAL_MASK) — a literal the instruction cannot carry inline; a loop that
// takes many reciprocals hands over one it keeps in a scalar register (lsd_rcp_mask()) instead of one move per use.
#define LSD_RCP_NORMAL_MASK 0x108
__device__ __forceinline__ int lsd_rcp_mask() {
  int m = LSD_RCP_NORMAL_MASK;
  asm volatile("" : "+s"(m));      // (opaque: stays in the register it is given, once)
  return m;
}
__device__ __forceinline__ float lsd_rcp_exact(float x, const int normalMask = LSD_RCP_NORMAL_MASK) {
#ifdef LSD_RCP_IEEE
  return 1.0f / x;
#else
  const float r0 = __builtin_amdgcn_rcpf(x);
  if (__builtin_expect(!__builtin_amdgcn_classf(r0, normalMask), 0)) return 1.0f / x;   // estimate not +-normal
  const float e = __builtin_fmaf(-x, r0, 1.0f);
  return __builtin_fmaf(r0, e, r0);
#endif
}
