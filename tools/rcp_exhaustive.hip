// Is v_rcp_f32 (1 ulp) + one Newton-Raphson correction with fused multiply-adds the correctly rounded 1/x?
// Exhaustive on the device itself: all 2^32 bit patterns against the IEEE division sequence hipcc emits for 1.0f / x, by class of input.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rcp_exhaustive.hip -o tools/rcp_exhaustive.bin && tools/rcp_exhaustive.bin
// Variants: nr1 = the bare correction (claim: inputs whose reciprocal is a normal number); exact = lsd_rcp_exact of
// lsd_slam_amd/csrc/rcp_exact.hpp, included here (claim: every input, NaNs compared as a class).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../lsd_slam_amd/csrc/rcp_exact.hpp"
__device__ __forceinline__ float rcp_nr1(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r0, 1.0f);
  return __builtin_fmaf(r0, e, r0);
}
__device__ __forceinline__ float rcp_exact(float x) { return lsd_rcp_exact(x); }   // the product's function (lsd_slam_amd/csrc/rcp_exact.hpp)
// classes: 0 zero, 1 denormal, 2 normal with normal reciprocal, 3 normal with denormal reciprocal, 4 normal with infinite reciprocal, 5 inf, 6 nan
__global__ void k_check(unsigned long long* out) {
  unsigned long long n[7] = {0}, bad1[7] = {0}, badx[7] = {0}, bad0[7] = {0};
  for (unsigned long long b = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < (1ull << 32); b += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)b);
    const float ref = 1.0f / x;
    const unsigned xe = ((unsigned)b >> 23) & 0xff, xm = (unsigned)b & 0x7fffff;
    const unsigned re = (__float_as_uint(ref) >> 23) & 0xff;
    int cls;
    if (xe == 0) cls = xm ? 1 : 0;
    else if (xe == 255) cls = xm ? 6 : 5;
    else cls = re == 0 ? 3 : (re == 255 ? 4 : 2);
    const float a = rcp_nr1(x), c = rcp_exact(x), r0 = __builtin_amdgcn_rcpf(x);
    const bool refnan = ref != ref;
    auto differs = [&](float v) { return refnan ? !(v != v) : __float_as_uint(v) != __float_as_uint(ref); };
    n[cls]++;
    if (differs(a)) bad1[cls]++;
    if (differs(c)) badx[cls]++;
    if (differs(r0)) bad0[cls]++;
  }
  for (int k = 0; k < 7; k++) { atomicAdd(&out[k], n[k]); atomicAdd(&out[7 + k], bad1[k]); atomicAdd(&out[14 + k], badx[k]); atomicAdd(&out[21 + k], bad0[k]); }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 28 * 8); hipMemset(d, 0, 28 * 8);
  hipLaunchKernelGGL(k_check, dim3(8192), dim3(256), 0, 0, d);
  unsigned long long h[28];
  hipMemcpy(h, d, 28 * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"zero", "denormal", "normal_recip_normal", "normal_recip_denormal", "normal_recip_inf", "inf", "nan"};
  printf("{");
  unsigned long long tot = 0, totx = 0;
  for (int k = 0; k < 7; k++) {
    printf("\"%s\": {\"inputs\": %llu, \"mismatch_rcp_hw\": %llu, \"mismatch_rcp_plus_one_correction\": %llu, \"mismatch_lsd_rcp_exact\": %llu}, ", names[k], h[k], h[21 + k], h[7 + k], h[14 + k]);
    tot += h[k]; totx += h[14 + k];
  }
  printf("\"inputs_checked\": %llu, \"mismatch_lsd_rcp_exact\": %llu}\n", tot, totx);
  return 0;
}
