// micro-test (developer tool): do workgroups of two streams that share a compute unit keep their LDS to themselves, also for 8- and
// 16-byte LDS accesses and when the CU's 160 KB are fully packed?  Stream A: 256-lane workgroups with 64 652 bytes of LDS
// (k_track_step's footprint); stream B: small workgroups with 9.6 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS, int BLOCK, int VEC>
__global__ __launch_bounds__(BLOCK) void lds_hold(unsigned tag, long long spin, unsigned* bad) {
  __shared__ __attribute__((aligned(16))) unsigned s[WORDS];
  const unsigned key = tag ^ blockIdx.x;
  if (VEC == 4) {
    for (int i = threadIdx.x; i < WORDS / 4; i += BLOCK) {
      uint4 v = make_uint4(key ^ (4 * i) * 2654435761u, key ^ (4 * i + 1) * 2654435761u, key ^ (4 * i + 2) * 2654435761u, key ^ (4 * i + 3) * 2654435761u);
      ((uint4*)s)[i] = v;
    }
    for (int i = (WORDS / 4) * 4 + threadIdx.x; i < WORDS; i += BLOCK) s[i] = key ^ i * 2654435761u;
  } else if (VEC == 2) {
    for (int i = threadIdx.x; i < WORDS / 2; i += BLOCK) ((uint2*)s)[i] = make_uint2(key ^ (2 * i) * 2654435761u, key ^ (2 * i + 1) * 2654435761u);
    for (int i = (WORDS / 2) * 2 + threadIdx.x; i < WORDS; i += BLOCK) s[i] = key ^ i * 2654435761u;
  } else {
    for (int i = threadIdx.x; i < WORDS; i += BLOCK) s[i] = key ^ i * 2654435761u;
  }
  __syncthreads();
  long long t0 = clock64();
  while (clock64() - t0 < spin) { }
  __syncthreads();
  unsigned e = 0;
  if (VEC == 4) {
    for (int i = threadIdx.x; i < WORDS / 4; i += BLOCK) {
      const uint4 v = ((uint4*)s)[(i * 7 + 3) % (WORDS / 4)];
      const int j = (i * 7 + 3) % (WORDS / 4);
      e += v.x != (key ^ (4 * j) * 2654435761u); e += v.y != (key ^ (4 * j + 1) * 2654435761u);
      e += v.z != (key ^ (4 * j + 2) * 2654435761u); e += v.w != (key ^ (4 * j + 3) * 2654435761u);
    }
  } else {
    for (int i = threadIdx.x; i < WORDS; i += BLOCK) e += s[i] != (key ^ i * 2654435761u);
  }
  if (e) atomicAdd(bad, e);
}
template <int VEC>
static void run(const char* name) {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  unsigned* bad;
  hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  hipDeviceSynchronize();
  for (int i = 0; i < 2000; i++) {
    hipLaunchKernelGGL((lds_hold<16163, 256, VEC>), dim3(400), dim3(256), 0, a, 0xA0000000u + i, 8000LL, bad);
    hipLaunchKernelGGL((lds_hold<2400, 256, 1>), dim3(1200), dim3(256), 0, b, 0xB0000000u + i, 3000LL, bad + 1);
    hipLaunchKernelGGL((lds_hold<2400, 64, 1>), dim3(4800), dim3(64), 0, b, 0xC0000000u + i, 2000LL, bad + 1);
  }
  hipDeviceSynchronize();
  unsigned h[2] = {9, 9};
  hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("%s: %u corrupted words in the 64 652-byte workgroups, %u in the 9.6 KB ones\n", name, h[0], h[1]);
  hipStreamDestroy(a); hipStreamDestroy(b);
}
int main() {
  run<1>("4-byte LDS accesses ");
  run<2>("8-byte LDS accesses ");
  run<4>("16-byte LDS accesses");
  return 0;
}
