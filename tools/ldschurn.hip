// micro-test (developer tool): k_track_step's launch shape — 400 workgroups with ~64 KB of static LDS of which most leave at once and a few
// reduce through LDS — beside a second stream whose workgroups use 9.6 KB of LDS each.  Do the few keep their LDS to themselves?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void big(unsigned tag, int busy, unsigned* bad, const int* gate) {
  __shared__ unsigned s[16163];
  if ((int)blockIdx.x >= busy && *gate == 0) return;            // most workgroups leave immediately (a data-dependent test, like st.done)
  const unsigned key = tag ^ blockIdx.x;
  for (int i = threadIdx.x; i < 16163; i += 256) s[i] = key ^ i * 2654435761u;
  __syncthreads();
  unsigned e = 0;
  for (int r = 0; r < 3; r++) {
    for (int i = threadIdx.x; i < 16163; i += 256) e += s[(i * 7 + r) % 16163] != (key ^ ((i * 7 + r) % 16163) * 2654435761u);
    __syncthreads();
  }
  if (e) atomicAdd(bad, e);
}
__global__ __launch_bounds__(256) void small_lds(float* p) {
  __shared__ float s[2400];
  for (int i = threadIdx.x; i < 2400; i += 256) s[i] = (float)(i + blockIdx.x);
  __syncthreads();
  float acc = 0;
  for (int r = 0; r < 40; r++) for (int i = threadIdx.x; i < 2400; i += 256) acc += s[(i * 7 + r) % 2400];
  p[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  unsigned* bad; int* gate; float* buf;
  hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  hipMalloc(&gate, 4); hipMemset(gate, 0, 4);
  hipMalloc(&buf, 1200 * 256 * 4);
  hipDeviceSynchronize();
  for (int i = 0; i < 4000; i++) {
    if ((i % 12) == 0) for (int k = 0; k < 4; k++) hipLaunchKernelGGL(small_lds, dim3(1200), dim3(256), 0, b, buf);
    hipLaunchKernelGGL(big, dim3(400), dim3(256), 0, a, 0xA0000000u + i, (i % 12) == 0 ? 5 : 144, bad, gate);
  }
  hipDeviceSynchronize();
  unsigned h = 9;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("LDS of the few busy 64 KB workgroups beside 9.6 KB workgroups of another stream: %u corrupted words\n", h);
  return 0;
}
