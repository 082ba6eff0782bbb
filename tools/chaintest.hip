// micro-test (developer tool): does the kernel-to-kernel hand-over INSIDE one stream (a chain of dependent launches, each reading what
// other workgroups / XCDs of the previous launch wrote, double-buffered like k_track_step's scratch) survive a second stream that
// runs unrelated kernels at the same time?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void link(const int* src, int* dst, int expect, int shift, int* bad) {
  const int b = (blockIdx.x + shift) % gridDim.x;
  const int v = src[b * 256 + threadIdx.x];
  if (v != expect) atomicAdd(bad, 1);
  dst[blockIdx.x * 256 + threadIdx.x] = expect + 1;
}
__global__ void stream_other(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.0f;
}
static int run(bool other, int grid, int links) {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  int *buf, *bad;
  float* big;
  const size_t nbig = 16u << 20;
  hipMalloc(&buf, (size_t)2 * grid * 256 * 4); hipMalloc(&bad, 4); hipMalloc(&big, nbig * 4);
  hipMemset(buf, 0, (size_t)2 * grid * 256 * 4); hipMemset(bad, 0, 4); hipMemset(big, 0, nbig * 4);
  hipDeviceSynchronize();
  for (int i = 0; i < links; i++) {
    hipLaunchKernelGGL(link, dim3(grid), dim3(256), 0, a, buf + (size_t)(i & 1) * grid * 256, buf + (size_t)((i + 1) & 1) * grid * 256, i, 1 + i % 5, bad);
    if (other && (i % 4) == 0) hipLaunchKernelGGL(stream_other, dim3(1024), dim3(256), 0, b, big, nbig);
  }
  hipDeviceSynchronize();
  int h = -1;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  // the first link reads zeros and expects 0: fine
  hipFree(buf); hipFree(bad); hipFree(big);
  hipStreamDestroy(a); hipStreamDestroy(b);
  return h;
}
int main() {
  for (int grid : {40, 304, 2048})
    printf("grid %4d, 4000 links: alone %d wrong words | with a second stream busy %d wrong words\n", grid, run(false, grid, 4000), run(true, grid, 4000));
  return 0;
}
