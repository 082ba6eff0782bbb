// micro-test (developer tool): is data handed from one stream to another through hipEventRecord / hipStreamWaitEvent visible on every
// XCD?  A producer kernel rewrites a buffer each iteration; a consumer kernel on the other stream — whose workgroups read the lines that
// OTHER workgroups (other XCDs) read in the previous iteration, so that stale clean lines can sit in their L2 — verifies it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void produce(int* buf, int v) { buf[blockIdx.x * 256 + threadIdx.x] = v + (int)threadIdx.x; }
__global__ void consume(const int* buf, int v, int shift, int* bad) {
  const int b = (blockIdx.x + shift) % gridDim.x;
  if (buf[b * 256 + threadIdx.x] != v + (int)threadIdx.x) atomicAdd(bad, 1);
}
__global__ void acquire_all(int* sink) {
  if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); if (sink == (int*)1) *sink = 0; }
}
static int run(int mode, unsigned evflags, int iters, int grid) {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t ea[8], eb[8];
  for (int i = 0; i < 8; i++) { hipEventCreateWithFlags(&ea[i], evflags); hipEventCreateWithFlags(&eb[i], evflags); }
  int *buf, *bad;
  hipMalloc(&buf, (size_t)grid * 256 * 4); hipMalloc(&bad, 4);
  hipMemset(buf, 0, (size_t)grid * 256 * 4); hipMemset(bad, 0, 4);
  hipDeviceSynchronize();
  for (int i = 0; i < iters; i++) {
    hipStream_t cs = mode == 3 ? a : b;
    hipLaunchKernelGGL(produce, dim3(grid), dim3(256), 0, a, buf, i * 1000);
    if (mode != 3) { hipEventRecord(ea[i % 8], a); hipStreamWaitEvent(b, ea[i % 8], 0); }
    if (mode == 4) hipLaunchKernelGGL(acquire_all, dim3(16), dim3(64), 0, b, (int*)nullptr);
    hipLaunchKernelGGL(consume, dim3(grid), dim3(256), 0, cs, buf, i * 1000, 1 + (i % 7), bad);
    if (mode != 3) { hipEventRecord(eb[i % 8], b); hipStreamWaitEvent(a, eb[i % 8], 0); }
  }
  hipDeviceSynchronize();
  int h = -1;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  hipFree(buf); hipFree(bad);
  hipStreamDestroy(a); hipStreamDestroy(b);
  return h;
}
int main() {
  const int iters = 400;
  for (int grid : {64, 1024, 8192}) {
    printf("grid %5d: events(DisableTiming) %d stale words | events(default) %d | same stream %d | events + acquire kernel %d\n", grid,
           run(1, hipEventDisableTiming, iters, grid), run(1, hipEventDefault, iters, grid), run(3, hipEventDisableTiming, iters, grid),
           run(4, hipEventDisableTiming, iters, grid));
  }
  return 0;
}
