// micro-test (developer tool): does a kernel queued behind a hipStreamWaitEvent / hipEventRecord marker still wait for the kernels queued
// BEFORE the marker on the same stream?  K1 spins, then writes; [marker]; K2 reads what K1 wrote.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k1(int* p, int v, long long spin) {
  long long t0 = clock64();
  while (clock64() - t0 < spin) { }
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}
__global__ void k2(const int* p, int v, int* bad) { if (threadIdx.x == 0 && blockIdx.x == 0 && *p != v) atomicAdd(bad, 1); }
__global__ void tiny(int* q) { if (threadIdx.x == 0) *q = 1; }
static int run(int mode, unsigned evflags, int iters) {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t ev[16];
  for (int i = 0; i < 16; i++) hipEventCreateWithFlags(&ev[i], evflags);
  int *p, *bad, *q;
  hipMalloc(&p, 4); hipMalloc(&bad, 4); hipMalloc(&q, 4);
  hipMemset(p, 0, 4); hipMemset(bad, 0, 4);
  hipDeviceSynchronize();
  for (int i = 1; i <= iters; i++) {
    if (mode == 1 || mode == 3) {          // an event of the OTHER stream, complete (mode 1) or about to be (mode 3) when A reaches the wait
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, q);
      hipEventRecord(ev[i % 16], b);
      if (mode == 1) hipStreamSynchronize(b);
    }
    hipLaunchKernelGGL(k1, dim3(4), dim3(64), 0, a, p, i, 40000LL);     // ~20 us
    if (mode == 1 || mode == 3) hipStreamWaitEvent(a, ev[i % 16], 0);
    if (mode == 2) hipEventRecord(ev[i % 16], a);
    hipLaunchKernelGGL(k2, dim3(4), dim3(64), 0, a, p, i, bad);
  }
  hipDeviceSynchronize();
  int h = -1;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  hipStreamDestroy(a); hipStreamDestroy(b);
  return h;
}
int main() {
  const int n = 2000;
  printf("K1 -> K2, no marker: %d of %d wrong\n", run(0, hipEventDisableTiming, n), n);
  printf("K1 -> [wait for a COMPLETE event of another stream] -> K2: %d wrong (DisableTiming), %d (default flags)\n", run(1, hipEventDisableTiming, n), run(1, hipEventDefault, n));
  printf("K1 -> [wait for a pending event of another stream] -> K2: %d wrong (DisableTiming), %d (default flags)\n", run(3, hipEventDisableTiming, n), run(3, hipEventDefault, n));
  printf("K1 -> [hipEventRecord on this stream] -> K2: %d wrong (DisableTiming), %d (default flags)\n", run(2, hipEventDisableTiming, n), run(2, hipEventDefault, n));
  return 0;
}
