// Latency micro-benchmarks that size the tracker design on MI355X (results: profiles/r01_microbench.txt).
//   1. straight-line code executed once by one lane vs the same instruction count in a loop  -> cold instruction fetch cost
//   2. dependent global-load chain (L2-resident)                                             -> memory round trip
//   3. grid barrier through device-scope atomics, G workgroups                               -> persistent-kernel sync cost
//   4. empty kernel back-to-back launch cadence
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N>
__global__ void k_straight(float* out, float a, float b) {
  float x0 = out[0], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
#pragma unroll
  for (int i = 0; i < N / 4; i++) {
    x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
  }
  if (threadIdx.x == 0) out[1 + blockIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void k_loop(float* out, float a, float b, int n) {
  float x0 = out[0], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
#pragma unroll 1
  for (int i = 0; i < n / 32; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) { x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b; }
  }
  if (threadIdx.x == 0) out[1 + blockIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void k_chase(const int* __restrict__ next, int steps, int* out) {
  int p = threadIdx.x;
  for (int i = 0; i < steps; i++) p = next[p];
  out[threadIdx.x] = p;
}
__global__ void k_empty(int* p) { if (p == nullptr) __builtin_trap(); }

// grid barrier: every workgroup increments `count` (release), then spins until it reaches G * (round + 1) (acquire)
__global__ void k_gridbar(unsigned* count, int rounds, int G, unsigned* fail) {
  for (int r = 0; r < rounds; r++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned target = (unsigned)G * (unsigned)(r + 1);
      int spins = 0;
      while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { *fail = 1; break; }
      }
    }
    __syncthreads();
  }
}
// leader/follower hand-off: all arrive (atomic add), leader (wg 0) waits for all, bumps epoch; others wait for epoch
__global__ void k_leader(unsigned* count, unsigned* epoch, int rounds, int G, unsigned* fail) {
  for (int r = 0; r < rounds; r++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      int spins = 0;
      if (blockIdx.x == 0) {
        unsigned target = (unsigned)(G - 1) * (unsigned)(r + 1);
        while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000) { *fail = 1; break; }
        }
        __hip_atomic_store(epoch, (unsigned)(r + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000) { *fail = 1; break; }
        }
      }
    }
    __syncthreads();
  }
}

template <typename F>
static float time_launches(hipStream_t s, int reps, F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; i++) f();
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b, s);
  hipStreamSynchronize(s);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1000.f / reps;
}

int main() {
  hipStream_t s;
  CHK(hipStreamCreate(&s));
  float* d_out;
  CHK(hipMalloc(&d_out, 4096 * 4));
  CHK(hipMemset(d_out, 0, 4096 * 4));
  const int reps = 200;
  printf("== back-to-back launch cadence (us per launch, %d launches) ==\n", reps);
  printf("empty kernel 1 WG            : %.2f\n", time_launches(s, reps, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int*)d_out); }));
  printf("empty kernel 152 WG x 512    : %.2f\n", time_launches(s, reps, [&] { hipLaunchKernelGGL(k_empty, dim3(152), dim3(512), 0, s, (int*)d_out); }));
  {
    // the same empty kernel as a captured graph of 24 nodes
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 24; i++) hipLaunchKernelGGL(k_empty, dim3(152), dim3(256), 0, s, (int*)d_out);
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float t = time_launches(s, 50, [&] { (void)hipGraphLaunch(ge, s); });
    printf("graph of 24 empty kernels (152 WG x 256): %.2f us per graph launch = %.2f us per kernel\n", t, t / 24);
    float t1 = time_launches(s, 50, [&] { for (int i = 0; i < 24; i++) hipLaunchKernelGGL(k_empty, dim3(152), dim3(256), 0, s, (int*)d_out); });
    printf("24 plain launches                        : %.2f us = %.2f us per kernel\n", t1, t1 / 24);
  }
  printf("== straight-line code, 1 WG of 64 lanes, N dependent-ish FMAs (4 chains) ==\n");
#define ST(N) printf("straight N=%-6d 1 WG  : %.2f us   152 WG: %.2f us\n", N, \
    time_launches(s, reps, [&] { hipLaunchKernelGGL(k_straight<N>, dim3(1), dim3(64), 0, s, d_out, 1.0001f, 0.5f); }), \
    time_launches(s, reps, [&] { hipLaunchKernelGGL(k_straight<N>, dim3(152), dim3(64), 0, s, d_out, 1.0001f, 0.5f); }))
  ST(256); ST(1024); ST(4096); ST(8192); ST(16384);
  for (int n : {256, 1024, 4096, 8192, 16384})
    printf("loop     N=%-6d 1 WG  : %.2f us   152 WG: %.2f us\n", n,
           time_launches(s, reps, [&] { hipLaunchKernelGGL(k_loop, dim3(1), dim3(64), 0, s, d_out, 1.0001f, 0.5f, n); }),
           time_launches(s, reps, [&] { hipLaunchKernelGGL(k_loop, dim3(152), dim3(64), 0, s, d_out, 1.0001f, 0.5f, n); }));

  printf("== dependent global load chain (64 lanes, table of 16K ints) ==\n");
  {
    std::vector<int> h(16384);
    for (int i = 0; i < 16384; i++) h[i] = (i * 2654435761u + 12345u) % 16384;
    int *d_next, *d_o;
    CHK(hipMalloc(&d_next, 16384 * 4)); CHK(hipMalloc(&d_o, 256 * 4));
    CHK(hipMemcpy(d_next, h.data(), 16384 * 4, hipMemcpyHostToDevice));
    float t0 = time_launches(s, reps, [&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s, d_next, 0, d_o); });
    for (int st : {1, 4, 16, 64}) {
      float t = time_launches(s, reps, [&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s, d_next, st, d_o); });
      printf("chase steps=%-3d : %.2f us  (%.0f ns / step over the 0-step kernel %.2f us)\n", st, t, (t - t0) * 1000.f / st, t0);
    }
  }
  printf("== grid barriers through device-scope atomics (us per barrier) ==\n");
  {
    unsigned *d_cnt, *d_ep, *d_fail;
    CHK(hipMalloc(&d_cnt, 4)); CHK(hipMalloc(&d_ep, 4)); CHK(hipMalloc(&d_fail, 4));
    for (int G : {8, 64, 152, 256}) {
      const int rounds = 200;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      float base = 0, ms = 0;
      for (int pass = 0; pass < 2; pass++) {
        int R = pass == 0 ? 0 : rounds;
        CHK(hipMemsetAsync(d_cnt, 0, 4, s)); CHK(hipMemsetAsync(d_fail, 0, 4, s));
        hipEventRecord(a, s);
        hipLaunchKernelGGL(k_gridbar, dim3(G), dim3(256), 0, s, d_cnt, R, G, d_fail);
        hipEventRecord(b, s);
        CHK(hipStreamSynchronize(s));
        hipEventElapsedTime(pass == 0 ? &base : &ms, a, b);
      }
      unsigned fail = 0; CHK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
      float lms = 0;
      CHK(hipMemsetAsync(d_cnt, 0, 4, s)); CHK(hipMemsetAsync(d_ep, 0, 4, s)); CHK(hipMemsetAsync(d_fail, 0, 4, s));
      hipEventRecord(a, s);
      hipLaunchKernelGGL(k_leader, dim3(G), dim3(256), 0, s, d_cnt, d_ep, rounds, G, d_fail);
      hipEventRecord(b, s);
      CHK(hipStreamSynchronize(s));
      hipEventElapsedTime(&lms, a, b);
      unsigned fail2 = 0; CHK(hipMemcpy(&fail2, d_fail, 4, hipMemcpyDeviceToHost));
      printf("G=%-4d all-to-all barrier: %.2f us   arrive->leader->release: %.2f us   (fail flags %u %u)\n", G,
             (ms - base) * 1000.f / rounds, (lms - base) * 1000.f / rounds, fail, fail2);
      hipEventDestroy(a); hipEventDestroy(b);
    }
  }
  return 0;
}
