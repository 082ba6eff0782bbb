// Context + device-resident Frame: image / gradient / maxGradient / inverse-depth pyramids (K-pyr-*), point-cloud
// export (K0), ground-truth depth initialisation.  gfx950 only.
//
// Reference behaviour restated on the device:
//   Frame::Frame                 C/DataStructures/Frame.cpp:35-48    uint8 -> float copy (no scaling)
//   Frame::initialize            Frame.cpp:397-459                   per-level intrinsics
//   Frame::buildImage            Frame.cpp:491-630                   2x2 box mean, SSE association (:532-544)
//   Frame::buildGradients        Frame.cpp:643-680                   linear-index walk incl. row wrap
//   Frame::buildMaxGradients     Frame.cpp:690-767                   |grad| + separable 3x3 max (linear-index ranges)
//   Frame::buildIDepthAndIDepthVar Frame.cpp:775-877                 inverse-variance 2x2 pooling
//   Frame::setDepthFromGroundTruth Frame.cpp:245-293
//   TrackingReference::makePointCloud C/Tracking/TrackingReference.cpp:96-147
// Unwritten pool memory of the reference is defined as 0 here (rows 0 / h-1 of gradients etc.).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include "lsdhip_internal.hpp"

// ---------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------

// One 16x16 level-0 tile per workgroup -> 8x8, 4x4, 2x2, 1x1 tiles of levels 1..4, all in one launch.
// HBM traffic: 1 B/px read (uint8) + 4 B * (1 + 1/4 + 1/16 + 1/64 + 1/256) written.
// The same workgroup also writes the level-0 gradient texels (gx, gy, I, 0) and |grad| of its 16x16 pixels, straight from
// the uint8 source (the float image is its exact conversion), with the reference's linear-index neighbour rule.
__device__ __forceinline__ void image_pyramid_tile(const uint8_t* __restrict__ gray, float* __restrict__ i0,
                                                   float* __restrict__ i1, float* __restrict__ i2,
                                                   float* __restrict__ i3, float* __restrict__ i4, int w, int h,
                                                   float4* __restrict__ grad0, float* __restrict__ absgrad0) {
  __shared__ float s0[16][17];
  __shared__ float s1[8][9];
  __shared__ float s2[4][5];
  __shared__ float s3[2][3];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int bx = blockIdx.x, by = blockIdx.y;
  {
    int x = bx * 16 + tx, y = by * 16 + ty;
    const int i = y * w + x;
    float v = (float)gray[i];
    i0[i] = v;
    s0[ty][tx] = v;
    // Frame::buildGradients (Frame.cpp:643-680): rows 1..h-2 by linear index, so x = 0 / w-1 wrap into the neighbouring rows
    // (grad0 == nullptr — every frame since round 6: the level-0 gradients and maxGradients are keyframe planes, built when a frame
    // becomes one: lsd_frames_require_level0 below)
    if (grad0) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool inner = (i >= w) && (i < w * (h - 1));
      if (inner) {
        g.x = 0.5f * ((float)gray[i + 1] - (float)gray[i - 1]);
        g.y = 0.5f * ((float)gray[i + w] - (float)gray[i - w]);
        g.z = v;
      }
      grad0[i] = g;
      absgrad0[i] = inner ? sqrtf(g.x * g.x + g.y * g.y) : 0.f;
    }
  }
  __syncthreads();
  if (tid < 64) {
    int ax = tid & 7, ay = tid >> 3;
    float c0 = s0[2 * ay][2 * ax] + s0[2 * ay + 1][2 * ax];          // (top + bottom) of the left column
    float c1 = s0[2 * ay][2 * ax + 1] + s0[2 * ay + 1][2 * ax + 1];  // right column
    float r = (c0 + c1) * 0.25f;
    s1[ay][ax] = r;
    i1[(by * 8 + ay) * (w >> 1) + bx * 8 + ax] = r;
  }
  __syncthreads();
  if (tid < 16) {
    int ax = tid & 3, ay = tid >> 2;
    float c0 = s1[2 * ay][2 * ax] + s1[2 * ay + 1][2 * ax];
    float c1 = s1[2 * ay][2 * ax + 1] + s1[2 * ay + 1][2 * ax + 1];
    float r = (c0 + c1) * 0.25f;
    s2[ay][ax] = r;
    i2[(by * 4 + ay) * (w >> 2) + bx * 4 + ax] = r;
  }
  __syncthreads();
  if (tid < 4) {
    int ax = tid & 1, ay = tid >> 1;
    float c0 = s2[2 * ay][2 * ax] + s2[2 * ay + 1][2 * ax];
    float c1 = s2[2 * ay][2 * ax + 1] + s2[2 * ay + 1][2 * ax + 1];
    float r = (c0 + c1) * 0.25f;
    s3[ay][ax] = r;
    i3[(by * 2 + ay) * (w >> 3) + bx * 2 + ax] = r;
  }
  __syncthreads();
  if (tid == 0) {
    float c0 = s3[0][0] + s3[1][0];
    float c1 = s3[0][1] + s3[1][1];
    i4[by * (w >> 4) + bx] = (c0 + c1) * 0.25f;
  }
}
__global__ __launch_bounds__(256) void k_image_pyramid(const uint8_t* __restrict__ gray, float* __restrict__ i0,
                                                        float* __restrict__ i1, float* __restrict__ i2,
                                                        float* __restrict__ i3, float* __restrict__ i4, int w, int h,
                                                        float4* __restrict__ grad0, float* __restrict__ absgrad0) {
  image_pyramid_tile(gray, i0, i1, i2, i3, i4, w, h, grad0, absgrad0);
}
// the same for the new frames of several sequences in one launch (blockIdx.z = frame; lsdhip_frame_create_batch)
struct ImagePyrItem {
  LSD_G const uint8_t* gray;
  LSD_G float* img[LSD_LEVELS];
  LSD_G float4* grad0;
  LSD_G float* absgrad0;
};
__global__ __launch_bounds__(256) void k_image_pyramid_batch(const ImagePyrItem* __restrict__ items, int w, int h) {
  const ImagePyrItem it = items[blockIdx.z];
  image_pyramid_tile(it.gray, it.img[0], it.img[1], it.img[2], it.img[3], it.img[4], w, h, it.grad0, it.absgrad0);
}

struct GradArgs {
  const float* img[LSD_LEVELS];
  float4* grad[LSD_LEVELS];
  float* absgrad0;
  int w[LSD_LEVELS], h[LSD_LEVELS];
};

// Second launch of a new frame: the gradient texels of levels 1..4 (4 B/px read + neighbours from L2, 16 B/px written); the level-1
// blocks also leave the frame's refPixelWasGood mask in its "never written" state (0xFF), so that the tracker needs no separate fill
// before its first use.  (Until round 6 the launch carried the level-0 maxGradients as extra blocks — hence the name; that plane is a
// keyframe plane now: k_maxgrad_candidates.)
struct GradMaxArgs {
  LSD_G const float* img[LSD_LEVELS];
  LSD_G float4* grad[LSD_LEVELS];
  int w[LSD_LEVELS], h[LSD_LEVELS];
  int blk0[LSD_LEVELS + 1];     // first block of level l's gradient range (levels 1..4); blk0[LSD_LEVELS] = gradBlocks
  LSD_G uint32_t* wasGoodWords;
  int nMaskWords;
};
// Frame::buildMaxGradients (Frame.cpp:682-747) for one pixel: separable 3x3 maximum of |grad| with the reference's linear-index validity
// ranges (the vertical pass covers [w + 1, w (h - 1) - 1), the horizontal pass the same range; the two border values it copies)
__device__ __forceinline__ float max_gradient_at(LSD_G const float* __restrict__ absg, const int w, const int h, const int i) {
  const int lo = w + 1, hi = w * (h - 1) - 1;
  auto vmax = [&](int j) -> float {
    if (j < lo || j >= hi) return 0.f;
    float g1 = absg[j - w];
    float g2 = absg[j];
    if (g1 < g2) g1 = g2;
    float g3 = absg[j + w];
    return (g1 < g3) ? g3 : g1;
  };
  float out = 0.f;
  if (i >= lo && i < hi) {
    float g1 = vmax(i - 1);
    float g2 = vmax(i);
    if (g1 < g2) g1 = g2;
    float g3 = vmax(i + 1);
    out = (g1 < g3) ? g3 : g1;
  } else if (i == w || i == hi) {
    out = absg[i];
  }
  return out;
}
__device__ __forceinline__ void gradients_max_block(const GradMaxArgs& a) {
  const int b = blockIdx.x;
  if (b < a.blk0[LSD_LEVELS]) {
    int l = 1;
#pragma unroll
    for (int k = 2; k < LSD_LEVELS; k++) if (b >= a.blk0[k]) l = k;
    const int w = a.w[l], h = a.h[l];
    const int i = (b - a.blk0[l]) * 256 + threadIdx.x;
    if (l == 1 && i < a.nMaskWords) a.wasGoodWords[i] = 0xFFFFFFFFu;   // the frame's level-1 refPixelWasGood in its "never written" state
    if (i >= w * h) return;
    const float* __restrict__ img = a.img[l];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool inner = (i >= w) && (i < w * (h - 1));
    if (inner) {
      g.x = 0.5f * (img[i + 1] - img[i - 1]);
      g.y = 0.5f * (img[i + w] - img[i - w]);
      g.z = img[i];
    }
    a.grad[l][i] = g;
  }
}
__global__ __launch_bounds__(256) void k_gradients_max(GradMaxArgs a) { gradients_max_block(a); }
__global__ __launch_bounds__(256) void k_gradients_max_batch(const GradMaxArgs* __restrict__ items) {
  __shared__ GradMaxArgs s_a;     // (a by-value copy of an indexed record would live in scratch: the level tables are indexed dynamically)
  const int* src = (const int*)(items + blockIdx.y);
  for (int i = threadIdx.x; i < (int)(sizeof(GradMaxArgs) / 4); i += 256) ((int*)&s_a)[i] = src[i];
  __syncthreads();
  gradients_max_block(s_a);
}

// Level-0 gradients (gx, gy, I, 0) and |grad| from the level-0 float image (the exact conversion of the 8-bit source): the arithmetic the
// image pyramid kernel did for every frame until round 6 (Frame::buildGradients, Frame.cpp:643-680, rows 1..h-2 by linear index).
// The reference builds Frame::gradients(0) and maxGradients(0) on demand, and only keyframes are ever asked for them (DepthMap's
// observe / regularise / propagate, setDepthFromGroundTruth); a tracked frame needs the levels >= 1.  So do we now: 7.3 of the 12.6 MB a
// 640x480 frame's pyramids moved were these two planes and the |grad| scratch.
struct Level0Item {
  LSD_G const float* img0;
  LSD_G float4* grad0;
  LSD_G float* absgrad;
};
__device__ __forceinline__ void level0_gradients_px(const Level0Item& a, const int w, const int h) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const float v = a.img0[i];
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool inner = (i >= w) && (i < w * (h - 1));
  if (inner) {
    g.x = 0.5f * (a.img0[i + 1] - a.img0[i - 1]);
    g.y = 0.5f * (a.img0[i + w] - a.img0[i - w]);
    g.z = v;
  }
  a.grad0[i] = g;
  a.absgrad[i] = inner ? sqrtf(g.x * g.x + g.y * g.y) : 0.f;
}
__global__ __launch_bounds__(256) void k_level0_gradients(Level0Item a, int w, int h) { level0_gradients_px(a, w, h); }
__global__ __launch_bounds__(256) void k_level0_gradients_batch(const Level0Item* __restrict__ items, int w, int h) {
  const Level0Item a = items[blockIdx.y];
  level0_gradients_px(a, w, h);
}

// Gradient candidates of a keyframe (lsdhip_frame::d_gradCand): one workgroup per group of 1024 consecutive pixels, four per lane; the
// offsets of the group's candidates — inside the 3-pixel border, !(maxGradients < minUseGrad): the two tests of observeDepthRow that
// depend on nothing but the keyframe (DepthMap.cpp:111-131) — compacted in pixel order, and their number.
struct GradCandItem {
  LSD_G const float* maxgrad;
  LSD_G uint16_t* cand;
};
__device__ __forceinline__ void grad_cand_group(const GradCandItem& a, const int w, const int h, const float th) {
  __shared__ int s_w[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = w * h, g = blockIdx.x;
  const int i0 = g * 1024 + tid * 4;
  unsigned m = 0;
  int y = i0 / w, x = i0 - y * w;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = i0 + k;
    if (i < n0 && !(x < 3 || x >= w - 3 || y < 3 || y >= h - 3) && !(a.maxgrad[i] < th)) m |= 1u << k;
    if (++x >= w) { x = 0; y++; }
  }
  const int cnt = __popc(m);
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int pos = incl - cnt;
#pragma unroll
  for (int q = 0; q < 4; q++) pos += q < wave ? s_w[q] : 0;
  LSD_G uint16_t* out = a.cand + (size_t)g * 1024;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if ((m >> k) & 1u) out[pos++] = (uint16_t)(tid * 4 + k);
  if (tid == 0) a.cand[(size_t)((n0 + 1023) >> 10) * 1024 + g] = (uint16_t)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}
__global__ __launch_bounds__(256) void k_grad_candidates(GradCandItem a, int w, int h, float th) { grad_cand_group(a, w, h, th); }
__global__ __launch_bounds__(256) void k_grad_candidates_batch(const GradCandItem* __restrict__ items, int w, int h, float th) {
  const GradCandItem a = items[blockIdx.y];
  grad_cand_group(a, w, h, th);
}

// maxGradients(0) and the gradient candidates in one launch (the keyframe planes' second launch): one workgroup of 1024 lanes per group of
// 1024 consecutive pixels, one pixel per lane.
struct MaxCandItem {
  LSD_G const float* absg;
  LSD_G float* maxgrad;
  LSD_G uint16_t* cand;
};
__device__ __forceinline__ void maxgrad_cand_group(const MaxCandItem& a, const int w, const int h, const float th) {
  __shared__ int s_w[16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = w * h, g = blockIdx.x;
  const int i = g * 1024 + tid;
  bool cand = false;
  if (i < n0) {
    const float mg = max_gradient_at(a.absg, w, h, i);
    a.maxgrad[i] = mg;
    const int y = i / w, x = i - y * w;
    cand = !(x < 3 || x >= w - 3 || y < 3 || y >= h - 3) && !(mg < th);
  }
  const unsigned long long bal = __ballot(cand);
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)), total = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) { pos += q < wave ? s_w[q] : 0; total += s_w[q]; }
  if (cand) a.cand[(size_t)g * 1024 + pos] = (uint16_t)tid;
  if (tid == 0) a.cand[(size_t)((n0 + 1023) >> 10) * 1024 + g] = (uint16_t)total;
}
__global__ __launch_bounds__(1024) void k_maxgrad_candidates(MaxCandItem a, int w, int h, float th) { maxgrad_cand_group(a, w, h, th); }
__global__ __launch_bounds__(1024) void k_maxgrad_candidates_batch(const MaxCandItem* __restrict__ items, int w, int h, float th) {
  const MaxCandItem a = items[blockIdx.y];
  maxgrad_cand_group(a, w, h, th);
}

// inverse-variance pooling of one 2x2 block, children in the order idx, idx+1, idx+sw, idx+sw+1
__device__ __forceinline__ void pool4(const float id[4], const float var[4], float& oid, float& ovar) {
  float idepthSumsSum = 0.f, ivarSumsSum = 0.f;
  int num = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (var[k] > 0) {
      float ivar = lsd_rcp_exact(var[k]);
      ivarSumsSum += ivar;
      idepthSumsSum += ivar * id[k];
      num++;
    }
  }
  if (num > 0) {
    float depth = ivarSumsSum / idepthSumsSum;
    oid = lsd_rcp_exact(depth);
    ovar = num / ivarSumsSum;
  } else {
    oid = -1.f;
    ovar = -1.f;
  }
}

struct DepthPyrArgs {
  LSD_G float* id[LSD_LEVELS];
  LSD_G float* var[LSD_LEVELS];
  int w0, h0;
  // optional passenger: one extra workgroup (blockIdx.y == gridDim.y - 1, blockIdx.x == 0 of an extra grid row) folds the
  // (sum, count) partials of the setDepth that produced level 0 into a pinned record (Frame::setDepth's meanIdepth / numPoints)
  LSD_G const double* redPartials;
  int redN;
  LSD_G double* redOut;
  LSD_G uint8_t* blk[LSD_LEVELS];   // reference blocks of levels >= 1 (k_ref_blocks)
};

// Levels 1..4 of (idepth, idepthVar) from level 0, one 32x32 level-0 tile per workgroup: every lane pools one 2x2 block of level 0
// straight from HBM (two 8-byte loads per plane) into its level-1 pixel, the coarser levels follow through LDS (16x16 -> 8x8 -> 4x4 ->
// 2x2).  (Rounds 1-4: 16x16 tiles, one level-0 pixel per lane — a quarter of the bytes per workgroup and four barriers for them: 41 %
// of the HBM peak with 32 maps per launch.)  Image sizes are multiples of 16, so a tile may hang over the right / lower edge by half.
__device__ __forceinline__ void idepth_pyramid_tile(const DepthPyrArgs& a) {
  __shared__ float tid1[16][17], tvar1[16][17];
  __shared__ float tid2[8][9], tvar2[8][9];
  __shared__ float tid3[4][5], tvar3[4][5];
  const int tid = threadIdx.x;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int w0 = a.w0, h0 = a.h0;
  if (a.redPartials && by == (int)gridDim.y - 1) {
    if (bx != 0) return;
    __shared__ double s_a[256], s_b[256];
    double sa = 0, sb = 0;
    for (int i = tid; i < a.redN; i += 256) { sa += a.redPartials[2 * i]; sb += a.redPartials[2 * i + 1]; }
    s_a[tid] = sa;
    s_b[tid] = sb;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) { s_a[tid] += s_a[tid + off]; s_b[tid] += s_b[tid + off]; }
      __syncthreads();
    }
    if (tid == 0) { a.redOut[0] = s_a[0]; a.redOut[1] = s_b[0]; a.redOut[2] = 0.0; }
    return;
  }
  {
    const int ax = tid & 15, ay = tid >> 4;
    const int x = bx * 32 + 2 * ax, y = by * 32 + 2 * ay;         // the block's upper left level-0 pixel
    float oi = -1.f, ov = -1.f;
    if (x < w0 && y < h0) {
      const float2 i0 = *(const float2*)(a.id[0] + (size_t)y * w0 + x), i1 = *(const float2*)(a.id[0] + (size_t)(y + 1) * w0 + x);
      const float2 v0 = *(const float2*)(a.var[0] + (size_t)y * w0 + x), v1 = *(const float2*)(a.var[0] + (size_t)(y + 1) * w0 + x);
      const float i4[4] = {i0.x, i0.y, i1.x, i1.y};
      const float v4[4] = {v0.x, v0.y, v1.x, v1.y};
      pool4(i4, v4, oi, ov);
      const int o = (by * 16 + ay) * (w0 >> 1) + bx * 16 + ax;
      a.id[1][o] = oi; a.var[1][o] = ov;
    }
    tid1[ay][ax] = oi; tvar1[ay][ax] = ov;
  }
  __syncthreads();
  if (tid < 64) {
    const int ax = tid & 7, ay = tid >> 3;
    float i4[4] = {tid1[2 * ay][2 * ax], tid1[2 * ay][2 * ax + 1], tid1[2 * ay + 1][2 * ax], tid1[2 * ay + 1][2 * ax + 1]};
    float v4[4] = {tvar1[2 * ay][2 * ax], tvar1[2 * ay][2 * ax + 1], tvar1[2 * ay + 1][2 * ax], tvar1[2 * ay + 1][2 * ax + 1]};
    float oi, ov;
    pool4(i4, v4, oi, ov);
    tid2[ay][ax] = oi; tvar2[ay][ax] = ov;
    if (bx * 8 + ax < (w0 >> 2) && by * 8 + ay < (h0 >> 2)) {
      const int o = (by * 8 + ay) * (w0 >> 2) + bx * 8 + ax;
      a.id[2][o] = oi; a.var[2][o] = ov;
    }
  }
  __syncthreads();
  if (tid < 16) {
    const int ax = tid & 3, ay = tid >> 2;
    float i4[4] = {tid2[2 * ay][2 * ax], tid2[2 * ay][2 * ax + 1], tid2[2 * ay + 1][2 * ax], tid2[2 * ay + 1][2 * ax + 1]};
    float v4[4] = {tvar2[2 * ay][2 * ax], tvar2[2 * ay][2 * ax + 1], tvar2[2 * ay + 1][2 * ax], tvar2[2 * ay + 1][2 * ax + 1]};
    float oi, ov;
    pool4(i4, v4, oi, ov);
    tid3[ay][ax] = oi; tvar3[ay][ax] = ov;
    if (bx * 4 + ax < (w0 >> 3) && by * 4 + ay < (h0 >> 3)) {
      const int o = (by * 4 + ay) * (w0 >> 3) + bx * 4 + ax;
      a.id[3][o] = oi; a.var[3][o] = ov;
    }
  }
  __syncthreads();
  if (tid < 4) {
    const int ax = tid & 1, ay = tid >> 1;
    float i4[4] = {tid3[2 * ay][2 * ax], tid3[2 * ay][2 * ax + 1], tid3[2 * ay + 1][2 * ax], tid3[2 * ay + 1][2 * ax + 1]};
    float v4[4] = {tvar3[2 * ay][2 * ax], tvar3[2 * ay][2 * ax + 1], tvar3[2 * ay + 1][2 * ax], tvar3[2 * ay + 1][2 * ax + 1]};
    float oi, ov;
    pool4(i4, v4, oi, ov);
    if (bx * 2 + ax < (w0 >> 4) && by * 2 + ay < (h0 >> 4)) {
      const int o = (by * 2 + ay) * (w0 >> 4) + bx * 2 + ax;
      a.id[4][o] = oi; a.var[4][o] = ov;
    }
  }
}
__global__ __launch_bounds__(256) void k_idepth_pyramid(DepthPyrArgs a) { idepth_pyramid_tile(a); }
__global__ __launch_bounds__(256) void k_idepth_pyramid_batch(const DepthPyrArgs* __restrict__ items) {
  const DepthPyrArgs a = items[blockIdx.z];
  idepth_pyramid_tile(a);
}

// Reference blocks of the idepth pyramid's levels >= 1: which pixels of a level are reference points of a tracking job — inside the
// one-pixel border, idepthVar > 0, idepth != 0: the test of TrackingReference::makePointCloud (C/Tracking/TrackingReference.cpp:120-131) —
// depends on the keyframe's planes alone, and a throughput-mode batch asks it of every pixel in every evaluation (8 bytes per pixel, four
// ballots per 1024 pixels, a barrier).  Answered once per pyramid instead: one wave per block of 256 consecutive pixels leaves the
// in-block offsets of the block's valid pixels, in pixel order, in the block's 256 bytes (slot-interleaved, below), and the count in the table behind the blocks;
// strips are multiples of 256 pixels, so a strip's list is the concatenation of its blocks' lists — no scan over the level.
__device__ __forceinline__ void ref_blocks_tile(const DepthPyrArgs& a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int gb = blockIdx.x * 4 + wave;
  int l = 1, w = a.w0 >> 1, h = a.h0 >> 1;
  for (; l < LSD_LEVELS; l++) {
    const int nb = (w * h + 255) >> 8;
    if (gb < nb) break;
    gb -= nb;
    w >>= 1; h >>= 1;
  }
  if (l >= LSD_LEVELS) return;
  const int work = w * h, nblk = (work + 255) >> 8;
  const int i0 = gb * 256 + lane * 4;
  LSD_G const float* id = nullptr;
  LSD_G const float* var = nullptr;
  LSD_G uint8_t* blk = nullptr;
#pragma unroll
  for (int q = 1; q < LSD_LEVELS; q++)          // (no dynamic index into the argument record: it would move to scratch memory)
    if (q == l) { id = a.id[q]; var = a.var[q]; blk = a.blk[q]; }
  unsigned m = 0;
  int y = i0 / w, x = i0 - y * w;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = i0 + k;
    if (i < work) {
      const float vv = var[i], dd = id[i];
      const bool ok = !(x < 1 || x >= w - 1 || y < 1 || y >= h - 1) && !(vv <= 0 || dd == 0);
      m |= (ok ? 1u : 0u) << k;
    }
    if (++x >= w) { x = 0; y++; }
  }
  const int cnt = __popc(m);
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  LSD_G uint8_t* out = blk + (size_t)gb * 256;
  // slot s of the block's list lives in byte (s mod 64) * 4 + s / 64: the reader's lane l takes one 4-byte word = slots l, l + 64, l + 128,
  // l + 192, so the lanes of a wave fill consecutive list entries (no LDS bank conflicts)
  int pos = incl - cnt;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if ((m >> k) & 1u) { out[((pos & 63) << 2) + (pos >> 6)] = (uint8_t)(lane * 4 + k); pos++; }
  if (lane == 63) ((LSD_G int*)(blk + (size_t)nblk * 256))[gb] = incl;
}
__global__ __launch_bounds__(256) void k_ref_blocks(DepthPyrArgs a) { ref_blocks_tile(a); }
__global__ __launch_bounds__(256) void k_ref_blocks_batch(const DepthPyrArgs* __restrict__ items) {
  const DepthPyrArgs a = items[blockIdx.z];
  ref_blocks_tile(a);
}
static int lsd_refblk_grid(const lsdhip_ctx* c) {
  int nb = 0;
  for (int l = 1; l < LSD_LEVELS; l++) nb += lsd_refblk_blocks(c->wl[l] * c->hl[l]);
  return (nb + 3) / 4;
}

// Frame::setDepthFromGroundTruth
__global__ __launch_bounds__(256) void k_set_depth_gt(const float* __restrict__ depth, const float* __restrict__ maxgrad,
                                                       float* __restrict__ id, float* __restrict__ var, int w, int h,
                                                       float cov_scale, float minUseGrad) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  int x = i % w, y = i / w;
  float d = depth[i];
  if (x > 0 && x < w - 1 && y > 0 && y < h - 1 && maxgrad[i] >= minUseGrad && !isnan(d) && d > 0) {
    id[i] = 1.0f / d;
    var[i] = 0.01f * 0.01f * cov_scale;  // VAR_GT_INIT_INITIAL * cov_scale (C/util/settings.h:75)
  } else {
    id[i] = -1.f;
    var[i] = -1.f;
  }
}

// ---- K0: TrackingReference::makePointCloud, x outer / y inner, order preserving ------------------------------
__global__ __launch_bounds__(256) void k_pc_count(const float* __restrict__ id, const float* __restrict__ var, int w, int h,
                                                   int* __restrict__ colCount) {
  int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= w) return;
  int c = 0;
  if (x >= 1 && x < w - 1)
    for (int y = 1; y < h - 1; y++) {
      int idx = x + y * w;
      if (!(var[idx] <= 0 || id[idx] == 0)) c++;
    }
  colCount[x] = c;
}
__global__ void k_pc_scan(int* __restrict__ colCount, int w, int* __restrict__ total) {
  // single thread exclusive scan over <= a few thousand columns (keyframe-rate export path, not the hot path)
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int x = 0; x < w; x++) { int c = colCount[x]; colCount[x] = acc; acc += c; }
    *total = acc;
  }
}
__global__ __launch_bounds__(256) void k_pc_write(const float* __restrict__ id, const float* __restrict__ var,
                                                   const float* __restrict__ img, const float4* __restrict__ grad, int w, int h,
                                                   float fxi, float fyi, float cxi, float cyi, const int* __restrict__ colOff,
                                                   float* __restrict__ pos, float* __restrict__ colvar,
                                                   float* __restrict__ gradOut, int* __restrict__ idxOut) {
  int x = blockIdx.x * 256 + threadIdx.x;
  if (x < 1 || x >= w - 1) return;
  int n = colOff[x];
  for (int y = 1; y < h - 1; y++) {
    int idx = x + y * w;
    float v = var[idx], d = id[idx];
    if (v <= 0 || d == 0) continue;
    float inv = 1.0f / d;
    pos[3 * n + 0] = inv * (fxi * x + cxi);
    pos[3 * n + 1] = inv * (fyi * y + cyi);
    pos[3 * n + 2] = inv * 1.0f;
    float4 g = grad[idx];
    gradOut[2 * n + 0] = g.x;
    gradOut[2 * n + 1] = g.y;
    colvar[2 * n + 0] = img[idx];
    colvar[2 * n + 1] = v;
    idxOut[n] = idx;
    n++;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------
#define LSD_FLAG_WGS 16
#ifdef LSD_DEVTOOLS
static void trace_dump(lsdhip_ctx* c);
#endif
extern "C" void lsdhip_default_params(lsdhip_params* p) {
  p->minUseGrad = 5;
  p->cameraPixelNoise2 = 4 * 4;
  p->depthSmoothingFactor = 1;
  p->allowNegativeIdepths = 1;
  p->useSubpixelStereo = 1;
  p->useAffineLightningEstimation = 1;
}

extern "C" int lsdhip_ctx_create(int device, int w, int h, const float K[4], const lsdhip_params* params, lsdhip_ctx** out) {
  if (!out || !K || w <= 0 || h <= 0 || (w % 16) != 0 || (h % 16) != 0) {
    lsd_set_error("lsdhip_ctx_create: image dimensions must be positive multiples of 16 (got %dx%d)", w, h);
    return LSDHIP_E_ARG;
  }
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    lsd_set_error("lsdhip_ctx_create: device %d not available (%d visible)", device, ndev);
    return LSDHIP_E_HIP;
  }
  HIPCHK(hipSetDevice(device));
  lsdhip_ctx* c = new lsdhip_ctx();
  LSD_CTX_LOCK(c);
  c->device = device;
  c->w = w;
  c->h = h;
  if (params) c->params = *params; else lsdhip_default_params(&c->params);
  // Frame::initialize (Frame.cpp:397-459)
  float fx[LSD_LEVELS], fy[LSD_LEVELS], cx[LSD_LEVELS], cy[LSD_LEVELS];
  fx[0] = K[0]; fy[0] = K[1]; cx[0] = K[2]; cy[0] = K[3];
  for (int l = 0; l < LSD_LEVELS; l++) {
    c->wl[l] = w >> l;
    c->hl[l] = h >> l;
    if (l > 0) {
      fx[l] = fx[l - 1] * 0.5;
      fy[l] = fy[l - 1] * 0.5;
      cx[l] = (cx[0] + 0.5) / ((int)1 << l) - 0.5;
      cy[l] = (cy[0] + 0.5) / ((int)1 << l) - 0.5;
    }
    float Kl[9] = {fx[l], 0.f, cx[l], 0.f, fy[l], cy[l], 0.f, 0.f, 1.f};
    float Ki[9];
    lsdm::inverse3_eigen(Kl, Ki);
    c->intr[l] = {fx[l], fy[l], cx[l], cy[l], Ki[0], Ki[4], Ki[2], Ki[5]};
    if (l == 0) { memcpy(c->K0, Kl, sizeof(Kl)); memcpy(c->K0inv, Ki, sizeof(Ki)); }
  }
  // (the tracking stream at the highest queue priority: 36.3 k against 37.1 k frames/s at S = 32, profiles/r05_notes.md — not kept)
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&c->ev_a));
  HIPCHK(hipEventCreate(&c->ev_b));
  HIPCHK(hipHostMalloc((void**)&c->h_slots, LSD_NUM_SLOTS * sizeof(DeferredSlot), hipHostMallocMapped));
  memset(c->h_slots, 0, LSD_NUM_SLOTS * sizeof(DeferredSlot));
  *out = c;
  return LSDHIP_OK;
}
// Developer instrumentation (LSDHIP_HOST_TRACE=1): host-side time between consecutive marks of the calling thread, summed per mark
// id and printed when a context is destroyed.  Costs one steady_clock read per mark when on, one branch when off.
#ifdef LSD_DEVTOOLS
static const bool g_hostTraceOn = getenv("LSDHIP_HOST_TRACE") != nullptr;
static long long g_htNs[32], g_htN[32], g_htHist[32][9];
static thread_local long long g_htLast = 0;
extern "C" void lsdhip_host_mark(int k) {
  if (!g_hostTraceOn) return;
  const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (k > 0 && k < 32 && g_htLast) {
    const long long d = now - g_htLast;
    g_htNs[k] += d; g_htN[k]++;
    static const long long edges[8] = {2000, 5000, 10000, 20000, 50000, 100000, 200000, 500000};
    int b = 0;
    while (b < 8 && d >= edges[b]) b++;
    g_htHist[k][b]++;
  }
  g_htLast = now;
}
static void host_trace_print() {
  if (!g_hostTraceOn) return;
  for (int k = 0; k < 32; k++)
    if (g_htN[k]) {
      fprintf(stderr, "HOSTTRACE mark %2d: n %6lld  mean %7.2f us   <2 <5 <10 <20 <50 <100 <200 <500 >=500 us:", k, g_htN[k], g_htNs[k] / 1e3 / g_htN[k]);
      for (int b = 0; b < 9; b++) fprintf(stderr, " %lld", g_htHist[k][b]);
      fprintf(stderr, "\n");
    }
}
#else
extern "C" void lsdhip_host_mark(int) {}      // (the C++ loop of include/lsd_slam_hip.hpp marks its phases: a no-op in the default library)
static void host_trace_print() {}
#endif
extern "C" void lsdhip_ctx_destroy(lsdhip_ctx* c) {
  if (!c) return;
  host_trace_print();
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->mstream) (void)hipStreamSynchronize(c->mstream);
#ifdef LSD_DEVTOOLS
  trace_dump(c);
#endif
  if (c->d_sums) (void)hipFree(c->d_sums);
  if (c->mstream) {
    for (int i = 0; i < LSD_EVR; i++) if (c->mEv[i]) (void)hipEventDestroy(c->mEv[i]);
    (void)hipStreamDestroy(c->mstream);
  }
  for (int i = 0; i < lsdhip_ctx::MAX_LANES; i++) if (c->lanes[i]) { (void)hipStreamSynchronize(c->lanes[i]); (void)hipEventDestroy(c->lane_done[i]); (void)hipStreamDestroy(c->lanes[i]); }
  if (c->lane_fork) (void)hipEventDestroy(c->lane_fork);
  if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipEventDestroy(c->aux_fork); (void)hipEventDestroy(c->aux_done); (void)hipStreamDestroy(c->aux_stream); }
  for (void* a : c->free_arenas) (void)hipFree(a);
  for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
  if (c->h_slots) (void)hipHostFree(c->h_slots);
  if (c->args.h) (void)hipHostFree(c->args.h);
  if (c->args.d) (void)hipFree(c->args.d);
  for (hipEvent_t e : c->args.ev) if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < LSD_BPROF_SLOTS; i++) if (c->bprof[i].a) { (void)hipEventDestroy(c->bprof[i].a); (void)hipEventDestroy(c->bprof[i].b); }
  if (c->d_obsBatchAcc) (void)hipFree(c->d_obsBatchAcc);
  if (c->d_gtStage) (void)hipFree(c->d_gtStage);
  if (c->d_flagArrive) (void)hipFree(c->d_flagArrive);
  if (c->d_gate) (void)hipFree(c->d_gate);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  (void)hipStreamDestroy(c->stream);
  delete c;
}
extern "C" void* lsdhip_ctx_stream(lsdhip_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int lsdhip_ctx_alloc_dev(lsdhip_ctx* c, size_t bytes, void** out) {
  if (!c || !out || bytes == 0) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(out, bytes));
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_free_dev(lsdhip_ctx* c, void* p) {
  if (!c) return LSDHIP_E_ARG;
  if (!p) return LSDHIP_OK;
  HIPCHK(hipSetDevice(c->device));
  if (int rc = lsd_sync_all(c)) return rc;
  HIPCHK(hipFree(p));
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_synchronize(lsdhip_ctx* c) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  if (c) {
    for (int i = 0; i < LSD_NUM_SLOTS; i++) {   // pick up every deferred result
      if (c->slot_stats_owner[i]) { int rc = lsd_frame_resolve(c->slot_stats_owner[i]); if (rc) return rc; }
      if (c->slot_rescale_owner[i]) { int rc = lsd_frame_resolve(c->slot_rescale_owner[i]); if (rc) return rc; }
    }
  }
  if (!c) return LSDHIP_E_ARG;
  if (c->aux_stream) HIPCHK(hipStreamSynchronize(c->aux_stream));
  return lsd_sync_all(c);
}
// ---- pipelined contexts: tracking stream beside mapping stream ---------------------------------------------------------------------
extern "C" int lsdhip_ctx_set_pipeline(lsdhip_ctx* c, int on) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (int rc = lsd_sync_all(c)) return rc;
  if (on && !c->mstream) {
    HIPCHK(hipStreamCreateWithFlags(&c->mstream, hipStreamNonBlocking));
    for (int i = 0; i < LSD_EVR; i++) {
      HIPCHK(hipEventCreateWithFlags(&c->mEv[i], hipEventDisableTiming));
    }
  }
  if (!on && c->pipeline) {
    // back to one stream: everything is drained, so every recorded point counts as passed and waited for
    c->mDoneSeq = c->tWaitedM = c->mSeq;
  }
  c->pipeline = on != 0;
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_pipeline(lsdhip_ctx* c) { return c ? (c->pipeline ? 1 : 0) : LSDHIP_E_ARG; }
extern "C" void* lsdhip_ctx_map_stream(lsdhip_ctx* c) { return c ? (void*)lsd_map_stream(c) : nullptr; }
int lsd_sync_all(lsdhip_ctx* c) {
  if (c->pipeline && !c->pendingMerges.empty()) { if (int rc = lsd_flush_merges(c)) return rc; }
  for (int i = 0; i < c->lanes_open; i++) if (c->lane_used[i]) HIPCHK(hipStreamSynchronize(c->lanes[i]));   // (inside an open lane region)
  if (c->mstream) {
    HIPCHK(hipStreamSynchronize(c->mstream));
    c->mDoneSeq = c->mSeq;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return LSDHIP_OK;
}
// Tracking -> mapping ordering.  A mapping-stream operation that consumes a tracking job's results is queued by the host AFTER it has
// seen the job's `done` (pinned memory, written behind a system-scope fence by the finishing launch); what the operation reads — the
// frame's mask, and through the host the pose — was written by launches that had completed before the finishing launch started, and
// the launches still queued behind it touch only the tracker's own state.  So no event is recorded on the tracking stream: a record
// behind every job is a barrier packet in the queue the launch chain runs through (the first form of the pipeline had one, plus a
// hipStreamWaitEvent on the mapping stream; profiles/r04_notes.md).
int lsd_m_begin(lsdhip_ctx* c) {
  if (!c->pendingMerges.empty()) return lsd_flush_merges(c);
  return LSDHIP_OK;
}
#ifdef LSD_DEVTOOLS
// Experiment: kernels that touch nothing but their own buffer, queued on the mapping stream right when a tracking job starts.
// kind 1: memory streaming (32 MB read-modify-write), 2: LDS-heavy workgroups (9.6 KB each, like k_reg_fused), 3: ALU spin.
static const int g_pipeDummy = getenv("LSDHIP_PIPE_DUMMY") ? atoi(getenv("LSDHIP_PIPE_DUMMY")) : 0;
__global__ __launch_bounds__(256) void k_dummy_stream(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ __launch_bounds__(256) void k_dummy_lds(float* p) {
  __shared__ float s[2400];
  for (int i = threadIdx.x; i < 2400; i += 256) s[i] = (float)(i + blockIdx.x);
  __syncthreads();
  float acc = 0;
  for (int r = 0; r < 40; r++) for (int i = threadIdx.x; i < 2400; i += 256) acc += s[(i * 7 + r) % 2400];
  p[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
// kind 4: workgroup barriers, no LDS; kind 5: LDS, no barrier (64-lane workgroups, lane-private slots); kind 6: like 2 with 40 KB
__global__ __launch_bounds__(256) void k_dummy_barrier(float* p) {
  float acc = (float)threadIdx.x;
  for (int r = 0; r < 400; r++) { acc = acc * 1.0001f + 0.5f; __syncthreads(); }
  p[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_dummy_lds_nobar(float* p) {
  __shared__ float s[2400];
  float acc = 0;
  for (int r = 0; r < 60; r++) {
    for (int i = threadIdx.x; i < 2400; i += 64) s[i] = (float)(i + r);
    for (int i = threadIdx.x; i < 2400; i += 64) acc += s[i];
  }
  p[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dummy_lds_big(float* p) {
  __shared__ float s[10000];
  for (int i = threadIdx.x; i < 10000; i += 256) s[i] = (float)(i + blockIdx.x);
  __syncthreads();
  float acc = 0;
  for (int r = 0; r < 10; r++) for (int i = threadIdx.x; i < 10000; i += 256) acc += s[(i * 7 + r) % 10000];
  p[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_dummy_spin(float* p, long long spin) {
  long long t0 = clock64();
  float a = (float)threadIdx.x;
  while (clock64() - t0 < spin) a = a * 1.0001f + 0.5f;
  p[(size_t)blockIdx.x * 64 + threadIdx.x] = a;
}
int lsd_pipe_dummy(lsdhip_ctx* c) {
  if (!g_pipeDummy || !c->pipeline) return LSDHIP_OK;
  static float* buf = nullptr;
  const size_t n = 8u << 20;
  if (!buf) { HIPCHK(hipMalloc((void**)&buf, n * 4)); HIPCHK(hipMemset(buf, 0, n * 4)); HIPCHK(hipStreamSynchronize(nullptr)); }
  for (int rep = 0; rep < 4; rep++) {
    if (g_pipeDummy == 1) hipLaunchKernelGGL(k_dummy_stream, dim3(2048), dim3(256), 0, c->mstream, buf, n);
    else if (g_pipeDummy == 2) hipLaunchKernelGGL(k_dummy_lds, dim3(1200), dim3(256), 0, c->mstream, buf);
    else if (g_pipeDummy == 4) hipLaunchKernelGGL(k_dummy_barrier, dim3(1200), dim3(256), 0, c->mstream, buf);
    else if (g_pipeDummy == 5) hipLaunchKernelGGL(k_dummy_lds_nobar, dim3(4800), dim3(64), 0, c->mstream, buf);
    else if (g_pipeDummy == 6) hipLaunchKernelGGL(k_dummy_lds_big, dim3(1200), dim3(256), 0, c->mstream, buf);
    else hipLaunchKernelGGL(k_dummy_spin, dim3(4800), dim3(64), 0, c->mstream, buf, 20000LL);
  }
  return LSDHIP_OK;
}
#endif   // LSD_DEVTOOLS
long long lsd_m_record(lsdhip_ctx* c) {
  if (!c->pipeline) return 0;
  if (c->lanes_open) { c->lane_record_pending = true; return c->mSeq + 1; }   // recorded once, where the lanes join (lsdhip_ctx_lanes_end)
  const long long s = c->mSeq + 1;
  if (hipEventRecord(c->mEv[s % LSD_EVR], c->mstream) != hipSuccess) { lsd_set_error("hipEventRecord on the mapping stream failed"); return LSDHIP_E_HIP; }
  c->mSeq = s;
  return s;
}
int lsd_t_wait_m(lsdhip_ctx* c, long long seq) {
  if (!c->pipeline || seq <= c->tWaitedM || seq <= c->mDoneSeq) return LSDHIP_OK;
  if (seq > c->mSeq) {
    // a sequence number beyond the last record point: handed out inside an open lane region (recorded where the lanes join) — there is
    // no event to wait for yet
    if (c->lanes_open) { lsd_set_error("tracking job depends on mapping work of an open lane region (call lsdhip_ctx_lanes_end first)"); return LSDHIP_E_STATE; }
    seq = c->mSeq;
  }
  // the ring slot holds the event of `seq` or, once the ring has wrapped, of a later point of the in-order stream: either orders us
  HIPCHK(hipStreamWaitEvent(c->stream, c->mEv[seq % LSD_EVR], 0));
  c->tWaitedM = seq;
  return LSDHIP_OK;
}
bool lsd_m_done(lsdhip_ctx* c, long long seq) {
  if (seq <= c->mDoneSeq) return true;
  if (seq > c->mSeq) return false;
  if (hipEventQuery(c->mEv[seq % LSD_EVR]) != hipSuccess) return false;
  // (after a wrap the slot's event is a later point: then more than `seq` is done, which is still true of `seq`)
  if (c->mSeq - seq < LSD_EVR) c->mDoneSeq = seq;
  return true;
}
#ifdef LSD_DEVTOOLS
static const char* g_traceSums = getenv("LSDHIP_TRACE_SUMS");
#define LSD_TRACE_SLOTS 65536
__global__ __launch_bounds__(256) void k_trace_sum(const uint32_t* __restrict__ p, size_t nwords, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256)
    acc += (unsigned long long)p[i] * (unsigned long long)(i * 2654435761ull + 1ull);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
void lsd_trace_sum(lsdhip_ctx* c, hipStream_t s, int kind, int id, const void* p, size_t bytes) {
  if (!g_traceSums) return;
  if (!c->d_sums) { if (hipMalloc((void**)&c->d_sums, LSD_TRACE_SLOTS * 8) != hipSuccess) return; (void)hipMemset(c->d_sums, 0, LSD_TRACE_SLOTS * 8); (void)hipStreamSynchronize(nullptr); }
  const size_t slot = c->sums_meta.size() / 2;
  if (slot >= LSD_TRACE_SLOTS) return;
  c->sums_meta.push_back(kind); c->sums_meta.push_back(id);
  c->sums_host.push_back(0);
  const size_t nwords = bytes / 4;
  int grid = (int)((nwords + 255) / 256); if (grid > 64) grid = 64; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_trace_sum, dim3(grid), dim3(256), 0, s, (const uint32_t*)p, nwords, c->d_sums + slot);
}
void lsd_trace_val(lsdhip_ctx* c, int kind, int id, unsigned long long v) {
  if (!g_traceSums) return;
  if (c->sums_meta.size() / 2 >= LSD_TRACE_SLOTS) return;
  c->sums_meta.push_back(-kind); c->sums_meta.push_back(id);
  c->sums_host.push_back(v);
}
static void trace_dump(lsdhip_ctx* c) {
  if (!g_traceSums || c->sums_meta.empty()) return;
  const size_t n = c->sums_meta.size() / 2;
  std::vector<unsigned long long> h(n, 0);
  if (c->d_sums) (void)hipMemcpy(h.data(), c->d_sums, n * 8, hipMemcpyDeviceToHost);
  if (FILE* f = fopen(g_traceSums, "a")) {
    for (size_t i = 0; i < n; i++) {
      const int kind = c->sums_meta[2 * i];
      fprintf(f, "%d %d %016llx\n", kind < 0 ? -kind : kind, c->sums_meta[2 * i + 1], kind < 0 ? c->sums_host[i] : h[i]);
    }
    fclose(f);
  }
}
// Developer hook (LSDHIP_PIPE_GATE=1): forces the overlap the pipelined mode is about, whatever the host's pace — a DepthMap::updateKeyframe
// queued on the mapping stream holds (a one-lane spin, bounded) until the NEXT tracking job's launches are about to start, so that its
// kernels run exactly beside that job's first launches even when the caller is a slow Python loop (tools/pipe_overlap_debug.py).
static const bool g_pipeGate = getenv("LSDHIP_PIPE_GATE") != nullptr;
__global__ void k_gate_wait(const int* flag, int value) {
  if (threadIdx.x != 0) return;
  for (unsigned spins = 0; spins < (1u << 22); spins++) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value) return;
    __builtin_amdgcn_s_sleep(4);
  }
}
__global__ void k_gate_open(int* flag, int value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
int lsd_gate_wait(lsdhip_ctx* c) {
  if (!g_pipeGate || !c->pipeline) return LSDHIP_OK;
  if (!c->d_gate) { HIPCHK(hipMalloc((void**)&c->d_gate, 64)); HIPCHK(hipMemset(c->d_gate, 0, 64)); HIPCHK(hipStreamSynchronize(nullptr)); }
  c->gateWaited = c->gateSeq + 1;
  hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, c->mstream, c->d_gate, c->gateWaited);
  return LSDHIP_OK;
}
int lsd_gate_open(lsdhip_ctx* c) {
  if (!g_pipeGate || !c->pipeline) return LSDHIP_OK;
  if (!c->d_gate) { HIPCHK(hipMalloc((void**)&c->d_gate, 64)); HIPCHK(hipMemset(c->d_gate, 0, 64)); HIPCHK(hipStreamSynchronize(nullptr)); }
  c->gateSeq++;
  hipLaunchKernelGGL(k_gate_open, dim3(1), dim3(64), 0, c->stream, c->d_gate, c->gateSeq);
  return LSDHIP_OK;
}
#endif   // LSD_DEVTOOLS
// TrackingReference::importFrame (C/Tracking/TrackingReference.cpp:71-87) as the tracking side's hand-over point: the newest
// Frame::setDepth result of the mapping stream becomes what SE3Tracker jobs read.  A no-op on non-pipelined contexts.
int lsd_frame_publish_depth(lsdhip_frame* f) {
  if (!f->depthPending) return LSDHIP_OK;
  for (int l = 0; l < LSD_LEVELS; l++) { std::swap(f->d_idepth[l], f->d_idepthW[l]); std::swap(f->d_idepthVar[l], f->d_idepthVarW[l]); std::swap(f->d_refBlk[l], f->d_refBlkW[l]); }
  std::swap(f->refBlkValid, f->refBlkValidW);
  f->depthPending = false;
  f->depthSeq = f->depthPendingSeq;
  f->hasIDepth = true;
  f->depthVersion++;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_publish_depth(lsdhip_frame* f) {
  if (!f) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(f->ctx);
  return lsd_frame_publish_depth(f);
}
extern "C" int lsdhip_ctx_copy_dev(lsdhip_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || !dst || !src) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, lsd_transport_stream(c)));
  return LSDHIP_OK;
}
// ---- lanes: the DepthMap call chains of DIFFERENT depth maps are independent of each other; between lanes_begin and lanes_end the
// caller routes each map's calls to one of n extra streams (lane_select), so that the chains of several sequences (finalizeKeyFrame +
// createKeyFrame: ~18 small dependent launches each) run side by side instead of one after the other.  Everything queued on the
// context's stream before lanes_begin is visible to every lane; lanes_end orders the context's stream behind all lanes.
extern "C" int lsdhip_ctx_lanes_begin(lsdhip_ctx* c, int n) {
  if (!c || n < 1 || n > lsdhip_ctx::MAX_LANES) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  if (c->lanes_open) { lsd_set_error("lsdhip_ctx_lanes_begin: already open"); return LSDHIP_E_STATE; }
  const hipStream_t base = c->pipeline ? c->mstream : c->stream;     // lanes branch off the stream the DepthMap calls run on
  HIPCHK(hipSetDevice(c->device));
  if (!c->lane_fork) HIPCHK(hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
  for (int i = 0; i < n; i++) {
    if (!c->lanes[i]) {
      HIPCHK(hipStreamCreateWithFlags(&c->lanes[i], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->lane_done[i], hipEventDisableTiming));
    }
    c->lane_used[i] = false;
  }
  // a lane waits for the fork point only: whatever the lanes' calls read must be queued before it — also the merges of speculative
  // trials' refPixelWasGood planes that the tracking calls have noted but not queued yet (createKeyFrame reads the new keyframe's mask)
  if (c->pipeline && !c->pendingMerges.empty()) { if (int rcf = lsd_flush_merges(c)) return rcf; }
  HIPCHK(hipEventRecord(c->lane_fork, base));
  c->lanes_open = n;
  c->lane_cur = -1;
  c->lane_record_pending = false;
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_lane_select(lsdhip_ctx* c, int lane) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  if (!c->lanes_open || lane < -1 || lane >= c->lanes_open) { lsd_set_error("lsdhip_ctx_lane_select: lane %d of %d", lane, c->lanes_open); return LSDHIP_E_ARG; }
  if (lane >= 0 && !c->lane_used[lane]) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamWaitEvent(c->lanes[lane], c->lane_fork, 0));
    c->lane_used[lane] = true;
  }
  c->lane_cur = lane;
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_lanes_end(lsdhip_ctx* c) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  if (!c->lanes_open) return LSDHIP_OK;
  HIPCHK(hipSetDevice(c->device));
  c->lane_cur = -1;
  const hipStream_t base = c->pipeline ? c->mstream : c->stream;
  for (int i = 0; i < c->lanes_open; i++) {
    if (!c->lane_used[i]) continue;
    HIPCHK(hipEventRecord(c->lane_done[i], c->lanes[i]));
    HIPCHK(hipStreamWaitEvent(base, c->lane_done[i], 0));
  }
  c->lanes_open = 0;
  // pipelined contexts: the calls inside the region were all given the mapping-stream sequence number of THIS point (lsd_m_record), the
  // first one at which the mapping stream is behind every lane
  if (c->lane_record_pending) { c->lane_record_pending = false; if (lsd_m_record(c) < 0) return LSDHIP_E_HIP; }
  return LSDHIP_OK;
}
// ---- transport stream: exchange under compute (halo rows of the row-band loop travel while the interior rows are computed) -------
extern "C" int lsdhip_ctx_aux_begin(lsdhip_ctx* c) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (c->aux_active) { lsd_set_error("lsdhip_ctx_aux_begin: already between begin and end"); return LSDHIP_E_STATE; }
  if (!c->aux_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->aux_done, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(c->aux_fork, c->stream));
  HIPCHK(hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0));
  c->aux_active = true;
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_aux_end(lsdhip_ctx* c) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  if (!c->aux_active) { lsd_set_error("lsdhip_ctx_aux_end without lsdhip_ctx_aux_begin"); return LSDHIP_E_STATE; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->aux_done, c->aux_stream));
  c->aux_active = false;
  c->aux_pending = true;
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_aux_join(lsdhip_ctx* c) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  if (c->aux_active) { lsd_set_error("lsdhip_ctx_aux_join between begin and end"); return LSDHIP_E_STATE; }
  if (!c->aux_pending) return LSDHIP_OK;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamWaitEvent(c->stream, c->aux_done, 0));
  c->aux_pending = false;
  return LSDHIP_OK;
}
extern "C" void* lsdhip_ctx_aux_stream(lsdhip_ctx* c) { return c ? (void*)c->aux_stream : nullptr; }
// ---- inter-process exchange on one node without RCCL (lsdhip_driver's second transport): IPC-mapped device memory + flags ---------
// A flag is an int in device memory that both processes map; values only grow.  Both operations are stream-ordered one-lane kernels:
// set publishes everything the stream did before it (system-scope release), wait spins — bounded — until the flag has reached the
// value and raises *fail otherwise.
// Both run as LSD_FLAG_WGS one-wave workgroups, dealt round-robin over the chip's 8 XCDs: the writes a set publishes may sit dirty in
// any XCD's L2 and the data a wait is followed by may be cached stale in any XCD's L2 — and the other side is another process, whose
// accesses this runtime's own cache bookkeeping between consecutive launches knows nothing about.  So every XCD runs the
// system-scope release (set: before the last arriving workgroup raises the flag) or acquire (wait: after it has seen the flag).
__global__ void k_flag_set(int* flag, int value, unsigned* arrive) {
  if (threadIdx.x != 0) return;
  __threadfence_system();
  const unsigned n = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (n + 1 == gridDim.x) {
    __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the next set on this stream starts from zero
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void k_flag_wait(const int* flag, int value, int* fail) {
  if (threadIdx.x != 0) return;
  for (unsigned spins = 0; spins < (1u << 26); spins++) {          // ~8 s (2 s proved too tight once for a peer process's first launches on a shared GPU)
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= value) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                // system scope: this XCD's L2 drops what it cached of foreign memory
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  *fail = value;
}
extern "C" int lsdhip_ctx_ipc_export(lsdhip_ctx* c, void* dev, unsigned char handle64[64]) {
  if (!c || !dev || !handle64) return LSDHIP_E_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, dev));
  memcpy(handle64, &h, 64);
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_ipc_open(lsdhip_ctx* c, const unsigned char handle64[64], void** out_dev) {
  if (!c || !handle64 || !out_dev) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  HIPCHK(hipIpcOpenMemHandle(out_dev, h, hipIpcMemLazyEnablePeerAccess));
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_ipc_close(lsdhip_ctx* c, void* dev) {
  if (!c || !dev) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipIpcCloseMemHandle(dev));
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_flag_set(lsdhip_ctx* c, int* flag_dev, int value) {
  if (!c || !flag_dev) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (!c->d_flagArrive) {
    HIPCHK(hipMalloc((void**)&c->d_flagArrive, 64));
    HIPCHK(hipMemset(c->d_flagArrive, 0, 64));
    // the memset runs on the null stream, which the (non-blocking) streams of this library do not wait for: without this the first
    // k_flag_set could count arrivals on top of whatever the allocation held, never reach its grid size and never raise its flag —
    // every wait of both processes then times out at value 1 (seen twice on the GPU box, tests/test_bands_gpu.py)
    HIPCHK(hipStreamSynchronize(nullptr));
  }
  // (sets of one context are stream-ordered, on either of its streams in turn: one arrival counter serves them all)
  hipLaunchKernelGGL(k_flag_set, dim3(LSD_FLAG_WGS), dim3(64), 0, lsd_transport_stream(c), flag_dev, value, c->d_flagArrive);
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_flag_wait(lsdhip_ctx* c, const int* flag_dev, int value, int* fail_dev) {
  if (!c || !flag_dev || !fail_dev) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_flag_wait, dim3(LSD_FLAG_WGS), dim3(64), 0, lsd_transport_stream(c), flag_dev, value, fail_dev);
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_memset_dev(lsdhip_ctx* c, void* dev, int byte, size_t bytes) {
  if (!c || !dev) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemsetAsync(dev, byte, bytes, c->stream));
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_read_dev(lsdhip_ctx* c, void* host, const void* dev, size_t bytes) {
  if (!c || !host || !dev) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (c->pipeline) { if (int rc = lsd_sync_all(c)) return rc; }
  HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return LSDHIP_OK;
}
extern "C" int lsdhip_ctx_set_async(lsdhip_ctx* c, int on) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  c->async = on != 0;
  return LSDHIP_OK;
}
int lsd_ctx_take_slot(lsdhip_ctx* c) {
  LSD_CTX_LOCK(c);
  const int i = c->slot_next;
  c->slot_next = (c->slot_next + 1) % LSD_NUM_SLOTS;
  if (c->slot_stats_owner[i]) { int rc = lsd_frame_resolve(c->slot_stats_owner[i]); if (rc) return rc; }
  if (c->slot_rescale_owner[i]) { int rc = lsd_frame_resolve(c->slot_rescale_owner[i]); if (rc) return rc; }
  c->slot_epoch[i] = c->enqEpoch;
  c->slot_mseq[i] = c->mSeq + 1;     // pipelined contexts: the calling DepthMap entry ends with the record point of this number
  return i;
}
int lsd_frame_resolve(lsdhip_frame* f) {
  if (f->pendStats < 0 && f->pendRescale < 0) return LSDHIP_OK;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  // the values are in pinned host memory once the kernels that write them have completed: known without a synchronisation when a
  // tracking job enqueued after them has been seen to finish (the ring of slots wraps onto a retired keyframe's every few keyframes)
  bool landed;
  if (c->pipeline) {
    // the slots are written on the mapping stream: done once the record point behind their DepthMap call has been passed
    landed = (f->pendRescale < 0 || lsd_m_done(c, c->slot_mseq[f->pendRescale])) && (f->pendStats < 0 || lsd_m_done(c, c->slot_mseq[f->pendStats]));
    if (!landed) {
      // (inside an open lane region the slot's record point does not exist yet — it is recorded where the lanes join — and the kernels
      // that write the slot were queued on a lane: drain the lanes in use as well)
      for (int i = 0; i < c->lanes_open; i++) if (c->lane_used[i]) HIPCHK(hipStreamSynchronize(c->lanes[i]));
      HIPCHK(hipStreamSynchronize(c->mstream));
      c->mDoneSeq = c->mSeq;
    }
  } else {
    landed = (f->pendRescale < 0 || c->doneEpoch > c->slot_epoch[f->pendRescale]) &&
             (f->pendStats < 0 || c->doneEpoch > c->slot_epoch[f->pendStats]);
    if (!landed) {
      for (int i = 0; i < c->lanes_open; i++) if (c->lane_used[i]) HIPCHK(hipStreamSynchronize(c->lanes[i]));   // (a slot written on a lane)
      HIPCHK(hipStreamSynchronize(c->stream));
      c->enqEpoch++;
      c->doneEpoch = c->enqEpoch;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  int rc = LSDHIP_OK;
  if (f->pendRescale >= 0) {
    const DeferredSlot& s = c->h_slots[f->pendRescale];
    // rescaleFactor = numIdepth / sumIdepth in float (DepthMap.cpp:1294), the expression k_rescale evaluated
    const float sumIdepth = (float)s.sum, numIdepth = (float)s.count;
    f->thisToParent_raw.s = numIdepth / sumIdepth;
    if (s.flag != 0) { lsd_set_error("propagateDepth: too many source hypotheses mapped to one target pixel"); rc = LSDHIP_E_CAPACITY; }
    c->slot_rescale_owner[f->pendRescale] = nullptr;
    f->pendRescale = -1;
  }
  if (f->pendStats >= 0) {
    const DeferredSlot& s = c->h_slots[f->pendStats];
    const float sumIdepth = (float)s.sum;
    const int numIdepth = (int)s.count;
    f->meanIdepth = sumIdepth / numIdepth;
    f->numPoints = numIdepth;
    c->slot_stats_owner[f->pendStats] = nullptr;
    f->pendStats = -1;
  }
  return rc;
}
extern "C" int lsdhip_ctx_intrinsics(lsdhip_ctx* c, int level, float out[8]) {
  if (!c || level < 0 || level >= LSD_LEVELS) return LSDHIP_E_ARG;
  memcpy(out, &c->intr[level], sizeof(LevelIntr));
  return LSDHIP_OK;
}
extern "C" int lsdhip_prof_enable(lsdhip_ctx* c, int on) { if (!c) return LSDHIP_E_ARG; c->prof_on = on != 0; return LSDHIP_OK; }
static void bprof_collect(lsdhip_ctx* c, bool wait) {
  for (int i = 0; i < LSD_BPROF_SLOTS; i++) {
    lsdhip_ctx::BProfSlot& p = c->bprof[i];
    if (!p.pending) continue;
    if (wait) (void)hipEventSynchronize(p.b);
    else if (hipEventQuery(p.b) != hipSuccess) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->bprof_ms[p.kind] += ms; c->bprof_units[p.kind] += p.units; c->bprof_calls[p.kind]++; }
    p.pending = false;
  }
}
int lsd_bprof_begin(lsdhip_ctx* c, int kind, hipStream_t s) {
  if (!c->prof_on || kind < 0 || kind >= LSD_BPROF_KINDS) return -1;
  if ((c->bprof_tick[kind]++ % LSD_BPROF_PERIOD) != 0) return -1;
  const int i = c->bprof_next;
  c->bprof_next = (c->bprof_next + 1) % LSD_BPROF_SLOTS;
  lsdhip_ctx::BProfSlot& p = c->bprof[i];
  if (p.pending) { bprof_collect(c, false); if (p.pending) { if (hipEventSynchronize(p.b) != hipSuccess) return LSDHIP_E_HIP; bprof_collect(c, false); } }
  if (!p.a) { if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return LSDHIP_E_HIP; }
  p.kind = kind;
  if (hipEventRecord(p.a, s) != hipSuccess) return LSDHIP_E_HIP;
  return i;
}
int lsd_bprof_end(lsdhip_ctx* c, int slot, hipStream_t s, double units) {
  if (slot < 0) return slot == -1 ? LSDHIP_OK : slot;
  lsdhip_ctx::BProfSlot& p = c->bprof[slot];
  if (hipEventRecord(p.b, s) != hipSuccess) return LSDHIP_E_HIP;
  p.units = units;
  p.pending = true;
  return LSDHIP_OK;
}
// ms / calls / units per kind (LSD_BPROF_KINDS of each) of the sampled shared launches since the last reset, and the sampled walk launches'
// work: obs[0] = launches counted, obs[1] = searches, obs[2] = walk steps.  Waits for the brackets still in flight.
extern "C" int lsdhip_ctx_batch_prof_read(lsdhip_ctx* c, double* ms, long long* calls, double* units, double obs[3]) {
  if (!c || !ms || !calls || !units) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  bprof_collect(c, true);
  for (int k = 0; k < LSD_BPROF_KINDS; k++) { ms[k] = c->bprof_ms[k]; calls[k] = c->bprof_calls[k]; units[k] = c->bprof_units[k]; }
  if (obs) {
    obs[0] = obs[1] = obs[2] = 0;
    if (c->d_obsBatchAcc) {
      unsigned long long h[66];
      if (int rc = lsd_sync_all(c)) return rc;
      HIPCHK(hipMemcpy(h, c->d_obsBatchAcc, sizeof(h), hipMemcpyDeviceToHost));
      obs[0] = (double)h[65]; obs[1] = (double)h[0];
      for (int i = 1; i <= 64; i++) obs[2] += (double)h[i];
    }
  }
  return LSDHIP_OK;
}
extern "C" int lsdhip_prof_reset(lsdhip_ctx* c) {
  if (!c) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  c->prof_ms = 0; c->prof_bytes = 0; c->prof_launches = 0;
  bprof_collect(c, true);
  for (int k = 0; k < LSD_BPROF_KINDS; k++) { c->bprof_ms[k] = 0; c->bprof_units[k] = 0; c->bprof_calls[k] = 0; }
  if (c->d_obsBatchAcc) HIPCHK(hipMemsetAsync(c->d_obsBatchAcc, 0, 66 * 8, lsd_map_stream(c)));
  return LSDHIP_OK;
}
extern "C" int lsdhip_prof_read(lsdhip_ctx* c, double* ms, long long* launches, double* bytes) {
  if (c) { int rc = lsd_prof_collect(c); if (rc) return rc; }
  if (!c) return LSDHIP_E_ARG;
  if (ms) *ms = c->prof_ms;
  if (launches) *launches = c->prof_launches;
  if (bytes) *bytes = c->prof_bytes;
  return LSDHIP_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int frame_alloc(lsdhip_ctx* c, int id, lsdhip_frame** out) {
  LSD_CTX_LOCK(c);
  lsdhip_frame* f = new lsdhip_frame();
  f->ctx = c;
  f->id = id;
  f->thisToParent_raw.q = {1, 0, 0, 0};
  f->thisToParent_raw.t[0] = f->thisToParent_raw.t[1] = f->thisToParent_raw.t[2] = 0;
  f->thisToParent_raw.s = 1;
  // one arena per frame: gray | image[l] | grad[l] | absgrad | maxgrad | idepth[l] | idepthVar[l] | wasGood
  size_t off = 0, offs[64];
  int k = 0;
  auto take = [&](size_t bytes) { off = align_up(off, 256); offs[k++] = off; off += bytes; };
  size_t n0 = (size_t)c->w * c->h;
  take(n0);
  for (int l = 0; l < LSD_LEVELS; l++) take((size_t)c->wl[l] * c->hl[l] * 4);
  for (int l = 0; l < LSD_LEVELS; l++) take((size_t)c->wl[l] * c->hl[l] * 16);
  take(n0 * 4);
  take(n0 * 4);
  for (int l = 0; l < LSD_LEVELS; l++) take((size_t)c->wl[l] * c->hl[l] * 4);
  for (int l = 0; l < LSD_LEVELS; l++) take((size_t)c->wl[l] * c->hl[l] * 4);
  take((size_t)c->wl[1] * c->hl[1]);
  take(n0 * 4);   // re-activation data (Frame::takeReActivationData): idepth, idepthVar, validity
  take(n0 * 4);
  take(n0);
  for (int l = 0; l < LSD_LEVELS; l++) take((size_t)c->wl[l] * c->hl[l] * 4);   // second depth plane set (pipelined contexts)
  for (int l = 0; l < LSD_LEVELS; l++) take((size_t)c->wl[l] * c->hl[l] * 4);
  for (int s2 = 0; s2 < 2; s2++)                                                 // reference blocks of levels >= 1, one set per depth plane set
    for (int l = 1; l < LSD_LEVELS; l++) take(lsd_refblk_bytes(c->wl[l] * c->hl[l]));
  take(lsd_gradcand_bytes((int)n0));                                             // gradient candidates (keyframes)
  char* base = nullptr;
  c->arena_bytes = align_up(off, 256);
  if (!c->free_arenas.empty()) {
    base = (char*)c->free_arenas.back();
    c->free_arenas.pop_back();
  } else {
    hipError_t e = hipMalloc((void**)&base, c->arena_bytes);
    if (e != hipSuccess) { lsd_set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); delete f; return LSDHIP_E_HIP; }
  }
  k = 0;
  f->d_gray = (uint8_t*)(base + offs[k++]);
  for (int l = 0; l < LSD_LEVELS; l++) f->d_image[l] = (float*)(base + offs[k++]);
  for (int l = 0; l < LSD_LEVELS; l++) f->d_grad[l] = (float4*)(base + offs[k++]);
  f->d_absgrad = (float*)(base + offs[k++]);
  f->d_maxgrad = (float*)(base + offs[k++]);
  for (int l = 0; l < LSD_LEVELS; l++) f->d_idepth[l] = (float*)(base + offs[k++]);
  for (int l = 0; l < LSD_LEVELS; l++) f->d_idepthVar[l] = (float*)(base + offs[k++]);
  f->d_wasGood = (uint8_t*)(base + offs[k++]);
  f->d_idepth_reAct = (float*)(base + offs[k++]);
  f->d_idepthVar_reAct = (float*)(base + offs[k++]);
  f->d_validity_reAct = (uint8_t*)(base + offs[k++]);
  for (int l = 0; l < LSD_LEVELS; l++) f->d_idepthW[l] = (float*)(base + offs[k++]);
  for (int l = 0; l < LSD_LEVELS; l++) f->d_idepthVarW[l] = (float*)(base + offs[k++]);
  for (int l = 1; l < LSD_LEVELS; l++) f->d_refBlk[l] = (uint8_t*)(base + offs[k++]);
  for (int l = 1; l < LSD_LEVELS; l++) f->d_refBlkW[l] = (uint8_t*)(base + offs[k++]);
  f->d_gradCand = (uint16_t*)(base + offs[k++]);
  *out = f;
  return LSDHIP_OK;
}

int lsd_frame_build_pyramids(lsdhip_frame* f, const uint8_t* src, hipStream_t stream) {
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  if (!stream) stream = lsd_map_stream(c);
  dim3 grid(c->w / 16, c->h / 16);
  lsdhip_host_mark(21);
  hipLaunchKernelGGL(k_image_pyramid, grid, dim3(256), 0, stream, src ? src : f->d_gray, f->d_image[0], f->d_image[1], f->d_image[2],
                     f->d_image[3], f->d_image[4], c->w, c->h, (float4*)nullptr, (float*)nullptr);
  f->level0Ready = false; f->gradCandTh = -1.0f;
  GradMaxArgs ga;
  int nb = 0;
  for (int l = 0; l < LSD_LEVELS; l++) {
    ga.img[l] = lsd_g(f->d_image[l]); ga.grad[l] = lsd_g(f->d_grad[l]); ga.w[l] = c->wl[l]; ga.h[l] = c->hl[l];
    ga.blk0[l] = nb;
    if (l >= 1) nb += (c->wl[l] * c->hl[l] + 255) / 256;
  }
  ga.blk0[LSD_LEVELS] = nb;
  ga.wasGoodWords = lsd_g((uint32_t*)f->d_wasGood); ga.nMaskWords = (c->wl[1] * c->hl[1] + 3) / 4;
  lsdhip_host_mark(22);
  hipLaunchKernelGGL(k_gradients_max, dim3(nb), dim3(256), 0, stream, ga);   // the gradient blocks only: maxGradients is a keyframe plane (lsd_frames_require_level0)
  lsdhip_host_mark(23);
  HIPCHK(hipGetLastError());
  f->wasGoodPristine = true;
  return LSDHIP_OK;
}

// The reference blocks of the PUBLISHED depth planes of the keyframes of a throughput-mode tracking batch, where they are missing: a context
// builds them behind every idepth pyramid only once it has run such a batch (a single-sequence loop never reads them and does not pay the
// launch).  Queued on the caller's stream — the tracking stream, which is already ordered behind the planes.
int lsd_frames_require_ref_blocks(lsdhip_frame** kfs, int n, hipStream_t stream) {
  if (n <= 0) return LSDHIP_OK;
  lsdhip_ctx* c = kfs[0]->ctx;
  LSD_CTX_LOCK(c);
  c->refBlocksWanted = true;
  std::vector<lsdhip_frame*> todo;
  for (int j = 0; j < n; j++)
    if (kfs[j] && !kfs[j]->refBlkValid && std::find(todo.begin(), todo.end(), kfs[j]) == todo.end()) todo.push_back(kfs[j]);
  if (todo.empty()) return LSDHIP_OK;
  std::vector<DepthPyrArgs> items(todo.size());
  for (size_t j = 0; j < todo.size(); j++) {
    lsdhip_frame* f = todo[j];
    DepthPyrArgs& a = items[j];
    memset((void*)&a, 0, sizeof(a));
    for (int l = 0; l < LSD_LEVELS; l++) { a.id[l] = lsd_g(f->d_idepth[l]); a.var[l] = lsd_g(f->d_idepthVar[l]); a.blk[l] = lsd_g(f->d_refBlk[l]); }
    a.w0 = c->w; a.h0 = c->h;
  }
  void* dev = nullptr;
  if (int rc = lsd_args_push(c, items.data(), sizeof(DepthPyrArgs) * items.size(), stream, &dev)) return rc;
  hipLaunchKernelGGL(k_ref_blocks_batch, dim3(lsd_refblk_grid(c), 1, (unsigned)todo.size()), dim3(256), 0, stream, (const DepthPyrArgs*)dev);
  HIPCHK(hipGetLastError());
  if (int rc = lsd_args_release(c, dev, stream)) return rc;
  for (lsdhip_frame* f : todo) f->refBlkValid = true;
  return LSDHIP_OK;
}

// Frame::gradients(0) / Frame::maxGradients(0) on demand (the reference's Frame::require, Frame.cpp:560-640): the level-0 gradient texels,
// |grad| and its 3x3 maximum, for the frames that are asked for them — the keyframes of a DepthMap, a frame given a ground-truth depth,
// a level-0 tracking job, a download.  Queued on the mapping stream (where the consumers are; a caller on the tracking stream moves the
// frame's readySeq behind them).  One launch pair for all frames of the call that do not have the planes yet.
int lsd_frames_require_level0(lsdhip_frame** fs, int n) {
  if (n <= 0) return LSDHIP_OK;
  lsdhip_ctx* c = fs[0]->ctx;
  LSD_CTX_LOCK(c);
  std::vector<lsdhip_frame*> todo;
  for (int j = 0; j < n; j++)
    if (fs[j] && !fs[j]->level0Ready && std::find(todo.begin(), todo.end(), fs[j]) == todo.end()) todo.push_back(fs[j]);
  if (todo.empty()) return LSDHIP_OK;
  const hipStream_t ms = lsd_map_stream(c);
  const int n0 = c->w * c->h, m = (int)todo.size();
  auto fill = [&](lsdhip_frame* f, Level0Item& it, MaxCandItem& mc) {
    it.img0 = lsd_g((const float*)f->d_image[0]); it.grad0 = lsd_g(f->d_grad[0]); it.absgrad = lsd_g(f->d_absgrad);
    mc.absg = lsd_g((const float*)f->d_absgrad); mc.maxgrad = lsd_g(f->d_maxgrad); mc.cand = lsd_g(f->d_gradCand);
  };
  if (m == 1) {
    Level0Item it;
    MaxCandItem mc;
    fill(todo[0], it, mc);
    hipLaunchKernelGGL(k_level0_gradients, dim3((n0 + 255) / 256), dim3(256), 0, ms, it, c->w, c->h);
    hipLaunchKernelGGL(k_maxgrad_candidates, dim3(lsd_gradcand_groups(n0)), dim3(1024), 0, ms, mc, c->w, c->h, c->params.minUseGrad);
  } else {
    const size_t itBytes = align_up(sizeof(Level0Item) * (size_t)m, 256);
    std::vector<uint8_t> blob(itBytes + sizeof(MaxCandItem) * (size_t)m);
    for (int j = 0; j < m; j++) fill(todo[j], ((Level0Item*)blob.data())[j], ((MaxCandItem*)(blob.data() + itBytes))[j]);
    void* dev = nullptr;
    if (int rc = lsd_args_push(c, blob.data(), blob.size(), ms, &dev)) return rc;
    hipLaunchKernelGGL(k_level0_gradients_batch, dim3((n0 + 255) / 256, m), dim3(256), 0, ms, (const Level0Item*)dev, c->w, c->h);
    hipLaunchKernelGGL(k_maxgrad_candidates_batch, dim3(lsd_gradcand_groups(n0), m), dim3(1024), 0, ms,
                       (const MaxCandItem*)((const uint8_t*)dev + itBytes), c->w, c->h, c->params.minUseGrad);
    if (int rc = lsd_args_release(c, dev, ms)) return rc;
  }
  HIPCHK(hipGetLastError());
  for (lsdhip_frame* f : todo) { f->level0Ready = true; f->gradCandTh = c->params.minUseGrad; }
  return LSDHIP_OK;
}
int lsd_frame_require_level0(lsdhip_frame* f) { return lsd_frames_require_level0(&f, 1); }
// ... for a consumer on the tracking stream (a level-0 tracking / Sim3 job) or the host (a download): the frame's planes are complete at a
// new point of the mapping stream
int lsd_frame_require_level0_for_tracking(lsdhip_frame* f) {
  if (f->level0Ready) return LSDHIP_OK;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  if (int rc = lsd_m_begin(c)) return rc;
  if (int rc = lsd_frame_require_level0(f)) return rc;
  if (c->pipeline) {
    const long long seq = lsd_m_record(c);
    if (seq < 0) return LSDHIP_E_HIP;
    if (seq > f->readySeq) f->readySeq = seq;
    if (int rc = lsd_t_wait_m(c, seq)) return rc;   // (the callers have already ordered the tracking stream behind the frame's older readySeq)
  }
  return LSDHIP_OK;
}

int lsd_frame_build_idepth_pyramid(lsdhip_frame* f, const double* redPartials, int redN, double* redOut) {
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  // non-pipelined: a tracking job that is topped up after this point would read the new planes (tracker.hip); pipelined: the job's
  // planes stay untouched, the version changes when the new ones are published
  if (!c->pipeline) f->depthVersion++;
  DepthPyrArgs a;
  float** id = lsd_depth_w(f);
  float** var = lsd_depthvar_w(f);
  for (int l = 0; l < LSD_LEVELS; l++) { a.id[l] = lsd_g(id[l]); a.var[l] = lsd_g(var[l]); }
  a.w0 = c->w; a.h0 = c->h;
  a.redPartials = lsd_g(redPartials); a.redN = redN; a.redOut = lsd_g(redOut);
  for (int l = 0; l < LSD_LEVELS; l++) a.blk[l] = lsd_g(lsd_refblk_w(f)[l]);
  hipLaunchKernelGGL(k_idepth_pyramid, dim3((c->w + 31) / 32, (c->h + 31) / 32 + (redPartials ? 1 : 0)), dim3(256), 0, lsd_map_stream(c), a);
  if (c->refBlocksWanted) hipLaunchKernelGGL(k_ref_blocks, dim3(lsd_refblk_grid(c)), dim3(256), 0, lsd_map_stream(c), a);
  (c->pipeline ? f->refBlkValidW : f->refBlkValid) = c->refBlocksWanted;
  HIPCHK(hipGetLastError());
  if (c->pipeline) { f->depthPending = true; f->depthPendingSeq = c->mSeq + 1; }   // complete at the caller's record point
  else f->hasIDepth = true;
  return LSDHIP_OK;
}

// Kernel-argument records of the batched launches: a ring of pinned slots with device twins.  A slot is handed out again only once the
// event recorded behind its last use has completed: lsd_args_commit records it behind the copy, lsd_args_release — called after the
// launches that read the device twin — moves it behind them (whatever stream they were queued on).
int lsd_args_begin(lsdhip_ctx* c, size_t bytes, void** host_out, void** dev_out) {
  LSD_CTX_LOCK(c);
  lsdhip_ctx::ArgRing& r = c->args;
  constexpr int NS = lsdhip_ctx::ArgRing::NS;
  if (bytes > r.slotBytes) {
    // grow: nothing may still read the old block
    HIPCHK(hipDeviceSynchronize());
    if (r.h) { (void)hipHostFree(r.h); r.h = nullptr; }
    if (r.d) { (void)hipFree(r.d); r.d = nullptr; }
    r.slotBytes = 0;
    const size_t want = align_up(bytes * 2 > 65536 ? bytes * 2 : 65536, 256);
    HIPCHK(hipHostMalloc((void**)&r.h, want * NS, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&r.d, want * NS));
    r.slotBytes = want;
    for (int i = 0; i < NS; i++) { r.used[i] = false; if (!r.ev[i]) HIPCHK(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming)); }
  }
  const int i = r.next;
  r.next = (r.next + 1) % NS;
  if (r.used[i]) HIPCHK(hipEventSynchronize(r.ev[i]));   // (NS - 1 uses ago: long finished)
  r.cur = i;
  r.curBytes = bytes;
  *host_out = r.h + (size_t)i * r.slotBytes;
  *dev_out = r.d + (size_t)i * r.slotBytes;
  return LSDHIP_OK;
}
int lsd_args_commit(lsdhip_ctx* c, hipStream_t s) {
  LSD_CTX_LOCK(c);
  lsdhip_ctx::ArgRing& r = c->args;
  const int i = r.cur;
  HIPCHK(hipMemcpyAsync(r.d + (size_t)i * r.slotBytes, r.h + (size_t)i * r.slotBytes, r.curBytes, hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(r.ev[i], s));
  r.used[i] = true;
  return LSDHIP_OK;
}
// `dev`: the device address lsd_args_begin / lsd_args_push handed out — the slot is identified by it, not by "the slot begun last": a
// helper that pushes arguments of its own between a commit and its release (lsd_frame_build_idepth_pyramid_batch inside the keyframe
// change) must not make the release guard the wrong slot
int lsd_args_release(lsdhip_ctx* c, const void* dev, hipStream_t s) {
  LSD_CTX_LOCK(c);
  lsdhip_ctx::ArgRing& r = c->args;
  constexpr int NS = lsdhip_ctx::ArgRing::NS;
  const uint8_t* p = (const uint8_t*)dev;
  if (!r.d || r.slotBytes == 0 || p < (const uint8_t*)r.d || p >= (const uint8_t*)r.d + r.slotBytes * NS) return LSDHIP_OK;   // (the ring was regrown since: the device was drained then)
  const int i = (int)((size_t)(p - (const uint8_t*)r.d) / r.slotBytes);
  if (!r.used[i]) return LSDHIP_OK;
  HIPCHK(hipEventRecord(r.ev[i], s));
  return LSDHIP_OK;
}
int lsd_args_push(lsdhip_ctx* c, const void* src, size_t bytes, hipStream_t s, void** dev_out) {
  LSD_CTX_LOCK(c);
  void* h = nullptr;
  if (int rc = lsd_args_begin(c, bytes, &h, dev_out)) return rc;
  memcpy(h, src, bytes);
  return lsd_args_commit(c, s);
}

int lsd_frame_build_idepth_pyramid_batch(lsdhip_frame** fs, int n, const double* const* redPartials, int redN, double* const* redOut, const int* redNs) {
  if (n <= 0) return LSDHIP_OK;
  lsdhip_ctx* c = fs[0]->ctx;
  LSD_CTX_LOCK(c);
  std::vector<DepthPyrArgs> items((size_t)n);
  for (int j = 0; j < n; j++) {
    lsdhip_frame* f = fs[j];
    if (!c->pipeline) f->depthVersion++;
    DepthPyrArgs& a = items[j];
    float** id = lsd_depth_w(f);
    float** var = lsd_depthvar_w(f);
    for (int l = 0; l < LSD_LEVELS; l++) { a.id[l] = lsd_g(id[l]); a.var[l] = lsd_g(var[l]); }
    a.w0 = c->w; a.h0 = c->h;
    a.redPartials = lsd_g(redPartials[j]); a.redN = redNs ? redNs[j] : redN; a.redOut = lsd_g(redOut[j]);
    for (int l = 0; l < LSD_LEVELS; l++) a.blk[l] = lsd_g(lsd_refblk_w(f)[l]);
  }
  void* dev = nullptr;
  int rc = lsd_args_push(c, items.data(), sizeof(DepthPyrArgs) * (size_t)n, lsd_map_stream(c), &dev);
  if (rc) return rc;
  hipLaunchKernelGGL(k_idepth_pyramid_batch, dim3((c->w + 31) / 32, (c->h + 31) / 32 + 1, n), dim3(256), 0, lsd_map_stream(c), (const DepthPyrArgs*)dev);
  if (c->refBlocksWanted) hipLaunchKernelGGL(k_ref_blocks_batch, dim3(lsd_refblk_grid(c), 1, n), dim3(256), 0, lsd_map_stream(c), (const DepthPyrArgs*)dev);
  for (int j = 0; j < n; j++) (c->pipeline ? fs[j]->refBlkValidW : fs[j]->refBlkValid) = c->refBlocksWanted;
  HIPCHK(hipGetLastError());
  rc = lsd_args_release(c, dev, lsd_map_stream(c));
  if (rc) return rc;
  for (int j = 0; j < n; j++) {
    if (c->pipeline) { fs[j]->depthPending = true; fs[j]->depthPendingSeq = c->mSeq + 1; }
    else fs[j]->hasIDepth = true;
  }
  return LSDHIP_OK;
}

int lsd_frame_ensure_wasgood(lsdhip_frame* f) {
  if (!f->wasGoodValid) {
    if (!f->wasGoodPristine) HIPCHK(hipMemsetAsync(f->d_wasGood, 0xFF, (size_t)f->ctx->wl[1] * f->ctx->hl[1], f->ctx->stream));
    f->wasGoodValid = true;
  }
  f->wasGoodPristine = false;   // the caller is about to write the mask
  return LSDHIP_OK;
}

extern "C" int lsdhip_frame_create_from_device(lsdhip_ctx* c, int id, const uint8_t* gray_dev, lsdhip_frame** out) {
  if (!c || !gray_dev || !out) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  lsdhip_frame* f = nullptr;
  int rc = frame_alloc(c, id, &f);
  if (rc) return rc;
  // the pyramid kernel reads the caller's device image directly (stream-ordered; nothing else needs the uint8 plane)
  rc = lsd_m_begin(c);
  if (rc == LSDHIP_OK) rc = lsd_frame_build_pyramids(f, gray_dev, nullptr);
  if (rc) { lsdhip_frame_destroy(f); return rc; }
  lsd_trace_sum(c, lsd_map_stream(c), 2, id, f->d_image[0], (size_t)((char*)f->d_idepth[0] - (char*)f->d_image[0]));
  f->readySeq = lsd_m_record(c);
  if (f->readySeq < 0) { lsdhip_frame_destroy(f); return LSDHIP_E_HIP; }
  *out = f;
  return LSDHIP_OK;
}
static int frame_create_host(lsdhip_ctx* c, int id, const uint8_t* gray_host, bool wait, lsdhip_frame** out) {
  if (!c || !gray_host || !out) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  lsdhip_frame* f = nullptr;
  int rc = frame_alloc(c, id, &f);
  if (rc) return rc;
  const hipStream_t ms = lsd_map_stream(c);
  rc = lsd_m_begin(c);
  if (rc) { lsdhip_frame_destroy(f); return rc; }
  hipError_t e = hipMemcpyAsync(f->d_gray, gray_host, (size_t)c->w * c->h, hipMemcpyHostToDevice, ms);
  if (e != hipSuccess) { lsd_set_error("lsdhip_frame_create: upload failed: %s", hipGetErrorString(e)); lsdhip_frame_destroy(f); return LSDHIP_E_HIP; }
  rc = lsd_frame_build_pyramids(f, nullptr, nullptr);
  if (rc) { lsdhip_frame_destroy(f); return rc; }
  f->readySeq = lsd_m_record(c);
  if (f->readySeq < 0) { lsdhip_frame_destroy(f); return LSDHIP_E_HIP; }
  if (wait) {
    e = hipStreamSynchronize(ms);  // the host buffer may be reused by the caller
    if (e != hipSuccess) { lsd_set_error("lsdhip_frame_create: %s", hipGetErrorString(e)); lsdhip_frame_destroy(f); return LSDHIP_E_HIP; }
    if (c->pipeline) c->mDoneSeq = c->mSeq;
  }
  *out = f;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_create(lsdhip_ctx* c, int id, const uint8_t* gray_host, lsdhip_frame** out) {
  return frame_create_host(c, id, gray_host, true, out);
}
extern "C" int lsdhip_frame_create_async(lsdhip_ctx* c, int id, const uint8_t* gray_host, lsdhip_frame** out) {
  return frame_create_host(c, id, gray_host, false, out);
}
// Frame-memory pool (the reference's FrameMemory keeps returned buffers for reuse, util/... FrameMemory.cpp): make sure n arenas are
// allocated and waiting, so that a loop which keeps its keyframes alive does not pay a hipMalloc (0.5 ms for 20 MB) per keyframe.
extern "C" int lsdhip_ctx_reserve_frames(lsdhip_ctx* c, int n) {
  if (!c || n < 0) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if ((size_t)n > c->arena_keep) c->arena_keep = (size_t)n;
  std::vector<lsdhip_frame*> tmp;
  int rc = LSDHIP_OK;
  const size_t have = c->free_arenas.size();
  for (size_t i = 0; i < (size_t)n && rc == LSDHIP_OK; i++) {   // the first `have` come out of the pool, the rest are new
    lsdhip_frame* f = nullptr;
    rc = frame_alloc(c, -1, &f);
    if (rc == LSDHIP_OK) tmp.push_back(f);
  }
  (void)have;
  for (lsdhip_frame* f : tmp) lsdhip_frame_destroy(f);
  return rc;
}
// Frame creation for the new frames of n sequences at once: two launches for all of them (blockIdx.z / .y = frame) instead of two per
// frame.  Same planes, bit for bit, as n lsdhip_frame_create_from_device / lsdhip_frame_create calls.
extern "C" int lsdhip_frame_create_batch(lsdhip_ctx* c, int n, const int* ids, const uint8_t* const* gray, int images_on_device, lsdhip_frame** out) {
  if (!c || n <= 0 || !ids || !gray || !out) return LSDHIP_E_ARG;
  for (int j = 0; j < n; j++) if (!gray[j]) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if ((size_t)(2 * n + 8) > c->arena_keep) c->arena_keep = (size_t)(2 * n + 8);   // a round of n frames retires n arenas at once
  for (int j = 0; j < n; j++) out[j] = nullptr;
  auto fail = [&](int rc) { for (int j = 0; j < n; j++) if (out[j]) { lsdhip_frame_destroy(out[j]); out[j] = nullptr; } return rc; };
  for (int j = 0; j < n; j++) { int rc = frame_alloc(c, ids[j], &out[j]); if (rc) return fail(rc); }
  const hipStream_t ms = lsd_map_stream(c);
  int rc = lsd_m_begin(c);
  if (rc) return fail(rc);
  std::vector<ImagePyrItem> pi((size_t)n);
  std::vector<GradMaxArgs> gi((size_t)n);
  int nb = 0;
  for (int j = 0; j < n; j++) {
    lsdhip_frame* f = out[j];
    if (!images_on_device) {
      hipError_t e = hipMemcpyAsync(f->d_gray, gray[j], (size_t)c->w * c->h, hipMemcpyHostToDevice, ms);
      if (e != hipSuccess) { lsd_set_error("lsdhip_frame_create_batch: upload failed: %s", hipGetErrorString(e)); return fail(LSDHIP_E_HIP); }
    }
    pi[j].gray = lsd_g(images_on_device ? gray[j] : f->d_gray);
    for (int l = 0; l < LSD_LEVELS; l++) pi[j].img[l] = lsd_g(f->d_image[l]);
    pi[j].grad0 = nullptr; pi[j].absgrad0 = nullptr;   // keyframe planes: lsd_frames_require_level0
    f->level0Ready = false; f->gradCandTh = -1.0f;
    GradMaxArgs& ga = gi[j];
    nb = 0;
    for (int l = 0; l < LSD_LEVELS; l++) {
      ga.img[l] = lsd_g(f->d_image[l]); ga.grad[l] = lsd_g(f->d_grad[l]); ga.w[l] = c->wl[l]; ga.h[l] = c->hl[l];
      ga.blk0[l] = nb;
      if (l >= 1) nb += (c->wl[l] * c->hl[l] + 255) / 256;
    }
    ga.blk0[LSD_LEVELS] = nb;
    ga.wasGoodWords = lsd_g((uint32_t*)f->d_wasGood); ga.nMaskWords = (c->wl[1] * c->hl[1] + 3) / 4;
    f->wasGoodPristine = true;
  }
  const size_t piBytes = align_up(sizeof(ImagePyrItem) * (size_t)n, 256);
  std::vector<uint8_t> blob(piBytes + sizeof(GradMaxArgs) * (size_t)n);
  memcpy(blob.data(), pi.data(), sizeof(ImagePyrItem) * (size_t)n);
  memcpy(blob.data() + piBytes, gi.data(), sizeof(GradMaxArgs) * (size_t)n);
  void* dev = nullptr;
  rc = lsd_args_push(c, blob.data(), blob.size(), ms, &dev);
  if (rc) return fail(rc);
  const int bp = lsd_bprof_begin(c, 0, ms);
  if (bp < -1) return fail(bp);
  hipLaunchKernelGGL(k_image_pyramid_batch, dim3(c->w / 16, c->h / 16, n), dim3(256), 0, ms, (const ImagePyrItem*)dev, c->w, c->h);
  const int n0 = c->w * c->h;
  hipLaunchKernelGGL(k_gradients_max_batch, dim3(nb, n), dim3(256), 0, ms, (const GradMaxArgs*)((const uint8_t*)dev + piBytes));
  rc = lsd_bprof_end(c, bp, ms, (double)n * n0);
  if (rc) return fail(rc);
  if (hipGetLastError() != hipSuccess) { lsd_set_error("lsdhip_frame_create_batch: launch failed"); return fail(LSDHIP_E_HIP); }
  rc = lsd_args_release(c, dev, ms);
  if (rc) return fail(rc);
  const long long seq = lsd_m_record(c);
  if (seq < 0) return fail(LSDHIP_E_HIP);
  for (int j = 0; j < n; j++) out[j]->readySeq = seq;
  if (!images_on_device) {
    hipError_t e = hipStreamSynchronize(ms);   // the host buffers may be reused by the caller
    if (e != hipSuccess) { lsd_set_error("lsdhip_frame_create_batch: %s", hipGetErrorString(e)); return fail(LSDHIP_E_HIP); }
    if (c->pipeline) c->mDoneSeq = c->mSeq;
  }
  return LSDHIP_OK;
}
extern "C" void lsdhip_frame_destroy(lsdhip_frame* f) {
  if (!f) return;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  if (c->free_arenas.size() < c->arena_keep) {
    c->free_arenas.push_back(f->d_gray);   // arena base; reuse is ordered by the stream itself
  } else {
    (void)hipSetDevice(c->device);
    (void)lsd_sync_all(c);
    (void)hipFree(f->d_gray);
  }
  if (f->pendStats >= 0) c->slot_stats_owner[f->pendStats] = nullptr;
  if (f->pendRescale >= 0) c->slot_rescale_owner[f->pendRescale] = nullptr;
  for (size_t i = 0; i < c->pendingMerges.size();)     // nobody will read this frame's mask any more
    if (c->pendingMerges[i].plane == f->d_wasGood) { *c->pendingMerges[i].doneSeq = 0; c->pendingMerges.erase(c->pendingMerges.begin() + i); }
    else i++;
  lsd_depthmaps_forget_frame(c, f);   // a depth map whose active keyframe this is becomes "no active keyframe"
  delete f;
}
extern "C" int lsdhip_frame_id(lsdhip_frame* f) { return f ? f->id : -1; }

extern "C" int lsdhip_frame_download(lsdhip_frame* f, int what, int level, float* out) {
  if (!f || !out || level < 0 || level >= LSD_LEVELS) return LSDHIP_E_ARG;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  size_t n = (size_t)c->wl[level] * c->hl[level];
  const void* src = nullptr;
  if ((what == 1 && level == 0) || what == 2 || what == 6) { if (int rc = lsd_frame_require_level0_for_tracking(f)) return rc; }   // built on demand
  if (c->pipeline) { if (int rc = lsd_sync_all(c)) return rc; }
  else if (lsd_map_stream(c) != c->stream) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));   // (an open lane region)
  switch (what) {
    case 0: src = f->d_image[level]; break;
    case 1: src = f->d_grad[level]; n *= 4; break;
    case 2: if (level != 0) return LSDHIP_E_ARG; src = f->d_maxgrad; break;
    // Frame::idepth / idepthVar: what the last Frame::setDepth left (on pipelined contexts possibly not yet published to the tracker)
    case 3: if (!f->hasIDepth && !f->depthPending) return LSDHIP_E_STATE; src = lsd_depth_latest(f)[level]; break;
    case 4: if (!f->hasIDepth && !f->depthPending) return LSDHIP_E_STATE; src = lsd_depthvar_latest(f)[level]; break;
    // the level's reference blocks (k_ref_blocks), as bytes: ceil(pixels / 256) x 256 offsets, then one int32 count per block
    case 5:
      if (!f->hasIDepth && !f->depthPending) return LSDHIP_E_STATE;
      if (level < 1) return LSDHIP_E_ARG;
      {
        // of the newest planes (published or not), built here if the context has not been building them (everything is drained above)
        const bool pend = f->depthPending;
        bool& valid = pend ? f->refBlkValidW : f->refBlkValid;
        uint8_t** blk = pend ? f->d_refBlkW : f->d_refBlk;
        if (!valid) {
          DepthPyrArgs pa;
          memset((void*)&pa, 0, sizeof(pa));
          for (int l = 0; l < LSD_LEVELS; l++) { pa.id[l] = lsd_g(lsd_depth_latest(f)[l]); pa.var[l] = lsd_g(lsd_depthvar_latest(f)[l]); pa.blk[l] = lsd_g(blk[l]); }
          pa.w0 = c->w; pa.h0 = c->h;
          hipLaunchKernelGGL(k_ref_blocks, dim3(lsd_refblk_grid(c)), dim3(256), 0, c->stream, pa);
          valid = true;
        }
        src = blk[level];
      }
      HIPCHK(hipMemcpyAsync(out, src, lsd_refblk_bytes((int)n), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      return LSDHIP_OK;
    // the keyframe planes' gradient candidates (k_grad_candidates), as uint16: ceil(pixels / 1024) groups of 1024 offsets, then one count per group
    case 6:
      if (level != 0) return LSDHIP_E_ARG;
      HIPCHK(hipMemcpyAsync(out, f->d_gradCand, lsd_gradcand_bytes((int)n), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      return LSDHIP_OK;
    default: return LSDHIP_E_ARG;
  }
  HIPCHK(hipMemcpyAsync(out, src, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return LSDHIP_OK;
}

extern "C" int lsdhip_frame_set_depth_gt(lsdhip_frame* f, const float* depth_host, float cov_scale) {
  if (!f || !depth_host) return LSDHIP_E_ARG;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  int n0 = c->w * c->h;
  // staging plane kept by the context (ground-truth depth arrives once per keyframe in the GT-initialised modes)
  if (!c->d_gtStage) HIPCHK(hipMalloc((void**)&c->d_gtStage, (size_t)n0 * 4));
  float* d_depth = c->d_gtStage;
  if (c->pipeline) { if (int rcs = lsd_sync_all(c)) return rcs; }
  const hipStream_t ms = lsd_map_stream(c);
  if (int rcl = lsd_frame_require_level0(f)) return rcl;   // Frame::setDepthFromGroundTruth reads maxGradients(0) (Frame.cpp:259)
  HIPCHK(hipMemcpyAsync(d_depth, depth_host, (size_t)n0 * 4, hipMemcpyHostToDevice, ms));
  hipLaunchKernelGGL(k_set_depth_gt, dim3((n0 + 255) / 256), dim3(256), 0, ms, d_depth, f->d_maxgrad, lsd_depth_w(f)[0],
                     lsd_depthvar_w(f)[0], c->w, c->h, cov_scale, c->params.minUseGrad);
  int rc = lsd_frame_build_idepth_pyramid(f);
  HIPCHK(hipStreamSynchronize(ms));   // the host buffer may be reused by the caller
  if (rc == LSDHIP_OK && c->pipeline) { f->depthPendingSeq = 0; rc = lsd_frame_publish_depth(f); }   // a synchronous call: visible to the tracker at once
  return rc;
}
extern "C" int lsdhip_frame_set_depth_planes(lsdhip_frame* f, const float* id, const float* var) {
  if (!f || !id || !var) return LSDHIP_E_ARG;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  size_t n0 = (size_t)c->w * c->h;
  if (c->pipeline) { if (int rcs = lsd_sync_all(c)) return rcs; }
  const hipStream_t ms = lsd_map_stream(c);
  HIPCHK(hipMemcpyAsync(lsd_depth_w(f)[0], id, n0 * 4, hipMemcpyHostToDevice, ms));
  HIPCHK(hipMemcpyAsync(lsd_depthvar_w(f)[0], var, n0 * 4, hipMemcpyHostToDevice, ms));
  int rc = lsd_frame_build_idepth_pyramid(f);
  HIPCHK(hipStreamSynchronize(ms));
  if (rc == LSDHIP_OK && c->pipeline) { f->depthPendingSeq = 0; rc = lsd_frame_publish_depth(f); }
  return rc;
}
// Test / synthetic-benchmark hook: overwrite the level-0 maxGradients plane (scene S3 of SURVEY.md §8(d) generates
// hypothesis maps and gradient masks directly, without images)
extern "C" int lsdhip_frame_set_maxgrad(lsdhip_frame* f, const float* maxgrad_host) {
  if (!f || !maxgrad_host) return LSDHIP_E_ARG;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (int rcl = lsd_frame_require_level0_for_tracking(f)) return rcl;   // (the planes count as built from here on: the overwrite below must be the last word)
  if (c->pipeline) { if (int rcs = lsd_sync_all(c)) return rcs; }
  else if (lsd_map_stream(c) != c->stream) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(f->d_maxgrad, maxgrad_host, (size_t)c->w * c->h * 4, hipMemcpyHostToDevice, c->stream));
  {
    GradCandItem gc;     // the keyframe's gradient candidates follow the plane
    gc.maxgrad = lsd_g((const float*)f->d_maxgrad); gc.cand = lsd_g(f->d_gradCand);
    hipLaunchKernelGGL(k_grad_candidates, dim3(lsd_gradcand_groups(c->w * c->h)), dim3(256), 0, c->stream, gc, c->w, c->h, c->params.minUseGrad);
    f->gradCandTh = c->params.minUseGrad;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_get_wasgood(lsdhip_frame* f, uint8_t* out) {
  if (!f || !out) return LSDHIP_E_ARG;
  if (!f->wasGoodValid) return 0;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  if (c->pipeline) { if (int rcs = lsd_sync_all(c)) return rcs; }
  HIPCHK(hipMemcpyAsync(out, f->d_wasGood, (size_t)c->wl[1] * c->hl[1], hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 1;
}
extern "C" int lsdhip_frame_set_wasgood(lsdhip_frame* f, const uint8_t* in) {
  if (!f || !in) return LSDHIP_E_ARG;
  lsdhip_ctx* c = f->ctx;
  LSD_CTX_LOCK(c);
  if (c->pipeline) { if (int rcs = lsd_sync_all(c)) return rcs; }
  HIPCHK(hipMemcpyAsync(f->d_wasGood, in, (size_t)c->wl[1] * c->hl[1], hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  f->wasGoodValid = true;
  f->wasGoodPristine = false;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_clear_wasgood(lsdhip_frame* f) {
  if (!f) return LSDHIP_E_ARG;
  f->wasGoodValid = false;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_set_pose(lsdhip_frame* f, const double s[8], lsdhip_frame* parent, float initialTrackedResidual) {
  if (!f || !s) return LSDHIP_E_ARG;
  f->thisToParent_raw.q = {s[0], s[1], s[2], s[3]};
  f->thisToParent_raw.t[0] = s[4]; f->thisToParent_raw.t[1] = s[5]; f->thisToParent_raw.t[2] = s[6];
  f->thisToParent_raw.s = s[7];
  f->trackingParent = parent;
  f->trackingParentID = parent ? parent->id : -1;
  f->initialTrackedResidual = initialTrackedResidual;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_get_pose(lsdhip_frame* f, double s[8]) {
  if (!f || !s) return LSDHIP_E_ARG;
  { int rc = lsd_frame_resolve(f); if (rc) return rc; }   // the Sim3 scale of a new keyframe is a deferred result
  s[0] = f->thisToParent_raw.q.w; s[1] = f->thisToParent_raw.q.x; s[2] = f->thisToParent_raw.q.y; s[3] = f->thisToParent_raw.q.z;
  s[4] = f->thisToParent_raw.t[0]; s[5] = f->thisToParent_raw.t[1]; s[6] = f->thisToParent_raw.t[2];
  s[7] = f->thisToParent_raw.s;
  return LSDHIP_OK;
}
// se3FromSim3(reference->getCamToWorld().inverse() * frame->getCamToWorld()) for the two pose-tree shapes the hot path produces
// (C/SlamSystem.cpp:918-920: the initial estimate of SlamSystem::trackFrame): `frame` was tracked on `reference`, or both were tracked
// on the same parent (the frame that followed a keyframe change: tracked on the old keyframe while the mapper promoted `reference`).
extern "C" int lsdhip_frame_relative_pose(lsdhip_frame* reference, lsdhip_frame* frame, double frameToReference[7]) {
  if (!reference || !frame || !frameToReference) return LSDHIP_E_ARG;
  lsdm::Sim3dH rel;
  if (frame->trackingParent == reference && frame->trackingParentID == reference->id) {
    rel = frame->thisToParent_raw;
  } else if (frame->trackingParent && frame->trackingParent == reference->trackingParent && frame->trackingParentID == reference->trackingParentID) {
    { int rc = lsd_frame_resolve(reference); if (rc) return rc; }   // the Sim3 scale of a new keyframe is a deferred result
    const lsdm::Sim3dH inv = lsdm::sim3_inverse(reference->thisToParent_raw);
    rel.q = lsdm::q_mul(inv.q, frame->thisToParent_raw.q);
    rel.s = inv.s * frame->thisToParent_raw.s;
    double rt[3];
    lsdm::q_rotate<lsdm::Quatd, double>(inv.q, frame->thisToParent_raw.t, rt);
    for (int i = 0; i < 3; i++) rel.t[i] = inv.s * rt[i] + inv.t[i];
  } else {
    lsd_set_error("lsdhip_frame_relative_pose: frame %d and frame %d are neither parent and child nor siblings in the pose tree", reference->id, frame->id);
    return LSDHIP_E_STATE;
  }
  lsdm::q_normalize(rel.q);     // se3FromSim3 -> SO3 constructor (so3.hpp:630-633)
  frameToReference[0] = rel.q.w; frameToReference[1] = rel.q.x; frameToReference[2] = rel.q.y; frameToReference[3] = rel.q.z;
  frameToReference[4] = rel.t[0]; frameToReference[5] = rel.t[1]; frameToReference[6] = rel.t[2];
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_depth_updated(lsdhip_frame* f) { return f ? (f->depthHasBeenUpdatedFlag ? 1 : 0) : LSDHIP_E_ARG; }
extern "C" int lsdhip_frame_clear_depth_updated(lsdhip_frame* f) {
  if (!f) return LSDHIP_E_ARG;
  f->depthHasBeenUpdatedFlag = false;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_stats(lsdhip_frame* f, float out[8]) {
  if (!f || !out) return LSDHIP_E_ARG;
  { int rc = lsd_frame_resolve(f); if (rc) return rc; }   // meanIdepth / numPoints are deferred results
  out[0] = f->initialTrackedResidual; out[1] = f->meanIdepth; out[2] = (float)f->numPoints;
  out[3] = (float)f->numFramesTrackedOnThis; out[4] = (float)f->numMappedOnThis; out[5] = (float)f->numMappedOnThisTotal;
  out[6] = f->depthHasBeenUpdatedFlag ? 1.f : 0.f; out[7] = 0.f;
  return LSDHIP_OK;
}
extern "C" int lsdhip_frame_set_counters(lsdhip_frame* f, int a, int b, int c2, int flag) {
  if (!f) return LSDHIP_E_ARG;
  f->numFramesTrackedOnThis = a; f->numMappedOnThis = b; f->numMappedOnThisTotal = c2; f->depthHasBeenUpdatedFlag = flag != 0;
  return LSDHIP_OK;
}

extern "C" int lsdhip_ref_pointcloud(lsdhip_frame* kf, int level, float* pos, float* colvar, float* grad, int* idx) {
  if (!kf || level < 0 || level >= LSD_LEVELS) { lsd_set_error("lsdhip_ref_pointcloud: bad arguments"); return LSDHIP_E_ARG; }
  if (!kf->hasIDepth) { lsd_set_error("lsdhip_ref_pointcloud: keyframe has no depth"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = kf->ctx;
  LSD_CTX_LOCK(c);
  int w = c->wl[level], h = c->hl[level];
  size_t nmax = (size_t)w * h;
  if (c->pipeline) { if (int rcs = lsd_sync_all(c)) return rcs; }
  char* scratch = nullptr;
  size_t bytes = (size_t)(w + 1) * 4 + nmax * (12 + 8 + 8 + 4) + 1024;
  HIPCHK(hipMalloc((void**)&scratch, bytes));
  int* d_col = (int*)scratch;
  int* d_total = d_col + w;
  float* d_pos = (float*)(scratch + align_up((size_t)(w + 1) * 4, 256));
  float* d_cv = d_pos + nmax * 3;
  float* d_gr = d_cv + nmax * 2;
  int* d_idx = (int*)(d_gr + nmax * 2);
  const LevelIntr& in = c->intr[level];
  hipLaunchKernelGGL(k_pc_count, dim3((w + 255) / 256), dim3(256), 0, c->stream, kf->d_idepth[level], kf->d_idepthVar[level], w, h, d_col);
  hipLaunchKernelGGL(k_pc_scan, dim3(1), dim3(64), 0, c->stream, d_col, w, d_total);
  hipLaunchKernelGGL(k_pc_write, dim3((w + 255) / 256), dim3(256), 0, c->stream, kf->d_idepth[level], kf->d_idepthVar[level],
                     kf->d_image[level], kf->d_grad[level], w, h, in.fxi, in.fyi, in.cxi, in.cyi, d_col, d_pos, d_cv, d_gr, d_idx);
  int total = 0;
  HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (pos) HIPCHK(hipMemcpy(pos, d_pos, (size_t)total * 12, hipMemcpyDeviceToHost));
  if (colvar) HIPCHK(hipMemcpy(colvar, d_cv, (size_t)total * 8, hipMemcpyDeviceToHost));
  if (grad) HIPCHK(hipMemcpy(grad, d_gr, (size_t)total * 8, hipMemcpyDeviceToHost));
  if (idx) HIPCHK(hipMemcpy(idx, d_idx, (size_t)total * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipFree(scratch));
  return total;
}
