// Device code shared by the tracking kernels (k_track_step in tracker.hip): per-point
// arithmetic of K0-K3 in the reference's operation order, the wave-parallel Levenberg-Marquardt step, small reductions.  gfx950 only.
//
// Reference behaviour restated:
//   TrackingReference::makePointCloud   C/Tracking/TrackingReference.cpp:128-138
//   SE3Tracker::calcResidualAndBuffers  C/Tracking/SE3Tracker.cpp:885-1029
//   SE3Tracker::calcWeightsAndResidualSSE  :492-575   (op order of the SSE path; _mm_rcp_ps -> IEEE 1/x)
//   SE3Tracker::calculateWarpUpdateSSE  :1033-1130 + LGS6::updateSSE / finish  C/Tracking/LGSX.h:205-386
//   SE3Tracker::trackFrame              :280-486 (the control flow between two evaluations: lm_wave)
#pragma once
#include "lsdhip_internal.hpp"
#include <type_traits>

// ---- wave64 sum via DPP (row_shr 1,2,3 / 4 / 8, row_bcast 15 / 31); result valid in lane 63 ----------------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}
__device__ __forceinline__ int dpp_max_i(int a, int b) { return a > b ? a : b; }
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_i(int v) {   // masked-off / out-of-row lanes read INT_MIN-like identity (-1)
  return __builtin_amdgcn_update_dpp(-1, v, CTRL, ROW_MASK, BANK_MASK, false);
}
// wave64 max of non-negative keys (identity -1); result valid in lane 63
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_i0(int v) {   // masked-off / out-of-row lanes read 0 (the identity of a sum)
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, true);
}
__device__ __forceinline__ int wave_max_to_lane63(int v) {
  int t = dpp_max_i(v, dpp_i<0x111, 0xf, 0xf>(v));
  t = dpp_max_i(t, dpp_i<0x112, 0xf, 0xf>(v));
  t = dpp_max_i(t, dpp_i<0x113, 0xf, 0xf>(v));
  t = dpp_max_i(t, dpp_i<0x114, 0xf, 0xe>(t));
  t = dpp_max_i(t, dpp_i<0x118, 0xf, 0xc>(t));
  t = dpp_max_i(t, dpp_i<0x142, 0xa, 0xf>(t));
  t = dpp_max_i(t, dpp_i<0x143, 0xc, 0xf>(t));
  return t;
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  float t = v + dpp_f<0x111, 0xf, 0xf>(v);  // row_shr:1
  t = t + dpp_f<0x112, 0xf, 0xf>(v);        // row_shr:2
  t = t + dpp_f<0x113, 0xf, 0xf>(v);        // row_shr:3
  t = t + dpp_f<0x114, 0xf, 0xe>(t);        // row_shr:4 bank_mask:0xe
  t = t + dpp_f<0x118, 0xf, 0xc>(t);        // row_shr:8 bank_mask:0xc
  t = t + dpp_f<0x142, 0xa, 0xf>(t);        // row_bcast:15 row_mask:0xa
  t = t + dpp_f<0x143, 0xc, 0xf>(t);        // row_bcast:31 row_mask:0xc
  return t;
}

// Workgroup-wide top-3 of non-negative unique integer keys.  Every lane contributes up to three candidates sorted
// descending (c0 >= c1 >= c2, -1 = none).  Result in s_out[0..2] (descending, -1 = none) after the final barrier.
__device__ __forceinline__ void block_top3(int c0, int c1, int c2, int (*s_wave)[3], int* s_out) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    int m = __builtin_amdgcn_readlane(wave_max_to_lane63(c0), 63);
    if (lane == 0) s_wave[wave][r] = m;
    if (c0 == m && m >= 0) { c0 = c1; c1 = c2; c2 = -1; }
  }
  __syncthreads();
  if (wave == 0) {
    const int nw = blockDim.x >> 6;
    int v = (lane < nw * 3) ? s_wave[lane / 3][lane % 3] : -1;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      int m = __builtin_amdgcn_readlane(wave_max_to_lane63(v), 63);
      if (lane == 0) s_out[r] = m;
      if (v == m) v = -1;
    }
  }
  __syncthreads();
}
// index of (r, c), c >= r, in the row-major upper triangle of the 6x6 normal matrix, as a constant expression (a running
// counter keeps the accumulator array from being promoted to registers)
__device__ __forceinline__ constexpr int tri_index(int r, int c) { return RS_A0 + r * 6 - (r * (r - 1)) / 2 + (c - r); }

// branch-free (a branchy version gets turned into an indexed store, which drags the three keys into scratch memory)
__device__ __forceinline__ void top3_insert(int k, int& c0, int& c1, int& c2) {
  const int n0 = max(c0, k);
  int t = min(c0, k);
  const int n1 = max(c1, t);
  t = min(c1, t);
  c2 = max(c2, t);
  c1 = n1;
  c0 = n0;
}

// ---- per-point arithmetic -----------------------------------------------------------------------------------------
// Pointers into HBM carry the global address space explicitly.  A job description that lives in memory (batches) hands the
// kernel generic pointers, and loads through a generic pointer are FLAT instructions: they count against the LDS counter
// (lgkmcnt) as well as vmcnt, so every wait for an LDS read also drains the global loads a software pipeline wants in flight.
typedef const __attribute__((address_space(1))) float gfloat;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f gv4f;
typedef __attribute__((address_space(1))) uint8_t gbyte;
struct EvalCtx {            // one level of a job + the pose under evaluation, in registers / SGPRs
  gfloat* kf_idepth;
  gfloat* kf_idepthVar;
  gfloat* kf_image;
  gfloat* fr_grad;          // texels (gx, gy, I, 0): 4 floats per pixel
  gfloat* pts_pos;
  gfloat* pts_colvar;
  int npts, w, h;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float R[9], t[3];
  float aff_a, aff_b;
  float cameraPixelNoise2, var_weight, huber_half;
};
struct PointOut {
  bool in_image;
  bool good;
  float res, c1, c2, hw;      // residual, affine terms, Huber weight of the affine estimator
  float usage;                // min(1, z_ref / z_new)
  float werr;                 // wh * w_p * r^2 (K2)
  float w;                    // wh * w_p
  float J[6];
};

// The evaluation of one point in three steps, so that a caller may interleave the steps of different points (the batch
// throughput mode keeps the texel fetches of the next point in flight while it finishes the current one).  eval_point
// below is the plain composition; operation order is that of the reference (see the comments inside).
struct PointWarp { float Wx, Wy, Wz, u_new, v_new; bool in_image; };
struct Texel3 { float x, y, z; };   // (gx, gy, I) of a 16-byte texel; the fourth word is never read
struct PointTexels { Texel3 t00, t10, t01, t11; };
__device__ __forceinline__ void eval_warp(const EvalCtx& a, float px, float py, float pz, PointWarp& q) {
  // Wxp = rotMat * p + transVec (Eigen coefficient product: ((r0*x + r1*y) + r2*z), then + t)
  q.Wx = ((a.R[0] * px + a.R[1] * py) + a.R[2] * pz) + a.t[0];
  q.Wy = ((a.R[3] * px + a.R[4] * py) + a.R[5] * pz) + a.t[1];
  q.Wz = ((a.R[6] * px + a.R[7] * py) + a.R[8] * pz) + a.t[2];
  q.u_new = (q.Wx / q.Wz) * a.fx + a.cx;
  q.v_new = (q.Wy / q.Wz) * a.fy + a.cy;
  q.in_image = (q.u_new > 1 && q.v_new > 1 && q.u_new < a.w - 2 && q.v_new < a.h - 2);
}
// the four texels of getInterpolatedElement43 (C/util/globalFuncs.h:63-77); `fetch` = false reads texel 0 instead
__device__ __forceinline__ void eval_fetch(const EvalCtx& a, const PointWarp& q, bool fetch, PointTexels& t) {
  const int ix = fetch ? (int)q.u_new : 0;
  const int iy = fetch ? (int)q.v_new : 0;
  gfloat* bp = a.fr_grad + 4 * (ix + __mul24(iy, a.w));     // 24-bit multiply: full rate (a 32-bit integer multiply is quarter rate)
  auto ld = [](gfloat* f) { Texel3 r = {f[0], f[1], f[2]}; return r; };
  t.t00 = ld(bp); t.t10 = ld(bp + 4); t.t01 = ld(bp + 4 * a.w); t.t11 = ld(bp + 4 + 4 * a.w);
}
// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32) for the quantities that only feed sums held to a tolerance —
// K2's weights and K3's Jacobian (the reference's SSE path uses the 12-bit _mm_rcp_ps there, SE3Tracker.cpp:519-553), the usage and
// affine-lighting statistics.  Everything that decides a mask bit or a count (the projection, the bilinear sample, isGood) stays
// IEEE-exact: an IEEE division is a ~12-instruction dependent chain, and eleven of them made up a third of the evaluation.
// LSD_EVAL_FMA (default 1): the multiply-adds of K2's weight and K3's Jacobian formulas fused (round 5).  They only feed sums held to a
// tolerance — but the weighted error drives the LM loop's accept / reject test, so the reference's separately rounded order stays
// buildable: lsd_slam_amd/build.py build_variant("nofma", ["LSD_EVAL_FMA=0"]), and tests/test_gpu_parity.py::test_unfused_build_holds_the_same_parity
// runs the kernel-level and trackFrame parity tests on that build so that a regression can be bisected between the two.
#ifndef LSD_EVAL_FMA
#define LSD_EVAL_FMA 1
#endif
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ void eval_finish(const EvalCtx& a, const PointWarp& q, const PointTexels& t, float pz, float I_ref, float var,
                                            PointOut& o) {
  const float Wx = q.Wx, Wy = q.Wy, Wz = q.Wz, u_new = q.u_new, v_new = q.v_new;
  const Texel3 t00 = t.t00, t10 = t.t10, t01 = t.t01, t11 = t.t11;
  int ix = (int)u_new;
  int iy = (int)v_new;
  float dx = u_new - ix;
  float dy = v_new - iy;
  float dxdy = dx * dy;
  float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  float rx = w11 * t11.x + w01 * t01.x + w10 * t10.x + w00 * t00.x;
  float ry = w11 * t11.y + w01 * t01.y + w10 * t10.y + w00 * t00.y;
  float rz = w11 * t11.z + w01 * t01.z + w10 * t10.z + w00 * t00.z;

  float c1 = a.aff_a * I_ref + a.aff_b;
  float c2 = rz;
  float residual = c1 - c2;
  o.res = residual; o.c1 = c1; o.c2 = c2;
  o.hw = fabsf(residual) < 5.0f ? 1 : 5.0f * frcp(fabsf(residual));
  // isGood = residual^2 / (MAX_DIFF_CONSTANT + MAX_DIFF_GRAD_MULT * |grad|^2) < 1 (SE3Tracker.cpp:996-1001): for a positive finite
  // denominator the rounded quotient is below 1 exactly when the numerator is below the denominator (the largest quotient of
  // floats x < y is 1 - ulp/2, which is representable), so the comparison needs no division
  o.good = residual * residual < (40.0f * 40.0f + 0.5f * 0.5f * (rx * rx + ry * ry));
  float gx = a.fx * rx;   // buf_warped_dx
  float gy = a.fy * ry;   // buf_warped_dy
  float d = frcp(pz);     // buf_d
  float z = frcp(Wz);
  float depthChange = pz * z;
  o.usage = depthChange < 1 ? depthChange : 1;

  // K2 / K3: the reference's formulas (SSE operation order) with the multiply-adds fused.  Like the 1-ulp reciprocals above, these
  // quantities only feed sums held to a tolerance (the weights and the Jacobian: the reference's own SSE path computes them from 12-bit
  // reciprocals); a fused multiply-add is one rounding and one instruction where the separate pair is two of each, and this loop is
  // bound by instruction issue (profiles/r05_notes.md).  Everything above that decides a mask bit or a count is untouched.
  float pz2d = frcp((Wz * Wz) * d);
#if LSD_EVAL_FMA
  float g0 = __builtin_fmaf(Wz, a.t[0], 0.0f - Wx * a.t[2]) * pz2d;
  float g1 = __builtin_fmaf(Wz, a.t[1], 0.0f - Wy * a.t[2]) * pz2d;
  float drpdd = __builtin_fmaf(g0, gx, g1 * gy);
  float w_p = frcp(__builtin_fmaf(drpdd, drpdd * (a.var_weight * var), a.cameraPixelNoise2));
#else
  // the reference's SSE operation order with every multiply and add rounded on its own (calcWeightsAndResidualSSE, SE3Tracker.cpp:519-553)
  float g0 = (Wz * a.t[0] - Wx * a.t[2]) * pz2d;
  float g1 = (Wz * a.t[1] - Wy * a.t[2]) * pz2d;
  float drpdd = g0 * gx + g1 * gy;
  float w_p = frcp(a.cameraPixelNoise2 + drpdd * (drpdd * (a.var_weight * var)));
#endif
  float wr = residual * fsqrt(w_p);
  wr = fmaxf(wr, 0.0f - wr);
  float wh = (wr < a.huber_half) ? 1.0f : a.huber_half * frcp(wr);
  o.werr = wh * (wr * wr);
  o.w = wh * w_p;

  o.J[0] = z * gx;
  o.J[1] = z * gy;
  float v1 = (Wx * gy) * z;
  float v2 = (Wy * gx) * z;
  o.J[5] = v1 - v2;
  float z2 = z * z;
  v1 = (Wx * gx) * z2;
  v2 = (Wy * gy) * z2;
  o.J[2] = 0.0f - (v1 + v2);
#if LSD_EVAL_FMA
  o.J[3] = 0.0f - __builtin_fmaf(v2, Wy, __builtin_fmaf(v1, Wy, gy));
  o.J[4] = __builtin_fmaf(v2, Wx, __builtin_fmaf(v1, Wx, gx));
#else
  o.J[3] = 0.0f - ((v2 * Wy) + (gy + v1 * Wy));        // calculateWarpUpdateSSE, SE3Tracker.cpp:1080-1104
  o.J[4] = (gx + v1 * Wx) + v2 * Wx;
#endif
}
// (px,py,pz) = reference point, I_ref / var = its colour and inverse-depth variance.
__device__ __forceinline__ void eval_point(const EvalCtx& a, float px, float py, float pz, float I_ref, float var, PointOut& o) {
  PointWarp q;
  eval_warp(a, px, py, pz, q);
  o.in_image = q.in_image;
  if (!o.in_image) return;
  PointTexels t;
  eval_fetch(a, q, true, t);
  eval_finish(a, q, t, pz, I_ref, var, o);
}

// fetch the reference point `i` (dense index into the keyframe level, or index into the explicit list)
__device__ __forceinline__ bool fetch_point(const EvalCtx& a, int i, float& px, float& py, float& pz, float& I_ref,
                                            float& var, int& maskIdx) {
  if (a.npts >= 0) {
    if (i >= a.npts) return false;
    px = a.pts_pos[3 * i]; py = a.pts_pos[3 * i + 1]; pz = a.pts_pos[3 * i + 2];
    I_ref = a.pts_colvar[2 * i]; var = a.pts_colvar[2 * i + 1];
    maskIdx = -1;
    return true;
  }
  if (i >= a.w * a.h) return false;
  int x = i % a.w, y = i / a.w;
  if (x < 1 || x >= a.w - 1 || y < 1 || y >= a.h - 1) return false;
  var = a.kf_idepthVar[i];
  float id = a.kf_idepth[i];
  if (var <= 0 || id == 0) return false;
  float inv = lsd_rcp_exact(id);
  px = inv * (a.fxi * x + a.cxi);
  py = inv * (a.fyi * y + a.cyi);
  pz = inv * 1.0f;
  I_ref = a.kf_image[i];
  maskIdx = i;
  return true;
}

// ---- Levenberg-Marquardt step, wave-parallel ------------------------------------------------------------------------
// Runs in wave 0 of every workgroup with all 64 lanes active.  "Uniform" values are computed redundantly by every lane
// (same inputs, same instructions); the 6x6 factorisation keeps row i of the matrix in lane i and exchanges data with
// v_readlane; the two sincos evaluations and the four quaternion divisions of a normalisation run in different lanes.
__device__ __forceinline__ void set_eval_pose(TrackState& s, const lsdm::SE3fH& T) {
  s.Tn = T;
  lsdm::quatf_to_rot(T.q, s.R);
  s.t[0] = T.t[0]; s.t[1] = T.t[1]; s.t[2] = T.t[2];
}
__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rli(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// 6x6 solve for the LM step, Gauss-Jordan on the augmented 6x7 system with ONE ELEMENT PER LANE (lane = 8 i + j, i < 6 rows,
// j < 7 columns) and the pivot row / pivot column read back through LDS: six steps of (write own element, read pivot, pivot-row
// and pivot-column entries, divide, multiply-subtract) instead of the ~460 dependent lane-exchange instructions of the
// row-per-lane LDL^T above (3800 -> ~1300 cycles of the single wave the whole launch waits for).  No pivoting: the matrix
// is J^T W J with its diagonal scaled by (1 + lambda), symmetric positive definite.  Same solution as A.ldlt().solve(b) up
// to rounding (tolerance-level, like every reduction feeding it); the pivoted LDL^T stays in pose_math.hpp for the host
// paths and as the CPU-checked reference of this routine (tests/test_host_math_cpu.py).
__device__ __forceinline__ void gj6_solve_wave(const float* A /*LDS, 6x6*/, const float* bvec /*LDS*/, const float damp, float* s_m /*LDS [6][8]*/,
                                               const int lane, float (&x)[6]) {
  const int i = lane >> 3, j = lane & 7;
  const bool act = i < 6 && j < 7;
  const int ii = act ? i : 0, jj = act ? (j < 6 ? j : 0) : 0;
  float m = (j == 6) ? -bvec[ii] : A[ii * 6 + jj];
  if (i == j) m *= damp;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (act) s_m[i * 8 + j] = m;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float d = s_m[k * 8 + k], rk = s_m[k * 8 + (act ? j : 0)], ck = s_m[ii * 8 + k];
    const float f = ck * frcp(d);                  // 1-ulp reciprocal: the solve is tolerance-level by construction (see above)
    const float upd = m - f * rk;
    m = (i == k) ? m : upd;
    __builtin_amdgcn_wave_barrier();
  }
  // now diagonal: x_i = rhs_i / m_ii
  if (act) s_m[i * 8 + j] = m;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < 6; r++) x[r] = s_m[r * 8 + 6] * frcp(s_m[r * 8 + r]);
}

// sin / cos for the small angles of an LM increment: Taylor polynomials (|x| < 0.5: truncation error < 2e-10 relative),
// the library routine beyond
__device__ __forceinline__ void sincos_small(const float xx, float* sn, float* cs) {
  if (fabsf(xx) < 0.5f) {
    const float x2 = xx * xx;
    *sn = xx * (1.0f + x2 * (-1.0f / 6.0f + x2 * (1.0f / 120.0f + x2 * (-1.0f / 5040.0f + x2 * (1.0f / 362880.0f + x2 * (-1.0f / 39916800.0f))))));
    *cs = 1.0f + x2 * (-0.5f + x2 * (1.0f / 24.0f + x2 * (-1.0f / 720.0f + x2 * (1.0f / 40320.0f + x2 * (-1.0f / 3628800.0f + x2 * (1.0f / 479001600.0f))))));
  } else {
    sincosf(xx, sn, cs);
  }
}

// Quaternion normalisation, the four divisions in four lanes (lsdm::q_normalize arithmetic)
__device__ __forceinline__ void q_normalize_wave(lsdm::Quatf& q, const int lane) {
  const float rn = __builtin_amdgcn_rsqf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);   // 1 / norm, 1 ulp
  const float c = (lane & 3) == 0 ? q.w : ((lane & 3) == 1 ? q.x : ((lane & 3) == 2 ? q.y : q.z));
  const float r = c * rn;
  q.w = rl(r, 0); q.x = rl(r, 1); q.y = rl(r, 2); q.z = rl(r, 3);
}

// Sophus SE3Group<float>::exp (lsdm::se3f_exp arithmetic), sincos(theta/2) in even lanes and sincos(theta) in odd lanes
__device__ __forceinline__ lsdm::SE3fH se3f_exp_wave(const float (&a)[6], const int lane) {
  const float eps = static_cast<float>(1e-5);
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = ox * ox + (oy * oy + oz * oz);
  const float theta = fsqrt(theta_sq);
  const float half_theta = 0.5f * theta;
  float sn, cs;
  sincos_small((lane & 1) ? theta : half_theta, &sn, &cs);
  const float sin_half = rl(sn, 0), cos_half = rl(cs, 0), sin_theta = rl(sn, 1), cos_theta = rl(cs, 1);
  float imag, real;
  if (theta < eps) {
    const float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - static_cast<float>(1.0 / 48.0) * theta_sq + static_cast<float>(1.0 / 3840.0) * theta_po4;
    real = 1.0f - 0.5f * theta_sq + static_cast<float>(1.0 / 384.0) * theta_po4;
  } else {
    imag = sin_half * frcp(theta);
    real = cos_half;
  }
  lsdm::SE3fH r;
  r.q = {real, imag * ox, imag * oy, imag * oz};
  q_normalize_wave(r.q, lane);
  const float Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  float Om2[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float acc = Om[i * 3 + 0] * Om[0 * 3 + j];
      acc += Om[i * 3 + 1] * Om[1 * 3 + j];
      acc += Om[i * 3 + 2] * Om[2 * 3 + j];
      Om2[i * 3 + j] = acc;
    }
  float V[9];
  if (theta < eps) {
    lsdm::q_to_rot<lsdm::Quatf, float>(r.q, V);
  } else {
    const float tsq = theta * theta;
    const float ca = (1.0f - cos_theta) * frcp(tsq);
    const float cb = (theta - sin_theta) * frcp(tsq * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = (((i % 4) == 0 ? 1.0f : 0.0f) + ca * Om[i]) + cb * Om2[i];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float acc = V[i * 3 + 0] * a[0];
    acc += V[i * 3 + 1] * a[1];
    acc += V[i * 3 + 2] * a[2];
    r.t[i] = acc;
  }
  return r;
}
// Sophus operator*: fastMultiply + normalize (lsdm::se3f_mul arithmetic)
__device__ __forceinline__ lsdm::SE3fH se3f_mul_wave(const lsdm::SE3fH& a, const lsdm::SE3fH& b, const int lane) {
  lsdm::SE3fH r = a;
  float rt[3];
  lsdm::q_rotate<lsdm::Quatf, float>(a.q, b.t, rt);
#pragma unroll
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.q = lsdm::q_mul(a.q, b.q);
  q_normalize_wave(r.q, lane);
  return r;
}

// The summary in pinned host memory: every field is stored AND added (position-weighted) into `check`, then a system-scope release fence,
// then the `done` word.  The host polls `done` and then validates `check` (summary_wait_consistent, tracker.hip): one in a few thousand
// polled jobs the tail of the record (levelEvals / numLaunches / lastCand — the stores issued last) was still the previous job's when
// `done` had already arrived.
__device__ __forceinline__ void write_summary(const TrackState& s, const float* tot, TrackSummary* out, const int doneWord) {
  unsigned chk = (unsigned)doneWord;
#define SUM_PUT_I(field, value) do { const int v_ = (value); out->field = v_; chk += lsd_summary_term((unsigned)(offsetof(TrackSummary, field) / 4), (unsigned)v_); } while (0)
#define SUM_PUT_F(field, value) do { const float v_ = (value); out->field = v_; chk += lsd_summary_term((unsigned)(offsetof(TrackSummary, field) / 4), __float_as_uint(v_)); } while (0)
#define SUM_PUT_IA(field, i, value) do { const int v_ = (value); out->field[i] = v_; chk += lsd_summary_term((unsigned)(offsetof(TrackSummary, field) / 4 + (i)), (unsigned)v_); } while (0)
#define SUM_PUT_FA(field, i, value) do { const float v_ = (value); out->field[i] = v_; chk += lsd_summary_term((unsigned)(offsetof(TrackSummary, field) / 4 + (i)), __float_as_uint(v_)); } while (0)
  SUM_PUT_I(diverged, s.diverged); SUM_PUT_I(level, s.level); SUM_PUT_I(numEvaluations, s.numEvaluations); SUM_PUT_I(numWarpUpdates, s.numWarpUpdates);
  SUM_PUT_IA(pad_, 0, 0); SUM_PUT_IA(pad_, 1, 0); SUM_PUT_IA(pad_, 2, 0);
  SUM_PUT_FA(q, 0, s.T.q.w); SUM_PUT_FA(q, 1, s.T.q.x); SUM_PUT_FA(q, 2, s.T.q.y); SUM_PUT_FA(q, 3, s.T.q.z);
  SUM_PUT_FA(t, 0, s.T.t[0]); SUM_PUT_FA(t, 1, s.T.t[1]); SUM_PUT_FA(t, 2, s.T.t[2]);
  SUM_PUT_F(lastResidual, s.last_residual); SUM_PUT_F(pointUsage, s.pointUsage); SUM_PUT_F(goodCount, s.goodCount); SUM_PUT_F(badCount, s.badCount);
  SUM_PUT_F(meanRes, s.meanRes); SUM_PUT_F(aff_a, s.aff_a); SUM_PUT_F(aff_b, s.aff_b); SUM_PUT_F(aff_a_lastIt, s.aff_a_lastIt); SUM_PUT_F(aff_b_lastIt, s.aff_b_lastIt);
  for (int i = 0; i < RS_NUM; i++) SUM_PUT_FA(sums, i, tot[i]);
  {
    const double b = (double)s.bytes;
    out->bytes = b;
    const unsigned long long u = (unsigned long long)__double_as_longlong(b);
    chk += lsd_summary_term((unsigned)(offsetof(TrackSummary, bytes) / 4), (unsigned)u) + lsd_summary_term((unsigned)(offsetof(TrackSummary, bytes) / 4 + 1), (unsigned)(u >> 32));
  }
  for (int l = 0; l < LSD_LEVELS; l++) SUM_PUT_IA(levelEvals, l, s.levelEvals[l]);
  SUM_PUT_I(numLaunches, s.numLaunches);
  SUM_PUT_I(lastCand, s.lastCand);
#undef SUM_PUT_I
#undef SUM_PUT_F
#undef SUM_PUT_IA
#undef SUM_PUT_FA
  out->check = chk;
  __threadfence_system();
  out->done = doneWord;
}

struct LmShared {
  float gj[48];        // scratch of the 6x6 solve
  float tot[RS_NUM];   // corrected raw sums of the evaluation being finished (what the summary reports)
};
// Job parameters the LM wave needs, staged in LDS by an otherwise idle lane while the partial sums are being added
// (fetched field by field from the kernel-argument segment they cost one scalar-cache round trip each).
struct LmPar {
  float lambdaInitial, stepSizeMin, convergenceEps, minWarped, lambdaSuccessFac, lambdaFailFac;
  int maxIts, w, h, writeMask, evalOnly, useAffine, tfSemantics, lastLevel, trials;
};
__device__ __forceinline__ void stage_lm_par(const TrackJob& job, int level, LmPar& p, const int trials = 1) {
  const TrackLevel& L = job.lv[level];
  LmPar v;
  v.lambdaInitial = L.lambdaInitial; v.stepSizeMin = L.stepSizeMin; v.convergenceEps = L.convergenceEps; v.minWarped = L.minWarped;
  v.lambdaSuccessFac = job.lambdaSuccessFac; v.lambdaFailFac = job.lambdaFailFac;
  v.maxIts = L.maxIts; v.w = L.w; v.h = L.h; v.writeMask = L.writeMask; v.evalOnly = job.evalOnly; v.useAffine = job.useAffine;
  v.tfSemantics = job.trackFrameSemantics; v.lastLevel = job.lastLevel;
  v.trials = trials;
  p = v;
}

// The control flow of SE3Tracker::trackFrame between two evaluations (SE3Tracker.cpp:324-447), run by wave 0 of every
// workgroup on identical inputs (so every workgroup reaches the same decision without talking to the others).
//   col  : this lane's column total of the evaluation's raw sums, tail-drop corrected (lane c < RS_END <-> column c)
//   S    : the job state in LDS (read and updated in place; every lane writes the same values)
//   tot  : LDS copy of the corrected sums (what the summary reports)
// `out` is non-null in workgroup 0 only.
#ifdef LSD_PHASE_TRACE
#define LM_MARK(k) do { if (trp && lane == 0) trp[k] = clock64(); } while (0)
#else
#define LM_MARK(k) do { } while (0)
#endif
// Reject-chain speculation: `consumed` = index (within the launch that produced the sums) of the trial being finished (the
// caller has already advanced lambda / incTry / counters past the plain rejections before it), `mycand` = the trial this
// workgroup evaluates next: `mycand` retries further down the chain that starts at the proposal made here.
// calcWeightsAndResidualSSE's return value: the weighted error over the points its loop visits (SE3Tracker.cpp:572-574).  One
// definition: the speculation scans and lm_wave must agree on it bit for bit.
__device__ __forceinline__ float lm_werr(float sumWeightedErr, int M) { return sumWeightedErr * frcp((float)((M >> 2) << 2)); }
__device__ __forceinline__ float lm_lambda_fail(float LM_lambda, int incTry, float lambdaFailFac) {
  if (LM_lambda == 0) return 0.2f;
  double p = 1.0;
  for (int i = 0; i < incTry; i++) p *= (double)lambdaFailFac;   // std::pow(lambdaFailFac, incTry)
  return (float)((double)LM_lambda * p);
}
template <bool SPEC = false>
__device__ __forceinline__ bool lm_wave(const LmPar& par, TrackState& S, const float col, float* tot, const int lane,
                                        TrackSummary* out, unsigned long long* trp, const int consumed = 0, const int mycand = 0, const int doneWord = 1) {
  float* const s_gj = tot - 48;   // LmShared::gj precedes tot
  S.lastCand = consumed;
  S.ncand = 1;
  const LmPar L = par;
  const int maxIts = L.maxIts;
  // calcResidualAndBuffers epilogue (:1016-1028)
  const int M = (int)rl(col, RS_M);
  const float refNum = rl(col, RS_NREF);
  const float goodCount = rl(col, RS_GOOD), badCount = rl(col, RS_BAD);
  float aff_a_lastIt, aff_b_lastIt;
  {
    const float sxx = rl(col, RS_SXX), syy = rl(col, RS_SYY), sx = rl(col, RS_SX), sy = rl(col, RS_SY), sw = rl(col, RS_SW);
    // (1-ulp reciprocals / square roots throughout the LM step: one wave, every dependent IEEE division is ~100 cycles the whole
    // launch waits for; the quantities are reductions and a pose increment, all held to a tolerance)
    const float isw = frcp(sw);
    aff_a_lastIt = fsqrt((syy - sy * sy * isw) * frcp(sxx - sx * sx * isw));
    aff_b_lastIt = (sy - aff_a_lastIt * sx) * isw;
  }
  S.pointUsage = rl(col, RS_USAGE) * frcp(refNum);
  S.goodCount = goodCount;
  S.badCount = badCount;
  S.meanRes = rl(col, RS_SUMSIGNED) * frcp(goodCount);
  S.aff_a_lastIt = aff_a_lastIt;
  S.aff_b_lastIt = aff_b_lastIt;
  S.numEvaluations = S.numEvaluations + 1;
  if (lane == 0) S.levelEvals[S.level] = S.levelEvals[S.level] + 1;
  {
    // algorithmic bytes of this evaluation: 20 N + [mask] 5 N + 12 min(w h, 4 N)
    const float N = refNum, wh = (float)L.w * (float)L.h;
    const float texels = 4.0f * N < wh ? 4.0f * N : wh;
    S.bytes = S.bytes + (20.0f * N + (L.writeMask ? 5.0f * N : 0.0f) + 12.0f * texels);
  }
  S.pending = 0;
  LM_MARK(12);
  if (L.evalOnly) { S.done = 1; if (out && lane == 0) write_summary(S, tot, out, doneWord); return false; }

  if (M < L.minWarped) {   // :324-329 / :369-374
    S.diverged = 1; S.done = 1;
    if (out && lane == 0) write_summary(S, tot, out, doneWord);
    return false;
  }
  // calcWeightsAndResidualSSE epilogue (:572-574)
  const float werr = lm_werr(rl(col, RS_WERR), M);
  const bool useAffine = L.useAffine != 0;
  const bool tfSemantics = L.tfSemantics != 0;

  float LM_lambda = S.LM_lambda, lastErr = S.lastErr;
  int iteration = S.iteration, incTry = S.incTry;
  lsdm::SE3fH T = S.T;
  bool propose = false, start_iteration = false, accepted = false, rejected = false;
  if (S.phase == 0) {
    accepted = true;
    lastErr = werr;
    LM_lambda = L.lambdaInitial;
    iteration = 0;
    start_iteration = true;
  } else {
    const float error = werr;
    if (error < lastErr) {
      accepted = true;
      T = S.Tn;
      S.T = T;
      if (error > L.convergenceEps * lastErr) iteration = maxIts;     // error / lastErr > convergenceEps (both positive)
      lastErr = error;
      if (tfSemantics) S.last_residual = error;
      if (LM_lambda <= 0.2) LM_lambda = 0;
      else LM_lambda *= L.lambdaSuccessFac;
      iteration++;
      start_iteration = true;
    } else {
      const float i0 = S.inc[0], i1 = S.inc[1], i2 = S.inc[2], i3 = S.inc[3], i4 = S.inc[4], i5 = S.inc[5];
      const float incdot = (i0 * i0 + (i1 * i1 + i2 * i2)) + (i3 * i3 + (i4 * i4 + i5 * i5));
      if (!(incdot > L.stepSizeMin)) {
        iteration = maxIts;
        iteration++;
        start_iteration = true;
      } else {
        LM_lambda = lm_lambda_fail(LM_lambda, incTry, L.lambdaFailFac);
        propose = true;
        rejected = true;
      }
    }
  }
  if (accepted) {
    if (useAffine) { S.aff_a = aff_a_lastIt; S.aff_b = aff_b_lastIt; }
    // the accepted (or first) evaluation's normal equations are what calculateWarpUpdate would build next: LGS6::finish
    // (A / n, b / n with the SSE constraint count n = 6 (M / 4), LGSX.h:319-325, :385), one entry per lane
    const float rn = frcp((float)((size_t)6 * (size_t)(M >> 2)));
    const int k = lane - RS_A0;
    if (k >= 0 && k < 21) {
      const int i = (k >= 6) + (k >= 11) + (k >= 15) + (k >= 18) + (k >= 20);
      const int j = k - (i * 6 - (i * (i - 1)) / 2) + i;
      const float v = (0.0f + col) * rn;
      S.A[i * 6 + j] = v;
      S.A[j * 6 + i] = v;
    }
    if (lane >= RS_B0 && lane < RS_B0 + 6) S.b[lane - RS_B0] = (0.0f - col) * rn;
  }
  if (start_iteration && iteration < maxIts) { S.numWarpUpdates = S.numWarpUpdates + 1; incTry = 0; propose = true; }
  S.lastErr = lastErr;
  S.LM_lambda = LM_lambda;
  S.iteration = iteration;
  LM_MARK(13);
  if (propose) {
    // this workgroup's own trial: `mycand` retries further down the chain (each retry: incTry++, lambda as after a rejection)
    const int extra = SPEC ? mycand : 0;
    for (int j = 0; j < extra; j++) { incTry++; LM_lambda = lm_lambda_fail(LM_lambda, incTry, L.lambdaFailFac); }
    S.ncand = L.trials;
    float inc[6];
    const float damp = 1 + LM_lambda;
    LM_MARK(14);
    gj6_solve_wave(S.A, S.b, damp, s_gj, lane, inc);
    LM_MARK(15);
    S.incTry = incTry + 1;
#pragma unroll
    for (int i = 0; i < 6; i++) S.inc[i] = inc[i];
    const lsdm::SE3fH Tn = se3f_mul_wave(se3f_exp_wave(inc, lane), T, lane);
    S.Tn = Tn;
    float R[9];
    lsdm::quatf_to_rot(Tn.q, R);
#pragma unroll
    for (int i = 0; i < 9; i++) S.R[i] = R[i];
    S.t[0] = Tn.t[0]; S.t[1] = Tn.t[1]; S.t[2] = Tn.t[2];
    S.phase = 1;
    LM_MARK(16);
    return true;
  }
  S.incTry = incTry;
  // level finished
  if (!tfSemantics) S.last_residual = lastErr;   // trackFrameOnPermaref: lastResidual = lastErr (:265)
  if (S.level == L.lastLevel) {
    S.done = 1;
    if (out && lane == 0) write_summary(S, tot, out, doneWord);
  } else {
    S.level = S.level - 1;
    S.phase = 0;
    S.Tn = T;
    float R[9];
    lsdm::quatf_to_rot(T.q, R);
#pragma unroll
    for (int i = 0; i < 9; i++) S.R[i] = R[i];
    S.t[0] = T.t[0]; S.t[1] = T.t[1]; S.t[2] = T.t[2];
  }
  return false;
}

// one in-image point's contribution to the running sums (K1 statistics, K2 weighted error, K3 normal equations)
__device__ __forceinline__ void accumulate_point(const PointOut& o, float (&acc)[RS_END]) {
  // running sums: the product is fused into the addition (one rounding instead of two, one instruction instead of two) — these are
  // reductions held to a tolerance, not per-pixel outputs; the counts stay plain additions of 1
  acc[RS_M] += 1.f;
  acc[RS_SXX] = __builtin_fmaf(o.c1 * o.c1, o.hw, acc[RS_SXX]);
  acc[RS_SYY] = __builtin_fmaf(o.c2 * o.c2, o.hw, acc[RS_SYY]);
  acc[RS_SX] = __builtin_fmaf(o.c1, o.hw, acc[RS_SX]);
  acc[RS_SY] = __builtin_fmaf(o.c2, o.hw, acc[RS_SY]);
  acc[RS_SW] += o.hw;
  if (o.good) { acc[RS_GOOD] += 1.f; acc[RS_SUMRES2] = __builtin_fmaf(o.res, o.res, acc[RS_SUMRES2]); acc[RS_SUMSIGNED] += o.res; }
  else acc[RS_BAD] += 1.f;
  acc[RS_USAGE] += o.usage;
  acc[RS_WERR] += o.werr;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    float Jw = o.J[r] * o.w;
#pragma unroll
    for (int c = r; c < 6; c++) acc[tri_index(r, c)] = __builtin_fmaf(Jw, o.J[c], acc[tri_index(r, c)]);
  }
  float resw = o.res * o.w;
#pragma unroll
  for (int r = 0; r < 6; r++) acc[RS_B0 + r] = __builtin_fmaf(resw, o.J[r], acc[RS_B0 + r]);
  acc[RS_ERR] = __builtin_fmaf(resw, o.res, acc[RS_ERR]);
}

__device__ __forceinline__ void make_ctx_dev(const TrackJob& jobr, const TrackState& S, int level, EvalCtx& a) {
  const TrackJob* job = &jobr;
  const TrackLevel& L = job->lv[level];
  a.kf_idepth = (gfloat*)L.kf_idepth; a.kf_idepthVar = (gfloat*)L.kf_idepthVar; a.kf_image = (gfloat*)L.kf_image; a.fr_grad = (gfloat*)L.fr_grad;
  a.pts_pos = (gfloat*)L.pts_pos; a.pts_colvar = (gfloat*)L.pts_colvar; a.npts = L.npts; a.w = L.w; a.h = L.h;
  a.fx = L.fx; a.fy = L.fy; a.cx = L.cx; a.cy = L.cy; a.fxi = L.fxi; a.fyi = L.fyi; a.cxi = L.cxi; a.cyi = L.cyi;
#pragma unroll
  for (int i = 0; i < 9; i++) a.R[i] = S.R[i];
#pragma unroll
  for (int i = 0; i < 3; i++) a.t[i] = S.t[i];
  a.aff_a = S.aff_a; a.aff_b = S.aff_b;
  a.cameraPixelNoise2 = job->cameraPixelNoise2; a.var_weight = job->var_weight; a.huber_half = job->huber_half;
}

template <int N>
__device__ __forceinline__ void copy_words(void* dst, const void* src, int tid, int nthreads) {
  const unsigned* s = (const unsigned*)src;
  unsigned* d = (unsigned*)dst;
  for (int i = tid; i < N; i += nthreads) d[i] = s[i];
}

// logical tile of workgroup b among nb participating ones: workgroups are dealt round-robin to the 8 XCDs, so giving
// XCD x the contiguous tile range [x nb/8, (x+1) nb/8) keeps the texels one band of the frame touches in one L2.
__device__ __forceinline__ int xcd_tile(int b, int nb) { return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b; }


#define RS_COLS 44   // RS_END rounded up: floats per partial-sum row
// Scratch of a tracker in HBM, double-buffered by launch parity (a launch reads [parity], writes [1 - parity]).
struct TrackScratch {
  float* sums;     // [2][RS_COLS][max_rows]  column-major partial sums: one row per workgroup tile
  int4* topkey;    // [2][max_rows]           each tile's three largest reference-order keys among in-image points (x>=y>=z)
  float* topval;   // [2][max_rows][3][32]    K2/K3 contributions of those points (single-pass levels only)
  int max_rows;    // multiple of 4
  int cmax;        // trial slots per parity (reject-chain speculation): sums [2][cmax][RS_COLS][max_rows], topkey / topval likewise
  float* recs;     // [2][cmax][32]  increment / pose of the trials > 0 of a launch (null when cmax == 1)
#ifdef LSD_PHASE_TRACE
  unsigned long long* trace;   // [0] = launch counter, then 20 words per launch (developer build only, tools/phase_trace.py)
#endif
};
