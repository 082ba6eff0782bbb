// Sim3Tracker on the device.  gfx950 only.
//
// Reference behaviour restated (C/ = lsd_slam_core/src/):
//   Sim3Tracker::calcSim3Buffers                 C/Tracking/Sim3Tracker.cpp:414-607
//   Sim3Tracker::calcSim3WeightsAndResidualSSE   :611-736   (_mm_rcp_ps -> IEEE 1/x)
//   Sim3Tracker::calcSim3LGSSSE                  :858-983 + LGS4 / LGS6 / LGS7  C/Tracking/LGSX.h:45-176, :184-402, :411-443
//   Sim3Tracker::trackFrameSim3                  :149-378
//
// One evaluation = buffers + weights + both least-squares systems fused (sim3_eval_strip: 1-10 reference pixels per lane, 54 sums
// reduced lane -> LDS -> one row per strip) + fixed-order row sums with the SSE tail drop (sim3_totals).
// trackFrameSim3 runs on the device, one launch per evaluation (k_sim3_fused, built like k_track_step of the SE3 tracker): every
// workgroup first totals the previous launch's rows, takes the Levenberg-Marquardt decision (the loop of Sim3Tracker.cpp:149-378 as
// a state machine; 7x7 system solved by one wave, Sim3 exp / composition in double as the reference) and then evaluates its strip of
// the next request.  The host queues a budget of launches and waits for one pinned word per job.  Rounds 2-4 drove the loop from
// the host: one round trip per evaluation, 28 us per evaluation of which 21 were kernels.
// lsdhip_sim3tracker_evaluate (a single evaluation for the tests) keeps the two-kernel form: k_sim3_eval + k_sim3_finalize -> pinned record.
//
// Quirks kept: the SSE loops ignore the last size % 4 in-image points (x-outer / y-inner order) for the residual sums and
// both systems; LGS6::updateSSE counts 6 and LGS4::updateSSE 4 constraints per group of four points.
#include "lsdhip_internal.hpp"
#include "track_device.hpp"   // wave / workgroup reductions (block_top3, top3_insert), global address-space pointer types
#include <algorithm>
#include <atomic>
#include <chrono>
#include <vector>

#define S3_BLOCK 256
enum {
  S3_M = 0, S3_USAGE, S3_SXX, S3_SYY, S3_SX, S3_SY, S3_SW, S3_NREF,
  S3_SUMRESP, S3_SUMRESD, S3_NUMD,          // first tail-corrected column
  S3_A6,                                    // 21 + 6 + 1
  S3_A4 = S3_A6 + 28,                       // 10 + 4 + 1
  S3_END = S3_A4 + 15
};
#define S3_TAIL0 S3_SUMRESP
#define S3_NTAIL (S3_END - S3_TAIL0)        // 46 contributions per point
static_assert(S3_END == 54 && S3_NTAIL == 46, "layout");

struct Sim3Job {
  const float* kf_idepth; const float* kf_idepthVar; const float* kf_image; const float4* kf_grad;
  const float4* fr_grad; const float* fr_idepth; const float* fr_idepthVar;
  int w, h, nblocks, ppl;                   // nblocks strips of ppl * 256 pixels (ppl: pixels per lane)
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float R[9], t[3];                         // rxso3().matrix() (scale * rotation) and translation, cast to float
  float xRoll0, xRoll1, yRoll0, yRoll1;     // in-plane rotation of the reference gradients (ESM, Sim3Tracker.cpp:455-464)
  float aff_a, aff_b, cameraPixelNoise2, var_weight, huber_d;
  float* rows;                              // [nblocks][64] partial sums
  int4* topkey;                             // [nblocks] three largest order keys of in-image points
  float* record;                            // pinned host: S3_END totals (tail-corrected); word 63 = seq, raised last
  int seq;
};

// Up to S3_MAXB independent evaluations per launch (blockIdx.y = slot); the job descriptions travel in the kernel arguments.
#define S3_MAXB 12
// Strips per level: at most S3_NBMAX workgroups, each lane ceil(pixels / (256 S3_NBMAX)) pixels — few rows for the totals that every
// workgroup of the next launch adds up, enough workgroups to spread a level's memory latency
#define S3_NBMAX 128
struct Sim3Batch { Sim3Job j[S3_MAXB]; };
static_assert(sizeof(Sim3Batch) <= 4096, "kernel-argument limit");

// One reference pixel: its contributions to the S3_END sums, added to (ADD) or stored in `acc`; returns the pixel's order key
// x * h + y when its point lands inside the image, -1 otherwise
template <bool ADD>
__device__ __forceinline__ int sim3_eval_pixel(const Sim3Job& a, const int i, float (&acc)[S3_END]) {
  auto put = [&](const int k, const float v) { acc[k] = ADD ? acc[k] + v : v; };
  // (a job description read from memory hands over generic pointers, and loads through those are FLAT instructions that count against
  // the LDS counter too — track_device.hpp)
  typedef gv4f gfloat4;
  gfloat* kf_idepthVar = (gfloat*)a.kf_idepthVar; gfloat* kf_idepth = (gfloat*)a.kf_idepth; gfloat* kf_image = (gfloat*)a.kf_image;
  gfloat4* kf_grad = (gfloat4*)a.kf_grad; gfloat4* fr_grad = (gfloat4*)a.fr_grad;
  gfloat* fr_idepthVar = (gfloat*)a.fr_idepthVar; gfloat* fr_idepth = (gfloat*)a.fr_idepth;
  int key = -1;
  const int x = i % a.w, y = i / a.w;
  if (i < a.w * a.h && x >= 1 && x < a.w - 1 && y >= 1 && y < a.h - 1) {
    // (the pixel's four keyframe planes in one memory round trip, whether it holds a hypothesis or not; then the four texels and the
    // target pixel's depth pair in a second one: the evaluation is bound by these dependent round trips, not by their bytes)
    const float var = kf_idepthVar[i];
    const float id = kf_idepth[i];
    float I_ref = kf_image[i];
    v4f gref = kf_grad[i];
    asm volatile("" : "+v"(I_ref), "+v"(gref.x), "+v"(gref.y));   // (keeps the two loads here instead of behind the branch on var / id)
    if (!(var <= 0 || id == 0)) {
      // TrackingReference::makePointCloud (TrackingReference.cpp:128-138)
      const float inv = lsd_rcp_exact(id);
      const float px = inv * (a.fxi * x + a.cxi), py = inv * (a.fyi * y + a.cyi), pz = inv * 1.0f;
      put(S3_NREF, 1.f);
      const float Wx = ((a.R[0] * px + a.R[1] * py) + a.R[2] * pz) + a.t[0];
      const float Wy = ((a.R[3] * px + a.R[4] * py) + a.R[5] * pz) + a.t[1];
      const float Wz = ((a.R[6] * px + a.R[7] * py) + a.R[8] * pz) + a.t[2];
      const float u_new = (Wx / Wz) * a.fx + a.cx;
      const float v_new = (Wy / Wz) * a.fy + a.cy;
      if (u_new > 1 && v_new > 1 && u_new < a.w - 2 && v_new < a.h - 2) {
        key = x * a.h + y;
        // getInterpolatedElement43
        const int ix = (int)u_new, iy = (int)v_new;
        const float dx = u_new - ix, dy = v_new - iy, dxdy = dx * dy;
        gfloat4* bp = fr_grad + ix + iy * a.w;
        const v4f t00 = bp[0], t10 = bp[1], t01 = bp[a.w], t11 = bp[1 + a.w];
        const int idx_rounded = (int)(u_new + 0.5f) + a.w * (int)(v_new + 0.5f);
        const float var_frameDepth = fr_idepthVar[idx_rounded];
        float id_frameDepth = fr_idepth[idx_rounded];
        asm volatile("" : "+v"(id_frameDepth));                    // (... and this one beside the texels instead of behind var > 0)
        const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
        const float rx = w11 * t11.x + w01 * t01.x + w10 * t10.x + w00 * t00.x;
        const float ry = w11 * t11.y + w01 * t01.y + w10 * t10.y + w00 * t00.y;
        const float rz = w11 * t11.z + w01 * t01.z + w10 * t10.z + w00 * t00.z;
        // ESM gradients (USE_ESM_TRACKING == 1)
        const float rotatedGradX = a.xRoll0 * gref.x + a.xRoll1 * gref.y;
        const float rotatedGradY = a.yRoll0 * gref.x + a.yRoll1 * gref.y;
        const float gx = a.fx * 0.5f * (rx + rotatedGradX);
        const float gy = a.fy * 0.5f * (ry + rotatedGradY);
        const float c1 = a.aff_a * I_ref + a.aff_b;
        const float c2 = rz;
        const float rp = c1 - c2;
        const float hwgt = fabsf(rp) < 2.0f ? 1 : 2.0f / fabsf(rp);
        put(S3_M, 1.f);
        put(S3_SXX, c1 * c1 * hwgt); put(S3_SYY, c2 * c2 * hwgt); put(S3_SX, c1 * hwgt); put(S3_SY, c2 * hwgt); put(S3_SW, hwgt);
        const float ref_idepth = lsd_rcp_exact(Wz);
        const float d = lsd_rcp_exact(pz);
        float rd = -1, sv = -1;
        if (var_frameDepth > 0) { rd = ref_idepth - id_frameDepth; sv = var_frameDepth; }
        const float depthChange = pz / Wz;
        put(S3_USAGE, depthChange < 1 ? depthChange : 1);

        // calcSim3WeightsAndResidualSSE, operation order of the SSE path
        const float pz2d = lsd_rcp_exact((Wz * Wz) * d);
        const float g0 = (Wz * a.t[0] - Wx * a.t[2]) * pz2d;
        const float g1 = (Wz * a.t[1] - Wy * a.t[2]) * pz2d;
        const float g2 = (Wz - a.t[2]) * pz2d;
        const float drpdd = g0 * gx + g1 * gy;
        const float s_ = a.var_weight * var;
        const float w_p = lsd_rcp_exact(a.cameraPixelNoise2 + drpdd * (drpdd * s_));
        const float w_d = lsd_rcp_exact(sv + g2 * (g2 * s_));
        float wrp = rp * sqrtf(w_p);
        wrp = fmaxf(wrp, 0.0f - wrp);
        float wrd = rd * sqrtf(w_d);
        wrd = fmaxf(wrd, 0.0f - wrd);
        const bool depthValid = 0.0f < sv;
        const float wabs = (depthValid ? wrd : 0.0f) + wrp;
        const float wh = (wabs < a.huber_d) ? 1.0f : a.huber_d * lsd_rcp_exact(wabs);
        put(S3_NUMD, depthValid ? 1.f : 0.f);
        put(S3_SUMRESD, depthValid ? wh * (wrd * wrd) : 0.f);
        put(S3_SUMRESP, wh * (wrp * wrp));
        const float wp = wh * w_p;
        const float wd = depthValid ? wh * w_d : 0.f;

        // calcSim3LGSSSE
        const float z = lsd_rcp_exact(Wz);
        float J6[6], J4[4];
        J4[3] = z;
        J6[0] = z * gx;
        J6[1] = z * gy;
        J6[5] = ((Wx * gy) * z) - ((Wy * gx) * z);
        const float z2 = z * z;
        J4[0] = z2;
        J4[1] = z2 * Wy;
        J4[2] = 0.0f - (z2 * Wx);
        const float val1 = (Wx * gx) * z2, val2 = (Wy * gy) * z2;
        J6[2] = 0.0f - (val1 + val2);
        J6[3] = 0.0f - ((val2 * Wy) + (gy + val1 * Wy));
        J6[4] = (gx + val1 * Wx) + val2 * Wx;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const float Jw = J6[r] * wp;
#pragma unroll
          for (int c = r; c < 6; c++) put(S3_A6 + (r * 6 - (r * (r - 1)) / 2 + (c - r)), Jw * J6[c]);
        }
        const float resw6 = rp * wp;
#pragma unroll
        for (int r = 0; r < 6; r++) put(S3_A6 + 21 + r, resw6 * J6[r]);
        put(S3_A6 + 27, resw6 * rp);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float Jw = J4[r] * wd;
#pragma unroll
          for (int c = r; c < 4; c++) put(S3_A4 + (r * 4 - (r * (r - 1)) / 2 + (c - r)), Jw * J4[c]);
        }
        const float resw4 = rd * wd;
#pragma unroll
        for (int r = 0; r < 4; r++) put(S3_A4 + 10 + r, resw4 * J4[r]);
        put(S3_A4 + 14, resw4 * rd);
      }
    }
  }
  return key;
}
// The strip of workgroup blockIdx.x: a.ppl pixels per lane (consecutive 256-pixel runs, coalesced), their sums reduced lane -> LDS ->
// one row; the strip's three largest order keys beside it (sim3_totals drops the last M % 4 points in column order, as the SSE loops do)
__device__ __forceinline__ void sim3_eval_strip(const Sim3Job& a) {
  __shared__ float s_red[S3_END * (S3_BLOCK + 1) + 8];
  __shared__ float s_sum[4][64];
  __shared__ int s_wtop[S3_BLOCK / 64][3];
  __shared__ int s_top[3];
  const int tid = threadIdx.x;
  const int npix = a.w * a.h;
  float acc[S3_END];
#pragma unroll
  for (int k = 0; k < S3_END; k++) acc[k] = 0.f;
  int k0 = -1, k1 = -1, k2 = -1;       // this lane's in-image keys, descending
  for (int p = 0; p < a.ppl; p++) {
    const int i = (blockIdx.x * a.ppl + p) * S3_BLOCK + tid;
    if (i < npix) top3_insert(sim3_eval_pixel<true>(a, i, acc), k0, k1, k2);
  }
  // workgroup reduction through LDS (same scheme as k_track_step)
  constexpr int RRUN = 64;
#pragma unroll
  for (int k = 0; k < S3_END; k++) s_red[k * (S3_BLOCK + 1) + tid] = acc[k];
  __syncthreads();
  {
    const int slice = tid >> 6, k = tid & 63;
    if (k < S3_END) {
      const float* row = s_red + k * (S3_BLOCK + 1) + slice * RRUN;
      float v[RRUN];
#pragma unroll
      for (int j = 0; j < RRUN; j++) v[j] = row[j];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < RRUN; j++) s += v[j];
      s_sum[slice][k] = s;
    }
  }
  block_top3(k0, k1, k2, s_wtop, s_top);      // (its barriers also publish s_sum)
  if (tid < 64) {
    float s = 0.f;
    if (tid < S3_END) s = ((s_sum[0][tid] + s_sum[1][tid]) + s_sum[2][tid]) + s_sum[3][tid];
    ((__attribute__((address_space(1))) float*)a.rows)[(size_t)blockIdx.x * 64 + tid] = s;
  }
  if (tid == 0) a.topkey[blockIdx.x] = make_int4(s_top[0], s_top[1], s_top[2], -1);
}
__global__ __launch_bounds__(S3_BLOCK) void k_sim3_eval(Sim3Batch batch) {
  const Sim3Job& a = batch.j[blockIdx.y];
  if ((int)blockIdx.x >= a.nblocks) return;
  sim3_eval_strip(a);
}

// fixed-order sum of the strips' rows, the evaluation's three largest order keys, SSE tail drop: the evaluation's totals (tail-corrected)
// in s_tot[0 .. S3_END).  `a` is the evaluation's own description: the points of the three largest keys are evaluated once more, one
// lane each, for what they added to the sums; the last M % 4 of them in column order are what the SSE loops never visit.
// Everything a workgroup of the next launch waits for before its own strip, so ordered by memory round trips: the strips' keys and
// all rows are requested at once, the keys are merged while the rows arrive, and three lanes re-evaluate while the others add up.
static_assert(S3_NBMAX <= 256 && S3_NBMAX % 16 == 0, "one strip's keys per thread; 16 slices of at most S3_NBMAX / 16 rows");
__device__ __forceinline__ void sim3_totals(const Sim3Job& a, float* s_tot) {
  __shared__ float s_part[16][64];
  __shared__ float s_tail[3][S3_NTAIL + 2];
  __shared__ int s_keys[3];
  __shared__ int s_wtop[4][3];
  typedef int iv4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) iv4 giv4;
  const int tid = threadIdx.x;
  const iv4 none = {-1, -1, -1, -1};
  const iv4 kv = tid < a.nblocks ? ((giv4*)a.topkey)[tid] : none;      // (a strip's three keys are stored in descending order)
  // thread (slice, quad): rows [slice R, slice R + R) of columns 4 quad .. 4 quad + 3, one 16-byte load per row
  constexpr int RMAX = S3_NBMAX / 16;
  const int quad = tid & 15, sl = tid >> 4;
  const int R = (a.nblocks + 15) / 16;
  const int r0 = sl * R, r1 = min((sl + 1) * R, a.nblocks);
  v4f v[RMAX];
  {
    gv4f* rows4 = (gv4f*)a.rows + quad;
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < RMAX; j++) v[j] = r0 + j < r1 ? rows4[(size_t)(r0 + j) * 16] : zero4;
  }
  block_top3(kv.x, kv.y, kv.z, s_wtop, s_keys);     // the three largest of the strips' keys
  if (tid < 3 && s_keys[tid] >= 0) {
    const int key = s_keys[tid];
    float c[S3_END];
#pragma unroll
    for (int k = 0; k < S3_END; k++) c[k] = 0.f;
    sim3_eval_pixel<false>(a, (key / a.h) + (key % a.h) * a.w, c);
#pragma unroll
    for (int k = 0; k < S3_NTAIL; k++) s_tail[tid][k] = c[S3_TAIL0 + k];
  }
  {
    // two interleaved accumulators over the slice's rows, in row order (rows past the slice's end are zero: x + 0 == x)
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f), o = e;
#pragma unroll
    for (int j = 0; j < RMAX; j += 2) {
      e.x += v[j].x; e.y += v[j].y; e.z += v[j].z; e.w += v[j].w;
      if (j + 1 < RMAX) { o.x += v[j + 1].x; o.y += v[j + 1].y; o.z += v[j + 1].z; o.w += v[j + 1].w; }
    }
    float* dst = &s_part[sl][quad * 4];
    dst[0] = e.x + o.x; dst[1] = e.y + o.y; dst[2] = e.z + o.z; dst[3] = e.w + o.w;
  }
  __syncthreads();
  if (tid < 64) {
    // (each of the 64 threads adds up the point count itself — same terms, same order, same value — instead of waiting for thread S3_M's)
    float s = s_part[0][tid], cnt = s_part[0][S3_M];
#pragma unroll
    for (int sl2 = 1; sl2 < 16; sl2++) { s += s_part[sl2][tid]; cnt += s_part[sl2][S3_M]; }
    const int need = ((int)cnt) & 3;
    if (tid >= S3_TAIL0 && tid < S3_END)
      for (int k = 0; k < need; k++)
        if (s_keys[k] >= 0) s -= s_tail[k][tid - S3_TAIL0];
    s_tot[tid] = tid < S3_END ? s : 0.f;
  }
  __syncthreads();
}
// ... to the pinned record of a host-driven evaluation (lsdhip_sim3tracker_evaluate)
__global__ __launch_bounds__(256) void k_sim3_finalize(Sim3Batch batch) {
  const Sim3Job& a = batch.j[blockIdx.y];
  if (a.nblocks <= 0) return;
  __shared__ float s_tot[64];
  const int tid = threadIdx.x;
  sim3_totals(a, s_tot);
  if (tid < 63) a.record[tid] = s_tot[tid];
  // the host polls word 63 instead of sleeping in hipStreamSynchronize (whose wake-up costs more than the evaluation).  The word is the
  // launch's sequence number PLUS the position-weighted sum of the record's words: the host accepts the record only when the words it reads
  // add up — the words of a pinned record have been seen to land after the flag stored behind the fence (profiles/r06_notes.md section 1)
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    unsigned chk = (unsigned)a.seq;
    for (unsigned i = 0; i < 63; i++) chk += lsd_summary_term(i, __float_as_uint(s_tot[i]));
    ((volatile unsigned*)a.record)[63] = chk;
    __threadfence_system();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Sim3 algebra (double, Sophus semantics) and 7x7 LDL^T — host + device
// ---------------------------------------------------------------------------------------------------------------
#define S3_HD __host__ __device__ inline
namespace {
struct Sim3H { lsdm::Quatd q; double t[3]; double s; };

S3_HD Sim3H sim3_identity() { Sim3H r; r.q = {1, 0, 0, 0}; r.t[0] = r.t[1] = r.t[2] = 0; r.s = 1; return r; }
S3_HD void qd_normalize(lsdm::Quatd& q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
S3_HD Sim3H sim3_inverse(const Sim3H& S) {   // sim3.hpp:169-173
  Sim3H r;
  r.q = lsdm::q_conj(S.q);
  r.s = 1.0 / S.s;
  double rt[3];
  lsdm::q_rotate<lsdm::Quatd, double>(r.q, S.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = -(rt[i] * r.s);
  return r;
}
S3_HD Sim3H sim3_mul(const Sim3H& a, const Sim3H& b) {   // sim3.hpp:160-163
  Sim3H r;
  double rt[3];
  lsdm::q_rotate<lsdm::Quatd, double>(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + a.s * rt[i];
  r.q = lsdm::q_mul(a.q, b.q);
  qd_normalize(r.q);
  r.s = a.s * b.s;
  return r;
}
// the five transcendental values of an increment's exponential (the device computes them in different lanes: k_sim3_fused)
struct Sim3Trig { double sin_theta, cos_theta, sin_half, cos_half, exp_sigma; };
S3_HD double sim3_theta(const double a[7]) { return std::sqrt(a[3] * a[3] + (a[4] * a[4] + a[5] * a[5])); }
S3_HD Sim3H sim3_exp(const double a[7], const Sim3Trig& tr) {   // sim3.hpp:417-428, rxso3.hpp:416-425, calcW sim3.hpp:608-650
  const double eps = 1e-10;
  const double ox = a[3], oy = a[4], oz = a[5], sigma = a[6];
  const double scale = tr.exp_sigma;
  const double theta_sq = ox * ox + (oy * oy + oz * oz);
  const double theta = std::sqrt(theta_sq);
  double imag, real;
  if (theta < eps) {
    const double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    imag = tr.sin_half / theta;
    real = tr.cos_half;
  }
  Sim3H r;
  r.q = {real, imag * ox, imag * oy, imag * oz};
  qd_normalize(r.q);
  r.s = scale;
  const double Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  double Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = Om[i * 3 + 0] * Om[0 * 3 + j];
      acc += Om[i * 3 + 1] * Om[1 * 3 + j];
      acc += Om[i * 3 + 2] * Om[2 * 3 + j];
      Om2[i * 3 + j] = acc;
    }
  double A, B, C;
  if (std::abs(sigma) < eps) {
    C = 1.0;
    if (std::abs(theta) < eps) { A = 0.5; B = 1.0 / 6.0; }
    else { A = (1.0 - tr.cos_theta) / theta_sq; B = (theta - tr.sin_theta) / (theta_sq * theta); }
  } else {
    C = (scale - 1.0) / sigma;
    if (std::abs(theta) < eps) {
      const double sigma_sq = sigma * sigma;
      A = ((sigma - 1.0) * scale + 1.0) / sigma_sq;
      B = ((0.5 * sigma * sigma - sigma + 1.0) * scale) / (sigma_sq * sigma);
    } else {
      const double sa = scale * tr.sin_theta, sb = scale * tr.cos_theta, c = theta_sq + sigma * sigma;
      A = (sa * sigma + (1.0 - sb) * theta) / (theta * c);
      B = (C - ((sb - 1.0) * sigma + sa * theta) / c) * 1.0 / theta_sq;
    }
  }
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    for (int j = 0; j < 3; j++) {
      const double W = A * Om[i * 3 + j] + B * Om2[i * 3 + j] + C * (i == j ? 1.0 : 0.0);
      acc = j == 0 ? W * a[j] : acc + W * a[j];
    }
    r.t[i] = acc;
  }
  return r;
}
S3_HD Sim3H sim3_exp(const double a[7]) {
  const double theta = sim3_theta(a);
  Sim3Trig tr;
  tr.sin_theta = std::sin(theta); tr.cos_theta = std::cos(theta);
  tr.sin_half = std::sin(0.5 * theta); tr.cos_half = std::cos(0.5 * theta);
  tr.exp_sigma = std::exp(a[6]);
  return sim3_exp(a, tr);
}
// 7x7 LDL^T with diagonal pivoting (Eigen A.ldlt().solve(b) semantics), as lsdm::ldlt6_solve
// (the factorisation indexes its matrix with run-time indices: the caller provides the storage — the stack on the host, LDS on the device,
// where a private array of that kind would live in scratch memory)
struct Ldlt7Scratch { float M[7][7]; int p[7]; float y[7]; };
S3_HD void ldlt7_solve(const float Ain[49], const float bin[7], float x[7], Ldlt7Scratch& W) {
  const int n = 7;
  float (&M)[7][7] = W.M;
  int (&p)[7] = W.p;
  for (int i = 0; i < n; i++) { p[i] = i; for (int j = 0; j < n; j++) M[i][j] = Ain[i * n + j]; }
  for (int k = 0; k < n; k++) {
    int piv = k;
    float big = fabsf(M[k][k]);
    for (int i = k + 1; i < n; i++) if (fabsf(M[i][i]) > big) { big = fabsf(M[i][i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; j++) { float t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
      for (int i = 0; i < n; i++) { float t = M[i][k]; M[i][k] = M[i][piv]; M[i][piv] = t; }
      int tp = p[k]; p[k] = p[piv]; p[piv] = tp;
    }
    float d = M[k][k];
    for (int j = 0; j < k; j++) d -= M[k][j] * M[k][j] * M[j][j];
    M[k][k] = d;
    for (int i = k + 1; i < n; i++) {
      float v = M[i][k];
      for (int j = 0; j < k; j++) v -= M[i][j] * M[k][j] * M[j][j];
      M[i][k] = d != 0.0f ? v / d : 0.0f;
    }
  }
  float (&y)[7] = W.y;
  for (int i = 0; i < n; i++) y[i] = bin[p[i]];
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= M[i][j] * y[j];
  for (int i = 0; i < n; i++) y[i] = M[i][i] != 0.0f ? y[i] / M[i][i] : 0.0f;
  for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= M[j][i] * y[j];
  for (int i = 0; i < n; i++) x[p[i]] = y[i];
}

struct Sim3Res { float sumResD, sumResP; int numTermsD, numTermsP; float meanD, meanP, mean; };
struct Eval {
  int M;
  float pointUsage, aff_a_lastIt, aff_b_lastIt;
  Sim3Res res;
  float A[49], b[7];
  size_t num_constraints;
};
// LGS6 / LGS4 finishNoDivide + LGS7::initializeFrom (LGSX.h:424-442) entry by entry, from an evaluation's (tail-corrected) totals: the 6x6
// photometric system on rows / columns 0..5, the 4x4 depth system added on rows / columns 2, 3, 4, 6
S3_HD float sim3_system_A(const float* r, const int i, const int j) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  float v = 0.f;
  if (hi < 6) v = 0.0f + r[S3_A6 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
  const int l4 = lo == 6 ? 3 : lo - 2, h4 = hi == 6 ? 3 : hi - 2;     // 2, 3, 4, 6 -> 0, 1, 2, 3
  if (lo >= 2 && lo != 5 && hi != 5) v += 0.0f + r[S3_A4 + l4 * 4 - (l4 * (l4 - 1)) / 2 + (h4 - l4)];
  return v;
}
S3_HD float sim3_system_b(const float* r, const int i) {
  float v = 0.f;
  if (i < 6) v = 0.0f - r[S3_A6 + 21 + i];
  if (i >= 2 && i != 5) v += 0.0f - r[S3_A4 + 10 + (i == 6 ? 3 : i - 2)];
  return v;
}
// the evaluation's outcome in the reference's terms but for the system (`part` of `parts` callers share its 56 entries)
S3_HD void sim3_eval_scalars(const float* r, Eval* ev) {
  ev->M = (int)r[S3_M];
  ev->pointUsage = r[S3_USAGE] / r[S3_NREF];
  {
    const float sxx = r[S3_SXX], syy = r[S3_SYY], sx = r[S3_SX], sy = r[S3_SY], sw = r[S3_SW];
    ev->aff_a_lastIt = sqrtf((syy - sy * sy / sw) / (sxx - sx * sx / sw));
    ev->aff_b_lastIt = (sy - ev->aff_a_lastIt * sx) / sw;
  }
  Sim3Res& s = ev->res;
  s.sumResP = r[S3_SUMRESP];
  s.numTermsP = (ev->M >> 2) << 2;
  s.sumResD = r[S3_SUMRESD];
  s.numTermsD = (int)r[S3_NUMD];
  s.mean = (s.sumResD + s.sumResP) / (s.numTermsD + s.numTermsP);
  s.meanD = s.sumResD / s.numTermsD;
  s.meanP = s.sumResP / s.numTermsP;
  ev->num_constraints = (size_t)6 * (size_t)(ev->M >> 2) + (size_t)4 * (size_t)(ev->M >> 2);
}
S3_HD void sim3_eval_from_totals(const float* r, Eval* ev) {
  sim3_eval_scalars(r, ev);
  for (int i = 0; i < 7; i++) {
    for (int j = 0; j < 7; j++) ev->A[i * 7 + j] = sim3_system_A(r, i, j);
    ev->b[i] = sim3_system_b(r, i);
  }
}
}  // namespace

struct Sim3Set;
namespace { struct Sim3Track; }
struct lsdhip_sim3tracker {
  lsdhip_ctx* ctx = nullptr;
  int maxItsPerLvl[LSD_LEVELS] = {5, 20, 50, 100, 100};
  float lambdaSuccessFac = 0.5f, lambdaFailFac = 2.0f, lambdaInitial = 0, stepSizeMin = 1e-8f, convergenceEps = 0.999f;
  float huber_d = 3, var_weight = 1.0f;
  float* d_rows = nullptr;     // [2][S3_MAXB][S3_NBMAX][64]  (two launch parities: k_sim3_fused)
  int4* d_topkey = nullptr;    // [2][S3_MAXB][S3_NBMAX]
  float* h_record = nullptr;   // [S3_MAXB][64] pinned, device-mapped
  float* d_record = nullptr;   // device alias of h_record
  int seq = 0;                 // launch counter, echoed by k_sim3_finalize in word 63 of every active slot's record
  // trackFrameSim3 on the device (k_sim3_fused): per batch slot the job's level descriptions and LM state
  Sim3Set* h_sets = nullptr;  Sim3Set* d_sets = nullptr;              // [S3_MAXB]; h_*: pinned staging
  Sim3Track* h_states = nullptr;  Sim3Track* d_states = nullptr;      // [S3_MAXB] ([2][S3_MAXB] on the device: launch parities)
  lsdhip_sim3_result* h_results = nullptr;  lsdhip_sim3_result* d_results = nullptr;   // [S3_MAXB] pinned, device-mapped
  int* h_done = nullptr;  int* d_done = nullptr;                              // [2][S3_MAXB] pinned, device-mapped: flags, then the records' check words
  long long lateRecords = 0;                                                  // result records that were incomplete when their flag arrived
  int recentRounds = 0;        // evaluations the longest job of the last call needed: the next call's launch budget
};

extern "C" int lsdhip_sim3tracker_create(lsdhip_ctx* c, lsdhip_sim3tracker** out) {
  if (!c || !out) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  lsdhip_sim3tracker* t = new lsdhip_sim3tracker();
  t->ctx = c;
  const size_t mb = (size_t)S3_NBMAX * S3_MAXB * 2;
  HIPCHK(hipMalloc((void**)&t->d_rows, mb * 64 * 4));
  HIPCHK(hipMalloc((void**)&t->d_topkey, mb * 16));
  HIPCHK(hipHostMalloc((void**)&t->h_record, S3_MAXB * 64 * 4, hipHostMallocMapped));
  memset(t->h_record, 0, S3_MAXB * 64 * 4);
  HIPCHK(hipHostGetDevicePointer((void**)&t->d_record, t->h_record, 0));
  *out = t;
  return LSDHIP_OK;
}
extern "C" void lsdhip_sim3tracker_destroy(lsdhip_sim3tracker* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  (void)hipFree(t->d_rows); (void)hipFree(t->d_topkey); (void)hipHostFree(t->h_record);
  (void)hipFree(t->d_sets);      // (one block: level descriptions + states, as they are uploaded)
  (void)hipHostFree(t->h_sets); (void)hipHostFree(t->h_results); (void)hipHostFree(t->h_done);
  delete t;
}
extern "C" int lsdhip_sim3tracker_set_max_its(lsdhip_sim3tracker* t, const int its[LSD_LEVELS]) {
  if (!t || !its) return LSDHIP_E_ARG;
  for (int l = 0; l < LSD_LEVELS; l++) t->maxItsPerLvl[l] = its[l];
  return LSDHIP_OK;
}

static void sim3_strips(int npix, int* nblocks, int* ppl) {
  *ppl = (npix + S3_BLOCK * S3_NBMAX - 1) / (S3_BLOCK * S3_NBMAX);
  *nblocks = (npix + S3_BLOCK * *ppl - 1) / (S3_BLOCK * *ppl);
}
// one evaluation = calcSim3Buffers + calcSim3WeightsAndResidualSSE + calcSim3LGSSSE at one transformation.
// Description of the evaluation for batch slot `slot`:
static int sim3_build_job(lsdhip_sim3tracker* t, int slot, lsdhip_frame* kf, lsdhip_frame* frame, const Sim3H& referenceToFrame, int level,
                          float aff_a, float aff_b, Sim3Job* out) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  Sim3Job a;
  if (level == 0) { (void)lsd_frame_require_level0_for_tracking(kf); (void)lsd_frame_require_level0_for_tracking(frame); }   // level-0 texels on demand
  a.kf_idepth = kf->d_idepth[level]; a.kf_idepthVar = kf->d_idepthVar[level]; a.kf_image = kf->d_image[level]; a.kf_grad = kf->d_grad[level];
  a.fr_grad = frame->d_grad[level]; a.fr_idepth = frame->d_idepth[level]; a.fr_idepthVar = frame->d_idepthVar[level];
  a.w = c->wl[level]; a.h = c->hl[level];
  sim3_strips(a.w * a.h, &a.nblocks, &a.ppl);
  const LevelIntr& in = c->intr[level];
  a.fx = in.fx; a.fy = in.fy; a.cx = in.cx; a.cy = in.cy; a.fxi = in.fxi; a.fyi = in.fyi; a.cxi = in.cxi; a.cyi = in.cyi;
  double Rd[9];
  lsdm::quatd_to_rot(referenceToFrame.q, Rd);
  float Ru[9];
  for (int i = 0; i < 9; i++) { a.R[i] = (float)(referenceToFrame.s * Rd[i]); Ru[i] = (float)Rd[i]; }
  for (int i = 0; i < 3; i++) a.t[i] = (float)referenceToFrame.t[i];
  {
    // Quaternionf::setFromTwoVectors(R * (0,0,-1), (0,0,-1)).toRotationMatrix() * R   (Sim3Tracker.cpp:455-464)
    const float rf[3] = {Ru[0] * 0.f + Ru[1] * 0.f + Ru[2] * -1.f, Ru[3] * 0.f + Ru[4] * 0.f + Ru[5] * -1.f, Ru[6] * 0.f + Ru[7] * 0.f + Ru[8] * -1.f};
    const float n0 = sqrtf(rf[0] * rf[0] + (rf[1] * rf[1] + rf[2] * rf[2]));
    const float v0[3] = {rf[0] / n0, rf[1] / n0, rf[2] / n0};
    const float v1[3] = {0, 0, -1};
    const float cdot = v1[0] * v0[0] + (v1[1] * v0[1] + v1[2] * v0[2]);
    lsdm::Quatf q;
    if (cdot < -1.0f + 1e-5f) {
      q = {0, 1, 0, 0};
    } else {
      const float ax = v0[1] * v1[2] - v0[2] * v1[1], ay = v0[2] * v1[0] - v0[0] * v1[2], az = v0[0] * v1[1] - v0[1] * v1[0];
      const float s = sqrtf((1.0f + cdot) * 2.0f);
      const float invs = 1.0f / s;
      q = {s * 0.5f, ax * invs, ay * invs, az * invs};
    }
    float Q[9];
    lsdm::quatf_to_rot(q, Q);
    float roll[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        float acc = Q[i * 3 + 0] * Ru[0 * 3 + j];
        acc += Q[i * 3 + 1] * Ru[1 * 3 + j];
        acc += Q[i * 3 + 2] * Ru[2 * 3 + j];
        roll[i * 3 + j] = acc;
      }
    a.xRoll0 = roll[0]; a.xRoll1 = roll[1]; a.yRoll0 = roll[3]; a.yRoll1 = roll[4];
  }
  a.aff_a = aff_a; a.aff_b = aff_b;
  a.cameraPixelNoise2 = c->params.cameraPixelNoise2; a.var_weight = t->var_weight; a.huber_d = t->huber_d;
  a.rows = t->d_rows + (size_t)slot * S3_NBMAX * 64;
  a.topkey = t->d_topkey + (size_t)slot * S3_NBMAX;
  a.record = t->d_record + (size_t)slot * 64;
  a.seq = 0;
  *out = a;
  return LSDHIP_OK;
}
// launches the slots with nblocks > 0 and waits until each of them has raised the launch's sequence number
static int sim3_run_batch(lsdhip_sim3tracker* t, Sim3Batch& batch, int nslots) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  LsdTrackJobScope tjob_(c, true);   // pipelined contexts: behind the mapping stream's products, and a record point for it afterwards
  if (tjob_.rc) return tjob_.rc;
  const int seq = ++t->seq;
  int grid = 0;
  for (int k = 0; k < S3_MAXB; k++) {
    if (k >= nslots) batch.j[k].nblocks = 0;
    batch.j[k].seq = seq;
    if (batch.j[k].nblocks > grid) grid = batch.j[k].nblocks;
  }
  if (grid == 0) return LSDHIP_OK;
  hipLaunchKernelGGL(k_sim3_eval, dim3(grid, nslots), dim3(S3_BLOCK), 0, c->stream, batch);
  hipLaunchKernelGGL(k_sim3_finalize, dim3(1, nslots), dim3(256), 0, c->stream, batch);
  HIPCHK(hipGetLastError());
  const auto tStart = std::chrono::steady_clock::now();
  for (int k = 0; k < nslots; k++) {
    if (batch.j[k].nblocks <= 0) continue;
    volatile const unsigned* rec = (volatile const unsigned*)(t->h_record + (size_t)k * 64);
    // word 63 = seq + position-weighted sum of the words before it (k_sim3_finalize): whole record or nothing
    auto landed = [&]() { unsigned chk = (unsigned)seq; for (unsigned i = 0; i < 63; i++) chk += lsd_summary_term(i, rec[i]); return rec[63] == chk; };
    unsigned spins = 0;
    while (!landed()) {
      if ((++spins & 0xFFFFFu) == 0) {   // safety net only: a stream query puts a marker packet into the queue (profiles/r03_notes.md §2b)
        hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess && q != hipErrorNotReady) { lsd_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return LSDHIP_E_HIP; }
        if (std::chrono::steady_clock::now() - tStart > std::chrono::seconds(5)) { HIPCHK(hipStreamSynchronize(c->stream)); break; }
      }
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!landed()) { lsd_set_error("Sim3 evaluation did not complete"); return LSDHIP_E_STATE; }
  }
  return LSDHIP_OK;
}
static void sim3_read_eval(lsdhip_sim3tracker* t, int slot, Eval* ev) { sim3_eval_from_totals(t->h_record + (size_t)slot * 64, ev); }

static Sim3H sim3_in(const double p[8]) { Sim3H T; T.q = {p[0], p[1], p[2], p[3]}; T.t[0] = p[4]; T.t[1] = p[5]; T.t[2] = p[6]; T.s = p[7]; return T; }
static void sim3_out(const Sim3H& T, double p[8]) { p[0] = T.q.w; p[1] = T.q.x; p[2] = T.q.y; p[3] = T.q.z; p[4] = T.t[0]; p[5] = T.t[1]; p[6] = T.t[2]; p[7] = T.s; }

// host-side algebra of the LM step, exported for CPU tests of the product's own arithmetic (no device involved)
extern "C" int lsdhip_host_sim3_step(const double increment[7], const double referenceToFrame[8], double out[8]) {
  if (!increment || !referenceToFrame || !out) return LSDHIP_E_ARG;
  sim3_out(sim3_mul(sim3_exp(increment), sim3_in(referenceToFrame)), out);
  return LSDHIP_OK;
}
extern "C" int lsdhip_host_ldlt7(const float A[49], const float b[7], float x[7]) {
  if (!A || !b || !x) return LSDHIP_E_ARG;
  Ldlt7Scratch W;
  ldlt7_solve(A, b, x, W);
  return LSDHIP_OK;
}

extern "C" int lsdhip_sim3tracker_evaluate(lsdhip_sim3tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const double refToFrame[8], int level,
                                           float aff_a, float aff_b, lsdhip_sim3_eval_record* out) {
  if (!t || !kf || !frame || !refToFrame || !out || level < 0 || level >= LSD_LEVELS) return LSDHIP_E_ARG;
  if (!kf->hasIDepth || !frame->hasIDepth) { lsd_set_error("Sim3 tracking needs inverse depth on both frames"); return LSDHIP_E_STATE; }
  HIPCHK(hipSetDevice(t->ctx->device));
  Sim3Batch batch;
  memset(&batch, 0, sizeof(batch));
  int rc = sim3_build_job(t, 0, kf, frame, sim3_in(refToFrame), level, aff_a, aff_b, &batch.j[0]);
  if (rc) return rc;
  rc = sim3_run_batch(t, batch, 1);
  if (rc) return rc;
  Eval ev;
  sim3_read_eval(t, 0, &ev);
  out->warped_size = ev.M; out->pointUsage = ev.pointUsage; out->affine_a_lastIt = ev.aff_a_lastIt; out->affine_b_lastIt = ev.aff_b_lastIt;
  out->sumResD = ev.res.sumResD; out->sumResP = ev.res.sumResP; out->numTermsD = ev.res.numTermsD; out->numTermsP = ev.res.numTermsP;
  out->meanD = ev.res.meanD; out->meanP = ev.res.meanP; out->mean = ev.res.mean;
  memcpy(out->A, ev.A, sizeof(ev.A)); memcpy(out->b, ev.b, sizeof(ev.b));
  out->num_constraints = (double)ev.num_constraints;
  return LSDHIP_OK;
}

// ---- the Levenberg-Marquardt loop of trackFrameSim3 (Sim3Tracker.cpp:149-378) as a state machine between two evaluations --------------
// `advance` takes the evaluation that was asked for and either asks for the next one (transformation + level) or finishes the job.
// It runs on the device (lane 0 of every workgroup of k_sim3_fused, the state staged in LDS); the host only starts a job with it
// (sim3_init_job).  7x7 float system, Sim3 exp / composition in double, as the reference.  Several jobs advance in lock step and share
// their launches (lsdhip_sim3tracker_track_batch: blockIdx.y = job).
namespace {
struct Sim3LM {                    // the tracker's settings as the state machine needs them
  int maxIts[LSD_LEVELS];
  float lambdaSuccessFac, lambdaFailFac, lambdaInitial, stepSizeMin, convergenceEps;
  int useAffine, w, h;
};
struct Sim3Track {                  // plain data: lives in HBM between the launches, in LDS while a launch works on it
  lsdhip_sim3_result res;          // the result record as far as it is known (workgroup 0 copies it to the pinned record at the end)
  lsdhip_sim3_result* hostOut;     // the caller's record
  enum Phase { LEVEL_FIRST, TRY, FINAL, DONE };
  int phase;
  int rc;
  Sim3H referenceToFrame, candidate;
  float aff_a, aff_b;
  int lvl, finalLevel, iteration, incTry, numEvaluations;
  float LM_lambda, absInc;
  int warp_update_up_to_date;
  Eval cur;
  Sim3Res lastErr, finalResidual;
  // the evaluation this job waits for
  Sim3H reqPose;
  int reqLevel;
  // ... as the strips read it (sim3_request)
  float reqR[9], reqT[3], reqRoll[4];
  int pendingEval;                 // the previous launch evaluated the request: its rows wait in the scratch of this launch's parity
};
static_assert(sizeof(Sim3Track) % 4 == 0, "copied by words");
struct Sim3Scratch { float inc[7]; int solve; float m[7][8]; Sim3Trig trig; };

// the requested transformation as the evaluation kernel wants it: rxso3().matrix() and translation in float, the in-plane roll of the
// reference gradients (ESM, Sim3Tracker.cpp:455-464)
S3_HD void sim3_request(Sim3Track& J) {
  const Sim3H& T = J.reqPose;
  double Rd[9];
  lsdm::quatd_to_rot(T.q, Rd);
  float Ru[9];
  for (int i = 0; i < 9; i++) { J.reqR[i] = (float)(T.s * Rd[i]); Ru[i] = (float)Rd[i]; }
  for (int i = 0; i < 3; i++) J.reqT[i] = (float)T.t[i];
  // Quaternionf::setFromTwoVectors(R * (0,0,-1), (0,0,-1)).toRotationMatrix() * R
  const float rf[3] = {Ru[0] * 0.f + Ru[1] * 0.f + Ru[2] * -1.f, Ru[3] * 0.f + Ru[4] * 0.f + Ru[5] * -1.f, Ru[6] * 0.f + Ru[7] * 0.f + Ru[8] * -1.f};
  const float n0 = sqrtf(rf[0] * rf[0] + (rf[1] * rf[1] + rf[2] * rf[2]));
  const float v0[3] = {rf[0] / n0, rf[1] / n0, rf[2] / n0};
  const float v1[3] = {0, 0, -1};
  const float cdot = v1[0] * v0[0] + (v1[1] * v0[1] + v1[2] * v0[2]);
  lsdm::Quatf q;
  if (cdot < -1.0f + 1e-5f) {
    q = {0, 1, 0, 0};
  } else {
    const float ax = v0[1] * v1[2] - v0[2] * v1[1], ay = v0[2] * v1[0] - v0[0] * v1[2], az = v0[0] * v1[1] - v0[1] * v1[0];
    const float sq = sqrtf((1.0f + cdot) * 2.0f);
    const float invs = 1.0f / sq;
    q = {sq * 0.5f, ax * invs, ay * invs, az * invs};
  }
  float Q[9];
  lsdm::quatf_to_rot(q, Q);
  // rows 0 and 1, columns 0 and 1 of Q * R
  J.reqRoll[0] = (Q[0] * Ru[0] + Q[1] * Ru[3]) + Q[2] * Ru[6];
  J.reqRoll[1] = (Q[0] * Ru[1] + Q[1] * Ru[4]) + Q[2] * Ru[7];
  J.reqRoll[2] = (Q[3] * Ru[0] + Q[4] * Ru[3]) + Q[5] * Ru[6];
  J.reqRoll[3] = (Q[3] * Ru[1] + Q[4] * Ru[4]) + Q[5] * Ru[7];
}

S3_HD void sim3_out8(const Sim3H& T, double p[8]) { p[0] = T.q.w; p[1] = T.q.x; p[2] = T.q.y; p[3] = T.q.z; p[4] = T.t[0]; p[5] = T.t[1]; p[6] = T.t[2]; p[7] = T.s; }
S3_HD void sim3_finish(Sim3Track& J) {
  lsdhip_sim3_result* out = &J.res;
  for (int i = 0; i < 49; i++) out->lastSim3Hessian[i] = J.cur.A[i];
  out->numEvaluations = J.numEvaluations;
  out->pointUsage = J.cur.pointUsage;
  out->affineEstimation_a = J.aff_a; out->affineEstimation_b = J.aff_b;
  J.phase = Sim3Track::DONE;
  if (J.referenceToFrame.s <= 0) { out->diverged = 1; J.rc = LSDHIP_DIVERGED; return; }
  out->lastResidual = J.finalResidual.mean;
  out->lastDepthResidual = J.finalResidual.meanD;
  out->lastPhotometricResidual = J.finalResidual.meanP;
  sim3_out8(sim3_inverse(J.referenceToFrame), out->frameToReference);
  J.rc = LSDHIP_OK;
}
S3_HD void sim3_diverge(Sim3Track& J, bool setFlag) {
  // Sim3() is already in the result; `diverged` is only raised where the reference raises it (too few points, scale <= 0)
  if (setFlag) J.res.diverged = 1;
  J.res.numEvaluations = J.numEvaluations;
  J.phase = Sim3Track::DONE;
  J.rc = LSDHIP_DIVERGED;
}
// enter the next level that has iterations (or the final re-evaluation / the end)
S3_HD void sim3_next_level(const Sim3LM& P, Sim3Track& J) {
  while (J.lvl >= J.finalLevel && P.maxIts[J.lvl] == 0) J.lvl--;
  if (J.lvl >= J.finalLevel) {
    J.phase = Sim3Track::LEVEL_FIRST;
    J.reqPose = J.referenceToFrame; J.reqLevel = J.lvl;
    return;
  }
  // the system at the accepted transformation is recomputed on the final level when the last evaluation was accepted
  // (the affine parameters have moved since) or belonged to another level (Sim3Tracker.cpp:354-360)
  if (!J.warp_update_up_to_date) {
    J.phase = Sim3Track::FINAL;
    J.reqPose = J.referenceToFrame; J.reqLevel = J.finalLevel;
    return;
  }
  sim3_finish(J);
}
// The next increment solves the damped system of the last accepted evaluation (A / n, b / n with the constraint count n, diagonal
// * (1 + lambda)): entry (i, j) of the augmented 7x8 system, j = 7 the right-hand side
S3_HD float sim3_damped_entry(const Sim3Track& J, const int i, const int j) {
  const float nc = (float)J.cur.num_constraints;
  if (j == 7) return -J.cur.b[i] / nc;
  float v = J.cur.A[i * 7 + j] / nc;
  if (i == j) v *= 1 + J.LM_lambda;
  return v;
}
S3_HD void sim3_propose(const Sim3LM& P, Sim3Track& J, Sim3Scratch& W) { W.solve = 1; }   // (the device solves it with the lanes of a wave)
// ... and, with the increment, the candidate transformation whose evaluation is asked for
S3_HD void sim3_propose_finish(const Sim3LM& P, Sim3Track& J, Sim3Scratch& W) {
  J.incTry++;
  float absInc = 0;
  for (int i = 0; i < 7; i++) absInc += W.inc[i] * W.inc[i];
  J.absInc = absInc;
  if (!(absInc >= 0 && absInc < 1)) { sim3_diverge(J, false); return; }   // returns Sim3(), Hessian zero
  const double incd[7] = {(double)W.inc[0], (double)W.inc[1], (double)W.inc[2], (double)W.inc[3], (double)W.inc[4], (double)W.inc[5], (double)W.inc[6]};
  J.candidate = sim3_mul(sim3_exp(incd, W.trig), J.referenceToFrame);
  J.phase = Sim3Track::TRY;
  J.reqPose = J.candidate; J.reqLevel = J.lvl;
}
// start of an iteration of the level's loop (`for iteration < maxIts`), or leave the level
S3_HD void sim3_iteration(const Sim3LM& P, Sim3Track& J, Sim3Scratch& W) {
  if (J.iteration >= P.maxIts[J.lvl]) { J.lvl--; sim3_next_level(P, J); return; }
  J.warp_update_up_to_date = 1;   // the system of the last accepted evaluation is what calcSim3LGS would build here
  J.incTry = 0;
  sim3_propose(P, J, W);
}
// the requested evaluation has arrived
S3_HD void sim3_advance(const Sim3LM& P, Sim3Track& J, const Eval& ev, Sim3Scratch& W) {
  const bool useAffine = P.useAffine != 0;
  J.numEvaluations++;
  if (J.phase == Sim3Track::FINAL) {
    J.cur = ev;
    J.finalResidual = ev.res;
    sim3_finish(J);
    return;
  }
  if (ev.M < 0.5 * 0.01 * (P.w >> J.lvl) * (P.h >> J.lvl) || ev.M < 10) { sim3_diverge(J, true); return; }
  if (J.phase == Sim3Track::LEVEL_FIRST) {
    J.cur = ev;
    J.lastErr = ev.res;
    if (useAffine) { J.aff_a = ev.aff_a_lastIt; J.aff_b = ev.aff_b_lastIt; }
    J.LM_lambda = P.lambdaInitial;
    J.warp_update_up_to_date = 0;
    J.iteration = 0;
    sim3_iteration(P, J, W);
    return;
  }
  // TRY
  if (ev.res.mean < J.lastErr.mean) {
    J.referenceToFrame = J.candidate;
    J.cur = ev;
    J.warp_update_up_to_date = 0;
    if (useAffine) { J.aff_a = ev.aff_a_lastIt; J.aff_b = ev.aff_b_lastIt; }
    if (ev.res.mean / J.lastErr.mean > P.convergenceEps) J.iteration = P.maxIts[J.lvl];
    J.finalResidual = J.lastErr = ev.res;
    if (J.LM_lambda <= 0.2) J.LM_lambda = 0;
    else J.LM_lambda *= P.lambdaSuccessFac;
    J.iteration++;
    sim3_iteration(P, J, W);
  } else {
    if (!(J.absInc > P.stepSizeMin)) {
      J.iteration = P.maxIts[J.lvl];
      J.iteration++;
      sim3_iteration(P, J, W);
      return;
    }
    if (J.LM_lambda == 0) J.LM_lambda = 0.2;
    else J.LM_lambda *= pow((double)P.lambdaFailFac, (double)J.incTry);
    sim3_propose(P, J, W);
  }
}
}  // namespace

// ---- the job on the device ---------------------------------------------------------------------------------------------------------
struct Sim3Lvl {                  // what does not change during a job, per pyramid level
  const float* kf_idepth; const float* kf_idepthVar; const float* kf_image; const float4* kf_grad;
  const float4* fr_grad; const float* fr_idepth; const float* fr_idepthVar;
  int w, h, nblocks, ppl;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
};
struct Sim3Set {
  Sim3Lvl lv[LSD_LEVELS];
  float cameraPixelNoise2, var_weight, huber_d;
  float* rows[2]; int4* topkey[2];   // per launch parity
  lsdhip_sim3_result* result;     // pinned: the job's result record
  int* done;                      // pinned: seq * 256 + 255 (254: diverged) once the job has finished; seq * 256 + b when the b-th launch budget of the call
  int seq;                        // ends with the job unfinished (the host then queues another budget)
};
__device__ __forceinline__ void sim3_job_view(const Sim3Set& set, const Sim3Track& J, const int parity, Sim3Job& a) {
  const Sim3Lvl& L = set.lv[J.reqLevel];
  a.kf_idepth = L.kf_idepth; a.kf_idepthVar = L.kf_idepthVar; a.kf_image = L.kf_image; a.kf_grad = L.kf_grad;
  a.fr_grad = L.fr_grad; a.fr_idepth = L.fr_idepth; a.fr_idepthVar = L.fr_idepthVar;
  a.w = L.w; a.h = L.h; a.nblocks = L.nblocks; a.ppl = L.ppl;
  a.fx = L.fx; a.fy = L.fy; a.cx = L.cx; a.cy = L.cy; a.fxi = L.fxi; a.fyi = L.fyi; a.cxi = L.cxi; a.cyi = L.cyi;
#pragma unroll
  for (int i = 0; i < 9; i++) a.R[i] = J.reqR[i];
#pragma unroll
  for (int i = 0; i < 3; i++) a.t[i] = J.reqT[i];
  a.xRoll0 = J.reqRoll[0]; a.xRoll1 = J.reqRoll[1]; a.yRoll0 = J.reqRoll[2]; a.yRoll1 = J.reqRoll[3];
  a.aff_a = J.aff_a; a.aff_b = J.aff_b;
  a.cameraPixelNoise2 = set.cameraPixelNoise2; a.var_weight = set.var_weight; a.huber_d = set.huber_d;
  a.rows = set.rows[parity]; a.topkey = set.topkey[parity];
  a.record = nullptr; a.seq = 0;
}
// 7x7 solve of the LM step with one element of the augmented 7x8 system per lane (lane = 8 i + j), pivot row / column read back through
// LDS — gj6_solve_wave of the SE3 tracker (track_device.hpp) with one more row and DIAGONAL PIVOTING: at every step the unused row with
// the largest |diagonal| is the pivot (what Eigen's LDLT does, C/Tracking/Sim3Tracker.cpp:300 `A.ldlt().solve(b)`), and a pivot that is
// exactly zero eliminates nothing and leaves its unknown at 0 (Eigen: the pseudo-inverse of D).  After an accepted step LM_lambda falls back
// to 0 (Sim3Tracker.cpp:336-337), so the system is only positive SEMI-definite when the scale / depth rows are weakly constrained: the
// unpivoted form of round 5 divided by whatever stood on the diagonal and returned inf / NaN increments there, i.e. a job
// reported as diverged where the reference takes a finite step.  Same solution as the reference's factorisation up to rounding
// (ldlt7_solve above stays as the CPU-checked form, tests/test_host_math_cpu.py); tests/test_sim3_gpu.py::test_sim3_weak_depth_constraints.
__device__ __forceinline__ void gj7_solve_wave(const Sim3Track& J, Sim3Scratch& W, const int lane) {
  const int i = lane >> 3, j = lane & 7;
  const bool act = i < 7;
  const int ii = act ? i : 0;
  float m = sim3_damped_entry(J, ii, j);
  unsigned used = 0;                       // rows that have been pivots (uniform)
#pragma unroll
  for (int k = 0; k < 7; k++) {
    if (act) W.m[i][j] = m;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the pivot: largest |diagonal| among the rows not used yet, lowest index first (all seven reads are issued together)
    float dg[7];
#pragma unroll
    for (int r = 0; r < 7; r++) dg[r] = W.m[r][r];
    int p = 0;
    float best = -1.0f;
#pragma unroll
    for (int r = 0; r < 7; r++) {
      const float a = fabsf(dg[r]);
      const bool take = !((used >> r) & 1u) && a > best;
      best = take ? a : best;
      p = take ? r : p;
    }
    used |= 1u << p;
    const float d = W.m[p][p], rk = W.m[p][j], ck = W.m[ii][p];
    const float f = ck * lsd_rcp_exact(d);
    const float upd = m - f * rk;
    m = (i == p || d == 0.0f) ? m : upd;
    __builtin_amdgcn_wave_barrier();
  }
  if (act) W.m[i][j] = m;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 7) { const float d = W.m[lane][lane]; W.inc[lane] = d != 0.0f ? W.m[lane][7] * lsd_rcp_exact(d) : 0.0f; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// One launch per evaluation, as k_track_step does it for the SE3 tracker: every workgroup of a job first brings the job's state up to
// date — the totals of the evaluation the previous launch left in the rows of this launch's parity, the Levenberg-Marquardt decision
// and the next request, computed redundantly (same inputs, same instructions, same result in every workgroup; workgroup 0 writes it
// to the state of the other parity) — and then evaluates its strip of the request into the rows of the other parity.  A level is at
// most S3_NBMAX strips, so the redundant part reads little.  A finished job's launches leave at once.
// Measured and dropped on the way (profiles/r05_notes.md): 256-pixel tiles (300 workgroups on level 1 of a 640x480 keyframe, each
// adding up 300 rows before 4 us of tile work: no faster than the host-driven loop, 2.3x slower for eight jobs per launch), and a ticket
// per job with only the last workgroup adding up and deciding (the release fence and ticket of every workgroup and the cold reads behind
// the acquire cost more than the redundancy: 32 us per launch).
#ifdef LSD_DEVTOOLS
// phase stamps of workgroup 0 of job 0, per launch of a budget (LSDHIP_S3_TRACE=1 prints them: where a launch's time goes)
__device__ unsigned long long g_s3trace[64][8];
#define S3_MARK(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && traceSlot >= 0 && traceSlot < 64) g_s3trace[traceSlot][k] = wall_clock64(); } while (0)
#else
#define S3_MARK(k) do { } while (0)
#endif
__global__ __launch_bounds__(S3_BLOCK) void k_sim3_fused(const Sim3Set* __restrict__ sets, Sim3Track* __restrict__ st, const Sim3LM P, const int parity,
                                                          const int budgetEnd, const int traceSlot) {
  __shared__ Sim3Track s_J;
  __shared__ Eval s_ev;
  __shared__ Sim3Scratch s_W;
  __shared__ float s_tot[64];
  const int tid = threadIdx.x, job = blockIdx.y;
  const Sim3Set& set = sets[job];
  const Sim3Track* src = st + parity * S3_MAXB + job;
  Sim3Track* dst = st + (parity ^ 1) * S3_MAXB + job;
  S3_MARK(0);
  for (int i = tid; i < (int)(sizeof(Sim3Track) / 4); i += S3_BLOCK) ((unsigned*)&s_J)[i] = ((const unsigned*)src)[i];
  __syncthreads();
  const bool wasDone = s_J.phase == Sim3Track::DONE;
  if ((int)blockIdx.x >= set.lv[s_J.finalLevel].nblocks) return;   // no strip on any level to come (the final level is the finest)
  S3_MARK(1);
  if (!wasDone && s_J.pendingEval) {
    Sim3Job a;
    sim3_job_view(set, s_J, parity, a);
    sim3_totals(a, s_tot);
    S3_MARK(2);
    if (tid < 49) s_ev.A[tid] = sim3_system_A(s_tot, tid / 7, tid % 7);
    else if (tid < 56) s_ev.b[tid - 49] = sim3_system_b(s_tot, tid - 49);
    else if (tid == 64) sim3_eval_scalars(s_tot, &s_ev);
    __syncthreads();
    if (tid == 0) {
      s_W.solve = 0;
      sim3_advance(P, s_J, s_ev, s_W);
    }
    __syncthreads();
    S3_MARK(3);
    if (s_W.solve) {
      if (tid < 64) {
        gj7_solve_wave(s_J, s_W, tid);
        // the exponential's transcendental values side by side: sin / cos of theta and of theta / 2 in two lanes, exp(sigma) in a third
        if (tid < 3) {
          const double incd[7] = {0, 0, 0, (double)s_W.inc[3], (double)s_W.inc[4], (double)s_W.inc[5], (double)s_W.inc[6]};
          const double theta = sim3_theta(incd);
          if (tid < 2) {
            double sn, cs;
            sincos(tid == 0 ? theta : 0.5 * theta, &sn, &cs);
            if (tid == 0) { s_W.trig.sin_theta = sn; s_W.trig.cos_theta = cs; }
            else { s_W.trig.sin_half = sn; s_W.trig.cos_half = cs; }
          } else {
            s_W.trig.exp_sigma = exp(incd[6]);
          }
        }
      }
      __syncthreads();
      S3_MARK(4);
      if (tid == 0) sim3_propose_finish(P, s_J, s_W);
    } else {
      S3_MARK(4);
    }
    if (tid == 0 && s_J.phase != Sim3Track::DONE) sim3_request(s_J);
    __syncthreads();
  }
  S3_MARK(5);
  const bool done = s_J.phase == Sim3Track::DONE;
  if (blockIdx.x == 0) {
    if (tid == 0) s_J.pendingEval = done ? 0 : 1;
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(Sim3Track) / 4); i += S3_BLOCK) ((unsigned*)dst)[i] = ((const unsigned*)&s_J)[i];
    if ((done && !wasDone) || (!done && budgetEnd >= 0)) {
      if (done)
        for (int i = tid; i < (int)(sizeof(lsdhip_sim3_result) / 4); i += S3_BLOCK) ((unsigned*)set.result)[i] = ((const unsigned*)&s_J.res)[i];
      const int word = set.seq * 256 + (done ? (s_J.rc == LSDHIP_OK ? 255 : 254) : budgetEnd);
      if (done && tid == 0) {
        // check word of the record (position-weighted sum of its words + the flag's value): the host takes the record only when it adds up
        unsigned chk = (unsigned)word;
        for (unsigned i = 0; i < (unsigned)(sizeof(lsdhip_sim3_result) / 4); i++) chk += lsd_summary_term(i, ((const unsigned*)&s_J.res)[i]);
        *(volatile unsigned*)(set.done + S3_MAXB) = chk;
      }
      __threadfence_system();                   // the result record (pinned) before the flag
      __syncthreads();
      if (tid == 0) {
        *(volatile int*)set.done = word;
        __threadfence_system();
      }
    }
  }
  S3_MARK(6);
  if (done) return;
  Sim3Job a;
  sim3_job_view(set, s_J, parity ^ 1, a);
  if ((int)blockIdx.x >= a.nblocks) return;
  sim3_eval_strip(a);
  S3_MARK(7);
}

// the device-side job storage of a tracker (first trackFrameSim3 call)
static int sim3_device_storage(lsdhip_sim3tracker* t) {
  if (t->d_sets) return LSDHIP_OK;
  // one block on each side — [S3_MAXB] level descriptions, then the states ([2][S3_MAXB] on the device) — so that a call uploads both in one copy
  static_assert((sizeof(Sim3Set) * S3_MAXB) % alignof(Sim3Track) == 0, "the states behind the level descriptions keep their alignment");
  uint8_t *dblk = nullptr, *hblk = nullptr;
  HIPCHK(hipMalloc((void**)&dblk, sizeof(Sim3Set) * S3_MAXB + sizeof(Sim3Track) * S3_MAXB * 2));
  HIPCHK(hipHostMalloc((void**)&hblk, sizeof(Sim3Set) * S3_MAXB + sizeof(Sim3Track) * S3_MAXB, hipHostMallocDefault));
  t->d_sets = (Sim3Set*)dblk; t->d_states = (Sim3Track*)(dblk + sizeof(Sim3Set) * S3_MAXB);
  t->h_sets = (Sim3Set*)hblk; t->h_states = (Sim3Track*)(hblk + sizeof(Sim3Set) * S3_MAXB);
  HIPCHK(hipHostMalloc((void**)&t->h_results, sizeof(lsdhip_sim3_result) * S3_MAXB, hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&t->d_results, t->h_results, 0));
  HIPCHK(hipHostMalloc((void**)&t->h_done, sizeof(int) * S3_MAXB * 2, hipHostMallocMapped));   // [k]: flag, [S3_MAXB + k]: check word of result k
  memset(t->h_done, 0, sizeof(int) * S3_MAXB * 2);
  HIPCHK(hipHostGetDevicePointer((void**)&t->d_done, t->h_done, 0));
  return LSDHIP_OK;
}
static Sim3LM sim3_lm_params(const lsdhip_sim3tracker* t) {
  Sim3LM P;
  for (int l = 0; l < LSD_LEVELS; l++) P.maxIts[l] = t->maxItsPerLvl[l];
  P.lambdaSuccessFac = t->lambdaSuccessFac; P.lambdaFailFac = t->lambdaFailFac; P.lambdaInitial = t->lambdaInitial;
  P.stepSizeMin = t->stepSizeMin; P.convergenceEps = t->convergenceEps;
  P.useAffine = t->ctx->params.useAffineLightningEstimation; P.w = t->ctx->w; P.h = t->ctx->h;
  return P;
}
// n independent trackFrameSim3 jobs: their states and level descriptions go to the device, a budget of (evaluation, step) launch pairs is
// queued — every pair advances all unfinished jobs by one evaluation (at most S3_MAXB jobs per launch) —, the host waits for the jobs'
// `done` words (pinned) and tops the budget up if some job needs more
static int sim3_track_jobs(lsdhip_sim3tracker* t, std::vector<Sim3Track>& jobs, const std::vector<std::pair<lsdhip_frame*, lsdhip_frame*>>& frames) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  const int n = (int)jobs.size();
  const Sim3LM P = sim3_lm_params(t);
  LsdTrackJobScope tjob_(c, true);   // pipelined contexts: behind the mapping stream's products, and a record point for it afterwards
  if (tjob_.rc) return tjob_.rc;
  { int rc = sim3_device_storage(t); if (rc) return rc; }
  for (int base = 0; base < n; base += S3_MAXB) {
    const int m = std::min(S3_MAXB, n - base);
    const int seq = t->seq = (t->seq % 0x3FFFFF) + 1;
    int grid = 0, pending = 0;
    for (int k = 0; k < m; k++) {
      Sim3Track& J = jobs[base + k];
      Sim3Set& S = t->h_sets[k];
      for (int l = 0; l < LSD_LEVELS; l++) {
        Sim3Lvl& L = S.lv[l];
        lsdhip_frame* kf = frames[base + k].first;
        lsdhip_frame* fr = frames[base + k].second;
        if (l == 0) { (void)lsd_frame_require_level0_for_tracking(kf); (void)lsd_frame_require_level0_for_tracking(fr); }
        L.kf_idepth = kf->d_idepth[l]; L.kf_idepthVar = kf->d_idepthVar[l]; L.kf_image = kf->d_image[l]; L.kf_grad = kf->d_grad[l];
        L.fr_grad = fr->d_grad[l]; L.fr_idepth = fr->d_idepth[l]; L.fr_idepthVar = fr->d_idepthVar[l];
        L.w = c->wl[l]; L.h = c->hl[l];
        sim3_strips(L.w * L.h, &L.nblocks, &L.ppl);
        const LevelIntr& in = c->intr[l];
        L.fx = in.fx; L.fy = in.fy; L.cx = in.cx; L.cy = in.cy; L.fxi = in.fxi; L.fyi = in.fyi; L.cxi = in.cxi; L.cyi = in.cyi;
        if (l == J.finalLevel && J.phase != Sim3Track::DONE && L.nblocks > grid) grid = L.nblocks;
      }
      S.cameraPixelNoise2 = c->params.cameraPixelNoise2; S.var_weight = t->var_weight; S.huber_d = t->huber_d;
      for (int par = 0; par < 2; par++) {
        S.rows[par] = t->d_rows + ((size_t)par * S3_MAXB + k) * S3_NBMAX * 64;
        S.topkey[par] = t->d_topkey + ((size_t)par * S3_MAXB + k) * S3_NBMAX;
      }
      S.result = t->d_results + k;     // pinned, written by the device when the job ends
      S.done = t->d_done + k;
      S.seq = seq;
      if (J.phase != Sim3Track::DONE) { sim3_request(J); pending++; }
      J.pendingEval = 0;
      t->h_states[k] = J;
    }
    if (!pending) continue;
    HIPCHK(hipMemcpyAsync(t->d_sets, t->h_sets, sizeof(Sim3Set) * S3_MAXB + sizeof(Sim3Track) * (size_t)m, hipMemcpyHostToDevice, c->stream));
    int budget = t->recentRounds > 0 ? t->recentRounds + 3 : 24, parity = 0;   // evaluations + 1 launches end a job
    for (int b = 0;; b++) {
      if (b >= 250) { lsd_set_error("Sim3 tracking did not terminate"); return LSDHIP_E_STATE; }
      for (int i = 0; i < budget; i++, parity ^= 1)
        hipLaunchKernelGGL(k_sim3_fused, dim3(grid, m), dim3(S3_BLOCK), 0, c->stream, (const Sim3Set*)t->d_sets, t->d_states, P, parity,
                           i == budget - 1 ? b : -1, b == 0 ? i : -1);
      HIPCHK(hipGetLastError());
      // every pending job raises its word when it finishes or when the budget's last step leaves it unfinished; no stream query in
      // the wait (each one puts a marker packet into the queue the chain runs through) but as a safety net
      const auto tStart = std::chrono::steady_clock::now();
      bool unfinished = false;
      for (int k = 0; k < m; k++) {
        if (jobs[base + k].phase == Sim3Track::DONE) continue;
        volatile const int* flag = (volatile const int*)t->h_done + k;
        unsigned spins = 0;
        int v;
        auto settled = [&](int x) { return x == seq * 256 + 255 || x == seq * 256 + 254 || x == seq * 256 + b; };
        while (!settled(v = *flag)) {
          if ((++spins & 0xFFFFFu) == 0) {
            hipError_t q = hipStreamQuery(c->stream);
            if (q != hipSuccess && q != hipErrorNotReady) { lsd_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return LSDHIP_E_HIP; }
            if (std::chrono::steady_clock::now() - tStart > std::chrono::seconds(10)) {
              HIPCHK(hipStreamSynchronize(c->stream));
              if (!settled(v = *flag)) { lsd_set_error("Sim3 tracking did not complete"); return LSDHIP_E_STATE; }
              break;
            }
          }
          __builtin_ia32_pause();
        }
        if (v == seq * 256 + b) unfinished = true;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (!unfinished) break;
      budget = 8;
    }
    // results: the pinned records the device wrote before it raised the words (launches of the budget still queued behind a job's end
    // leave at once; the stream orders them before whatever comes next)
    int rounds = 0;
    for (int k = 0; k < m; k++) {
      Sim3Track& J = jobs[base + k];
      if (J.phase == Sim3Track::DONE) continue;             // finished before any evaluation (no level with iterations)
      {
        // the record is taken once its words add up to the check word the device stored with it (they are separate posted writes: the
        // flag has been seen to overtake the tail of such a record, profiles/r06_notes.md section 1)
        const unsigned word = (unsigned)((volatile int*)t->h_done)[k];
        volatile const unsigned* rw = (volatile const unsigned*)&t->h_results[k];
        const auto tv0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
          unsigned chk = word;
          for (unsigned i = 0; i < (unsigned)(sizeof(lsdhip_sim3_result) / 4); i++) chk += lsd_summary_term(i, rw[i]);
          if (chk == ((volatile unsigned*)t->h_done)[S3_MAXB + k]) break;
          if (spins == 0) t->lateRecords++;
          if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - tv0 > std::chrono::seconds(2)) {
            lsd_set_error("Sim3 result record in pinned memory never became consistent"); return LSDHIP_E_STATE;
          }
          __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      *J.hostOut = t->h_results[k];
      J.rc = ((volatile int*)t->h_done)[k] == seq * 256 + 255 ? LSDHIP_OK : LSDHIP_DIVERGED;
      J.phase = Sim3Track::DONE;
      if (J.hostOut->numEvaluations > rounds) rounds = J.hostOut->numEvaluations;
    }
    t->recentRounds = rounds;
#ifdef LSD_DEVTOOLS
    if (getenv("LSDHIP_S3_TRACE")) {
      HIPCHK(hipStreamSynchronize(c->stream));
      static unsigned long long h[64][8];
      HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_s3trace), sizeof(h)));
      const char* names[7] = {"state", "totals", "system+advance", "solve", "finish+request", "publish", "strip"};
      double sum[8] = {0}; int cnt = 0; double span = 0;
      for (int i = 1; i + 1 < rounds && i < 63; i++) {     // launches with a pending evaluation and a strip of their own
        for (int k = 0; k < 7; k++) sum[k] += (double)(h[i][k + 1] - h[i][k]) * 0.01;
        span += (double)(h[i + 1][0] - h[i][0]) * 0.01;
        cnt++;
      }
      if (cnt) {
        fprintf(stderr, "[s3trace] %d launches, start-to-start %.2f us:", cnt, span / cnt);
        for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.2f", names[k], sum[k] / cnt);
        fprintf(stderr, " (us, workgroup 0 of job 0)\n");
      }
    }
#endif
  }
  return LSDHIP_OK;
}
static int sim3_init_job(lsdhip_sim3tracker* t, Sim3Track& J, lsdhip_frame* kf, lsdhip_frame* frame, const double init[8], int startLevel,
                         int finalLevel, lsdhip_sim3_result* out) {
  if (!kf || !frame || !init || !out) return LSDHIP_E_ARG;
  if (!kf->hasIDepth || !frame->hasIDepth) { lsd_set_error("Sim3 tracking needs inverse depth on both frames"); return LSDHIP_E_STATE; }
  memset(out, 0, sizeof(*out));
  out->frameToReference[0] = 1; out->frameToReference[7] = 1;   // Sim3() on failure
  memset(&J, 0, sizeof(J));
  J.res = *out; J.hostOut = out;
  J.phase = Sim3Track::DONE;
  J.aff_a = 1; J.aff_b = 0;
  J.referenceToFrame = sim3_inverse(sim3_in(init));
  memset(&J.cur, 0, sizeof(J.cur));
  memset(&J.finalResidual, 0, sizeof(J.finalResidual));
  memset(&J.lastErr, 0, sizeof(J.lastErr));
  J.warp_update_up_to_date = 0;
  J.numEvaluations = 0;
  J.lvl = startLevel; J.finalLevel = finalLevel;
  J.rc = LSDHIP_OK;
  sim3_next_level(sim3_lm_params(t), J);
  if (J.phase == Sim3Track::DONE) *out = J.res;   // no level with iterations
  return LSDHIP_OK;
}

extern "C" int lsdhip_sim3tracker_track(lsdhip_sim3tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const double init[8], int startLevel,
                                        int finalLevel, lsdhip_sim3_result* out) {
  if (!t || !kf || !frame || !init || !out || startLevel < finalLevel || finalLevel < 0 || startLevel >= LSD_LEVELS) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(t->ctx->device));
  std::vector<Sim3Track> jobs(1);
  int rc = sim3_init_job(t, jobs[0], kf, frame, init, startLevel, finalLevel, out);
  if (rc) return rc;
  rc = sim3_track_jobs(t, jobs, {{kf, frame}});
  if (rc) return rc;
  return jobs[0].rc;
}

extern "C" int lsdhip_sim3tracker_track_batch(lsdhip_sim3tracker* t, int n, lsdhip_frame** keyframes, lsdhip_frame** frames, const double* inits,
                                              int startLevel, int finalLevel, lsdhip_sim3_result* results) {
  if (!t || n <= 0 || !keyframes || !frames || !inits || !results || startLevel < finalLevel || finalLevel < 0 || startLevel >= LSD_LEVELS)
    return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(t->ctx->device));
  std::vector<Sim3Track> jobs((size_t)n);
  std::vector<std::pair<lsdhip_frame*, lsdhip_frame*>> pairs((size_t)n);
  for (int j = 0; j < n; j++) {
    int rc = sim3_init_job(t, jobs[j], keyframes[j], frames[j], inits + 8 * (size_t)j, startLevel, finalLevel, &results[j]);
    if (rc) return rc;
    pairs[j] = {keyframes[j], frames[j]};
  }
  int rc = sim3_track_jobs(t, jobs, pairs);
  if (rc) return rc;
  int rcAll = LSDHIP_OK;
  for (int j = 0; j < n; j++) if (jobs[j].rc == LSDHIP_DIVERGED) rcAll = LSDHIP_DIVERGED;
  return rcAll;
}
