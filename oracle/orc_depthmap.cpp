// ORACLE — TEST INFRASTRUCTURE ONLY (see lsd_oracle.hpp).  PARITY UNPINNED.
// Restates C/DepthEstimation/DepthMap.cpp (everything except debug drawing / statistics) and the row-chunk
// scheduling of C/util/IndexThreadReduce.h:68-187.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "lsd_oracle.hpp"

namespace orc {
long long orc_walk_hist[64] = {0};

// ---- C/util/settings.h:34-35, :50-174 ---------------------------------------------------------
#define DIVISION_EPS 1e-10f
#define UNZERO(val) (val < 0 ? (val > -1e-10 ? -1e-10 : val) : (val < 1e-10 ? 1e-10 : val))
#define VALIDITY_COUNTER_MAX (5.0f)
#define VALIDITY_COUNTER_MAX_VARIABLE (250.0f)
#define VALIDITY_COUNTER_INC 5
#define VALIDITY_COUNTER_DEC 5
#define VALIDITY_COUNTER_INITIAL_OBSERVE 5
#define VAL_SUM_MIN_FOR_CREATE (30)
#define VAL_SUM_MIN_FOR_KEEP (24)
#define VAL_SUM_MIN_FOR_UNBLACKLIST (100)
#define MIN_BLACKLIST -1
#define SUCC_VAR_INC_FAC (1.01f)
#define FAIL_VAR_INC_FAC 1.1f
#define MAX_VAR (0.5f * 0.5f)
#define VAR_GT_INIT_INITIAL 0.01f * 0.01f
#define VAR_RANDOM_INIT_INITIAL (0.5f * MAX_VAR)
#define MIN_DEPTH 0.05f
#define MAX_EPL_LENGTH_CROP 30.0f
#define MIN_EPL_LENGTH_CROP (3.0f)
#define GRADIENT_SAMPLE_DIST 1.0f
#define SAMPLE_POINT_TO_BORDER 7
#define MAX_ERROR_STEREO (1300.0f)
#define MIN_DISTANCE_ERROR_STEREO (1.5f)
#define STEREO_EPL_VAR_FAC 2.0f
#define REG_DIST_VAR (0.075f * 0.075f * params.depthSmoothingFactor * params.depthSmoothingFactor)
#define DIFF_FAC_SMOOTHING (1.0f * 1.0f)
#define DIFF_FAC_OBSERVE (1.0f * 1.0f)
#define DIFF_FAC_PROP_MERGE (1.0f * 1.0f)
#define MIN_EPL_GRAD_SQUARED (2.0f * 2.0f)
#define MIN_EPL_LENGTH_SQUARED (1.0f * 1.0f)
#define MIN_EPL_ANGLE_SQUARED (0.3f * 0.3f)
#define MIN_ABS_GRAD_CREATE (params.minUseGrad)
#define MIN_ABS_GRAD_DECREASE (params.minUseGrad)
#define MAX_DIFF_CONSTANT (40.0f * 40.0f)
#define MAX_DIFF_GRAD_MULT (0.5f * 0.5f)
#define SE3TRACKING_MIN_LEVEL ORC_SE3TRACKING_MIN_LEVEL

// DepthMap.cpp:41-83
DepthMap::DepthMap(int w, int h, const float Kin[4], const Params& p) : params(p), width(w), height(h) {
  otherDepthMap = new DepthMapPixelHypothesis[(size_t)width * height];
  currentDepthMap = new DepthMapPixelHypothesis[(size_t)width * height];
  memset((void*)otherDepthMap, 0, sizeof(DepthMapPixelHypothesis) * width * height);
  memset((void*)currentDepthMap, 0, sizeof(DepthMapPixelHypothesis) * width * height);
  validityIntegralBuffer = (int*)calloc((size_t)width * height, sizeof(int));
  memset(&K, 0, sizeof(K));
  K(0, 0) = Kin[0]; K(1, 1) = Kin[1]; K(0, 2) = Kin[2]; K(1, 2) = Kin[3]; K(2, 2) = 1;
  fx = K(0, 0); fy = K(1, 1); cx = K(0, 2); cy = K(1, 2);
  KInv = inverse3(K);
  fxi = KInv(0, 0); fyi = KInv(1, 1); cxi = KInv(0, 2); cyi = KInv(1, 2);
  reset();
}
DepthMap::~DepthMap() {
  delete[] otherDepthMap;
  delete[] currentDepthMap;
  free(validityIntegralBuffer);
}
// DepthMap.cpp:102-108
void DepthMap::reset() {
  for (int i = 0; i < width * height; i++) { otherDepthMap[i].isValid = false; currentDepthMap[i].isValid = false; }
}

// IndexThreadReduce.h:68-123 / :147-187: a PERSISTENT pool of MAPPING_THREADS workers, woken per reduce() call, each pulling
// [todo, todo + stepSize) chunks under the pool mutex until the range is exhausted; reduce() returns when all are idle again.
// (Restated with std::thread / condition_variable; the calling thread only waits, as in the reference.)
struct DepthMap::Pool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable todo_signal, done_signal;
  std::function<void(int, int)> call;
  int nextIndex = 0, maxIndex = 0, stepSize = 1, busy = 0;
  unsigned generation = 0;
  bool running = true;
  explicit Pool(int n) {
    for (int i = 0; i < n; i++) workers.emplace_back([this] { loop(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> l(m); running = false; }
    todo_signal.notify_all();
    for (auto& t : workers) t.join();
  }
  void loop() {
    std::unique_lock<std::mutex> lock(m);
    while (running) {
      if (nextIndex < maxIndex) {
        const int todo = nextIndex;
        nextIndex += stepSize;
        busy++;
        lock.unlock();
        call(todo, std::min(todo + stepSize, maxIndex));
        lock.lock();
        busy--;
        if (nextIndex >= maxIndex && busy == 0) done_signal.notify_all();
      } else {
        todo_signal.wait(lock);
      }
    }
  }
  void run(std::function<void(int, int)> f, int first, int end, int step) {
    std::unique_lock<std::mutex> lock(m);
    call = std::move(f);
    nextIndex = first; maxIndex = end; stepSize = step;
    todo_signal.notify_all();
    done_signal.wait(lock, [this] { return nextIndex >= maxIndex && busy == 0; });
    nextIndex = maxIndex = 0;
  }
};
template <typename F> void DepthMap::reduce(F f, int first, int end, int stepSize) {
  if (!params.multiThreading || numThreads <= 1) { f(first, end); return; }
  if (stepSize == 0) stepSize = ((end - first) + numThreads - 1) / numThreads;
  if (!pool || (int)pool->workers.size() != numThreads) pool.reset(new Pool(numThreads));
  pool->run(f, first, end, stepSize);
}

// DepthMap.cpp:111-146
void DepthMap::observeDepthRow(int yMin, int yMax) {
  const float* keyFrameMaxGradBuf = activeKeyFrame->maxGradients(0);
  for (int y = yMin; y < yMax; y++)
    for (int x = 3; x < width - 3; x++) {
      int idx = x + y * width;
      DepthMapPixelHypothesis* target = currentDepthMap + idx;
      bool hasHypothesis = target->isValid;
      if (hasHypothesis && keyFrameMaxGradBuf[idx] < MIN_ABS_GRAD_DECREASE) { target->isValid = false; continue; }
      if (keyFrameMaxGradBuf[idx] < MIN_ABS_GRAD_CREATE || target->blacklisted < MIN_BLACKLIST) continue;
      if (!hasHypothesis) observeDepthCreate(x, y, idx);
      else observeDepthUpdate(x, y, idx, keyFrameMaxGradBuf);
    }
}
// DepthMap.cpp:147-150
void DepthMap::observeDepth() {
  activeKeyFrame->maxGradients(0);
  activeKeyFrame->gradients(0);
  for (Frame* f : referenceFrameByID) { f->image(0); }
  reduce([this](int a, int b) { observeDepthRow(a, b); }, 3, height - 3, 10);
}

// DepthMap.cpp:184-234
bool DepthMap::makeAndCheckEPL(const int x, const int y, const Frame* const ref, float* pepx, float* pepy) {
  int idx = x + y * width;
  float epx = -fx * ref->thisToOther_t[0] + ref->thisToOther_t[2] * (x - cx);
  float epy = -fy * ref->thisToOther_t[1] + ref->thisToOther_t[2] * (y - cy);
  if (std::isnan(epx + epy)) return false;
  float eplLengthSquared = epx * epx + epy * epy;
  if (eplLengthSquared < MIN_EPL_LENGTH_SQUARED) return false;
  float gx = activeKeyFrameImageData[idx + 1] - activeKeyFrameImageData[idx - 1];
  float gy = activeKeyFrameImageData[idx + width] - activeKeyFrameImageData[idx - width];
  float eplGradSquared = gx * epx + gy * epy;
  eplGradSquared = eplGradSquared * eplGradSquared / eplLengthSquared;
  if (eplGradSquared < MIN_EPL_GRAD_SQUARED) return false;
  if (eplGradSquared / (gx * gx + gy * gy) < MIN_EPL_ANGLE_SQUARED) return false;
  // unqualified sqrt() in the reference (DepthMap.cpp:229) is the C library's double sqrt(double): with the libstdc++ of
  // the reference's era <cmath> leaves only that overload in the global namespace, so the quotient is a double
  float fac = (float)((double)GRADIENT_SAMPLE_DIST / sqrt((double)eplLengthSquared));
  *pepx = epx * fac;
  *pepy = epy * fac;
  return true;
}

// DepthMap.cpp:237-292
bool DepthMap::observeDepthCreate(int x, int y, int idx) {
  DepthMapPixelHypothesis* target = currentDepthMap + idx;
  Frame* refFrame = activeKeyFrameIsReactivated ? newest_referenceFrame : oldest_referenceFrame;
  if (refFrame->trackingParent == activeKeyFrame) {
    uint8_t* wasGoodDuringTracking = refFrame->refPixelWasGoodNoCreate();
    if (wasGoodDuringTracking != 0 &&
        !wasGoodDuringTracking[(x >> SE3TRACKING_MIN_LEVEL) + (width >> SE3TRACKING_MIN_LEVEL) * (y >> SE3TRACKING_MIN_LEVEL)])
      return false;
  }
  float epx, epy;
  bool isGood = makeAndCheckEPL(x, y, refFrame, &epx, &epy);
  if (!isGood) return false;
  float new_u = x;
  float new_v = y;
  float result_idepth, result_var, result_eplLength;
  float error = doLineStereo(new_u, new_v, epx, epy, 0.0f, 1.0f, 1.0f / MIN_DEPTH, refFrame, refFrame->image(0),
                             result_idepth, result_var, result_eplLength);
  if (error == -3 || error == -2) target->blacklisted--;
  if (error < 0 || result_var > MAX_VAR) return false;
  result_idepth = UNZERO(result_idepth);
  *target = DepthMapPixelHypothesis(result_idepth, result_var, VALIDITY_COUNTER_INITIAL_OBSERVE);
  return true;
}

// DepthMap.cpp:294-473
bool DepthMap::observeDepthUpdate(int x, int y, int idx, const float* keyFrameMaxGradBuf) {
  DepthMapPixelHypothesis* target = currentDepthMap + idx;
  Frame* refFrame;
  if (!activeKeyFrameIsReactivated) {
    if ((int)target->nextStereoFrameMinID - referenceFrameByID_offset >= (int)referenceFrameByID.size()) return false;
    if ((int)target->nextStereoFrameMinID - referenceFrameByID_offset < 0) refFrame = oldest_referenceFrame;
    else refFrame = referenceFrameByID[(int)target->nextStereoFrameMinID - referenceFrameByID_offset];
  } else
    refFrame = newest_referenceFrame;

  if (refFrame->trackingParent == activeKeyFrame) {
    uint8_t* wasGoodDuringTracking = refFrame->refPixelWasGoodNoCreate();
    if (wasGoodDuringTracking != 0 &&
        !wasGoodDuringTracking[(x >> SE3TRACKING_MIN_LEVEL) + (width >> SE3TRACKING_MIN_LEVEL) * (y >> SE3TRACKING_MIN_LEVEL)])
      return false;
  }
  float epx, epy;
  bool isGood = makeAndCheckEPL(x, y, refFrame, &epx, &epy);
  if (!isGood) return false;

  float sv = (float)sqrt((double)target->idepth_var_smoothed);  // double sqrt, one rounding: == sqrtf
  float min_idepth = target->idepth_smoothed - sv * STEREO_EPL_VAR_FAC;
  float max_idepth = target->idepth_smoothed + sv * STEREO_EPL_VAR_FAC;
  if (min_idepth < 0) min_idepth = 0;
  if (max_idepth > 1 / MIN_DEPTH) max_idepth = 1 / MIN_DEPTH;

  float result_idepth = 0, result_var = 0, result_eplLength = 0;  // uninitialised in the reference (:344)
  float error = doLineStereo(x, y, epx, epy, min_idepth, target->idepth_smoothed, max_idepth, refFrame, refFrame->image(0),
                             result_idepth, result_var, result_eplLength);
  float diff = result_idepth - target->idepth_smoothed;

  if (error == -1) {
    return false;
  } else if (error == -2) {
    target->validity_counter -= VALIDITY_COUNTER_DEC;
    if (target->validity_counter < 0) target->validity_counter = 0;
    target->nextStereoFrameMinID = 0;
    target->idepth_var *= FAIL_VAR_INC_FAC;
    if (target->idepth_var > MAX_VAR) { target->isValid = false; target->blacklisted--; }
    return false;
  } else if (error == -3) {
    return false;
  } else if (error == -4) {
    return false;
  } else if (DIFF_FAC_OBSERVE * diff * diff > result_var + target->idepth_var_smoothed) {
    target->idepth_var *= FAIL_VAR_INC_FAC;
    if (target->idepth_var > MAX_VAR) target->isValid = false;
    return false;
  } else {
    float id_var = target->idepth_var * SUCC_VAR_INC_FAC;
    float w = result_var / (result_var + id_var);
    float new_idepth = (1 - w) * result_idepth + w * target->idepth;
    target->idepth = UNZERO(new_idepth);
    id_var = id_var * w;
    if (id_var < target->idepth_var) target->idepth_var = id_var;
    target->validity_counter += VALIDITY_COUNTER_INC;
    float absGrad = keyFrameMaxGradBuf[idx];
    if (target->validity_counter > VALIDITY_COUNTER_MAX + absGrad * (VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f)
      target->validity_counter = VALIDITY_COUNTER_MAX + absGrad * (VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f;
    if (result_eplLength < MIN_EPL_LENGTH_CROP) {
      float inc = activeKeyFrame->numFramesTrackedOnThis / (float)(activeKeyFrame->numMappedOnThis + 5);
      if (inc < 3) inc = 3;
      inc += ((int)(result_eplLength * 10000) % 2);
      if (result_eplLength < 0.5 * MIN_EPL_LENGTH_CROP) inc *= 3;
      target->nextStereoFrameMinID = refFrame->id() + inc;
    }
    return true;
  }
}

// DepthMap.cpp:475-653
void DepthMap::propagateDepth(Frame* new_keyframe) {
  for (int i = 0; i < width * height; i++) { otherDepthMap[i].isValid = false; otherDepthMap[i].blacklisted = 0; }

  // SE3 oldToNew_SE3 = se3FromSim3(new_keyframe->pose->thisToParent_raw).inverse();  (double, :503)
  SE3d newToOld;
  // se3FromSim3 = SE3(sim3.quaternion(), translation): the scaled quaternion, normalised by the SO3 constructor (so3.hpp:630-633)
  newToOld.q = new_keyframe->thisToParent_raw.q;
  { const double s_ = new_keyframe->thisToParent_raw.s; newToOld.q.w *= s_; newToOld.q.x *= s_; newToOld.q.y *= s_; newToOld.q.z *= s_; }
  qnormalize(newToOld.q);
  newToOld.t = new_keyframe->thisToParent_raw.t;
  SE3d oldToNew_SE3 = newToOld.inverse();
  V3f trafoInv_t = mk3<float>((float)oldToNew_SE3.t[0], (float)oldToNew_SE3.t[1], (float)oldToNew_SE3.t[2]);
  M3d Rd = oldToNew_SE3.rotationMatrix();
  M3f trafoInv_R;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) trafoInv_R(i, j) = (float)Rd(i, j);

  const uint8_t* trackingWasGood = new_keyframe->trackingParent == activeKeyFrame ? new_keyframe->refPixelWasGoodNoCreate() : 0;
  const float* activeKFImageData = activeKeyFrame->image(0);
  const float* newKFMaxGrad = new_keyframe->maxGradients(0);
  const float* newKFImageData = new_keyframe->image(0);

  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      DepthMapPixelHypothesis* source = currentDepthMap + x + y * width;
      if (!source->isValid) continue;
      V3f ray = mk3<float>(x * fxi + cxi, y * fyi + cyi, 1.0f);
      V3f Rr = matvec(trafoInv_R, ray);
      V3f pn = mk3<float>(Rr[0] / source->idepth_smoothed + trafoInv_t[0], Rr[1] / source->idepth_smoothed + trafoInv_t[1],
                          Rr[2] / source->idepth_smoothed + trafoInv_t[2]);
      float new_idepth = 1.0f / pn[2];
      float u_new = pn[0] * new_idepth * fx + cx;
      float v_new = pn[1] * new_idepth * fy + cy;
      if (!(u_new > 2.1f && v_new > 2.1f && u_new < width - 3.1f && v_new < height - 3.1f)) continue;
      int newIDX = (int)(u_new + 0.5f) + ((int)(v_new + 0.5f)) * width;
      float destAbsGrad = newKFMaxGrad[newIDX];
      if (trackingWasGood != 0) {
        if (!trackingWasGood[(x >> SE3TRACKING_MIN_LEVEL) + (width >> SE3TRACKING_MIN_LEVEL) * (y >> SE3TRACKING_MIN_LEVEL)] ||
            destAbsGrad < MIN_ABS_GRAD_DECREASE)
          continue;
      } else {
        float sourceColor = activeKFImageData[x + y * width];
        float destColor = getInterpolatedElement(newKFImageData, u_new, v_new, width);
        float residual = destColor - sourceColor;
        if (residual * residual / (MAX_DIFF_CONSTANT + MAX_DIFF_GRAD_MULT * destAbsGrad * destAbsGrad) > 1.0f ||
            destAbsGrad < MIN_ABS_GRAD_DECREASE)
          continue;
      }
      DepthMapPixelHypothesis* targetBest = otherDepthMap + newIDX;
      float idepth_ratio_4 = new_idepth / source->idepth_smoothed;
      idepth_ratio_4 *= idepth_ratio_4;
      idepth_ratio_4 *= idepth_ratio_4;
      float new_var = idepth_ratio_4 * source->idepth_var;
      if (targetBest->isValid) {
        float diff = targetBest->idepth - new_idepth;
        if (DIFF_FAC_PROP_MERGE * diff * diff > new_var + targetBest->idepth_var) {
          if (new_idepth < targetBest->idepth) continue;
          else targetBest->isValid = false;
        }
      }
      if (!targetBest->isValid) {
        *targetBest = DepthMapPixelHypothesis(new_idepth, new_var, source->validity_counter);
      } else {
        float w = new_var / (targetBest->idepth_var + new_var);
        float merged_new_idepth = w * targetBest->idepth + (1.0f - w) * new_idepth;
        int merged_validity = source->validity_counter + targetBest->validity_counter;
        if (merged_validity > VALIDITY_COUNTER_MAX + (VALIDITY_COUNTER_MAX_VARIABLE))
          merged_validity = VALIDITY_COUNTER_MAX + (VALIDITY_COUNTER_MAX_VARIABLE);
        *targetBest = DepthMapPixelHypothesis(merged_new_idepth, 1.0f / (1.0f / targetBest->idepth_var + 1.0f / new_var), merged_validity);
      }
    }
  std::swap(currentDepthMap, otherDepthMap);
}

// DepthMap.cpp:656-703
void DepthMap::regularizeDepthMapFillHolesRow(int yMin, int yMax) {
  const float* keyFrameMaxGradBuf = activeKeyFrame->maxGradients(0);
  for (int y = yMin; y < yMax; y++)
    for (int x = 3; x < width - 2; x++) {
      int idx = x + y * width;
      DepthMapPixelHypothesis* dest = otherDepthMap + idx;
      if (dest->isValid) continue;
      if (keyFrameMaxGradBuf[idx] < MIN_ABS_GRAD_DECREASE) continue;
      int* io = validityIntegralBuffer + idx;
      int val = io[2 + 2 * width] - io[2 - 3 * width] - io[-3 + 2 * width] + io[-3 - 3 * width];
      if ((dest->blacklisted >= MIN_BLACKLIST && val > VAL_SUM_MIN_FOR_CREATE) || val > VAL_SUM_MIN_FOR_UNBLACKLIST) {
        float sumIdepthObs = 0, sumIVarObs = 0;
        DepthMapPixelHypothesis* s1max = otherDepthMap + (x - 2) + (y + 3) * width;
        for (DepthMapPixelHypothesis* s1 = otherDepthMap + (x - 2) + (y - 2) * width; s1 < s1max; s1 += width)
          for (DepthMapPixelHypothesis* source = s1; source < s1 + 5; source++) {
            if (!source->isValid) continue;
            sumIdepthObs += source->idepth / source->idepth_var;
            sumIVarObs += 1.0f / source->idepth_var;
          }
        float idepthObs = sumIdepthObs / sumIVarObs;
        idepthObs = UNZERO(idepthObs);
        currentDepthMap[idx] = DepthMapPixelHypothesis(idepthObs, VAR_RANDOM_INIT_INITIAL, 0);
      }
    }
}
// DepthMap.cpp:706-718
void DepthMap::regularizeDepthMapFillHoles() {
  activeKeyFrame->maxGradients(0);
  buildRegIntegralBuffer();
  memcpy((void*)otherDepthMap, (void*)currentDepthMap, (size_t)width * height * sizeof(DepthMapPixelHypothesis));
  reduce([this](int a, int b) { regularizeDepthMapFillHolesRow(a, b); }, 3, height - 2, 10);
}
// DepthMap.cpp:722-740
void DepthMap::buildRegIntegralBufferRow1(int yMin, int yMax) {
  int* validityIntegralBufferPT = validityIntegralBuffer + yMin * width;
  DepthMapPixelHypothesis* ptSrc = currentDepthMap + yMin * width;
  for (int y = yMin; y < yMax; y++) {
    int validityIntegralBufferSUM = 0;
    for (int x = 0; x < width; x++) {
      if (ptSrc->isValid) validityIntegralBufferSUM += ptSrc->validity_counter;
      *(validityIntegralBufferPT++) = validityIntegralBufferSUM;
      ptSrc++;
    }
  }
}
// DepthMap.cpp:743-754
void DepthMap::buildRegIntegralBuffer() {
  reduce([this](int a, int b) { buildRegIntegralBufferRow1(a, b); }, 0, height, 0);
  int* validityIntegralBufferPT = validityIntegralBuffer;
  int* validityIntegralBufferPT_T = validityIntegralBuffer + width;
  int wh = height * width;
  for (int idx = width; idx < wh; idx++) *(validityIntegralBufferPT_T++) += *(validityIntegralBufferPT++);
}

// DepthMap.cpp:758-848
template <bool removeOcclusions> void DepthMap::regularizeDepthMapRow(int validityTH, int yMin, int yMax) {
  const int regularize_radius = 2;
  const float regDistVar = REG_DIST_VAR;
  for (int y = yMin; y < yMax; y++)
    for (int x = regularize_radius; x < width - regularize_radius; x++) {
      DepthMapPixelHypothesis* dest = currentDepthMap + x + y * width;
      DepthMapPixelHypothesis* destRead = otherDepthMap + x + y * width;
      if (!destRead->isValid) continue;
      float sum = 0, val_sum = 0, sumIvar = 0;
      int numOccluding = 0, numNotOccluding = 0;
      for (int dx = -regularize_radius; dx <= regularize_radius; dx++)
        for (int dy = -regularize_radius; dy <= regularize_radius; dy++) {
          DepthMapPixelHypothesis* source = destRead + dx + dy * width;
          if (!source->isValid) continue;
          float diff = source->idepth - destRead->idepth;
          if (DIFF_FAC_SMOOTHING * diff * diff > source->idepth_var + destRead->idepth_var) {
            if (removeOcclusions) { if (source->idepth > destRead->idepth) numOccluding++; }
            continue;
          }
          val_sum += source->validity_counter;
          if (removeOcclusions) numNotOccluding++;
          float distFac = (float)(dx * dx + dy * dy) * regDistVar;
          float ivar = 1.0f / (source->idepth_var + distFac);
          sum += source->idepth * ivar;
          sumIvar += ivar;
        }
      if (val_sum < validityTH) {
        dest->isValid = false;
        dest->blacklisted--;
        continue;
      }
      if (removeOcclusions) {
        if (numOccluding > numNotOccluding) { dest->isValid = false; continue; }
      }
      sum = sum / sumIvar;
      sum = UNZERO(sum);
      dest->idepth_smoothed = sum;
      dest->idepth_var_smoothed = 1.0f / sumIvar;
    }
}
// DepthMap.cpp:853-869
void DepthMap::regularizeDepthMap(bool removeOcclusions, int validityTH) {
  memcpy((void*)otherDepthMap, (void*)currentDepthMap, (size_t)width * height * sizeof(DepthMapPixelHypothesis));
  if (removeOcclusions) reduce([this, validityTH](int a, int b) { regularizeDepthMapRow<true>(validityTH, a, b); }, 2, height - 2, 10);
  else reduce([this, validityTH](int a, int b) { regularizeDepthMapRow<false>(validityTH, a, b); }, 2, height - 2, 10);
}

// DepthMap.cpp:883-916 (no srand in the reference => glibc default seed)
void DepthMap::initializeRandomly(Frame* new_frame) {
  activeKeyFrame = new_frame;
  activeKeyFrameImageData = activeKeyFrame->image(0);
  activeKeyFrameIsReactivated = false;
  const float* maxGradients = new_frame->maxGradients();
  for (int y = 1; y < height - 1; y++)
    for (int x = 1; x < width - 1; x++) {
      if (maxGradients[x + y * width] > MIN_ABS_GRAD_CREATE) {
        float idepth = 0.5f + 1.0f * ((rand() % 100001) / 100000.0f);
        currentDepthMap[x + y * width] = DepthMapPixelHypothesis(idepth, idepth, VAR_RANDOM_INIT_INITIAL, VAR_RANDOM_INIT_INITIAL, 20);
      } else {
        currentDepthMap[x + y * width].isValid = false;
        currentDepthMap[x + y * width].blacklisted = 0;
      }
    }
  activeKeyFrame->setDepth(currentDepthMap);
}

// DepthMap.cpp:920-962
void DepthMap::setFromExistingKF(Frame* kf) {
  activeKeyFrame = kf;
  const float* idepth = kf->idepth_reAct.data();
  const float* idepthVar = kf->idepthVar_reAct.data();
  const unsigned char* validity = kf->validity_reAct.data();
  DepthMapPixelHypothesis* pt = currentDepthMap;
  activeKeyFrame->numMappedOnThis = 0;
  activeKeyFrame->numFramesTrackedOnThis = 0;
  activeKeyFrameImageData = activeKeyFrame->image(0);
  activeKeyFrameIsReactivated = true;
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      if (*idepthVar > 0) {
        *pt = DepthMapPixelHypothesis(*idepth, *idepthVar, *validity);
      } else {
        currentDepthMap[x + y * width].isValid = false;
        currentDepthMap[x + y * width].blacklisted = (*idepthVar == -2) ? MIN_BLACKLIST - 1 : 0;
      }
      idepth++; idepthVar++; validity++; pt++;
    }
  regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP);
}

// DepthMap.cpp:965-1018
void DepthMap::initializeFromGTDepth(Frame* new_frame) {
  activeKeyFrame = new_frame;
  activeKeyFrameImageData = activeKeyFrame->image(0);
  activeKeyFrameIsReactivated = false;
  const float* idepth = new_frame->idepth();
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      float idepthValue = idepth[x + y * width];
      if (!std::isnan(idepthValue) && idepthValue > 0) {
        currentDepthMap[x + y * width] = DepthMapPixelHypothesis(idepthValue, idepthValue, VAR_GT_INIT_INITIAL, VAR_GT_INIT_INITIAL, 20);
      } else {
        currentDepthMap[x + y * width].isValid = false;
        currentDepthMap[x + y * width].blacklisted = 0;
      }
    }
  activeKeyFrame->setDepth(currentDepthMap);
}

// DepthMap.cpp:1079-1105 (the part of updateKeyframe that prepares the reference frames)
void DepthMap::setReferenceFrames(std::deque<std::shared_ptr<Frame>>& referenceFrames) {
  oldest_referenceFrame = referenceFrames.front().get();
  newest_referenceFrame = referenceFrames.back().get();
  referenceFrameByID.clear();
  referenceFrameByID_offset = oldest_referenceFrame->id();
  for (std::shared_ptr<Frame> frame : referenceFrames) {
    // only frames tracked on the active keyframe are in scope (the other branch, :1099, needs the pose graph)
    Sim3d refToKf = frame->thisToParent_raw;
    frame->prepareForStereoWith(activeKeyFrame, refToKf, K, 0);
    while ((int)referenceFrameByID.size() + referenceFrameByID_offset <= frame->id()) referenceFrameByID.push_back(frame.get());
  }
}

// DepthMap.cpp:1072-1213
void DepthMap::updateKeyframe(std::deque<std::shared_ptr<Frame>> referenceFrames) {
  setReferenceFrames(referenceFrames);
  observeDepth();
  regularizeDepthMapFillHoles();
  regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP);
  if (!activeKeyFrame->depthHasBeenUpdatedFlag) activeKeyFrame->setDepth(currentDepthMap);
  activeKeyFrame->numMappedOnThis++;
  activeKeyFrame->numMappedOnThisTotal++;
}

// DepthMap.cpp:1222-1327
void DepthMap::createKeyFrame(Frame* new_keyframe) {
  SE3d newToOld;
  // se3FromSim3 = SE3(sim3.quaternion(), translation): the scaled quaternion, normalised by the SO3 constructor (so3.hpp:630-633)
  newToOld.q = new_keyframe->thisToParent_raw.q;
  { const double s_ = new_keyframe->thisToParent_raw.s; newToOld.q.w *= s_; newToOld.q.x *= s_; newToOld.q.y *= s_; newToOld.q.z *= s_; }
  qnormalize(newToOld.q);
  newToOld.t = new_keyframe->thisToParent_raw.t;
  SE3d oldToNew_SE3 = newToOld.inverse();

  propagateDepth(new_keyframe);
  activeKeyFrame = new_keyframe;
  activeKeyFrameImageData = new_keyframe->image(0);
  activeKeyFrameIsReactivated = false;

  regularizeDepthMap(true, VAL_SUM_MIN_FOR_KEEP);
  regularizeDepthMapFillHoles();
  regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP);

  float sumIdepth = 0, numIdepth = 0;
  for (DepthMapPixelHypothesis* source = currentDepthMap; source < currentDepthMap + width * height; source++) {
    if (!source->isValid) continue;
    sumIdepth += source->idepth_smoothed;
    numIdepth++;
  }
  float rescaleFactor = numIdepth / sumIdepth;
  float rescaleFactor2 = rescaleFactor * rescaleFactor;
  for (DepthMapPixelHypothesis* source = currentDepthMap; source < currentDepthMap + width * height; source++) {
    if (!source->isValid) continue;
    source->idepth *= rescaleFactor;
    source->idepth_smoothed *= rescaleFactor;
    source->idepth_var *= rescaleFactor2;
    source->idepth_var_smoothed *= rescaleFactor2;
  }
  lastRescaleFactor = rescaleFactor;
  SE3d back = oldToNew_SE3.inverse();
  activeKeyFrame->thisToParent_raw.q = back.q;
  qnormalize(activeKeyFrame->thisToParent_raw.q);   // sim3FromSE3 -> Sim3::setScale normalises (rxso3.hpp:332-335)
  activeKeyFrame->thisToParent_raw.t = back.t;
  activeKeyFrame->thisToParent_raw.s = rescaleFactor;
  activeKeyFrame->setDepth(currentDepthMap);
}

// DepthMap.cpp:1363-1395
void DepthMap::finalizeKeyFrame() {
  regularizeDepthMapFillHoles();
  regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP);
  activeKeyFrame->setDepth(currentDepthMap);
  activeKeyFrame->takeReActivationData(currentDepthMap);
}

// C/util/globalFuncs.h:95-109
static inline void interp42(const float* mat4, float x, float y, int width, float out[2]) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat4 + 4 * (ix + iy * width);
  for (int c = 0; c < 2; c++)
    out[c] = dxdy * bp[4 * (1 + width) + c] + (dy - dxdy) * bp[4 * width + c] + (dx - dxdy) * bp[4 + c] + (1 - dx - dy + dxdy) * bp[c];
}

// DepthMap.cpp:1442-1972.  Returns the matching error (>= 0) or a status code -1 … -4.
// Deviation: the epipolar walk is capped at 1000 steps (the reference loop is unbounded and would hang on
// a NaN increment that slips through the :1510 test); no finite input reaches the cap.
float DepthMap::doLineStereo(const float u, const float v, const float epxn, const float epyn, const float min_idepth,
                             const float prior_idepth, float max_idepth, const Frame* const referenceFrame,
                             const float* referenceFrameImage, float& result_idepth, float& result_var,
                             float& result_eplLength) {
  V3f KinvP = mk3<float>(fxi * u + cxi, fyi * v + cyi, 1.0f);
  V3f pInf = matvec(referenceFrame->K_otherToThis_R, KinvP);
  V3f pReal = mk3<float>(pInf[0] / prior_idepth + referenceFrame->K_otherToThis_t[0],
                         pInf[1] / prior_idepth + referenceFrame->K_otherToThis_t[1],
                         pInf[2] / prior_idepth + referenceFrame->K_otherToThis_t[2]);
  float rescaleFactor = pReal[2] * prior_idepth;

  float firstX = u - 2 * epxn * rescaleFactor;
  float firstY = v - 2 * epyn * rescaleFactor;
  float lastX = u + 2 * epxn * rescaleFactor;
  float lastY = v + 2 * epyn * rescaleFactor;
  if (firstX <= 0 || firstX >= width - 2 || firstY <= 0 || firstY >= height - 2 || lastX <= 0 || lastX >= width - 2 ||
      lastY <= 0 || lastY >= height - 2)
    return -1;
  if (!(rescaleFactor > 0.7f && rescaleFactor < 1.4f)) return -1;

  float realVal_p1 = getInterpolatedElement(activeKeyFrameImageData, u + epxn * rescaleFactor, v + epyn * rescaleFactor, width);
  float realVal_m1 = getInterpolatedElement(activeKeyFrameImageData, u - epxn * rescaleFactor, v - epyn * rescaleFactor, width);
  float realVal = getInterpolatedElement(activeKeyFrameImageData, u, v, width);
  float realVal_m2 = getInterpolatedElement(activeKeyFrameImageData, u - 2 * epxn * rescaleFactor, v - 2 * epyn * rescaleFactor, width);
  float realVal_p2 = getInterpolatedElement(activeKeyFrameImageData, u + 2 * epxn * rescaleFactor, v + 2 * epyn * rescaleFactor, width);

  const V3f& Kt = referenceFrame->K_otherToThis_t;
  V3f pClose = mk3<float>(pInf[0] + Kt[0] * max_idepth, pInf[1] + Kt[1] * max_idepth, pInf[2] + Kt[2] * max_idepth);
  if (pClose[2] < 0.001f) {
    max_idepth = (0.001f - pInf[2]) / Kt[2];
    pClose = mk3<float>(pInf[0] + Kt[0] * max_idepth, pInf[1] + Kt[1] * max_idepth, pInf[2] + Kt[2] * max_idepth);
  }
  { float z = pClose[2]; pClose = mk3<float>(pClose[0] / z, pClose[1] / z, pClose[2] / z); }

  V3f pFar = mk3<float>(pInf[0] + Kt[0] * min_idepth, pInf[1] + Kt[1] * min_idepth, pInf[2] + Kt[2] * min_idepth);
  if (pFar[2] < 0.001f || max_idepth < min_idepth) return -1;
  { float z = pFar[2]; pFar = mk3<float>(pFar[0] / z, pFar[1] / z, pFar[2] / z); }

  if (std::isnan((float)(pFar[0] + pClose[0]))) return -4;

  float incx = pClose[0] - pFar[0];
  float incy = pClose[1] - pFar[1];
  float eplLength = (float)sqrt((double)(incx * incx + incy * incy));  // == sqrtf
  if (!eplLength > 0 || std::isinf(eplLength)) return -4;  // sic: true only for eplLength == 0 (or inf)

  if (eplLength > MAX_EPL_LENGTH_CROP) {
    pClose[0] = pFar[0] + incx * MAX_EPL_LENGTH_CROP / eplLength;
    pClose[1] = pFar[1] + incy * MAX_EPL_LENGTH_CROP / eplLength;
  }
  incx *= GRADIENT_SAMPLE_DIST / eplLength;
  incy *= GRADIENT_SAMPLE_DIST / eplLength;

  pFar[0] -= incx;
  pFar[1] -= incy;
  pClose[0] += incx;
  pClose[1] += incy;

  if (eplLength < MIN_EPL_LENGTH_CROP) {
    float pad = (MIN_EPL_LENGTH_CROP - (eplLength)) / 2.0f;
    pFar[0] -= incx * pad;
    pFar[1] -= incy * pad;
    pClose[0] += incx * pad;
    pClose[1] += incy * pad;
  }

  if (pFar[0] <= SAMPLE_POINT_TO_BORDER || pFar[0] >= width - SAMPLE_POINT_TO_BORDER || pFar[1] <= SAMPLE_POINT_TO_BORDER ||
      pFar[1] >= height - SAMPLE_POINT_TO_BORDER)
    return -1;

  if (pClose[0] <= SAMPLE_POINT_TO_BORDER || pClose[0] >= width - SAMPLE_POINT_TO_BORDER || pClose[1] <= SAMPLE_POINT_TO_BORDER ||
      pClose[1] >= height - SAMPLE_POINT_TO_BORDER) {
    if (pClose[0] <= SAMPLE_POINT_TO_BORDER) {
      float toAdd = (SAMPLE_POINT_TO_BORDER - pClose[0]) / incx;
      pClose[0] += toAdd * incx;
      pClose[1] += toAdd * incy;
    } else if (pClose[0] >= width - SAMPLE_POINT_TO_BORDER) {
      float toAdd = (width - SAMPLE_POINT_TO_BORDER - pClose[0]) / incx;
      pClose[0] += toAdd * incx;
      pClose[1] += toAdd * incy;
    }
    if (pClose[1] <= SAMPLE_POINT_TO_BORDER) {
      float toAdd = (SAMPLE_POINT_TO_BORDER - pClose[1]) / incy;
      pClose[0] += toAdd * incx;
      pClose[1] += toAdd * incy;
    } else if (pClose[1] >= height - SAMPLE_POINT_TO_BORDER) {
      float toAdd = (height - SAMPLE_POINT_TO_BORDER - pClose[1]) / incy;
      pClose[0] += toAdd * incx;
      pClose[1] += toAdd * incy;
    }
    float fincx = pClose[0] - pFar[0];
    float fincy = pClose[1] - pFar[1];
    float newEplLength = (float)sqrt((double)(fincx * fincx + fincy * fincy));  // == sqrtf
    if (pClose[0] <= SAMPLE_POINT_TO_BORDER || pClose[0] >= width - SAMPLE_POINT_TO_BORDER || pClose[1] <= SAMPLE_POINT_TO_BORDER ||
        pClose[1] >= height - SAMPLE_POINT_TO_BORDER || newEplLength < 8.0f)
      return -1;
  }

  float cpx = pFar[0];
  float cpy = pFar[1];
  float val_cp_m2 = getInterpolatedElement(referenceFrameImage, cpx - 2.0f * incx, cpy - 2.0f * incy, width);
  float val_cp_m1 = getInterpolatedElement(referenceFrameImage, cpx - incx, cpy - incy, width);
  float val_cp = getInterpolatedElement(referenceFrameImage, cpx, cpy, width);
  float val_cp_p1 = getInterpolatedElement(referenceFrameImage, cpx + incx, cpy + incy, width);
  float val_cp_p2;

  int loopCounter = 0;
  float best_match_x = -1;
  float best_match_y = -1;
  float best_match_err = 1e50;         // sic: +inf after the double -> float conversion
  float second_best_match_err = 1e50;
  float best_match_errPre = NAN, best_match_errPost = NAN, best_match_DiffErrPre = NAN, best_match_DiffErrPost = NAN;
  bool bestWasLastLoop = false;
  float eeLast = -1;
  float e1A = NAN, e1B = NAN, e2A = NAN, e2B = NAN, e3A = NAN, e3B = NAN, e4A = NAN, e4B = NAN, e5A = NAN, e5B = NAN;
  int loopCBest = -1, loopCSecond = -1;
  while ((((incx < 0) == (cpx > pClose[0]) && (incy < 0) == (cpy > pClose[1])) || loopCounter == 0) && loopCounter < 1000) {
    val_cp_p2 = getInterpolatedElement(referenceFrameImage, cpx + 2 * incx, cpy + 2 * incy, width);
    float ee = 0;
    if (loopCounter % 2 == 0) {
      e1A = val_cp_p2 - realVal_p2; ee += e1A * e1A;
      e2A = val_cp_p1 - realVal_p1; ee += e2A * e2A;
      e3A = val_cp - realVal;       ee += e3A * e3A;
      e4A = val_cp_m1 - realVal_m1; ee += e4A * e4A;
      e5A = val_cp_m2 - realVal_m2; ee += e5A * e5A;
    } else {
      e1B = val_cp_p2 - realVal_p2; ee += e1B * e1B;
      e2B = val_cp_p1 - realVal_p1; ee += e2B * e2B;
      e3B = val_cp - realVal;       ee += e3B * e3B;
      e4B = val_cp_m1 - realVal_m1; ee += e4B * e4B;
      e5B = val_cp_m2 - realVal_m2; ee += e5B * e5B;
    }
    if (ee < best_match_err) {
      second_best_match_err = best_match_err;
      loopCSecond = loopCBest;
      best_match_err = ee;
      loopCBest = loopCounter;
      best_match_errPre = eeLast;
      best_match_DiffErrPre = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B;
      best_match_errPost = -1;
      best_match_DiffErrPost = -1;
      best_match_x = cpx;
      best_match_y = cpy;
      bestWasLastLoop = true;
    } else {
      if (bestWasLastLoop) {
        best_match_errPost = ee;
        best_match_DiffErrPost = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B;
        bestWasLastLoop = false;
      }
      if (ee < second_best_match_err) {
        second_best_match_err = ee;
        loopCSecond = loopCounter;
      }
    }
    eeLast = ee;
    val_cp_m2 = val_cp_m1; val_cp_m1 = val_cp; val_cp = val_cp_p1; val_cp_p1 = val_cp_p2;
    cpx += incx;
    cpy += incy;
    loopCounter++;
  }

  orc_walk_hist[loopCounter < 63 ? loopCounter : 63]++;   // instrumentation (tools: walk-length distribution)
  if (best_match_err > 4.0f * (float)MAX_ERROR_STEREO) return -3;
  if (abs(loopCBest - loopCSecond) > 1.0f && MIN_DISTANCE_ERROR_STEREO * best_match_err > second_best_match_err) return -2;

  bool didSubpixel = false;
  if (params.useSubpixelStereo) {
    float gradPre_pre = -(best_match_errPre - best_match_DiffErrPre);
    float gradPre_this = +(best_match_err - best_match_DiffErrPre);
    float gradPost_this = -(best_match_err - best_match_DiffErrPost);
    float gradPost_post = +(best_match_errPost - best_match_DiffErrPost);
    bool interpPost = false;
    bool interpPre = false;
    // the first branch (:1782) is guarded by enablePrintDebugInfo == false in Release (NDEBUG) builds
    if ((gradPost_this < 0) ^ (gradPre_this < 0)) {
      // zero-crossing exactly in between: no interpolation
    } else if ((gradPre_pre < 0) ^ (gradPre_this < 0)) {
      if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      } else
        interpPre = true;
    } else if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      interpPost = true;
    }
    if (interpPre) {
      float d = gradPre_this / (gradPre_this - gradPre_pre);
      best_match_x -= d * incx;
      best_match_y -= d * incy;
      best_match_err = best_match_err - 2 * d * gradPre_this - (gradPre_pre - gradPre_this) * d * d;
      didSubpixel = true;
    } else if (interpPost) {
      float d = gradPost_this / (gradPost_this - gradPost_post);
      best_match_x += d * incx;
      best_match_y += d * incy;
      best_match_err = best_match_err + 2 * d * gradPost_this + (gradPost_post - gradPost_this) * d * d;
      didSubpixel = true;
    }
  }

  float sampleDist = GRADIENT_SAMPLE_DIST * rescaleFactor;
  float gradAlongLine = 0;
  float tmp = realVal_p2 - realVal_p1; gradAlongLine += tmp * tmp;
  tmp = realVal_p1 - realVal;          gradAlongLine += tmp * tmp;
  tmp = realVal - realVal_m1;          gradAlongLine += tmp * tmp;
  tmp = realVal_m1 - realVal_m2;       gradAlongLine += tmp * tmp;
  gradAlongLine /= sampleDist * sampleDist;

  if (best_match_err > (float)MAX_ERROR_STEREO + sqrtf(gradAlongLine) * 20) return -3;

  float idnew_best_match;
  float alpha;
  if (incx * incx > incy * incy) {
    float oldX = fxi * best_match_x + cxi;
    float nominator = (oldX * referenceFrame->otherToThis_t[2] - referenceFrame->otherToThis_t[0]);
    float dot0 = dot3(KinvP, referenceFrame->otherToThis_R_row0);
    float dot2 = dot3(KinvP, referenceFrame->otherToThis_R_row2);
    idnew_best_match = (dot0 - oldX * dot2) / nominator;
    alpha = incx * fxi * (dot0 * referenceFrame->otherToThis_t[2] - dot2 * referenceFrame->otherToThis_t[0]) / (nominator * nominator);
  } else {
    float oldY = fyi * best_match_y + cyi;
    float nominator = (oldY * referenceFrame->otherToThis_t[2] - referenceFrame->otherToThis_t[1]);
    float dot1 = dot3(KinvP, referenceFrame->otherToThis_R_row1);
    float dot2 = dot3(KinvP, referenceFrame->otherToThis_R_row2);
    idnew_best_match = (dot1 - oldY * dot2) / nominator;
    alpha = incy * fyi * (dot1 * referenceFrame->otherToThis_t[2] - dot2 * referenceFrame->otherToThis_t[1]) / (nominator * nominator);
  }

  if (idnew_best_match < 0) {
    if (!params.allowNegativeIdepths) return -2;
  }

  float photoDispError = 4.0f * params.cameraPixelNoise2 / (gradAlongLine + DIVISION_EPS);
  float trackingErrorFac = 0.25f * (1.0f + referenceFrame->initialTrackedResidual);
  float gradsInterp[2];
  interp42(const_cast<Frame*>(activeKeyFrame)->gradients(0), u, v, width, gradsInterp);
  float geoDispError = (gradsInterp[0] * epxn + gradsInterp[1] * epyn) + DIVISION_EPS;
  geoDispError = trackingErrorFac * trackingErrorFac * (gradsInterp[0] * gradsInterp[0] + gradsInterp[1] * gradsInterp[1]) /
                 (geoDispError * geoDispError);
  result_var = alpha * alpha * ((didSubpixel ? 0.05f : 0.5f) * sampleDist * sampleDist + geoDispError + photoDispError);
  result_idepth = idnew_best_match;
  result_eplLength = eplLength;
  return best_match_err;
}

}  // namespace orc
