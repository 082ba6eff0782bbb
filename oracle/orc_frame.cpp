// ORACLE — TEST INFRASTRUCTURE ONLY (see lsd_oracle.hpp).  PARITY UNPINNED.
// Restates C/DataStructures/Frame.cpp (pyramid builders, setDepth, prepareForStereoWith, re-activation
// data) and C/Tracking/TrackingReference.cpp:96-147.
#include <cmath>
#include <cstring>
#include "lsd_oracle.hpp"

namespace orc {

DenseDepthTrackerSettings::DenseDepthTrackerSettings() {  // C/util/settings.h:358-386
  lambdaSuccessFac = 0.5f;
  lambdaFailFac = 2.0f;
  const float stepSizeMinc[6] = {1e-8, 1e-8, 1e-8, 1e-8, 1e-8, 1e-8};
  const int maxIterations[6] = {5, 20, 50, 100, 100, 100};
  for (int level = 0; level < ORC_PYRAMID_LEVELS; ++level) {
    lambdaInitial[level] = 0;
    stepSizeMin[level] = stepSizeMinc[level];
    convergenceEps[level] = 0.999f;
    maxItsPerLvl[level] = maxIterations[level];
  }
  lambdaInitialTestTrack = 0;
  stepSizeMinTestTrack = 1e-3;
  convergenceEpsTestTrack = 0.98;
  maxItsTestTrack = 5;
  var_weight = 1.0;
  huber_d = 3;
}

// C/DataStructures/Frame.cpp:35-48 (uint8 -> float copy) + :397-459 (per-level intrinsics)
Frame::Frame(int id, int width, int height, const float Kin[4], const unsigned char* image) {
  id_ = id;
  M3f K0;
  memset(&K0, 0, sizeof(K0));
  K0(0, 0) = Kin[0]; K0(1, 1) = Kin[1]; K0(0, 2) = Kin[2]; K0(1, 2) = Kin[3]; K0(2, 2) = 1;
  K[0] = K0;
  fx[0] = K0(0, 0); fy[0] = K0(1, 1); cx[0] = K0(0, 2); cy[0] = K0(1, 2);
  KInv[0] = inverse3(K0);
  fxInv[0] = KInv[0](0, 0); fyInv[0] = KInv[0](1, 1); cxInv[0] = KInv[0](0, 2); cyInv[0] = KInv[0](1, 2);
  for (int level = 0; level < ORC_PYRAMID_LEVELS; ++level) {
    w_[level] = width >> level;
    h_[level] = height >> level;
    imageValid_[level] = gradValid_[level] = maxGradValid_[level] = idepthValid_[level] = false;
    if (level > 0) {
      fx[level] = fx[level - 1] * 0.5;  // double product, exact
      fy[level] = fy[level - 1] * 0.5;
      cx[level] = (cx[0] + 0.5) / ((int)1 << level) - 0.5;  // double arithmetic, then float (Frame.cpp:449)
      cy[level] = (cy[0] + 0.5) / ((int)1 << level) - 0.5;
      M3f Kl;
      memset(&Kl, 0, sizeof(Kl));
      Kl(0, 0) = fx[level]; Kl(0, 2) = cx[level]; Kl(1, 1) = fy[level]; Kl(1, 2) = cy[level]; Kl(2, 2) = 1;
      K[level] = Kl;
      KInv[level] = inverse3(Kl);
      fxInv[level] = KInv[level](0, 0); fyInv[level] = KInv[level](1, 1);
      cxInv[level] = KInv[level](0, 2); cyInv[level] = KInv[level](1, 2);
    }
  }
  image_[0].resize((size_t)w_[0] * h_[0]);
  for (size_t i = 0; i < image_[0].size(); i++) image_[0][i] = image[i];
  imageValid_[0] = true;
}

const float* Frame::image(int l) { if (!imageValid_[l]) buildImage(l); return image_[l].data(); }
const float* Frame::gradients(int l) { if (!gradValid_[l]) buildGradients(l); return grad_[l].data(); }
void Frame::overrideMaxGradients(const float* plane) {
  maxGradients(0);   // allocate / build, then replace
  std::copy(plane, plane + (size_t)width(0) * height(0), maxGrad_[0].begin());
}
const float* Frame::maxGradients(int l) { if (!maxGradValid_[l]) buildMaxGradients(l); return maxGrad_[l].data(); }
const float* Frame::idepth(int l) { if (!idepthValid_[l]) buildIDepthAndIDepthVar(l); return idepth_[l].data(); }
const float* Frame::idepthVar(int l) { if (!idepthValid_[l]) buildIDepthAndIDepthVar(l); return idepthVar_[l].data(); }

// Frame.cpp:491-630.  SSE branch (:516-553) sums columns first: ((s[0]+s[w]) + (s[1]+s[1+w]))*0.25;
// scalar fallback (:614-627): (((s[0]+s[1])+s[w])+s[1+w])*0.25.
void Frame::buildImage(int level) {
  if (level == 0) return;
  image(level - 1);
  int width = w_[level - 1], height = h_[level - 1];
  const float* source = image_[level - 1].data();
  image_[level].assign((size_t)w_[level] * h_[level], 0.f);
  float* dest = image_[level].data();
  if (sseImagePyramid && width % 8 == 0) {
    for (int y = 0; y < height; y += 2)
      for (int x = 0; x < width; x += 2) {
        const float* s = source + x + y * width;
        float c0 = s[0] + s[width];      // _mm_add_ps(top, bot), then hadd-style pair sum
        float c1 = s[1] + s[1 + width];
        *dest++ = (c0 + c1) * 0.25f;
      }
  } else {
    int wh = width * height;
    for (int y = 0; y < wh; y += width * 2)
      for (int x = 0; x < width; x += 2) {
        const float* s = source + x + y;
        *dest++ = (s[0] + s[1] + s[width] + s[1 + width]) * 0.25f;
      }
  }
  imageValid_[level] = true;
}

// Frame.cpp:643-680: linear walk from index w to w*(h-1); columns 0 / w-1 wrap across rows.
void Frame::buildGradients(int level) {
  image(level);
  int width = w_[level], height = h_[level];
  grad_[level].assign((size_t)width * height * 4, 0.f);
  const float* img = image_[level].data();
  const float* img_pt = img + width;
  const float* img_pt_max = img + width * (height - 1);
  float* g = grad_[level].data() + 4 * width;
  float val_m1 = *(img_pt - 1);
  float val_00 = *img_pt;
  float val_p1;
  for (; img_pt < img_pt_max; img_pt++, g += 4) {
    val_p1 = *(img_pt + 1);
    g[0] = 0.5f * (val_p1 - val_m1);
    g[1] = 0.5f * (*(img_pt + width) - *(img_pt - width));
    g[2] = val_00;
    val_m1 = val_00;
    val_00 = val_p1;
  }
  gradValid_[level] = true;
}

// Frame.cpp:690-767
void Frame::buildMaxGradients(int level) {
  gradients(level);
  int width = w_[level], height = h_[level];
  maxGrad_[level].assign((size_t)width * height, 0.f);
  std::vector<float> tmp((size_t)width * height, 0.f);
  float* mg = maxGrad_[level].data();
  const float* g = grad_[level].data() + 4 * width;
  float* maxgrad_pt = mg + width;
  float* maxgrad_pt_max = mg + width * (height - 1);
  for (; maxgrad_pt < maxgrad_pt_max; maxgrad_pt++, g += 4) {
    float dx = g[0], dy = g[1];
    *maxgrad_pt = sqrtf(dx * dx + dy * dy);
  }
  maxgrad_pt = mg + width + 1;
  maxgrad_pt_max = mg + width * (height - 1) - 1;
  float* t = tmp.data() + width + 1;
  for (; maxgrad_pt < maxgrad_pt_max; maxgrad_pt++, t++) {
    float g1 = maxgrad_pt[-width];
    float g2 = maxgrad_pt[0];
    if (g1 < g2) g1 = g2;
    float g3 = maxgrad_pt[width];
    if (g1 < g3) *t = g3; else *t = g1;
  }
  float numMappable = 0;
  maxgrad_pt = mg + width + 1;
  maxgrad_pt_max = mg + width * (height - 1) - 1;
  t = tmp.data() + width + 1;
  for (; maxgrad_pt < maxgrad_pt_max; maxgrad_pt++, t++) {
    float g1 = t[-1];
    float g2 = t[0];
    if (g1 < g2) g1 = g2;
    float g3 = t[1];
    if (g1 < g3) { *maxgrad_pt = g3; if (g3 >= minUseGradForMappable_) numMappable++; }
    else { *maxgrad_pt = g1; if (g1 >= minUseGradForMappable_) numMappable++; }
  }
  if (level == 0) numMappablePixels = numMappable;
  maxGradValid_[level] = true;
}

// Frame.cpp:775-877
void Frame::buildIDepthAndIDepthVar(int level) {
  if (!hasIDepth_ || level == 0) return;
  idepth(level - 1);
  int width = w_[level], height = h_[level];
  idepth_[level].assign((size_t)width * height, 0.f);
  idepthVar_[level].assign((size_t)width * height, 0.f);
  int sw = w_[level - 1];
  const float* idepthSource = idepth_[level - 1].data();
  const float* idepthVarSource = idepthVar_[level - 1].data();
  float* idepthDest = idepth_[level].data();
  float* idepthVarDest = idepthVar_[level].data();
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      int idx = 2 * (x + y * sw);
      int idxDest = x + y * width;
      float idepthSumsSum = 0, ivarSumsSum = 0;
      int num = 0;
      const int offs[4] = {0, 1, sw, sw + 1};
      for (int k = 0; k < 4; k++) {
        float var = idepthVarSource[idx + offs[k]];
        if (var > 0) {
          float ivar = 1.0f / var;
          ivarSumsSum += ivar;
          idepthSumsSum += ivar * idepthSource[idx + offs[k]];
          num++;
        }
      }
      if (num > 0) {
        float depth = ivarSumsSum / idepthSumsSum;
        idepthDest[idxDest] = 1.0f / depth;
        idepthVarDest[idxDest] = num / ivarSumsSum;
      } else {
        idepthDest[idxDest] = -1;
        idepthVarDest[idxDest] = -1;
      }
    }
  idepthValid_[level] = true;
}

void Frame::releaseIDepthPyramid() {
  for (int l = 1; l < ORC_PYRAMID_LEVELS; l++) idepthValid_[l] = false;
}

// Frame.cpp:199-243
void Frame::setDepth(const DepthMapPixelHypothesis* newDepth) {
  size_t n = (size_t)w_[0] * h_[0];
  idepth_[0].resize(n);
  idepthVar_[0].resize(n);
  float sumIdepth = 0;
  int numIdepth = 0;
  for (size_t i = 0; i < n; i++, newDepth++) {
    if (newDepth->isValid && newDepth->idepth_smoothed >= -0.05) {
      idepth_[0][i] = newDepth->idepth_smoothed;
      idepthVar_[0][i] = newDepth->idepth_var_smoothed;
      numIdepth++;
      sumIdepth += newDepth->idepth_smoothed;
    } else {
      idepth_[0][i] = -1;
      idepthVar_[0][i] = -1;
    }
  }
  meanIdepth = sumIdepth / numIdepth;
  numPoints = numIdepth;
  idepthValid_[0] = true;
  releaseIDepthPyramid();
  hasIDepth_ = true;
  depthHasBeenUpdatedFlag = true;
}

// Frame.cpp:245-293
void Frame::setDepthFromGroundTruth(const float* depth, float cov_scale, float minUseGrad) {
  minUseGradForMappable_ = minUseGrad;
  const float* pyrMaxGradient = maxGradients(0);
  int width0 = w_[0], height0 = h_[0];
  idepth_[0].resize((size_t)width0 * height0);
  idepthVar_[0].resize((size_t)width0 * height0);
  float* pyrIDepth = idepth_[0].data();
  float* pyrIDepthVar = idepthVar_[0].data();
  for (int y = 0; y < height0; y++)
    for (int x = 0; x < width0; x++) {
      if (x > 0 && x < width0 - 1 && y > 0 && y < height0 - 1 && pyrMaxGradient[x + y * width0] >= minUseGrad &&
          !std::isnan(*depth) && *depth > 0) {
        *pyrIDepth = 1.0f / *depth;
        *pyrIDepthVar = 0.01f * 0.01f * cov_scale;  // VAR_GT_INIT_INITIAL * cov_scale
      } else {
        *pyrIDepth = -1;
        *pyrIDepthVar = -1;
      }
      ++depth; ++pyrIDepth; ++pyrIDepthVar;
    }
  idepthValid_[0] = true;
  releaseIDepthPyramid();
  hasIDepth_ = true;
}

void Frame::setDepthPlanes(const float* id, const float* var) {
  size_t n = (size_t)w_[0] * h_[0];
  idepth_[0].assign(id, id + n);
  idepthVar_[0].assign(var, var + n);
  idepthValid_[0] = true;
  releaseIDepthPyramid();
  hasIDepth_ = true;
}

// Frame.cpp:295-317.  The Sim3 algebra runs in double (Sophus Sim3d), results are cast to float.
void Frame::prepareForStereoWith(Frame* other, const Sim3d& thisToOther, const M3f& Kf, int) {
  Sim3d otherToThis = thisToOther.inverse();
  M3d R_o2t = otherToThis.rotationMatrix();
  M3f R_o2t_f;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_o2t_f(i, j) = (float)R_o2t(i, j);
  K_otherToThis_R = scale(matmat(Kf, R_o2t_f), (float)otherToThis.s);
  otherToThis_t = mk3<float>((float)otherToThis.t[0], (float)otherToThis.t[1], (float)otherToThis.t[2]);
  K_otherToThis_t = matvec(Kf, otherToThis_t);

  thisToOther_t = mk3<float>((float)thisToOther.t[0], (float)thisToOther.t[1], (float)thisToOther.t[2]);
  K_thisToOther_t = matvec(Kf, thisToOther_t);
  M3d R_t2o = thisToOther.rotationMatrix();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) thisToOther_R(i, j) = (float)R_t2o(i, j) * (float)thisToOther.s;
  otherToThis_R_row0 = mk3<float>(thisToOther_R(0, 0), thisToOther_R(1, 0), thisToOther_R(2, 0));
  otherToThis_R_row1 = mk3<float>(thisToOther_R(0, 1), thisToOther_R(1, 1), thisToOther_R(2, 1));
  otherToThis_R_row2 = mk3<float>(thisToOther_R(0, 2), thisToOther_R(1, 2), thisToOther_R(2, 2));
  distSquared = (float)dot3(otherToThis.t, otherToThis.t);
  referenceID = other->id();
}

// Frame.cpp:107-145
void Frame::takeReActivationData(const DepthMapPixelHypothesis* depthMap) {
  size_t n = (size_t)w_[0] * h_[0];
  if (idepth_reAct.size() != n) { idepth_reAct.assign(n, 0.f); idepthVar_reAct.assign(n, 0.f); validity_reAct.assign(n, 0); }
  for (size_t i = 0; i < n; i++, depthMap++) {
    if (depthMap->isValid) {
      idepth_reAct[i] = depthMap->idepth;
      idepthVar_reAct[i] = depthMap->idepth_var;
      validity_reAct[i] = depthMap->validity_counter;
    } else if (depthMap->blacklisted < -1 /*MIN_BLACKLIST*/) {
      idepthVar_reAct[i] = -2;
    } else {
      idepthVar_reAct[i] = -1;
    }
  }
}

// Frame.h:421-437
uint8_t* Frame::refPixelWasGood() {
  if (wasGood_.empty()) wasGood_.assign((size_t)w_[ORC_SE3TRACKING_MIN_LEVEL] * h_[ORC_SE3TRACKING_MIN_LEVEL], 0xFF);
  return wasGood_.data();
}

// ---------------------------------------------------------------------------------------------
// C/Tracking/TrackingReference.cpp:71-87
void TrackingReference::importFrame(Frame* kf) {
  keyframe = kf;
  for (int l = 0; l < ORC_PYRAMID_LEVELS; l++) numData[l] = 0;
}

// C/Tracking/TrackingReference.cpp:96-147: x outer, y inner.
void TrackingReference::makePointCloud(int level) {
  if (numData[level] > 0) return;
  int w = keyframe->width(level), h = keyframe->height(level);
  float fxInvLevel = keyframe->fxInv[level], fyInvLevel = keyframe->fyInv[level];
  float cxInvLevel = keyframe->cxInv[level], cyInvLevel = keyframe->cyInv[level];
  const float* pyrIdepthSource = keyframe->idepth(level);
  const float* pyrIdepthVarSource = keyframe->idepthVar(level);
  const float* pyrColorSource = keyframe->image(level);
  const float* pyrGradSource = keyframe->gradients(level);
  posData[level].resize((size_t)w * h * 3);
  gradData[level].resize((size_t)w * h * 2);
  colorAndVarData[level].resize((size_t)w * h * 2);
  pointPosInXYGrid[level].resize((size_t)w * h);
  int n = 0;
  for (int x = 1; x < w - 1; x++)
    for (int y = 1; y < h - 1; y++) {
      int idx = x + y * w;
      if (pyrIdepthVarSource[idx] <= 0 || pyrIdepthSource[idx] == 0) continue;
      float inv = 1.0f / pyrIdepthSource[idx];
      posData[level][3 * n + 0] = inv * (fxInvLevel * x + cxInvLevel);
      posData[level][3 * n + 1] = inv * (fyInvLevel * y + cyInvLevel);
      posData[level][3 * n + 2] = inv * 1.0f;
      gradData[level][2 * n + 0] = pyrGradSource[4 * idx + 0];
      gradData[level][2 * n + 1] = pyrGradSource[4 * idx + 1];
      colorAndVarData[level][2 * n + 0] = pyrColorSource[idx];
      colorAndVarData[level][2 * n + 1] = pyrIdepthVarSource[idx];
      pointPosInXYGrid[level][n] = idx;
      n++;
    }
  numData[level] = n;
}

// C/util/globalFuncs.h:43-61
float getInterpolatedElement(const float* mat, float x, float y, int width) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat + ix + iy * width;
  float res = dxdy * bp[1 + width] + (dy - dxdy) * bp[width] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
  return res;
}

}  // namespace orc
