// lsd_slam_hip_io.hpp — host-side data formats either side of the hot path, header-only C++:
//   * the camera calibration file of lsd_slam_core (calib/*.cfg, parsed by util/Undistorter.cpp:43-140) in its
//     distortion-free form, with the reference's K convention (Undistorter.cpp:340-344);
//   * binary PGM (P5) image files for the image-folder driver (the reference reads through OpenCV, which is not here);
//   * the keyframe message of lsd_slam_viewer (msg/keyframeMsg.msg; InputPointDense payload V/KeyFrameDisplay.h:39-44,
//     filled as C/IOWrapper/ROS/ROSOutput3DWrapper.cpp:70-111) in ROS 1 wire serialisation;
//   * the viewer's point-cloud export (V/KeyFrameDisplay.cpp:269-340 flushPC + V/KeyFrameGraphDisplay.cpp:60-94 PLY header).
// Plain host code: nothing here touches pixels on the hot path.
#ifndef LSD_SLAM_HIP_IO_HPP
#define LSD_SLAM_HIP_IO_HPP

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "lsd_slam_hip.hpp"

namespace lsd_slam_hip {

struct Calibration {
  int width = 0, height = 0;
  Mat3f K;
};

// First line "fx fy cx cy dist", second "in_width in_height", third crop mode, fourth "out_width out_height".
// Values of fx..cy below 1 are relative to the image size.  Only dist == 0 with equal input / output size is
// accepted: undistortion (the rest of Undistorter.cpp) is outside the hot path.
inline Calibration parseCalibration(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw Error(LSDHIP_E_ARG, "cannot open calibration file " + path);
  std::string l1, l2, l3, l4;
  std::getline(f, l1); std::getline(f, l2); std::getline(f, l3); std::getline(f, l4);
  float fx, fy, cx, cy, dist = 0;
  int iw, ih, ow, oh;
  std::istringstream s1(l1), s2(l2), s4(l4);
  if (!(s1 >> fx >> fy >> cx >> cy)) throw Error(LSDHIP_E_ARG, "calibration: cannot read fx fy cx cy");
  s1 >> dist;
  if (!(s2 >> iw >> ih)) throw Error(LSDHIP_E_ARG, "calibration: cannot read the input size");
  if (!(s4 >> ow >> oh)) { ow = iw; oh = ih; }
  if (dist != 0 || ow != iw || oh != ih)
    throw Error(LSDHIP_E_ARG, "calibration needs undistortion / resizing, which is outside the accelerated path: rectify the images first");
  if (iw % 16 || ih % 16) throw Error(LSDHIP_E_ARG, "image size must be a multiple of 16 (C/SlamSystem.cpp:55-59)");
  Calibration c;
  c.width = iw; c.height = ih;
  if (cx < 1 && cy < 1) {  // relative calibration (Undistorter.cpp:340-344)
    fx *= iw; fy *= ih;
    cx = cx * iw - 0.5f; cy = cy * ih - 0.5f;
  }
  c.K = Mat3f::intrinsics(fx, fy, cx, cy);
  return c;
}

inline bool readPGM(const std::string& path, int w, int h, std::vector<unsigned char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  int iw = 0, ih = 0, maxv = 0;
  auto skip = [&]() { int c; while ((c = fgetc(f)) != EOF) { if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} } else if (!isspace(c)) { ungetc(c, f); break; } } };
  bool ok = fscanf(f, "%2s", magic) == 1 && std::string(magic) == "P5";
  if (ok) { skip(); ok = fscanf(f, "%d", &iw) == 1; }
  if (ok) { skip(); ok = fscanf(f, "%d", &ih) == 1; }
  if (ok) { skip(); ok = fscanf(f, "%d", &maxv) == 1 && maxv == 255; }
  if (ok) { fgetc(f); ok = iw == w && ih == h; }
  if (ok) { out.resize((size_t)w * h); ok = fread(out.data(), 1, out.size(), f) == out.size(); }
  fclose(f);
  return ok;
}
inline bool writePGM(const std::string& path, int w, int h, const unsigned char* data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f, "P5\n%d %d\n255\n", w, h);
  bool ok = fwrite(data, 1, (size_t)w * h, f) == (size_t)w * h;
  fclose(f);
  return ok;
}

// V/KeyFrameDisplay.h:39-44
struct InputPointDense {
  float idepth;
  float idepth_var;
  unsigned char color[4];
};
static_assert(sizeof(InputPointDense) == 12, "InputPointDense is 12 bytes on the wire");

// lsd_slam_viewer/msg/keyframeMsg.msg
struct KeyframeMsg {
  int32_t id = 0;
  double time = 0;
  bool isKeyframe = true;
  float camToWorld[7] = {0, 0, 0, 1, 0, 0, 0};  // Sophus Sim3f::data(): quaternion x y z w with norm = scale, then translation
  float fx = 0, fy = 0, cx = 0, cy = 0;
  uint32_t height = 0, width = 0;
  std::vector<InputPointDense> pointcloud;
};

// The vendored Sophus stores a Sim3 as a non-unit quaternion (x, y, z, w) whose norm is the scale (thirdparty/Sophus/
// sophus/rxso3.hpp:311-313), followed by the translation (sim3.hpp:740-754); ROSOutput3DWrapper memcpys those 7 floats.
inline void sim3ToWire(const Sim3& T, float out[7]) {
  const double r = T.s;
  out[0] = (float)(T.q[1] * r); out[1] = (float)(T.q[2] * r); out[2] = (float)(T.q[3] * r); out[3] = (float)(T.q[0] * r);
  out[4] = (float)T.t[0]; out[5] = (float)T.t[1]; out[6] = (float)T.t[2];
}

// ROSOutput3DWrapper::publishKeyframe (ROSOutput3DWrapper.cpp:70-111) at publishLvl 0, from the device planes
inline KeyframeMsg makeKeyframeMsg(const Frame& kf, const Sim3& camToWorld, const Mat3f& K) {
  KeyframeMsg m;
  m.id = kf.id();
  m.time = kf.timestamp();
  sim3ToWire(camToWorld, m.camToWorld);
  m.fx = K.fx(); m.fy = K.fy(); m.cx = K.cx(); m.cy = K.cy();
  m.width = (uint32_t)kf.width(0); m.height = (uint32_t)kf.height(0);
  const std::vector<float> id = kf.idepth(0), var = kf.idepthVar(0), img = kf.image(0);
  m.pointcloud.resize(id.size());
  for (size_t i = 0; i < id.size(); i++) {
    m.pointcloud[i].idepth = id[i];
    m.pointcloud[i].idepth_var = var[i];
    const unsigned char c = (unsigned char)img[i];
    m.pointcloud[i].color[0] = m.pointcloud[i].color[1] = m.pointcloud[i].color[2] = m.pointcloud[i].color[3] = c;
  }
  return m;
}

// ROS 1 serialisation of keyframeMsg (little-endian fields in declaration order, uint8[] with a uint32 length prefix)
inline std::vector<unsigned char> serializeKeyframeMsg(const KeyframeMsg& m) {
  std::vector<unsigned char> b;
  auto put = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; b.insert(b.end(), c, c + n); };
  put(&m.id, 4); put(&m.time, 8);
  const unsigned char kfFlag = m.isKeyframe ? 1 : 0;
  put(&kfFlag, 1);
  put(m.camToWorld, 28);
  put(&m.fx, 4); put(&m.fy, 4); put(&m.cx, 4); put(&m.cy, 4);
  put(&m.height, 4); put(&m.width, 4);
  const uint32_t len = (uint32_t)(m.pointcloud.size() * sizeof(InputPointDense));
  put(&len, 4);
  put(m.pointcloud.data(), len);
  return b;
}

// KeyFrameDisplay::flushPC (V/KeyFrameDisplay.cpp:269-340) with the viewer's default thresholds (V/settings.cpp:36-40:
// scaledDepthVarTH = absDepthVarTH = 1, minNearSupport = 5, sparsifyFactor = 1); appends (x, y, z, intensity) floats.
inline int flushPointCloud(const KeyframeMsg& m, std::vector<float>& xyzi, float scaledTH = 1.f, float absTH = 1.f, int minNearSupport = 5) {
  const int w = (int)m.width, h = (int)m.height;
  const float fxi = 1 / m.fx, fyi = 1 / m.fy, cxi = -m.cx / m.fx, cyi = -m.cy / m.fy;
  // camToWorld: rotation-and-scale quaternion (x y z w) + translation
  const float qx = m.camToWorld[0], qy = m.camToWorld[1], qz = m.camToWorld[2], qw = m.camToWorld[3];
  const float n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  const float scale = n;
  const float ux = qx / n, uy = qy / n, uz = qz / n, uw = qw / n;
  int num = 0;
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      const InputPointDense& p = m.pointcloud[x + y * w];
      if (p.idepth <= 0) continue;
      const float depth = 1 / p.idepth;
      float depth4 = depth * depth;
      depth4 *= depth4;
      if (p.idepth_var * depth4 > scaledTH) continue;
      if (p.idepth_var * depth4 * scale * scale > absTH) continue;
      if (minNearSupport > 1) {
        int nearSupport = 0;
        for (int dx = -1; dx < 2; dx++)
          for (int dy = -1; dy < 2; dy++) {
            const InputPointDense& q = m.pointcloud[x + dx + (y + dy) * w];
            if (q.idepth > 0) {
              const float diff = q.idepth - 1.0f / depth;
              if (diff * diff < 2 * p.idepth_var) nearSupport++;
            }
          }
        if (nearSupport < minNearSupport) continue;
      }
      const float v[3] = {(x * fxi + cxi) * depth * scale, (y * fyi + cyi) * depth * scale, depth * scale};
      // rotate by the unit quaternion, then translate
      const float tx = 2 * (uy * v[2] - uz * v[1]), ty = 2 * (uz * v[0] - ux * v[2]), tz = 2 * (ux * v[1] - uy * v[0]);
      xyzi.push_back(v[0] + uw * tx + (uy * tz - uz * ty) + m.camToWorld[4]);
      xyzi.push_back(v[1] + uw * ty + (uz * tx - ux * tz) + m.camToWorld[5]);
      xyzi.push_back(v[2] + uw * tz + (ux * ty - uy * tx) + m.camToWorld[6]);
      xyzi.push_back(p.color[2] / 255.0f);
      num++;
    }
  return num;
}
// PLY file as KeyFrameGraphDisplay.cpp:74-88 writes it
inline bool writePLY(const std::string& path, const std::vector<float>& xyzi) {
  std::ofstream f(path, std::ios::binary);
  if (!f) return false;
  f << "ply\nformat binary_little_endian 1.0\nelement vertex " << xyzi.size() / 4
    << "\nproperty float x\nproperty float y\nproperty float z\nproperty float intensity\nend_header\n";
  f.write((const char*)xyzi.data(), (std::streamsize)(xyzi.size() * sizeof(float)));
  return (bool)f;
}

}  // namespace lsd_slam_hip
#endif
