// Exercises include/lsd_slam_hip_io.hpp without a GPU: calibration parsing, keyframeMsg wire format, point-cloud export.
#include <cstdio>
#include <cstring>
#include <iostream>
#include "lsd_slam_hip_io.hpp"
using namespace lsd_slam_hip;
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  Calibration c = parseCalibration(argv[1]);
  printf("K %d %d %.6f %.6f %.6f %.6f\n", c.width, c.height, c.K.fx(), c.K.fy(), c.K.cx(), c.K.cy());
  // a hand-made keyframe: 32 x 16, a slanted plane, every third pixel invalid
  KeyframeMsg m;
  m.id = 7; m.time = 1.25; m.width = 32; m.height = 16;
  m.fx = c.K.fx() * 32.0f / c.width; m.fy = c.K.fy() * 16.0f / c.height; m.cx = 15.5f; m.cy = 7.5f;
  Sim3 T;
  T.q[0] = 0.9238795325112867; T.q[3] = 0.3826834323650898;   // 45 degrees about z
  T.t[0] = 1; T.t[1] = -2; T.t[2] = 0.5; T.s = 2.0;
  sim3ToWire(T, m.camToWorld);
  m.pointcloud.resize(32 * 16);
  for (int y = 0; y < 16; y++)
    for (int x = 0; x < 32; x++) {
      InputPointDense& p = m.pointcloud[x + y * 32];
      p.idepth = ((x + y) % 3 == 0) ? -1.f : 0.5f + 0.01f * x;
      p.idepth_var = 0.001f * (1 + (x % 4));
      p.color[0] = p.color[1] = p.color[2] = p.color[3] = (unsigned char)(8 * x);
    }
  std::vector<unsigned char> wire = serializeKeyframeMsg(m);
  FILE* f = fopen((std::string(argv[2]) + "/kf.msg").c_str(), "wb");
  fwrite(wire.data(), 1, wire.size(), f);
  fclose(f);
  std::vector<float> xyzi;
  int n = flushPointCloud(m, xyzi);
  writePLY(std::string(argv[2]) + "/pc.ply", xyzi);
  printf("points %d\n", n);
  std::vector<unsigned char> img(64 * 48);
  for (size_t i = 0; i < img.size(); i++) img[i] = (unsigned char)(i * 7);
  writePGM(std::string(argv[2]) + "/t.pgm", 64, 48, img.data());
  std::vector<unsigned char> back;
  printf("pgm %d\n", readPGM(std::string(argv[2]) + "/t.pgm", 64, 48, back) && back == img ? 1 : 0);
  return 0;
}
