// dataset_slam — the image-folder driver of the reference (lsd_slam_core/src/main_on_images.cpp:128-279) without ROS:
// calibration file + list of PGM images -> random initialisation on the first image, trackFrame + mapping for the rest
// (doSlam = false, blockUntilMapped = true, keyframes chosen by the reference's distance / usage score), on the GPU
// through include/lsd_slam_hip.hpp.  Outputs: a trajectory text file, one keyframeMsg (ROS 1 wire format) per finished
// keyframe, and the viewer's PLY point cloud.
//
//   dataset_slam <calib.cfg> <image_list.txt> <out_dir> [--kf-every N] [--device D] [--constraints 1]
//
// --constraints 1 additionally aligns every new keyframe with the keyframe it replaces by Sim3Tracker::trackFrameSim3 (the
// tracking-parent edge the reference's constraint search always tests, C/SlamSystem.cpp:1253-1262) and writes constraints.txt.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "../../include/lsd_slam_hip_io.hpp"

using namespace lsd_slam_hip;

// camToWorld of a frame = camToWorld(parent keyframe) * thisToParent (Sim3 composition; scale only changes at keyframes)
static Sim3 sim3_mul(const Sim3& a, const Sim3& b) {
  Sim3 r;
  r.q[0] = a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2] - a.q[3] * b.q[3];
  r.q[1] = a.q[0] * b.q[1] + a.q[1] * b.q[0] + a.q[2] * b.q[3] - a.q[3] * b.q[2];
  r.q[2] = a.q[0] * b.q[2] + a.q[2] * b.q[0] + a.q[3] * b.q[1] - a.q[1] * b.q[3];
  r.q[3] = a.q[0] * b.q[3] + a.q[3] * b.q[0] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
  const double* q = a.q;
  const double v[3] = {b.t[0] * a.s, b.t[1] * a.s, b.t[2] * a.s};
  const double ux = 2 * (q[2] * v[2] - q[3] * v[1]), uy = 2 * (q[3] * v[0] - q[1] * v[2]), uz = 2 * (q[1] * v[1] - q[2] * v[0]);
  r.t[0] = v[0] + q[0] * ux + (q[2] * uz - q[3] * uy) + a.t[0];
  r.t[1] = v[1] + q[0] * uy + (q[3] * ux - q[1] * uz) + a.t[1];
  r.t[2] = v[2] + q[0] * uz + (q[1] * uy - q[2] * ux) + a.t[2];
  r.s = a.s * b.s;
  return r;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::cerr << "usage: dataset_slam <calib.cfg> <image_list.txt> <out_dir> [--kf-every N] [--device D] [--constraints 1]\n";
    return 2;
  }
  int kfEvery = 0, device = 0, constraints = 0;
  for (int i = 4; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--kf-every")) kfEvery = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "--constraints")) constraints = atoi(argv[i + 1]);
  }
  try {
    const Calibration cal = parseCalibration(argv[1]);
    std::vector<std::string> files;
    {
      std::ifstream f(argv[2]);
      std::string line;
      while (std::getline(f, line)) if (!line.empty()) files.push_back(line);
    }
    if (files.empty()) { std::cerr << "no images listed in " << argv[2] << "\n"; return 2; }
    const std::string outDir = argv[3];
    Context::defaultDevice() = device;
    const int w = cal.width, h = cal.height;
    std::vector<unsigned char> img;
    if (!readPGM(files[0], w, h, img)) { std::cerr << "cannot read " << files[0] << " as " << w << "x" << h << " P5\n"; return 2; }
    srand(0);   // initializeRandomly draws from rand() in pixel order (DepthMap.cpp:883-916)
    SlamLoop loop(w, h, cal.K, img.data(), false, nullptr, kfEvery);
    std::ofstream traj(outDir + "/trajectory.txt");
    traj << "# frame keyframe frameToKeyframe(qw qx qy qz tx ty tz) camToWorld(qw qx qy qz tx ty tz s) pointUsage goodRatio trackingWasGood\n";
    Sim3 kfToWorld;   // identity
    std::vector<float> cloud;
    int nKeyframes = 0;
    int keyframeId = 0;
    loop.onKeyframeFinished = [&](Frame& kf, DepthMap&) {
      KeyframeMsg m = makeKeyframeMsg(kf, kfToWorld, cal.K);
      const std::vector<unsigned char> wire = serializeKeyframeMsg(m);
      std::ofstream f(outDir + "/keyframe_" + std::to_string(kf.id()) + ".msg", std::ios::binary);
      f.write((const char*)wire.data(), (std::streamsize)wire.size());
      flushPointCloud(m, cloud);
      nKeyframes++;
    };
    int good = 0, nConstraints = 0;
    std::unique_ptr<Sim3Tracker> sim3;
    std::ofstream cons;
    if (constraints) {
      sim3.reset(new Sim3Tracker(w, h, cal.K));
      cons.open(outDir + "/constraints.txt");
      cons << "# parentKeyframe newKeyframe init(qw qx qy qz tx ty tz s) estimate(qw qx qy qz tx ty tz s) residual depthResidual photoResidual usage diverged\n";
    }
    for (size_t i = 1; i < files.size(); i++) {
      if (!readPGM(files[i], w, h, img)) { std::cerr << "skipping " << files[i] << " (wrong size or unreadable)\n"; continue; }
      std::shared_ptr<Frame> parentKF = loop.keyframe;
      SE3 f2k = loop.step(img.data());
      if (constraints && loop.newKeyframe) {
        // new keyframe -> replaced keyframe, starting from the tracked pose with the depth rescale as scale
        TrackingReference parentRef;
        parentRef.importFrame(parentKF.get());
        const Sim3 init = loop.keyframe->thisToParent_raw();
        const Sim3 est = sim3->trackFrameSim3(&parentRef, loop.keyframe.get(), init, 3, 1);
        cons << parentKF->id() << " " << loop.keyframe->id();
        for (int k = 0; k < 4; k++) cons << " " << init.q[k];
        for (int k = 0; k < 3; k++) cons << " " << init.t[k];
        cons << " " << init.s;
        for (int k = 0; k < 4; k++) cons << " " << est.q[k];
        for (int k = 0; k < 3; k++) cons << " " << est.t[k];
        cons << " " << est.s << " " << sim3->lastResidual << " " << sim3->lastDepthResidual << " " << sim3->lastPhotometricResidual << " "
             << sim3->pointUsage << " " << (sim3->diverged ? 1 : 0) << "\n";
        nConstraints++;
      }
      Sim3 rel;
      for (int k = 0; k < 4; k++) rel.q[k] = f2k.q[k];
      for (int k = 0; k < 3; k++) rel.t[k] = f2k.t[k];
      Sim3 c2w = sim3_mul(kfToWorld, rel);
      if (loop.newKeyframe) {
        // the new keyframe's pose relative to its parent now carries the depth rescale (DepthMap.cpp:1305)
        Sim3 tp = loop.keyframe->thisToParent_raw();
        kfToWorld = sim3_mul(kfToWorld, tp);
        c2w = kfToWorld;
      }
      const float gr = loop.tracker.lastGoodCount / (loop.tracker.lastGoodCount + loop.tracker.lastBadCount);
      traj << i << " " << keyframeId << " " << f2k.q[0] << " " << f2k.q[1] << " " << f2k.q[2] << " " << f2k.q[3] << " " << f2k.t[0] << " "
           << f2k.t[1] << " " << f2k.t[2] << " " << c2w.q[0] << " " << c2w.q[1] << " " << c2w.q[2] << " " << c2w.q[3] << " " << c2w.t[0] << " "
           << c2w.t[1] << " " << c2w.t[2] << " " << c2w.s << " " << loop.tracker.pointUsage << " " << gr << " " << (loop.tracker.trackingWasGood ? 1 : 0) << "\n";
      if (loop.newKeyframe) keyframeId = loop.keyframe->id();
      good += loop.tracker.trackingWasGood ? 1 : 0;
    }
    // the last keyframe: SlamSystem::finalize -> finishCurrentKeyframe
    loop.map.finalizeKeyFrame();
    loop.onKeyframeFinished(*loop.keyframe, loop.map);
    writePLY(outDir + "/pc.ply", cloud);
    std::cout << "frames " << files.size() - 1 << " tracked_good " << good << " keyframes " << nKeyframes << " points " << cloud.size() / 4
              << " constraints " << nConstraints << "\n";
    return 0;
  } catch (const Error& e) {
    std::cerr << "dataset_slam: " << e.what() << "\n";
    return 1;
  }
}
