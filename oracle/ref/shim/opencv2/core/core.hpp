// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN for the OpenCV vocabulary the reference's hot-path sources touch: the
// debug images of SE3Tracker / DepthMap (C/Tracking/SE3Tracker.h:52-56, C/DepthEstimation/DepthMap.h:76-79), which are
// only written when the plot* settings are on.  cv::Mat here is a plain owned/borrowed byte buffer; drawing / colour
// conversion / file output are no-ops.  OpenCV is an external dependency absent from this machine.
#pragma once
#include <cstdint>
#include <cstring>
#include <deque>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_GRAY2RGB 8
#define CV_FONT_HERSHEY_SIMPLEX 0

namespace cv {

template <typename T, int N> struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; i++) val[i] = T(0); }
  Vec(T a, T b, T c) { static_assert(N == 3, "3-vector"); val[0] = a; val[1] = b; val[2] = c; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<unsigned char, 3> Vec3b;
struct Scalar {
  double val[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};
struct Point2f { float x, y; Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {} };
struct Size { int width, height; };
struct Point { int x, y; Point(int x_ = 0, int y_ = 0) : x(x_), y(y_) {} };

class Mat {
 public:
  int rows = 0, cols = 0, type_ = 0;
  unsigned char* data = nullptr;
  std::shared_ptr<std::vector<unsigned char>> own;
  static int elem(int type) { int depth = type & 7, cn = (type >> 3) + 1; return (depth == CV_32F ? 4 : 1) * cn; }
  Mat() {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
    own = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elem(type));
    data = own->data();
  }
  Mat(int r, int c, int type, void* ext) : rows(r), cols(c), type_(type), data((unsigned char*)ext) {}
  int type() const { return type_; }
  Size size() const { return Size{cols, rows}; }
  void release() { own.reset(); data = nullptr; rows = cols = 0; }
  template <typename T> T& at(int y, int x) { return *(T*)(data + ((size_t)y * cols + x) * elem(type_)); }
  template <typename T> const T& at(int y, int x) const { return *(const T*)(data + ((size_t)y * cols + x) * elem(type_)); }
  void convertTo(Mat& dst, int type) const { if (dst.rows != rows || dst.cols != cols || dst.type_ != type) dst = Mat(rows, cols, type); }
  Mat clone() const { Mat m(rows, cols, type_); if (data) memcpy(m.data, data, (size_t)rows * cols * elem(type_)); return m; }
  void setTo(const Scalar&) {}
};
inline Mat operator*(float, const Mat& m) { return m; }
inline Mat operator*(const Mat& m, float) { return m; }
inline Mat operator+(const Mat& a, const Mat&) { return a; }

inline void line(Mat&, Point2f, Point2f, const Scalar&, int = 1, int = 8, int = 0) {}
inline void cvtColor(const Mat& src, Mat& dst, int) { if (dst.rows != src.rows || dst.cols != src.cols || dst.type() != CV_8UC3) dst = Mat(src.rows, src.cols, CV_8UC3); }
inline bool imwrite(const std::string&, const Mat&) { return true; }

}  // namespace cv
