// eigen_sophus_on_cpu.h — the Eigen / Sophus / df names that the reference's pose-refinement bodies use
// (icpKernel, lib/kinect_fusion/src/optimization/icp.cu:20-137; Poly3CameraModel::project + the CameraModel helpers,
// include/df/camera/poly3.h:36-88, cameraModel.h:72-82; optEnergy, lib/synthesize/synthesize.cpp:2476-2526), defined so
// that g++ compiles those bodies UNCHANGED. Eigen, Sophus, thrust and CUDA are absent from this image; nothing here is
// copied from them. Where the result of a float expression depends on the library's evaluation order, the stand-in
// follows the order the libraries PUBLISH for these fixed-size, non-vectorisable types (Eigen 3.3-era, the
// reference's time; `DontAlign` 3-vectors are never vectorised):
//
//   * reductions (dot, squaredNorm, the rows of a small product): Eigen's `redux_novec_unroller` splits a range of
//     length L into halves of length L/2 and L - L/2 and combines them with one operation, recursively — for three
//     terms  t0 + (t1 + t2),  for four  (t0 + t1) + (t2 + t3);
//   * small fixed-size products are coefficient-based ("lazy"): C(i,j) = redux over k of A(i,k) * B(k,j); a scalar
//     factor on the left operand is applied to that operand first ((s A) B), as written in the reference's source;
//   * normalized(): v / sqrt(squaredNorm) per component (a true division, Eigen 3.3's div_assign_op);
//   * cross(): (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0);
//   * Quaternion::_transformVector (what Sophus::SO3::operator*(point) calls): uv = 2 (q.vec x v);
//     result = (v + w uv) + q.vec x uv;  Sophus::SE3::operator*(point) = so3 * p + translation;
//   * Sophus::SE3(Quaternion, Point) normalises the quaternion: coeffs / sqrt(squaredNorm), 4-term reduction above
//     over the coefficient order (x, y, z, w).
// What CUDA would add on the reference's own hardware — FMA contraction of a * b + c — is NOT reproduced: the canonical
// arithmetic of this repo is one IEEE rounding per operation, the same stance as for the five ops.
// float -> int conversions of NaN / out-of-range values (`const int u = projected + 0.5`; undefined in C++, NaN -> 0 and
// saturating in PTX): x86's cvttss2si yields INT_MIN for all of them. Either way `(u <= border) || (u >= width-1-border)`
// is true and the pixel leaves through the SAME exit (the border test), so the bodies run as they are.
// TEST INFRASTRUCTURE ONLY (compiled into oracle/_ref/libposecnn_ref.so by oracle/ref_shim/Makefile).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define __global__
#define __device__
#define __host__

typedef unsigned int uint;
typedef unsigned char uchar;

struct Uint3Shim { uint x, y, z; };
static Uint3Shim threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
static const Uint3Shim blockDim = {1, 1, 1};

using std::fabs;
using std::min;

namespace Eigen {

enum { DontAlign = 2, RowMajor = 1 };

// Eigen's complete-unrolling reduction order (Core/Redux.h, redux_novec_unroller)
template <typename T>
static inline T redux_sum(const T* t, int start, int len)
{
  if (len == 1) return t[start];
  const int half = len / 2;
  return redux_sum(t, start, half) + redux_sum(t, start + half, len - half);
}

template <typename T, int R, int C, int Opt = 0>
struct Matrix;

template <typename T, int R, int C, int Opt>
struct CommaInit {
  Matrix<T, R, C, Opt>* m;
  int n;
  CommaInit& operator,(const T& s) { m->v[n++] = s; return *this; }
};

template <typename T, int R, int C, int Opt>
struct Matrix {
  T v[R * C];   // row-major (only observable through the comma initialiser, which fills row by row like Eigen's)
  Matrix() {}
  Matrix(const T& a, const T& b) { static_assert(R * C == 2, "size"); v[0] = a; v[1] = b; }
  Matrix(const T& a, const T& b, const T& c) { static_assert(R * C == 3, "size"); v[0] = a; v[1] = b; v[2] = c; }
  Matrix(const T& a, const T& b, const T& c, const T& d) { static_assert(R * C == 4, "size"); v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
  template <int O2>
  Matrix(const Matrix<T, R, C, O2>& o) { for (int i = 0; i < R * C; i++) v[i] = o.v[i]; }
  static Matrix Zero() { Matrix m; for (int i = 0; i < R * C; i++) m.v[i] = T(0); return m; }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
  T& operator()(int i, int j) { return v[i * C + j]; }
  const T& operator()(int i, int j) const { return v[i * C + j]; }
  const T* data() const { return v; }
  template <int N>
  Matrix<T, N, 1, Opt> head() const { Matrix<T, N, 1, Opt> h; for (int i = 0; i < N; i++) h.v[i] = v[i]; return h; }
  Matrix<T, C, R, Opt> transpose() const
  {
    Matrix<T, C, R, Opt> t;
    for (int i = 0; i < R; i++)
      for (int j = 0; j < C; j++) t.v[j * R + i] = v[i * C + j];
    return t;
  }
  template <int O2>
  T dot(const Matrix<T, R, C, O2>& o) const
  {
    T t[R * C];
    for (int i = 0; i < R * C; i++) t[i] = v[i] * o.v[i];
    return redux_sum(t, 0, R * C);
  }
  T squaredNorm() const
  {
    T t[R * C];
    for (int i = 0; i < R * C; i++) t[i] = v[i] * v[i];
    return redux_sum(t, 0, R * C);
  }
  T norm() const { return std::sqrt(squaredNorm()); }
  Matrix normalized() const
  {
    const T z = squaredNorm();
    Matrix n = *this;
    if (z > T(0)) { const T s = std::sqrt(z); for (int i = 0; i < R * C; i++) n.v[i] = v[i] / s; }
    return n;
  }
  template <int O2>
  Matrix<T, 3, 1, Opt> cross(const Matrix<T, R, C, O2>& b) const
  {
    static_assert(R * C == 3, "cross");
    return Matrix<T, 3, 1, Opt>(v[1] * b.v[2] - v[2] * b.v[1], v[2] * b.v[0] - v[0] * b.v[2], v[0] * b.v[1] - v[1] * b.v[0]);
  }
  CommaInit<T, R, C, Opt> operator<<(const T& s) { v[0] = s; return CommaInit<T, R, C, Opt>{this, 1}; }
};

template <typename T, int R, int C, int O1, int O2>
Matrix<T, R, C, O1> operator-(const Matrix<T, R, C, O1>& a, const Matrix<T, R, C, O2>& b)
{
  Matrix<T, R, C, O1> m;
  for (int i = 0; i < R * C; i++) m.v[i] = a.v[i] - b.v[i];
  return m;
}
template <typename T, int R, int C, int O1, int O2>
Matrix<T, R, C, O1> operator+(const Matrix<T, R, C, O1>& a, const Matrix<T, R, C, O2>& b)
{
  Matrix<T, R, C, O1> m;
  for (int i = 0; i < R * C; i++) m.v[i] = a.v[i] + b.v[i];
  return m;
}
template <typename T, int R, int C, int O>
Matrix<T, R, C, O> operator*(const T& s, const Matrix<T, R, C, O>& a)
{
  Matrix<T, R, C, O> m;
  for (int i = 0; i < R * C; i++) m.v[i] = s * a.v[i];
  return m;
}
template <typename T, int R, int C, int O>
Matrix<T, R, C, O> operator*(const Matrix<T, R, C, O>& a, const T& s)
{
  Matrix<T, R, C, O> m;
  for (int i = 0; i < R * C; i++) m.v[i] = a.v[i] * s;
  return m;
}
// coefficient-based product of small fixed-size matrices
template <typename T, int R, int K, int C, int O1, int O2>
Matrix<T, R, C, 0> operator*(const Matrix<T, R, K, O1>& a, const Matrix<T, K, C, O2>& b)
{
  Matrix<T, R, C, 0> m;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) {
      T t[K];
      for (int k = 0; k < K; k++) t[k] = a(i, k) * b(k, j);
      m(i, j) = redux_sum(t, 0, K);
    }
  return m;
}

typedef Matrix<int, 2, 1, 0> Vector2i;
typedef Matrix<float, 2, 1, 0> Vector2f;
typedef Matrix<float, 3, 1, 0> Vector3f;

template <typename Scalar, int D>
using UnalignedVec = Matrix<Scalar, D, 1, DontAlign>;
template <typename Scalar>
using UnalignedVec2 = Matrix<Scalar, 2, 1, DontAlign>;
template <typename Scalar>
using UnalignedVec3 = Matrix<Scalar, 3, 1, DontAlign>;
template <typename Scalar>
using UnalignedVec4 = Matrix<Scalar, 4, 1, DontAlign>;

// Quaternion with Eigen's constructor order (w, x, y, z) and coefficient order (x, y, z, w)
template <typename T>
struct Quaternion {
  T x_, y_, z_, w_;
  Quaternion() : x_(0), y_(0), z_(0), w_(1) {}
  Quaternion(const T& w, const T& x, const T& y, const T& z) : x_(x), y_(y), z_(z), w_(w) {}
  T w() const { return w_; }
  Matrix<T, 3, 1, 0> vec() const { return Matrix<T, 3, 1, 0>(x_, y_, z_); }
  T squaredNorm() const
  {
    const T t[4] = {x_ * x_, y_ * y_, z_ * z_, w_ * w_};
    return redux_sum(t, 0, 4);
  }
  void normalize()
  {
    const T n = std::sqrt(squaredNorm());
    x_ = x_ / n; y_ = y_ / n; z_ = z_ / n; w_ = w_ / n;
  }
  template <int O>
  Matrix<T, 3, 1, 0> _transformVector(const Matrix<T, 3, 1, O>& v) const
  {
    Matrix<T, 3, 1, 0> uv = vec().cross(v);
    uv = uv + uv;
    return (Matrix<T, 3, 1, 0>(v) + w() * uv) + vec().cross(uv);
  }
};
typedef Quaternion<float> Quaternionf;

}  // namespace Eigen

namespace Sophus {
template <typename Scalar>
struct SE3 {
  typedef Eigen::Matrix<Scalar, 3, 1, 0> Point;
  Eigen::Quaternion<Scalar> q;
  Point t;
  SE3() : t(Scalar(0), Scalar(0), Scalar(0)) {}
  // SE3(Quaternion, Point): SO3's constructor normalises the quaternion
  SE3(const Eigen::Quaternion<Scalar>& quat, const Point& trans) : q(quat), t(trans) { q.normalize(); }
  // (shim only) the members exactly as given: the pin tests hand over a pose that is already an SE3f's content
  static SE3 raw(const Eigen::Quaternion<Scalar>& quat, const Point& trans) { SE3 s; s.q = quat; s.t = trans; return s; }
  template <int O>
  Point operator*(const Eigen::Matrix<Scalar, 3, 1, O>& p) const { return q._transformVector(p) + t; }
};
typedef SE3<float> SE3f;
}  // namespace Sophus

namespace df {

// Tensor<2, T> on plain memory: (d0, d1) -> data[d0 + dim0 * d1] (df/util/tensor.h:841-851, offsetXD)
template <typename T>
struct DeviceTensor2 {
  uint dims[2];
  T* data_;
  DeviceTensor2(uint w, uint h, T* d) : data_(d) { dims[0] = w; dims[1] = h; }
  uint dimensionSize(uint d) const { return dims[d]; }
  T& operator()(uint d0, uint d1) const { return data_[d0 + (size_t)dims[0] * d1]; }
  T& operator()(const Eigen::Vector2i& p) const { return (*this)(p(0), p(1)); }
};
template <typename T>
using ManagedHostTensor2 = DeviceTensor2<T>;

namespace internal {
// df/optimization/linearSystems.h:25-29 (the 1 x ModelDim specialisation)
template <typename Scalar, uint ResidualDim, uint ModelDim>
struct JacobianAndResidual {
  Eigen::Matrix<Scalar, 1, ModelDim, Eigen::DontAlign | Eigen::RowMajor> J;
  Scalar r;
};
}  // namespace internal

// df/util/debugHelpers.h:11-34: the per-pixel exit reason of icpKernel, as a colour
template <typename... DebugArgTs>
struct PixelDebugger {
  static void debugPixel(const Eigen::Vector2i&, const Eigen::UnalignedVec4<uchar>&, DebugArgTs...) {}
};
template <>
struct PixelDebugger<DeviceTensor2<Eigen::UnalignedVec4<unsigned char> > > {
  static void debugPixel(const Eigen::Vector2i& pixel, const Eigen::UnalignedVec4<uchar>& color,
                         DeviceTensor2<Eigen::UnalignedVec4<unsigned char> > debugArg)
  {
    debugArg(pixel) = color;
  }
};

}  // namespace df
