// cuda_on_cpu.h — the few CUDA / Eigen names the reference's __global__ kernel BODIES use, defined so
// that g++ compiles those bodies unchanged. A launch of n threads is emulated faithfully as n
// sequential calls of the kernel function with blockIdx.x = 0..n-1, blockDim.x = 1, gridDim.x = n
// (PCNN_LAUNCH_1D), so CUDA_1D_KERNEL_LOOP runs exactly one iteration per "thread" — which matters:
// ROIPoolForward advances its `bottom_data` parameter inside the loop
// (roi_pooling_op_gpu.cu.cc:81). Threads run in ascending index order, which is exactly the
// canonical ordering DESIGN.md §4 specifies (atomicAdd hands out slots in ascending index order). TEST INFRASTRUCTURE ONLY; written for this repo, not copied.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct Dim3Shim { int x, y, z; };
static Dim3Shim blockIdx = {0, 0, 0}, gridDim = {1, 1, 1};
static const Dim3Shim threadIdx = {0, 0, 0}, blockDim = {1, 1, 1};

#define PCNN_LAUNCH_1D(n, call)                 \
  do {                                          \
    const int n__ = (n);                        \
    gridDim.x = n__ > 0 ? n__ : 1;              \
    for (int t__ = 0; t__ < n__; t__++) {       \
      blockIdx.x = t__;                         \
      call;                                     \
    }                                           \
    blockIdx.x = 0;                             \
    gridDim.x = 1;                              \
  } while (0)

// the macro the reference defines in every .cu.cc (e.g. hough_voting_gpu_op.cu.cc:16-18); the
// extraction starts below that definition, so it is restated here verbatim in meaning
#define CUDA_1D_KERNEL_LOOP(i, n) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x)

static inline int atomicAdd(int* p, int v) { int old = *p; *p += v; return old; }

// CUDA's overloaded math on float resolves to the float versions; <cmath> does the same for
// std::, bring them into the global namespace the way the CUDA headers do.
using std::ceil; using std::exp; using std::fabs; using std::floor; using std::fmax; using std::fmin;
using std::max; using std::min; using std::round; using std::sqrt;

// expf: CUDA's bits are not reproducible; the canonical expf of DESIGN.md §4 is substituted (the ONLY
// arithmetic substitution in the shim).
extern "C" float oracle_expf(float);
static inline float exp_shim(float x) { return oracle_expf(x); }
#define exp(x) exp_shim(x)

namespace Eigen {
// Minimal fixed-size stand-ins for what compute_box_overlap uses
// (hough_voting_gpu_op.cu.cc:129-142): Matrix<float,R,C,DontAlign>, operator()(i,j), transpose(),
// 3x3 * 3x8 product, Quaternionf(w,x,y,z).toRotationMatrix() with Eigen's published formula
// (Eigen/src/Geometry/Quaternion.h: tx = 2x ... R(0,0) = 1 - (tyy + tzz) ...).
enum { DontAlign = 0 };
template <typename T, int R, int C, int Opt = 0>
struct Matrix {
  T v[R][C];
  T& operator()(int i, int j) { return v[i][j]; }
  const T& operator()(int i, int j) const { return v[i][j]; }
  Matrix<T, C, R, Opt> transpose() const
  {
    Matrix<T, C, R, Opt> t;
    for (int i = 0; i < R; i++)
      for (int j = 0; j < C; j++) t.v[j][i] = v[i][j];
    return t;
  }
};
template <typename T, int R, int K, int C, int O1, int O2>
Matrix<T, R, C, O2> operator*(const Matrix<T, R, K, O1>& a, const Matrix<T, K, C, O2>& b)
{
  Matrix<T, R, C, O2> m;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) {
      T acc = a.v[i][0] * b.v[0][j];
      for (int k = 1; k < K; k++) acc = acc + a.v[i][k] * b.v[k][j];
      m.v[i][j] = acc;
    }
  return m;
}
typedef Matrix<float, 3, 3, 0> Matrix3f;
struct Quaternionf {
  float w_, x_, y_, z_;
  Quaternionf(float w, float x, float y, float z) : w_(w), x_(x), y_(y), z_(z) {}
  Matrix3f toRotationMatrix() const
  {
    Matrix3f r;
    const float tx = 2 * x_, ty = 2 * y_, tz = 2 * z_;
    const float twx = tx * w_, twy = ty * w_, twz = tz * w_;
    const float txx = tx * x_, txy = ty * x_, txz = tz * x_;
    const float tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
    r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
    return r;
  }
};
}  // namespace Eigen
