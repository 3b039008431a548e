// bilinear.h — tap tables of PoseCNN's fixed bilinear "deconv" layers (lib/networks/network.py:141-157
// make_deconv_filter), shared by the deconv / label-head kernels (upscore.hip) and by the Hough
// voting kernel that interpolates the 1/8-resolution vertex field on the fly (hough_voting.hip).
// Every user accumulates in the same canonical order — input rows ascending, input columns
// ascending, acc = acc + (wy*wx)*in, then + bias — so all of them produce identical bits.
#pragma once
#include "pcnn_device.h"

namespace pcnn {

// network.py:144-150: f = ceil(k/2), c = (2f - 1 - f%2) / (2f), w[t] = 1 - |t/f - c|
// (evaluated in double like the numpy code, then stored as f32 like tf.constant_initializer)
__host__ __device__ inline float bilinear_tap(int t, int k)
{
  const int f = (k + 1) / 2;
  const double c = (double)(2 * f - 1 - f % 2) / (2.0 * (double)f);
  return (float)(1.0 - fabs((double)t / (double)f - c));
}

struct Taps {
  int i0, n;       // first contributing input index and count (<= 2 when k == 2s); may be clipped
  float w[4];      // tap weights for i0, i0+1, ...
};

// output index o of a SAME conv2d_transpose (pad = (k - s) / 2): o = s*i + t - pad, 0 <= t < k
__device__ __forceinline__ Taps make_taps(int o, int k, int s, int pad, int n_in)
{
  Taps T;
  const int a = o + pad;
  int lo = (a - k + 1 + s - 1) / s;  // ceil((a - k + 1) / s) for a - k + 1 possibly negative
  if (a - k + 1 < 0) lo = -((k - 1 - a) / s);
  int hi = a / s;
  if (lo < 0) lo = 0;
  if (hi > n_in - 1) hi = n_in - 1;
  T.i0 = lo;
  T.n = hi - lo + 1;
  if (T.n < 0) T.n = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) T.w[j] = j < T.n ? bilinear_tap(a - s * (lo + j), k) : 0.f;
  return T;
}

// make_taps with the k tap values bilinear_tap(t, k), t = 0 .. k-1, read from a table (LDS) instead of being evaluated in
// double precision per call: same integers, same f32 weights.
__device__ __forceinline__ Taps make_taps_tab(int o, int k, int s, int pad, int n_in, const float* tab)
{
  Taps T;
  const int a = o + pad;
  int lo = (a - k + 1 + s - 1) / s;
  if (a - k + 1 < 0) lo = -((k - 1 - a) / s);
  int hi = a / s;
  if (lo < 0) lo = 0;
  if (hi > n_in - 1) hi = n_in - 1;
  T.i0 = lo;
  T.n = hi - lo + 1;
  if (T.n < 0) T.n = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) T.w[j] = j < T.n ? tab[a - s * (lo + j)] : 0.f;
  return T;
}

// One output element of deconv(in)[+ bias] for channel c at output pixel (oy, ox); `inb` points at
// the image's [H, W, C] low-resolution plane. Same arithmetic as deconv_bilinear_kernel.
__device__ __forceinline__ float bilinear_at(const float* __restrict__ inb, const Taps& ty,
                                             const Taps& tx, int W, int C, int c)
{
  float acc = 0.f;
#pragma unroll
  for (int jy = 0; jy < 4; jy++) {
    if (jy < ty.n) {
      const float* row = inb + (size_t)(ty.i0 + jy) * W * C + c;
#pragma unroll
      for (int jx = 0; jx < 4; jx++) {
        if (jx < tx.n) {
          const float w = ty.w[jy] * tx.w[jx];
          acc = acc + w * row[(size_t)(tx.i0 + jx) * C];
        }
      }
    }
  }
  return acc;
}

// bilinear_at for THREE consecutive channels c, c+1, c+2 of one output pixel (the Hough layer's (u, v, log d) triple): the
// twelve loads of the <= 2 x 2 taps are issued together (absent taps read a present one's address and are skipped in the
// sums) instead of one trip to memory per tap and channel behind a guard each. Per channel the same terms in the same
// order as bilinear_at: same bits.
__device__ __forceinline__ void bilinear_at3(const float* __restrict__ inb, const Taps& ty, const Taps& tx, int W, int C,
                                             int c, float& o0, float& o1, float& o2)
{
  if (ty.n >= 1 && ty.n <= 2 && tx.n >= 1 && tx.n <= 2) {
    float v[4][3];
#pragma unroll
    for (int jy = 0; jy < 2; jy++)
#pragma unroll
      for (int jx = 0; jx < 2; jx++) {
        const int yy = ty.i0 + (jy < ty.n ? jy : ty.n - 1), xx = tx.i0 + (jx < tx.n ? jx : tx.n - 1);
        const float* q = inb + ((size_t)yy * W + xx) * C + c;
        v[jy * 2 + jx][0] = q[0]; v[jy * 2 + jx][1] = q[1]; v[jy * 2 + jx][2] = q[2];
      }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int jy = 0; jy < 2; jy++)
#pragma unroll
      for (int jx = 0; jx < 2; jx++)
        if (jy < ty.n && jx < tx.n) {
          const float w = ty.w[jy] * tx.w[jx];
          a0 = a0 + w * v[jy * 2 + jx][0];
          a1 = a1 + w * v[jy * 2 + jx][1];
          a2 = a2 + w * v[jy * 2 + jx][2];
        }
    o0 = a0; o1 = a1; o2 = a2;
  } else {
    o0 = bilinear_at(inb, ty, tx, W, C, c);
    o1 = bilinear_at(inb, ty, tx, W, C, c + 1);
    o2 = bilinear_at(inb, ty, tx, W, C, c + 2);
  }
}

}  // namespace pcnn
