// winograd.hip — the data transforms of a Winograd F(2x2, 3x3) evaluation of the VGG trunk's deep
// 3x3 convolutions (`Network.conv`, lib/networks/network.py:159-187; vgg16_convs.py:42-52).
//
// The trunk is 85 % of a frame and runs at 83 % of the fp32 MFMA peak as a direct convolution:
// the only way to make an fp32 3x3 convolution cheaper is to do fewer multiplies. F(2x2, 3x3)
// computes a 2x2 output tile from a 4x4 input tile with 16 multiplies per (cin, cout) instead of
// 36 — exact in real arithmetic, all-f32 here (what cuDNN itself picks for these layers):
//
//   V[k][t][ci] = (B^T d B)[k]          pcnn_winograd_input_fwd    (this file, HBM stream)
//   M[k] = V[k] (T x Cin) * U[k]        16 fp32 GEMMs              (library batched GEMM, MFMA)
//   Y tile = A^T M A + bias, ReLU       pcnn_winograd_output_fwd   (this file, HBM stream;
//                                                                   optional fused 2x2 max-pool)
//   U[k][ci][co] = (G g G^T)[k]         once per filter            (host, float64 -> f32)
//
// with t = (b, ty, tx) over the (H/2) x (W/2) output tiles, k = 4*i + j over the 4x4 transform
// domain. It pays where the 4x larger V / M tensors are cheap against the saved MACs: C >= 256.
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]      A^T = [1 1 1 0; 0 1 -1 -1]
// Canonical order: rows first, then columns; sums left to right; + bias last.
#include <cstdlib>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x,
                                                         float* __restrict__ v, int H, int W, int C,
                                                         long long total, long long plane)
{
  const int cv = C / 4;
  const int Ht = H / 2, Wt = W / 2;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * 4;
    const long long t = idx / cv;
    const int tx = (int)(t % Wt);
    const int ty = (int)((t / Wt) % Ht);
    const long long b = t / ((long long)Wt * Ht);
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    const float* xb = x + b * H * W * (long long)C + c;
    f4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int yy = y0 + r;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int xx = x0 + s;
        f4 val = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
          val = *reinterpret_cast<const f4*>(xb + ((long long)yy * W + xx) * C);
        d[r][s] = val;
      }
    }
    f4 tmp[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      tmp[0][s] = d[0][s] - d[2][s];
      tmp[1][s] = d[1][s] + d[2][s];
      tmp[2][s] = d[2][s] - d[1][s];
      tmp[3][s] = d[1][s] - d[3][s];
    }
    float* vo = v + t * C + c;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *reinterpret_cast<f4*>(vo + (4 * i + 0) * plane) = tmp[i][0] - tmp[i][2];
      *reinterpret_cast<f4*>(vo + (4 * i + 1) * plane) = tmp[i][1] + tmp[i][2];
      *reinterpret_cast<f4*>(vo + (4 * i + 2) * plane) = tmp[i][2] - tmp[i][1];
      *reinterpret_cast<f4*>(vo + (4 * i + 3) * plane) = tmp[i][1] - tmp[i][3];
    }
  }
}

__device__ __forceinline__ f4 relu4(f4 a)
{
  f4 r;
  r.x = a.x > 0.f ? a.x : 0.f;
  r.y = a.y > 0.f ? a.y : 0.f;
  r.z = a.z > 0.f ? a.z : 0.f;
  r.w = a.w > 0.f ? a.w : 0.f;
  return r;
}

__device__ __forceinline__ f4 max4(f4 a, f4 b)
{
  f4 r;
  r.x = b.x > a.x ? b.x : a.x;
  r.y = b.y > a.y ? b.y : a.y;
  r.z = b.z > a.z ? b.z : a.z;
  r.w = b.w > a.w ? b.w : a.w;
  return r;
}

template <bool POOL>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ m,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int H, int W, int C,
                                                          int relu, long long total, long long plane)
{
  const int cv = C / 4;
  const int Ht = H / 2, Wt = W / 2;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * 4;
    const long long t = idx / cv;
    const int tx = (int)(t % Wt);
    const int ty = (int)((t / Wt) % Ht);
    const long long b = t / ((long long)Wt * Ht);
    const float* mi = m + t * C + c;
    f4 q[4][4];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k >> 2][k & 3] = *reinterpret_cast<const f4*>(mi + k * plane);
    f4 tmp[2][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      tmp[0][j] = q[0][j] + q[1][j] + q[2][j];
      tmp[1][j] = q[1][j] - q[2][j] - q[3][j];
    }
    const f4 bq = *reinterpret_cast<const f4*>(bias + c);
    f4 o[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      o[a][0] = tmp[a][0] + tmp[a][1] + tmp[a][2] + bq;
      o[a][1] = tmp[a][1] - tmp[a][2] - tmp[a][3] + bq;
      if (relu) { o[a][0] = relu4(o[a][0]); o[a][1] = relu4(o[a][1]); }
    }
    if (POOL) {
      // the 2x2 output tile is exactly one max_pool(2,2,2,2) window (same window order as
      // bias_relu_pool2_kernel: (y,x), (y,x+1), (y+1,x), (y+1,x+1))
      const f4 p = max4(max4(max4(o[0][0], o[0][1]), o[1][0]), o[1][1]);
      *reinterpret_cast<f4*>(y + ((b * Ht + ty) * Wt + tx) * C + c) = p;
    } else {
      float* yo = y + ((b * H + 2 * ty) * W + 2 * tx) * C + c;
      *reinterpret_cast<f4*>(yo) = o[0][0];
      *reinterpret_cast<f4*>(yo + C) = o[0][1];
      *reinterpret_cast<f4*>(yo + (long long)W * C) = o[1][0];
      *reinterpret_cast<f4*>(yo + (long long)W * C + C) = o[1][1];
    }
  }
}

// ---- F(4x4, 3x3): 36 multiplies per 16 outputs (4x fewer than direct), transform-domain tensors
// only 2.25x the activation. Interpolation points 0, +-1, +-2, inf (Lavin & Gray 2015):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Tiles are 4x4 outputs; the tile grid is ceil(H/4) x ceil(W/4), partial tiles read zeros and skip
// the stores outside the image. Canonical expressions (one f32 rounding per written operation):
//   in : r0 = (4 d0 - 5 d2) + d4;  a = d4 - 4 d2, b = d3 - 4 d1: r1 = a + b, r2 = a - b;
//        c = d4 - d2, e = 2 (d3 - d1): r3 = c + e, r4 = c - e;  r5 = (4 d1 - 5 d3) + d5
//   out: s = m1 + m2, d = m1 - m2, S = m3 + m4, D = m3 - m4:
//        y0 = (m0 + s) + S;  y1 = d + 2 D;  y2 = s + 4 S;  y3 = (d + 8 D) + m5
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename VT>
__device__ __forceinline__ VT vrelu(VT a)
{
  VT r;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(VT) / 4); i++) r[i] = a[i] > 0.f ? a[i] : 0.f;
  return r;
}

template <typename VT>
__device__ __forceinline__ VT vmax(VT a, VT b)
{
  VT r;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(VT) / 4); i++) r[i] = b[i] > a[i] ? b[i] : a[i];
  return r;
}

template <typename VT>
__device__ __forceinline__ void bt6(const VT* d, VT* r)
{
  r[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
  const VT a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1];
  r[1] = a + b;
  r[2] = a - b;
  const VT c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
  r[3] = c + e;
  r[4] = c - e;
  r[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
}

template <typename VT>
__device__ __forceinline__ void at6(const VT* m, VT* y)
{
  const VT s = m[1] + m[2], d = m[1] - m[2], S = m[3] + m[4], D = m[3] - m[4];
  y[0] = (m[0] + s) + S;
  y[1] = d + 2.f * D;
  y[2] = s + 4.f * S;
  y[3] = (d + 8.f * D) + m[5];
}

// (Round 5 measured non-temporal stores for V — written once here, read back by wino43_mfma_kernel: the 12 transforms 4.57 ->
//  4.45 ms alone (conv4_1 0.113 -> 0.085, conv5_x 0.051 -> 0.044, conv2_2 0.819 -> 0.797) and the MFMA kernels behind them no
//  slower (14.37 -> 14.29 ms) — but inside the three-stream step the transform came out 3 % SLOWER (258 -> 265 us per launch),
//  the MFMA kernel 0.7 % faster, the step 805 vs 808 frames/s: nothing; plain stores stay.)
template <typename VT>
__global__ __launch_bounds__(256) void wino43_input_kernel(const float* __restrict__ x,
                                                           float* __restrict__ v, int H, int W, int C,
                                                           int Ht, int Wt, long long total,
                                                           long long plane, int blocked)
{
  constexpr int VW = sizeof(VT) / 4;
  const int cv = C / VW;
  // `blocked` (C % 64 == 0, float4): a workgroup = 4x4 neighbouring tiles x 64 channels, so the
  // overlapping halves of neighbouring 6x6 patches are re-read by the same CU (L1 / its XCD's L2)
  // instead of by workgroups the dispatcher has scattered over all 8 XCDs (PMC: 2.16x -> the
  // fetched bytes of X with the linear map). total then counts workgroups * 256.
  const int nbx = (Wt + 3) / 4, nby = (Ht + 3) / 4, ncg = C / 64;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    int c, tx, ty;
    long long b, t;
    if (blocked) {
      long long wg = idx >> 8;
      const int l = (int)(idx & 255);
      const int cgi = (int)(wg % ncg); wg /= ncg;
      const int bx = (int)(wg % nbx); wg /= nbx;
      const int by = (int)(wg % nby);
      b = wg / nby;
      c = cgi * 64 + (l & 15) * 4;
      tx = bx * 4 + ((l >> 4) & 3);
      ty = by * 4 + (l >> 6);
      if (tx >= Wt || ty >= Ht) continue;
      t = (b * Ht + ty) * Wt + tx;
    } else {
      c = (int)(idx % cv) * VW;
      t = idx / cv;
      tx = (int)(t % Wt);
      ty = (int)((t / Wt) % Ht);
      b = t / ((long long)Wt * Ht);
    }
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const float* xb = x + b * H * W * (long long)C + c;
    VT tmp[6][6];  // tmp[i][s] = (B^T d)[i][s]
    // The patch's 36 loads are issued together, from clamped (always valid) addresses, and the pixels outside the image
    // are zeroed afterwards. With `if (inside) load` every load sat under its own exec branch and the compiler waited for
    // each patch column before transforming it: six trips to memory per thread, one after the other, at two waves per
    // SIMD (round 5; same values, same arithmetic).
    VT d[6][6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const int yy = min(max(y0 + r, 0), H - 1);
#pragma unroll
      for (int s2 = 0; s2 < 6; s2++) {
        const int xx = min(max(x0 + s2, 0), W - 1);
        d[r][s2] = *reinterpret_cast<const VT*>(xb + ((long long)yy * W + xx) * C);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 6; s2++) {
      const int xx = x0 + s2;
      VT col[6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const int yy = y0 + r;
        const VT zero = {};
        col[r] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? d[r][s2] : zero;
      }
      VT o[6];
      bt6(col, o);
#pragma unroll
      for (int i = 0; i < 6; i++) tmp[i][s2] = o[i];
    }
    float* vo = v + t * C + c;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      VT o[6];
      bt6(tmp[i], o);
#pragma unroll
      for (int j = 0; j < 6; j++) *reinterpret_cast<VT*>(vo + (6 * i + j) * plane) = o[j];
    }
  }
}

// POOL: 0 = y [B,H,W,C]; 1 = only the 2x2 max-pooled tensor (into y); 2 = both (y and ypool) — for a
// layer like conv4_3 whose un-pooled activation other layers read as well
template <int POOL, typename VT>
__global__ __launch_bounds__(256) void wino43_output_kernel(const float* __restrict__ m,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ y, float* __restrict__ ypool,
                                                            int H, int W, int C,
                                                            int Ht, int Wt, int relu, long long total,
                                                            long long plane)
{
  constexpr int VW = sizeof(VT) / 4;
  const int cv = C / VW;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * VW;
    const long long t = idx / cv;
    const int tx = (int)(t % Wt);
    const int ty = (int)((t / Wt) % Ht);
    const long long b = t / ((long long)Wt * Ht);
    const float* mi = m + t * C + c;
    VT tmp[4][6];  // tmp[a][j] = (A^T m)[a][j]
#pragma unroll
    for (int j = 0; j < 6; j++) {
      VT col[6];
#pragma unroll
      for (int i = 0; i < 6; i++) col[i] = *reinterpret_cast<const VT*>(mi + (6 * i + j) * plane);
      VT o[4];
      at6(col, o);
#pragma unroll
      for (int a = 0; a < 4; a++) tmp[a][j] = o[a];
    }
    const VT bq = *reinterpret_cast<const VT*>(bias + c);
    VT out[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      at6(tmp[a], out[a]);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        out[a][e] = out[a][e] + bq;
        if (relu) out[a][e] = vrelu(out[a][e]);
      }
    }
    const int oy0 = 4 * ty, ox0 = 4 * tx;
    if (POOL != 0) {
      // a 4x4 output tile holds 2x2 pooling windows (H, W even)
      float* yp = POOL == 1 ? y : ypool;
      const int Hp = H / 2, Wp = W / 2;
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int py = 2 * ty + a, px = 2 * tx + e;
          if (py < Hp && px < Wp) {
            const VT p = vmax(vmax(vmax(out[2 * a][2 * e], out[2 * a][2 * e + 1]), out[2 * a + 1][2 * e]), out[2 * a + 1][2 * e + 1]);
            *reinterpret_cast<VT*>(yp + ((b * Hp + py) * Wp + px) * C + c) = p;
          }
        }
    }
    if (POOL != 1) {
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (oy0 + a < H && ox0 + e < W)
            *reinterpret_cast<VT*>(y + ((b * H + oy0 + a) * W + ox0 + e) * C + c) = out[a][e];
    }
  }
}

int validate(int B, int H, int W, int C)
{
  PCNN_REQUIRE(B >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, PCNN_EINVAL,
               "winograd: need even height/width (got %dx%dx%d)", B, H, W);
  PCNN_REQUIRE(C >= 4 && C % 4 == 0, PCNN_EINVAL, "winograd: channels must be a multiple of 4 (got %d)", C);
  return PCNN_OK;
}

inline unsigned grid_for(long long total)
{
  long long b = (total + 255) / 256;
  return (unsigned)(b < 256 * 64 ? b : 256 * 64);
}

}  // namespace

extern "C" int pcnn_winograd43_input_fwd(const float* x, int B, int H, int W, int C, float* v,
                                         void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1, PCNN_EINVAL, "winograd43: bad shape %dx%dx%d", B, H, W);
  PCNN_REQUIRE(C >= 4 && C % 4 == 0, PCNN_EINVAL, "winograd43: channels must be a multiple of 4 (got %d)", C);
  PCNN_REQUIRE(x && v, PCNN_ENULL, "winograd43_input: NULL pointer");
  PCNN_REQUIRE(aligned16(x) && aligned16(v), PCNN_EINVAL, "winograd43_input: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const long long T = (long long)B * Ht * Wt;
  const long long total = T * (C / 4);
  // (a float2-per-thread instance — half the registers, twice the waves — measures the same: the
  // transforms sit at the HBM streaming rate, not at an occupancy limit)
  if (C % 64 == 0) {
    const long long wgs = (long long)B * ((Ht + 3) / 4) * ((Wt + 3) / 4) * (C / 64);
    PCNN_LAUNCH(wino43_input_kernel<f4>, dim3(grid_for(wgs * 256)), dim3(256), 0, stream, x, v, H, W, C, Ht, Wt, wgs * 256, T * C, 1);
  } else {
    PCNN_LAUNCH(wino43_input_kernel<f4>, dim3(grid_for(total)), dim3(256), 0, stream, x, v, H, W, C, Ht, Wt, total, T * C, 0);
  }
  return check_launch("winograd43_input_fwd");
}

extern "C" int pcnn_winograd43_output_fwd(const float* m, const float* bias, int B, int H, int W, int C,
                                          int relu, int pool, float* y, void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1, PCNN_EINVAL, "winograd43: bad shape %dx%dx%d", B, H, W);
  PCNN_REQUIRE(C >= 4 && C % 4 == 0, PCNN_EINVAL, "winograd43: channels must be a multiple of 4 (got %d)", C);
  PCNN_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), PCNN_EINVAL, "winograd43_output: pooling needs even height/width");
  PCNN_REQUIRE(m && bias && y, PCNN_ENULL, "winograd43_output: NULL pointer");
  PCNN_REQUIRE(aligned16(m) && aligned16(y) && aligned16(bias), PCNN_EINVAL, "winograd43_output: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const long long T = (long long)B * Ht * Wt;
  const long long total = T * (C / 4);
  if (pool)
    PCNN_LAUNCH((wino43_output_kernel<1, f4>), dim3(grid_for(total)), dim3(256), 0, stream, m, bias, y, (float*)nullptr, H, W, C, Ht, Wt, relu, total, T * C);
  else
    PCNN_LAUNCH((wino43_output_kernel<0, f4>), dim3(grid_for(total)), dim3(256), 0, stream, m, bias, y, (float*)nullptr, H, W, C, Ht, Wt, relu, total, T * C);
  return check_launch("winograd43_output_fwd");
}

extern "C" int pcnn_winograd43_output_both_fwd(const float* m, const float* bias, int B, int H, int W,
                                               int C, int relu, float* y, float* y_pool, void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, PCNN_EINVAL,
               "winograd43_output_both: need even height/width (got %dx%dx%d)", B, H, W);
  PCNN_REQUIRE(C >= 4 && C % 4 == 0, PCNN_EINVAL, "winograd43: channels must be a multiple of 4 (got %d)", C);
  PCNN_REQUIRE(m && bias && y && y_pool, PCNN_ENULL, "winograd43_output_both: NULL pointer");
  PCNN_REQUIRE(aligned16(m) && aligned16(y) && aligned16(y_pool) && aligned16(bias), PCNN_EINVAL,
               "winograd43_output_both: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const long long T = (long long)B * Ht * Wt;
  const long long total = T * (C / 4);
  PCNN_LAUNCH((wino43_output_kernel<2, f4>), dim3(grid_for(total)), dim3(256), 0, stream, m, bias, y, y_pool, H, W, C, Ht, Wt, relu, total, T * C);
  return check_launch("winograd43_output_both_fwd");
}

extern "C" int pcnn_winograd_input_fwd(const float* x, int B, int H, int W, int C, float* v,
                                       void* stream_)
{
  int st = validate(B, H, W, C);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(x && v, PCNN_ENULL, "winograd_input: NULL pointer");
  PCNN_REQUIRE(aligned16(x) && aligned16(v), PCNN_EINVAL, "winograd_input: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const long long T = (long long)B * (H / 2) * (W / 2);
  const long long total = T * (C / 4);
  PCNN_LAUNCH(wino_input_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, v, H, W, C, total, T * C);
  return check_launch("winograd_input_fwd");
}

extern "C" int pcnn_winograd_output_fwd(const float* m, const float* bias, int B, int H, int W, int C,
                                        int relu, int pool, float* y, void* stream_)
{
  int st = validate(B, H, W, C);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(m && bias && y, PCNN_ENULL, "winograd_output: NULL pointer");
  PCNN_REQUIRE(aligned16(m) && aligned16(y) && aligned16(bias), PCNN_EINVAL, "winograd_output: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const long long T = (long long)B * (H / 2) * (W / 2);
  const long long total = T * (C / 4);
  if (pool)
    PCNN_LAUNCH(wino_output_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, m, bias, y, H, W, C, relu, total, T * C);
  else
    PCNN_LAUNCH(wino_output_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, m, bias, y, H, W, C, relu, total, T * C);
  return check_launch("winograd_output_fwd");
}
