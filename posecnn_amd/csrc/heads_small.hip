// heads_small.hip — the small stuff between the trunk and the custom layers, one launch each instead of a dozen
// framework kernels (VERDICT r2 "weak" #9: at batch 1 those launches were ~10 % of a frame):
//
//   head_lowres_kernel    vgg16_convs.py:128-142 (label head) and :151-163 (vertex head) in the fused-heads form of
//                         posecnn_amd/networks.py:  t = score_conv4 + deconv_{4,2}(score_conv5) [+ planted scene];
//                         z = conv1x1(t) (no bias, no ReLU: both ride in the full-resolution epilogue kernels).
//                         `add_score` / `dropout` (= t, keep_prob 1) is still written — other layers read it.
//                         Replaces deconv_bilinear + add + add + library 1x1 convolution.
//   det_assemble_kernel   lib/fcn/test.py:197-211 on the device: poses[i, :4] = poses_tanh[i, 4 c : 4 c + 4] with
//                         c = int(rois[i, 1]); rows = box7 | quaternion4 | translation3, zeros past the device-side
//                         count; `stride` 9 picks the un-jittered first row of each training-mode group.
#include <algorithm>

#include "bilinear.h"
#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int HL_PX = 32;   // low-resolution pixels per workgroup

__global__ __launch_bounds__(256) void head_lowres_kernel(
    const float* __restrict__ a, const float* __restrict__ b5, const float* __restrict__ planted,
    const float* __restrict__ wT, float* __restrict__ add_out, float* __restrict__ z, int B, int h, int w, int U,
    int Cout, int k, int s)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tL = smem;                  // [HL_PX][U]
  float* wL = smem + HL_PX * U;      // [U][Cout]
  __shared__ Taps s_ty[HL_PX], s_tx[HL_PX];   // the interpolation taps of a pixel: once per pixel (double arithmetic), not per element
  __shared__ int s_img[HL_PX];
  const int tid = threadIdx.x;
  const long long total = (long long)B * h * w;
  const long long gp0 = (long long)blockIdx.x * HL_PX;
  const int h5 = h / s, w5 = w / s, pad = (k - s) / 2;
  if (tid < HL_PX) {
    const long long gp = gp0 + tid < total ? gp0 + tid : total - 1;
    s_tx[tid] = make_taps((int)(gp % w), k, s, pad, w5);
    s_ty[tid] = make_taps((int)((gp / w) % h), k, s, pad, h5);
    s_img[tid] = (int)(gp / ((long long)w * h));
  }
  for (int i = tid; i < U * Cout; i += 256) wL[i] = wT[i];
  __syncthreads();
  for (int idx = tid; idx < HL_PX * U; idx += 256) {
    const int p = idx / U, c = idx - p * U;
    const long long gp = gp0 + p;
    float t = 0.f;
    if (gp < total) {
      const float up = bilinear_at(b5 + (size_t)s_img[p] * h5 * w5 * U, s_ty[p], s_tx[p], w5, U, c);
      t = a[gp * U + c] + up;                       // add_score = score_conv4 + upscore_conv5   (tf.add_n order)
      if (planted) t = t + planted[gp * U + c];     // bench aid: the planted scene
      add_out[gp * U + c] = t;
    }
    tL[idx] = t;
  }
  __syncthreads();
  // z[p][co] = sum_k t[p][k] W[k][co], k ascending; a thread owns 4 pixels x 1 output channel and walks K four at a
  // time (one 128-bit LDS read per pixel — all lanes of a pixel quad read the same address: a broadcast)
  const int items = (HL_PX / 4) * Cout;
  for (int it = tid; it < items; it += 256) {
    const int pq = it / Cout, co = it - pq * Cout;
    const float* t0 = tL + (4 * pq) * U;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int kk = 0; kk < U; kk += 4) {
      const float4 x0 = *reinterpret_cast<const float4*>(t0 + kk), x1 = *reinterpret_cast<const float4*>(t0 + U + kk);
      const float4 x2 = *reinterpret_cast<const float4*>(t0 + 2 * U + kk), x3 = *reinterpret_cast<const float4*>(t0 + 3 * U + kk);
      const float w0 = wL[kk * Cout + co], w1 = wL[(kk + 1) * Cout + co], w2 = wL[(kk + 2) * Cout + co], w3 = wL[(kk + 3) * Cout + co];
      acc0 = __builtin_fmaf(x0.x, w0, acc0); acc0 = __builtin_fmaf(x0.y, w1, acc0); acc0 = __builtin_fmaf(x0.z, w2, acc0); acc0 = __builtin_fmaf(x0.w, w3, acc0);
      acc1 = __builtin_fmaf(x1.x, w0, acc1); acc1 = __builtin_fmaf(x1.y, w1, acc1); acc1 = __builtin_fmaf(x1.z, w2, acc1); acc1 = __builtin_fmaf(x1.w, w3, acc1);
      acc2 = __builtin_fmaf(x2.x, w0, acc2); acc2 = __builtin_fmaf(x2.y, w1, acc2); acc2 = __builtin_fmaf(x2.z, w2, acc2); acc2 = __builtin_fmaf(x2.w, w3, acc2);
      acc3 = __builtin_fmaf(x3.x, w0, acc3); acc3 = __builtin_fmaf(x3.y, w1, acc3); acc3 = __builtin_fmaf(x3.z, w2, acc3); acc3 = __builtin_fmaf(x3.w, w3, acc3);
    }
    const long long gp = gp0 + 4 * pq;
    if (gp < total) z[gp * Cout + co] = acc0;
    if (gp + 1 < total) z[(gp + 1) * Cout + co] = acc1;
    if (gp + 2 < total) z[(gp + 2) * Cout + co] = acc2;
    if (gp + 3 < total) z[(gp + 3) * Cout + co] = acc3;
  }
}

__global__ __launch_bounds__(256) void det_assemble_kernel(
    const float* __restrict__ rois, const float* __restrict__ poses_tanh, const float* __restrict__ top_pose,
    const int* __restrict__ count_dev, int rows_in, int stride, int C, float* __restrict__ rows_out,
    int* __restrict__ count_out, int rows_out_n)
{
  const int count = min(max(count_dev[0], 0), rows_in);
  if (blockIdx.x == 0 && threadIdx.x == 0) count_out[0] = count / stride;
  const int total = rows_out_n * 14;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int ro = i / 14, col = i - ro * 14;
    const int ri = ro * stride;
    float v = 0.f;
    if (ri < count) {
      if (col < 7) v = rois[(size_t)ri * 7 + col];
      else if (col < 11) {
        int cls = (int)rois[(size_t)ri * 7 + 1];
        cls = cls < 0 ? 0 : (cls > C - 1 ? C - 1 : cls);
        v = poses_tanh[(size_t)ri * 4 * C + 4 * cls + (col - 7)];
      } else v = top_pose[(size_t)ri * 7 + 4 + (col - 11)];
    }
    rows_out[i] = v;
  }
}

}  // namespace

extern "C" int pcnn_head_lowres_fwd(const float* score4, const float* score5, const float* planted,
                                    const float* weights_t, int B, int h, int w, int units, int out_channels,
                                    int kernel, int stride, float* add_out, float* z, void* stream_)
{
  PCNN_REQUIRE(B >= 1 && h >= 1 && w >= 1 && units >= 4 && units % 4 == 0 && out_channels >= 1, PCNN_EINVAL,
               "head_lowres: bad shape (units must be a multiple of 4, got %d)", units);
  PCNN_REQUIRE(stride >= 1 && kernel >= stride && (kernel - stride) % 2 == 0 && kernel <= 2 * stride && h % stride == 0 && w % stride == 0,
               PCNN_EINVAL, "head_lowres: need stride <= kernel <= 2 stride, (kernel - stride) even, %dx%d divisible by the stride %d", h, w, stride);
  PCNN_REQUIRE(score4 && score5 && weights_t && add_out && z, PCNN_ENULL, "head_lowres: NULL pointer");
  const size_t lds = sizeof(float) * ((size_t)HL_PX * units + (size_t)units * out_channels);
  PCNN_REQUIRE(lds <= 60 * 1024, PCNN_EINVAL, "head_lowres: %d units x %d outputs exceed the kernel's 64 KB of LDS", units, out_channels);
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * h * w;
  const long long blocks = (total + HL_PX - 1) / HL_PX;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "head_lowres: grid too large");
  PCNN_LAUNCH(head_lowres_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, score4, score5, planted, weights_t, add_out, z,
              B, h, w, units, out_channels, kernel, stride);
  return check_launch("head_lowres_fwd");
}

extern "C" int pcnn_det_assemble_fwd(const float* rois, const float* poses_tanh, const float* top_pose,
                                     const int32_t* num_rows_dev, int rows, int row_stride, int num_classes,
                                     float* det_rows, int32_t* det_count, void* stream_)
{
  PCNN_REQUIRE(rows >= 0 && row_stride >= 1 && num_classes >= 1, PCNN_EINVAL, "det_assemble: bad shape");
  PCNN_REQUIRE(num_rows_dev && det_count, PCNN_ENULL, "det_assemble: NULL count pointer");
  const int n_out = (rows + row_stride - 1) / row_stride;
  PCNN_REQUIRE(n_out == 0 || (rois && poses_tanh && top_pose && det_rows), PCNN_ENULL, "det_assemble: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int blocks = std::max(1, std::min(64, (n_out * 14 + 255) / 256));
  PCNN_LAUNCH(det_assemble_kernel, dim3(blocks), dim3(256), 0, stream, rois, poses_tanh, top_pose, num_rows_dev, rows, row_stride,
              num_classes, det_rows, det_count, n_out);
  return check_launch("det_assemble_fwd");
}
