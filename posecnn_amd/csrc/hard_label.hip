// hard_label.hip — gfx950 hard-example label weights and the label-head epilogue.
//
//   pcnn_hard_label_*        replaces TF1 ops "Hardlabel"/"HardlabelGrad"
//                            (lib/hard_label_layer/hard_label_op.cc:143-188,
//                             hard_label_op_gpu.cu.cc:17-29,55-63 — GPU-kernel semantics)
//   pcnn_softmax_argmax_fwd  replaces softmax_high_dimension + argmax_2d
//                            (lib/networks/network.py:474-488,432-434; vgg16_convs.py:144-146)
//
// Both are pure HBM streams. The reference's HardlabelForward has every thread zero and patch its
// own 88-byte row (C = 22 floats), i.e. lanes write 88 bytes apart. Here a workgroup owns 256
// consecutive pixels: the per-pixel decision (which channel, if any, becomes 1) is made once by
// the pixel's thread and parked in LDS, then the 256*C output floats are written as one contiguous
// run of dwordx4 stores.
#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int HL_PIX = 256;

__global__ __launch_bounds__(256) void hard_label_fwd_kernel(const float* __restrict__ prob,
                                                             const int* __restrict__ gt,
                                                             float* __restrict__ out, long long N,
                                                             int C, float threshold)
{
  __shared__ int s_hot[HL_PIX];  // channel set to 1 for the pixel, or -1
  const int tid = threadIdx.x;
  for (long long p0 = (long long)blockIdx.x * HL_PIX; p0 < N; p0 += (long long)gridDim.x * HL_PIX) {
    const long long pix = p0 + tid;
    int hot = -1;
    if (pix < N) {
      const int g = gt[pix];
      // hard_label_op_gpu.cu.cc:25-27; labels outside [-1, C) index out of bounds there — ignored here
      if (g >= 0 && g < C && (g > 0 || prob[pix * C + g] < threshold)) hot = g;
    }
    s_hot[tid] = hot;
    __syncthreads();
    const long long npix = (N - p0) < HL_PIX ? (N - p0) : HL_PIX;
    const long long nfl = npix * C;
    float* o = out + p0 * C;
    if ((nfl & 3) == 0 && ((p0 * C) & 3) == 0) {
      for (long long i4 = tid; i4 < nfl / 4; i4 += 256) {
        const int i = (int)(i4 * 4);
        float4 v;
        int p = i / C, c = i - p * C;
        v.x = (s_hot[p] == c) ? 1.f : 0.f; if (++c == C) { c = 0; ++p; }
        v.y = (s_hot[p] == c) ? 1.f : 0.f; if (++c == C) { c = 0; ++p; }
        v.z = (s_hot[p] == c) ? 1.f : 0.f; if (++c == C) { c = 0; ++p; }
        v.w = (s_hot[p] == c) ? 1.f : 0.f;
        *reinterpret_cast<float4*>(o + i) = v;
      }
    } else {
      for (long long i = tid; i < nfl; i += 256) {
        const int p = (int)(i / C), c = (int)(i - (long long)p * C);
        o[i] = (s_hot[p] == c) ? 1.f : 0.f;
      }
    }
    __syncthreads();
  }
}

// softmax over the channel axis + first argmax of the probabilities. A thread owns one pixel's C
// scores (C <= 64 kept in registers via a fixed-trip loop over LDS-free reloads from global: the
// row is 88 B, read as scalars that the compiler merges; the kernel is bound by the 2 x N*C*4 B
// of traffic either way).
template <int CMAX>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ score,
                                                             float* __restrict__ prob,
                                                             int* __restrict__ label, long long N,
                                                             int C)
{
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < N;
       pix += (long long)gridDim.x * 256) {
    const float* s = score + pix * C;
    float e[CMAX];
    float m = s[0];
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) { e[c] = s[c]; m = fmaxf(m, e[c]); }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) { e[c] = exp_softmax_f32(e[c] - m); sum += e[c]; }
    int best = 0;
    float bestp = div_rn(e[0], sum);
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) {
        float p = div_rn(e[c], sum);
        if (prob) prob[pix * C + c] = p;
        if (p > bestp) { bestp = p; best = c; }
      }
    label[pix] = best;
  }
}

}  // namespace

extern "C" int pcnn_hard_label_fwd(const float* prob, const int32_t* gt, int64_t N, int C,
                                   float threshold, float* out, void* stream_)
{
  // attribute check, hard_label_op.cc:150-155
  PCNN_REQUIRE(threshold > 0, PCNN_EINVAL, "hard_label: Need threshold > 0, got %g", (double)threshold);
  PCNN_REQUIRE(N >= 0 && C >= 1, PCNN_EINVAL, "hard_label: bad shape N=%lld C=%d", (long long)N, C);
  if (N == 0) return PCNN_OK;
  PCNN_REQUIRE(prob && gt && out, PCNN_ENULL, "hard_label: NULL pointer");
  PCNN_REQUIRE(aligned16(out), PCNN_EINVAL, "hard_label: out must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  long long blocks = (N + HL_PIX - 1) / HL_PIX;
  if (blocks > 256 * 16) blocks = 256 * 16;
  PCNN_LAUNCH(hard_label_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, prob, gt,
                     out, (long long)N, C, threshold);
  return pcnn::check_launch("hard_label_fwd");
}

extern "C" int pcnn_hard_label_bwd(float* grad_prob, float* grad_gt, int64_t N, int C,
                                   void* stream_)
{
  PCNN_REQUIRE(N >= 0 && C >= 1, PCNN_EINVAL, "hard_label_bwd: bad shape");
  if (N == 0) return PCNN_OK;
  PCNN_REQUIRE(grad_prob && grad_gt, PCNN_ENULL, "hard_label_bwd: NULL output");
  hipStream_t stream = (hipStream_t)stream_;
  // HardlabelBackward, hard_label_op_gpu.cu.cc:55-63: zeros
  int st = pcnn::zero_async(grad_prob, sizeof(float) * (size_t)N * C, stream, "hard_label_bwd");
  if (st == PCNN_OK) st = pcnn::zero_async(grad_gt, sizeof(float) * (size_t)N, stream, "hard_label_bwd");
  if (st != PCNN_OK) return st;
  return PCNN_OK;
}

extern "C" int pcnn_softmax_argmax_fwd(const float* score, int64_t N, int C, float* prob,
                                       int32_t* label, void* stream_)
{
  PCNN_REQUIRE(N >= 0 && C >= 1 && C <= PCNN_MAX_CLASSES, PCNN_EINVAL,
               "softmax_argmax: need 1 <= num_classes <= %d (got %d)", PCNN_MAX_CLASSES, C);
  if (N == 0) return PCNN_OK;
  PCNN_REQUIRE(score && label, PCNN_ENULL, "softmax_argmax: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  long long blocks = (N + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (C <= 24)
    PCNN_LAUNCH(softmax_argmax_kernel<24>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       score, prob, label, (long long)N, C);
  else
    PCNN_LAUNCH(softmax_argmax_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       score, prob, label, (long long)N, C);
  return pcnn::check_launch("softmax_argmax_fwd");
}
