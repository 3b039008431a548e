// losses.hip — the vertex regression loss of PoseCNN's training graph (lib/fcn/train.py:564-573
// smooth_l1_loss_vertex) as two HBM-streaming passes instead of ~10 framework element-wise ops over
// three [B,H,W,3C] tensors (1.3 GB each at B=16, 640x480, C=22):
//
//   forward   reads pred, target, weight once -> loss = sum(in_loss) / (sum(weight) + 1e-10)
//   backward  reads them once more and writes d loss / d pred
//
// Canonical reduction order (TF's reduce_sum has none): SL1_BLOCKS x 256 threads, thread (blk, t)
// adds elements (blk*256 + t) + m*SL1_BLOCKS*256 for m ascending into f32 accumulators; a 256-leaf
// halving tree in LDS, then a SL1_BLOCKS-leaf halving tree in a second one-block kernel. The CPU
// checker restates exactly this order, so the loss matches bit for bit.
#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int SL1_BLOCKS = 1024;

__device__ __forceinline__ void sl1_elem(float p, float t, float w, float sigma2, float& in_loss,
                                         float& dpred)
{
  const float diff = w * (p - t);
  const float ad = fabsf(diff);
  const float inv = div_rn(1.0f, sigma2);
  if (ad < inv) {
    in_loss = (diff * diff) * div_rn(sigma2, 2.0f);
    dpred = w * (sigma2 * diff);
  } else {
    in_loss = ad - div_rn(0.5f, sigma2);
    dpred = w * (diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f));
  }
}

__global__ __launch_bounds__(256) void sl1_partial_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ weight,
                                                          long long n, float sigma2,
                                                          float* __restrict__ partial)
{
  __shared__ float sl[256], sw[256];
  const int t = threadIdx.x;
  float al = 0.f, aw = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)SL1_BLOCKS * 256) {
    float il, dp;
    const float w = weight[i];
    sl1_elem(pred[i], target[i], w, sigma2, il, dp);
    al = al + il;
    aw = aw + w;
  }
  sl[t] = al;
  sw[t] = aw;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (t < st) {
      sl[t] = sl[t] + sl[t + st];
      sw[t] = sw[t] + sw[t + st];
    }
    __syncthreads();
  }
  if (t == 0) {
    partial[blockIdx.x] = sl[0];
    partial[SL1_BLOCKS + blockIdx.x] = sw[0];
  }
}

__global__ __launch_bounds__(SL1_BLOCKS / 2) void sl1_final_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ out)
{
  __shared__ float sl[SL1_BLOCKS], sw[SL1_BLOCKS];
  const int t = threadIdx.x;
  sl[t] = partial[t];
  sl[t + SL1_BLOCKS / 2] = partial[t + SL1_BLOCKS / 2];
  sw[t] = partial[SL1_BLOCKS + t];
  sw[t + SL1_BLOCKS / 2] = partial[SL1_BLOCKS + t + SL1_BLOCKS / 2];
  __syncthreads();
  for (int st = SL1_BLOCKS / 2; st >= 1; st >>= 1) {
    if (t < st) {
      sl[t] = sl[t] + sl[t + st];
      sw[t] = sw[t] + sw[t + st];
    }
    __syncthreads();
  }
  if (t == 0) {
    const float denom = sw[0] + 1e-10f;
    out[0] = div_rn(sl[0], denom);
    out[1] = sl[0];
    out[2] = sw[0];
  }
}

__global__ __launch_bounds__(256) void sl1_bwd_kernel(const float* __restrict__ pred,
                                                      const float* __restrict__ target,
                                                      const float* __restrict__ weight,
                                                      const float* __restrict__ sums,
                                                      const float* __restrict__ upstream,
                                                      long long n, float sigma2,
                                                      float* __restrict__ grad)
{
  const float denom = sums[2] + 1e-10f;
  const float g = upstream ? upstream[0] : 1.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    float il, dp;
    sl1_elem(pred[i], target[i], weight[i], sigma2, il, dp);
    grad[i] = div_rn(dp, denom) * g;
  }
}

}  // namespace

extern "C" int pcnn_smooth_l1_vertex_workspace_bytes(size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "smooth_l1_vertex: NULL bytes");
  *bytes = sizeof(float) * 2 * SL1_BLOCKS;
  return PCNN_OK;
}

extern "C" int pcnn_smooth_l1_vertex_fwd(const float* pred, const float* target, const float* weight,
                                         int64_t n, float sigma, float* out, void* workspace,
                                         size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(n >= 0, PCNN_EINVAL, "smooth_l1_vertex: negative size");
  PCNN_REQUIRE(sigma > 0.f, PCNN_EINVAL, "smooth_l1_vertex: sigma must be positive");
  PCNN_REQUIRE(out && (n == 0 || (pred && target && weight)), PCNN_ENULL, "smooth_l1_vertex: NULL pointer");
  PCNN_REQUIRE(workspace && workspace_bytes >= sizeof(float) * 2 * SL1_BLOCKS, PCNN_EWORKSPACE,
               "smooth_l1_vertex: workspace NULL or too small");
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = (float*)workspace;
  PCNN_LAUNCH(sl1_partial_kernel, dim3(SL1_BLOCKS), dim3(256), 0, stream, pred, target, weight,
              (long long)n, sigma * sigma, partial);
  PCNN_LAUNCH(sl1_final_kernel, dim3(1), dim3(SL1_BLOCKS / 2), 0, stream, partial, out);
  return check_launch("smooth_l1_vertex_fwd");
}

extern "C" int pcnn_smooth_l1_vertex_bwd(const float* pred, const float* target, const float* weight,
                                         const float* out, const float* upstream, int64_t n,
                                         float sigma, float* grad_pred, void* stream_)
{
  PCNN_REQUIRE(n >= 0, PCNN_EINVAL, "smooth_l1_vertex_bwd: negative size");
  PCNN_REQUIRE(sigma > 0.f, PCNN_EINVAL, "smooth_l1_vertex_bwd: sigma must be positive");
  if (n == 0) return PCNN_OK;
  PCNN_REQUIRE(pred && target && weight && out && grad_pred, PCNN_ENULL, "smooth_l1_vertex_bwd: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  PCNN_LAUNCH(sl1_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pred, target, weight, out,
              upstream, (long long)n, sigma * sigma, grad_pred);
  return check_launch("smooth_l1_vertex_bwd");
}
