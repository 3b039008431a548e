// average_distance.hip — gfx950 PoseCNN pose loss (PLoss / SLoss) forward + analytic dq gradient
// (replaces TF1 ops "Averagedistance"/"AveragedistanceGrad",
//  lib/average_distance_loss/average_distance_loss_op.cc:253-314,
//  average_distance_loss_op_gpu.cu.cc:35-206 (AveragedistanceForward), :210-252
//  (sum_losses_gradients), :256-343 (launcher), :347-377 (backward)).
//
// The reference writes 54 floats of per-(roi, point) rotation scratch, 88 floats of per-point
// gradient scratch and a loss per point to global memory (~1.5 MB per ROI), then sums them with one
// thread per output channel. Here:
//   adl_terms   grid (point slabs, rois): rotations live in registers; for a symmetric class the
//               nearest-neighbour search walks the gt-rotated model points staged through LDS in
//               1024-point tiles (every lane reads the same LDS address -> broadcast); per point
//               only the 5 non-zero terms {loss, dq_s, dq_u, dq_v, dq_w} go to the workspace.
//   adl_sum     grid (rois): the canonical ascending-p sums (sum_losses_gradients :237-248) from
//               an LDS copy of the ROI's 5 x P terms; writes the whole 4C-wide gradient row.
//   adl_total   thrust::reduce over ROIs (:333-335), ascending.
#include <cfloat>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int ADL_THREADS = 256;
constexpr int ADL_QTILE = 1024;

// quaternion (s,u,v,w) -> rotation, average_distance_loss_op_gpu.cu.cc:62-71
__device__ __forceinline__ void quat_rot(float s, float u, float v, float w, float* r)
{
  r[0] = s * s + u * u - v * v - w * w;
  r[1] = 2 * (u * v - s * w);
  r[2] = 2 * (u * w + s * v);
  r[3] = 2 * (u * v + s * w);
  r[4] = s * s - u * u + v * v - w * w;
  r[5] = 2 * (v * w - s * u);
  r[6] = 2 * (u * w - s * v);
  r[7] = 2 * (v * w + s * u);
  r[8] = s * s - u * u - v * v + w * w;
}

// first class whose weight is positive (average_distance_loss_op_gpu.cu.cc:52-60), found by the whole wave at once:
// lane l looks at classes l, l + 64, ...; a ballot picks the lowest (the serial form was a chain of up to C dependent
// loads in front of every block). Must be called by all lanes of the wave.
__device__ __forceinline__ int find_class(const float* __restrict__ weight, int n, int C)
{
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane_id();
    const bool hit = c < C && weight[(size_t)n * PCNN_POSE_CHANNELS * C + PCNN_POSE_CHANNELS * c] > 0;
    const unsigned long long m = __ballot(hit);
    if (m) return c0 + __ffsll((long long)m) - 1;
  }
  return -1;
}

// terms layout: [R][5][P]
__global__ __launch_bounds__(ADL_THREADS) void adl_terms_kernel(
    const float* __restrict__ prediction, const float* __restrict__ target,
    const float* __restrict__ weight, const float* __restrict__ point,
    const float* __restrict__ symmetry, float* __restrict__ terms, int R_cap, int C, int P,
    float margin, const int* __restrict__ num_rows_dev)
{
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float s_qx[ADL_QTILE], s_qy[ADL_QTILE], s_qz[ADL_QTILE];   // gt-rotated model points, coordinate-major
  const int n = blockIdx.y;
  const int p = blockIdx.x * ADL_THREADS + threadIdx.x;
  // R = the op's row count: the buffers' row capacity, or (capacity-sized buffers of the sync-free
  // Hough op) the device-side count; rows past it do not exist for the loss
  const int R = num_rows_dev ? min(R_cap, num_rows_dev[0]) : R_cap;
  const int cls = n < R ? find_class(weight, n, C) : -1;
  float* tn = terms + (size_t)n * 5 * P;
  if (cls < 0) return;   // no target (or past the row count): adl_sum writes this row's zeros without reading any terms
                         // (a train-mode buffer of 3024 rows holds ~470 with targets: 134 MB of zeros were written here
                         //  and read back there)
  const int qi = n * PCNN_POSE_CHANNELS * C + PCNN_POSE_CHANNELS * cls;
  float rg[9], ru[9];
  quat_rot(target[qi], target[qi + 1], target[qi + 2], target[qi + 3], rg);
  const float s = prediction[qi], u = prediction[qi + 1], v = prediction[qi + 2], w = prediction[qi + 3];
  quat_rot(s, u, v, w, ru);
  const float* pts = point + (size_t)cls * P * 3;
  const bool valid = p < P;
  float pt0 = 0, pt1 = 0, pt2 = 0;
  if (valid) { pt0 = pts[p * 3]; pt1 = pts[p * 3 + 1]; pt2 = pts[p * 3 + 2]; }
  const float x1 = ru[0] * pt0 + ru[1] * pt1 + ru[2] * pt2;
  const float y1 = ru[3] * pt0 + ru[4] * pt1 + ru[5] * pt2;
  const float z1 = ru[6] * pt0 + ru[7] * pt1 + ru[8] * pt2;

  int qmin = p;
  if (symmetry[cls] > 0) {
    // closest gt-rotated model point, strict '<' from FLT_MAX, first wins (:155-168)
    float dmin = FLT_MAX;
    for (int q0 = 0; q0 < P; q0 += ADL_QTILE) {
      __syncthreads();
      for (int j = threadIdx.x; j < ADL_QTILE; j += ADL_THREADS) {
        int q = q0 + j;
        // past the last point: +inf coordinates -> distance +inf (or NaN), never '<' anything: the scan below runs over
        // whole groups of four without a tail
        float qx = __builtin_inff(), qy = qx, qz = qx;
        if (q < P) {
          float a = pts[q * 3], b = pts[q * 3 + 1], c = pts[q * 3 + 2];
          qx = rg[0] * a + rg[1] * b + rg[2] * c;
          qy = rg[3] * a + rg[4] * b + rg[5] * c;
          qz = rg[6] * a + rg[7] * b + rg[8] * c;
        }
        s_qx[j] = qx; s_qy[j] = qy; s_qz[j] = qz;
      }
      __syncthreads();
      const int lim = min(ADL_QTILE, P - q0);
      // Four candidates per trip: one 128-bit LDS read per coordinate (every lane reads the same address: a broadcast)
      // and the squared distance of two candidates per packed-f32 instruction — v_pk_add / v_pk_mul round each half
      // like the scalar form, so every distance is ((ex ex) + (ey ey)) + (ez ez) bit for bit (:160-163); the strict '<'
      // walks the four in ascending q. Round 4's loop took 3 LDS reads + 14 vector instructions per candidate, this one
      // 0.75 + ~7: the kernel is bound by exactly this instruction stream (P^2 = 6.9 M candidates per symmetric RoI).
      const v2f x1v = (v2f){x1, x1}, y1v = (v2f){y1, y1}, z1v = (v2f){z1, z1};
      // SIXTEEN candidates per trip, their twelve LDS reads issued before the first use: a symmetric row's workgroup is one
      // wave per SIMD with nobody to run under an LDS round trip, and with one read-then-use group of four per trip the scan
      // cost 78 cycles per candidate — ~120 of LDS latency per trip — i.e. 85 us for a single symmetric row
      // (tools/probe_adl.py). The tile is padded with +inf to a multiple of 16 (ADL_QTILE is one).
      for (int j = 0; j < lim; j += 16) {
        v4f qx[4], qy[4], qz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          qx[u] = *reinterpret_cast<const v4f*>(&s_qx[j + 4 * u]);
          qy[u] = *reinterpret_cast<const v4f*>(&s_qy[j + 4 * u]);
          qz[u] = *reinterpret_cast<const v4f*>(&s_qz[j + 4 * u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const v2f ex0 = x1v - qx[u].xy, ey0 = y1v - qy[u].xy, ez0 = z1v - qz[u].xy;
          const v2f ex1 = x1v - qx[u].zw, ey1 = y1v - qy[u].zw, ez1 = z1v - qz[u].zw;
          const v2f d0 = (ex0 * ex0 + ey0 * ey0) + ez0 * ez0;
          const v2f d1 = (ex1 * ex1 + ey1 * ey1) + ez1 * ez1;
          const int q4 = q0 + j + 4 * u;
          if (d0.x < dmin) { dmin = d0.x; qmin = q4; }
          if (d0.y < dmin) { dmin = d0.y; qmin = q4 + 1; }
          if (d1.x < dmin) { dmin = d1.x; qmin = q4 + 2; }
          if (d1.y < dmin) { dmin = d1.y; qmin = q4 + 3; }
        }
      }
    }
  }
  if (!valid) return;
  const float qa = pts[qmin * 3], qb = pts[qmin * 3 + 1], qc = pts[qmin * 3 + 2];
  const float x2 = rg[0] * qa + rg[1] * qb + rg[2] * qc;
  const float y2 = rg[3] * qa + rg[4] * qb + rg[5] * qc;
  const float z2 = rg[6] * qa + rg[7] * qb + rg[8] * qc;
  const float ex = x1 - x2, ey = y1 - y2, ez = z1 - z2;
  const float distance = ex * ex + ey * ey + ez * ez;
  float loss = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
  if (!(distance < margin)) {
    loss = (float)((double)(distance - margin) / (2.0 * R * P));
    // derivatives of Ru w.r.t. (s,u,v,w), :96-139
    const float d0[9] = {2 * s, -2 * w, 2 * v, 2 * w, 2 * s, -2 * u, -2 * v, 2 * u, 2 * s};
    const float d1[9] = {2 * u, 2 * v, 2 * w, 2 * v, -2 * u, -2 * s, 2 * w, 2 * s, -2 * u};
    const float d2[9] = {-2 * v, 2 * u, 2 * s, 2 * u, 2 * v, 2 * w, -2 * s, 2 * w, -2 * v};
    const float d3[9] = {-2 * w, -2 * s, 2 * u, 2 * s, -2 * w, 2 * v, 2 * u, 2 * v, 2 * w};
    const float den = (float)(R * P);
    const float pt[3] = {pt0, pt1, pt2};
    const float df[3] = {ex, ey, ez};
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        g0 += div_rn(df[j] * pt[k] * d0[j * 3 + k], den);
        g1 += div_rn(df[j] * pt[k] * d1[j * 3 + k], den);
        g2 += div_rn(df[j] * pt[k] * d2[j * 3 + k], den);
        g3 += div_rn(df[j] * pt[k] * d3[j * 3 + k], den);
      }
  }
  tn[p] = loss;
  tn[(size_t)P + p] = g0;
  tn[(size_t)2 * P + p] = g1;
  tn[(size_t)3 * P + p] = g2;
  tn[(size_t)4 * P + p] = g3;
}

// ascending-p sums; one wave per ROI, lanes 0..4 own one chain each
__global__ __launch_bounds__(64) void adl_sum_kernel(const float* __restrict__ terms,
                                                     const float* __restrict__ weight,
                                                     float* __restrict__ loss_batch,
                                                     float* __restrict__ bottom_diff, int C, int P,
                                                     int R_cap, const int* __restrict__ num_rows_dev)
{
  extern __shared__ __attribute__((aligned(16))) float s_t[];  // tile of [5][TILE]
  constexpr int TILE = 2048;
  const int n = blockIdx.x, lane = threadIdx.x;
  const int CH = PCNN_POSE_CHANNELS * C;
  const int R = num_rows_dev ? min(R_cap, num_rows_dev[0]) : R_cap;
  const int cls = n < R ? find_class(weight, n, C) : -1;
  for (int c = lane; c < CH; c += 64) bottom_diff[(size_t)n * CH + c] = 0.f;
  if (cls < 0) {         // every term of the row is +0 (never written by adl_terms): the sums are +0
    if (lane == 0) loss_batch[n] = 0.f;
    return;
  }
  const float* tn = terms + (size_t)n * 5 * P;
  float acc = 0.f;
  for (int p0 = 0; p0 < P; p0 += TILE) {
    const int lim = min(TILE, P - p0);
    __syncthreads();
    for (int k = 0; k < 5; k++)
      for (int j = lane; j < lim; j += 64) s_t[k * TILE + j] = tn[(size_t)k * P + p0 + j];
    __syncthreads();
    if (lane < 5) {
      const float* t = s_t + lane * TILE;
      for (int j = 0; j < lim; j++) acc += t[j];
    }
  }
  __syncthreads();
  if (lane == 0) loss_batch[n] = acc;
  if (lane >= 1 && lane < 5 && cls >= 0) bottom_diff[(size_t)n * CH + PCNN_POSE_CHANNELS * cls + (lane - 1)] = acc;
}

// thrust::reduce over the ROIs (:333-335), canonical order = ascending n: the wave stages 1024 terms at a
// time in LDS with coalesced loads, lane 0 adds them one by one from there (a sequential f32 sum, but at
// LDS latency instead of one dependent global load per term: 139 -> ~10 us at 3024 rows)
__global__ __launch_bounds__(64) void adl_total_kernel(const float* __restrict__ loss_batch, float* __restrict__ loss, int R_cap,
                                                       const int* __restrict__ num_rows_dev)
{
  __shared__ float s_l[1024];
  const int R = num_rows_dev ? min(R_cap, num_rows_dev[0]) : R_cap;   // rows past the count hold +0 terms
  const int lane = threadIdx.x;
  float total = 0.f;
  for (int n0 = 0; n0 < R; n0 += 1024) {
    const int lim = min(1024, R - n0);
    for (int j = lane; j < lim; j += 64) s_l[j] = loss_batch[n0 + j];
    __syncthreads();
    if (lane == 0)
      for (int j = 0; j < lim; j++) total += s_l[j];
    __syncthreads();
  }
  if (lane == 0) loss[0] = total;
}

__global__ __launch_bounds__(256) void adl_bwd_kernel(const float* __restrict__ grad,
                                                      const float* __restrict__ bottom_diff,
                                                      float* __restrict__ out, long long total)
{
  const float g = grad[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    out[i] = g * bottom_diff[i];
}

size_t adl_ws(int R, int P) { return align_up(sizeof(float) * (size_t)R * 5 * P, 256) + align_up(sizeof(float) * (size_t)(R > 0 ? R : 1), 256); }

}  // namespace

extern "C" int pcnn_average_distance_workspace_bytes(int R, int C, int P, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "average_distance_workspace_bytes: bytes is NULL");
  PCNN_REQUIRE(R >= 0 && C >= 1 && P >= 1, PCNN_EINVAL, "average_distance: bad shape R=%d C=%d P=%d", R, C, P);
  *bytes = adl_ws(R, P);
  return PCNN_OK;
}

extern "C" int pcnn_average_distance_fwd(const float* prediction, const float* target,
                                         const float* weight, const float* point,
                                         const float* symmetry, int R, int C, int P, float margin,
                                         const int32_t* num_rows_dev,
                                         float* loss, float* bottom_diff, void* workspace,
                                         size_t workspace_bytes, void* stream_)
{
  // attribute check, average_distance_loss_op.cc:262-267
  PCNN_REQUIRE(margin >= 0, PCNN_EINVAL, "average_distance: Need margin >= 0, got %g", (double)margin);
  PCNN_REQUIRE(R >= 0 && C >= 1 && P >= 1, PCNN_EINVAL, "average_distance: bad shape R=%d C=%d P=%d", R, C, P);
  PCNN_REQUIRE((long long)R * P < (1ll << 31), PCNN_EINVAL, "average_distance: R*P overflows int32");
  PCNN_REQUIRE(R <= 65535, PCNN_EINVAL, "average_distance: %d rows exceed the 65535 rows one launch addresses (grid.y); split the batch", R);
  PCNN_REQUIRE(loss, PCNN_ENULL, "average_distance: loss is NULL");
  hipStream_t stream = (hipStream_t)stream_;
  if (R == 0) {
    return pcnn::zero_async(loss, sizeof(float), stream, "average_distance");
  }
  PCNN_REQUIRE(prediction && target && weight && point && symmetry && bottom_diff, PCNN_ENULL,
               "average_distance: NULL pointer");
  PCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= adl_ws(R, P), PCNN_EWORKSPACE,
               "average_distance: workspace NULL, misaligned or too small (%zu < %zu)", workspace_bytes, adl_ws(R, P));
  float* terms = (float*)workspace;
  float* loss_batch = (float*)((char*)workspace + align_up(sizeof(float) * (size_t)R * 5 * P, 256));
  PCNN_LAUNCH(adl_terms_kernel, dim3((P + ADL_THREADS - 1) / ADL_THREADS, R), dim3(ADL_THREADS), 0,
                     stream, prediction, target, weight, point, symmetry, terms, R, C, P, margin, num_rows_dev);
  PCNN_LAUNCH(adl_sum_kernel, dim3(R), dim3(64), sizeof(float) * 5 * 2048, stream, terms, weight,
                     loss_batch, bottom_diff, C, P, R, num_rows_dev);
  PCNN_LAUNCH(adl_total_kernel, dim3(1), dim3(64), 0, stream, loss_batch, loss, R, num_rows_dev);
  return pcnn::check_launch("average_distance_fwd");
}

extern "C" int pcnn_average_distance_bwd(const float* grad, const float* bottom_diff, int R,
                                         int channels, float* out, void* stream_)
{
  PCNN_REQUIRE(R >= 0 && channels >= 1, PCNN_EINVAL, "average_distance_bwd: bad shape");
  if (R == 0) return PCNN_OK;
  PCNN_REQUIRE(grad && bottom_diff && out, PCNN_ENULL, "average_distance_bwd: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  long long total = (long long)R * channels;
  int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  PCNN_LAUNCH(adl_bwd_kernel, dim3(blocks), dim3(256), 0, stream, grad, bottom_diff, out, total);
  return pcnn::check_launch("average_distance_bwd");
}
