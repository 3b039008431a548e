// icp.hip — depth-based pose refinement, first slice: the projective point-to-plane ICP core
// behind Synthesizer::refinePose / solveICP (lib/synthesize/synthesize.cpp:2020-2026, :2052-2380; called from
// lib/fcn/test.py:1925-1933), i.e. df::icp (lib/kinect_fusion/src/optimization/icp.cpp:20-106) with its per-pixel
// kernel (src/optimization/icp.cu:25-136) and the masked depth -> vertex map step (synthesize.cpp:2139-2155 +
// src/image/backprojection.cu:10-27). The OpenGL renderer that produces the predicted vertex / normal maps, the
// PCL kd-tree hypothesis scoring and the nlopt refinement around it are NOT part of this slice: the entry takes the
// predicted maps as inputs, exactly like df::icp does.
//
// The reference runs, per iteration and per object: one kernel writing a 28-byte Jacobian/residual record for EVERY
// pixel of the frame (8.6 MB), cudaDeviceSynchronize, a thrust::transform_reduce over all of them, a second
// synchronise, a 6x6 solve on the host, and an H2D of the new pose — 8 to 50 times per object.
// Here an iteration is two launches and nothing leaves the device until the last one is done:
//   icp_terms_kernel   one thread per pixel of the predicted maps, all N objects of a frame in one grid (grid.y);
//                      the 29 sums a pixel contributes to (21 J^T J, 6 J^T r, inlier count, sum r^2) are reduced
//                      in LDS by a halving tree per 256-pixel block (blocks without a single contributing pixel —
//                      ~95 % of a frame — skip the tree) into one partial row per block;
//   icp_solve_kernel   one wave per object: partial rows added in ascending block order in f64 (29 lanes, coalesced),
//                      LDL^T solve, update = exp(solution), accumulated = update * accumulated — all in f64 by lane 0;
//                      the state stays in the workspace, the next iteration's terms kernel reads it from there.
// Arithmetic follows the canonical statement the CPU checker implements as well (same expression trees, same reduction
// order, sin/cos replaced by fixed Taylor polynomials), so the result is bit-identical to it (tests/test_gpu_icp.py).
#include <algorithm>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int ICP_BLOCK = 256;
constexpr int ICP_NSUM = 29;

__global__ __launch_bounds__(256) void icp_backproject_kernel(
    const uint16_t* __restrict__ depth, const int* __restrict__ label, long long P, int W, int obj_id, float factor,
    float fx, float fy, float px, float py, float* __restrict__ vertex_map)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)(i / W);
    const float d = (label == nullptr || label[i] == obj_id) ? (float)depth[i] / factor : 0.f;
    vertex_map[3 * i + 0] = ((float)x - px) / fx * d;
    vertex_map[3 * i + 1] = ((float)y - py) / fy * d;
    vertex_map[3 * i + 2] = d;
  }
}

__global__ __launch_bounds__(64) void icp_init_kernel(double* __restrict__ state, int N)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < 12 * N) {
    const int e = i % 12;
    state[i] = (e == 0 || e == 5 || e == 10) ? 1.0 : 0.0;
  }
}

// icpKernel (icp.cu:25-136) + the per-block part of the reduction
__global__ __launch_bounds__(ICP_BLOCK) void icp_terms_kernel(
    const float* __restrict__ live, const float* __restrict__ pred_v, const float* __restrict__ pred_n,
    const double* __restrict__ state, long long P, int H, int W, int pc, float fx, float fy, float px, float py,
    float znear, float zfar, float max_error, float* __restrict__ partial, int nblocks)
{
  __shared__ float red[ICP_NSUM][ICP_BLOCK];
  __shared__ int s_any;
  const int n = blockIdx.y, t = threadIdx.x;
  const long long p = (long long)blockIdx.x * ICP_BLOCK + t;
  if (t == 0) s_any = 0;
  __syncthreads();
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; i++) T[i] = (float)state[12 * n + i];
  float J[6], r = 0.f;
  bool ok = false;
  if (p < P) {
    const float* pv = pred_v + ((long long)n * P + p) * pc;
    const float* pn = pred_n + ((long long)n * P + p) * pc;
    const float* lv0 = live + (long long)n * P * 3;
    const float border = 2.f, ray_norm_dot_threshold = 0.1f;
    const float pvx = pv[0], pvy = pv[1], pvz = pv[2];
    if ((pvz >= znear) && !(pvz > zfar)) {
      const float ux = ((T[0] * pvx + T[1] * pvy) + T[2] * pvz) + T[3];
      const float uy = ((T[4] * pvx + T[5] * pvy) + T[6] * pvz) + T[7];
      const float uz = ((T[8] * pvx + T[9] * pvy) + T[10] * pvz) + T[11];
      const float projx = ux / uz * fx + px, projy = uy / uz * fy + py;
      if ((projx == projx) && (projy == projy) && !(fabsf(projx) > 1e8f) && !(fabsf(projy) > 1e8f)) {
        const int u = (int)(projx + 0.5f), v = (int)(projy + 0.5f);
        if (!(((float)u <= border) || ((float)u >= (float)(W - 1) - border) || ((float)v <= border) || ((float)v >= (float)(H - 1) - border))) {
          const float* lv = lv0 + 3 * ((long long)v * W + u);
          const float lx = lv[0], ly = lv[1], ldepth = lv[2];
          if ((ldepth >= znear) && !(ldepth > zfar)) {
            const float nrm = sqrt_rn((ux * ux + uy * uy) + uz * uz);
            const float rx = ux / nrm, ry = uy / nrm, rz = uz / nrm;
            const float nx = pn[0], ny = pn[1], nz = pn[2];
            const float dotrn = (rx * nx + ry * ny) + rz * nz;
            if (-dotrn >= ray_norm_dot_threshold) {
              const float ex = lx - ux, ey = ly - uy, ez = ldepth - uz;
              const float error = (nx * ex + ny * ey) + nz * ez;
              if (fabsf(error) <= max_error) {
                const float w = 1.f / ldepth;
                const float wx = w * nx, wy = w * ny, wz = w * nz;
                J[0] = wx; J[1] = wy; J[2] = wz;
                J[3] = wy * (-uz) + wz * uy;
                J[4] = wx * uz + wz * (-ux);
                J[5] = wx * (-uy) + wy * ux;
                r = w * error;
                ok = true;
              }
            }
          }
        }
      }
    }
  }
  if (ok) s_any = 1;    // (benign race: every writer stores 1)
  __syncthreads();
  float* out = partial + ((long long)n * nblocks + blockIdx.x) * ICP_NSUM;
  if (!s_any) {         // nothing contributes: the block's row is zero, no tree
    if (t < ICP_NSUM) out[t] = 0.f;
    return;
  }
  {
    int q = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i; j < 6; j++) { red[q][t] = ok ? J[i] * J[j] : 0.f; q++; }
#pragma unroll
    for (int i = 0; i < 6; i++) red[21 + i][t] = ok ? J[i] * r : 0.f;
    red[27][t] = ok ? 1.f : 0.f;
    red[28][t] = ok ? r * r : 0.f;
  }
  __syncthreads();
  for (int s = ICP_BLOCK / 2; s >= 1; s >>= 1) {
    if (t < s) {
#pragma unroll
      for (int q = 0; q < ICP_NSUM; q++) red[q][t] = red[q][t] + red[q][t + s];
    }
    __syncthreads();
  }
  if (t < ICP_NSUM) out[t] = red[t][0];
}

__device__ void icp_exp_se3(const double* xi, double* U)
{
  const double wx = xi[3], wy = xi[4], wz = xi[5];
  const double t2 = (wx * wx + wy * wy) + wz * wz;
  const double fa[10] = {1.0, 6.0, 120.0, 5040.0, 362880.0, 39916800.0, 6227020800.0, 1307674368000.0, 355687428096000.0, 121645100408832000.0};
  const double fb[10] = {2.0, 24.0, 720.0, 40320.0, 3628800.0, 479001600.0, 87178291200.0, 20922789888000.0, 6402373705728000.0, 2432902008176640000.0};
  const double fc[10] = {6.0, 120.0, 5040.0, 362880.0, 39916800.0, 6227020800.0, 1307674368000.0, 355687428096000.0, 121645100408832000.0, 51090942171709440000.0};
  double A = 0, B = 0, C = 0;
  for (int k = 9; k >= 0; k--) {
    const double s = (k & 1) ? -1.0 : 1.0;
    A = A * t2 + s / fa[k];
    B = B * t2 + s / fb[k];
    C = C * t2 + s / fc[k];
  }
  const double Wm[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W2[3 * i + j] = (Wm[3 * i] * Wm[j] + Wm[3 * i + 1] * Wm[3 + j]) + Wm[3 * i + 2] * Wm[6 + j];
  for (int i = 0; i < 3; i++) {
    double t = 0;
    for (int j = 0; j < 3; j++) {
      const double id = i == j ? 1.0 : 0.0;
      U[4 * i + j] = (id + A * Wm[3 * i + j]) + B * W2[3 * i + j];
      const double V = (id + B * Wm[3 * i + j]) + C * W2[3 * i + j];
      t = t + V * xi[j];
    }
    U[4 * i + 3] = t;
  }
}

// icp.cpp:58-100 on the device: sums -> 6x6 solve -> exp -> accumulated update
__global__ __launch_bounds__(64) void icp_solve_kernel(
    const float* __restrict__ partial, int nblocks, double* __restrict__ state, float* __restrict__ stats, int it, int iterations)
{
  __shared__ double S[ICP_NSUM];
  const int n = blockIdx.x, t = threadIdx.x;
  if (t < ICP_NSUM) {
    const float* src = partial + (long long)n * nblocks * ICP_NSUM + t;
    double acc = 0.0;
    for (int b = 0; b < nblocks; b++) acc = acc + (double)src[(long long)b * ICP_NSUM];
    S[t] = acc;
  }
  __syncthreads();
  if (t != 0) return;
  if (stats) {
    stats[((long long)n * iterations + it) * 2 + 0] = (float)S[27];
    stats[((long long)n * iterations + it) * 2 + 1] = (float)S[28];
  }
  double A[6][6], b[6], L[6][6], d[6], y[6], x[6];
  int q = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) { A[i][j] = A[j][i] = S[q]; q++; }
  for (int i = 0; i < 6; i++) b[i] = S[21 + i];
  // LDL^T without pivoting; a pivot below 1e-10 of the largest diagonal entry (an unconstrained direction, or no
  // inlier at all) is dropped: that variable stays 0 instead of the solve blowing up
  // (Eigen's pivoted LDLT of the reference does the equivalent for a rank-deficient system)
  double maxdiag = 0.0;
  int skip[6];
  for (int i = 0; i < 6; i++) if (A[i][i] > maxdiag) maxdiag = A[i][i];
  const double tol = 1e-10 * maxdiag;
  for (int j = 0; j < 6; j++) {
    double dj = A[j][j];
    for (int k = 0; k < j; k++) dj = dj - (L[j][k] * L[j][k]) * d[k];
    skip[j] = !(dj > tol) || !(dj < 1e300);
    d[j] = skip[j] ? 0.0 : dj;
    for (int i = j + 1; i < 6; i++) {
      double v = A[i][j];
      for (int k = 0; k < j; k++) v = v - (L[i][k] * L[j][k]) * d[k];
      L[i][j] = skip[j] ? 0.0 : v / dj;
    }
  }
  for (int i = 0; i < 6; i++) { double v = b[i]; for (int k = 0; k < i; k++) v = v - L[i][k] * y[k]; y[i] = v; }
  for (int i = 0; i < 6; i++) y[i] = skip[i] ? 0.0 : y[i] / d[i];
  for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v = v - L[k][i] * x[k]; x[i] = v; }
  double U[12], Nn[12];
  double* T = state + 12 * n;
  icp_exp_se3(x, U);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Nn[4 * i + j] = (U[4 * i] * T[j] + U[4 * i + 1] * T[4 + j]) + U[4 * i + 2] * T[8 + j];
    Nn[4 * i + 3] = ((U[4 * i] * T[3] + U[4 * i + 1] * T[7]) + U[4 * i + 2] * T[11]) + U[4 * i + 3];
  }
  for (int i = 0; i < 12; i++) T[i] = Nn[i];
}

size_t icp_ws_bytes(int N, int H, int W)
{
  const long long nblocks = ((long long)H * W + ICP_BLOCK - 1) / ICP_BLOCK;
  return align_up(sizeof(float) * (size_t)N * nblocks * ICP_NSUM, 256);
}

}  // namespace

extern "C" int pcnn_icp_backproject_fwd(const uint16_t* depth, const int32_t* label, int height, int width, int obj_id,
                                        float factor_depth, float fx, float fy, float px, float py, float* vertex_map,
                                        void* stream_)
{
  PCNN_REQUIRE(height >= 1 && width >= 1, PCNN_EINVAL, "icp_backproject: bad shape %dx%d", height, width);
  PCNN_REQUIRE(factor_depth > 0 && fx != 0 && fy != 0, PCNN_EINVAL, "icp_backproject: factor_depth must be positive, focal lengths non-zero");
  PCNN_REQUIRE(depth && vertex_map, PCNN_ENULL, "icp_backproject: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const long long P = (long long)height * width;
  const unsigned grid = (unsigned)std::min<long long>((P + 255) / 256, 4096);
  PCNN_LAUNCH(icp_backproject_kernel, dim3(grid), dim3(256), 0, stream, depth, label, P, width, obj_id, factor_depth, fx, fy, px, py, vertex_map);
  return check_launch("icp_backproject_fwd");
}

extern "C" int pcnn_icp_refine_workspace_bytes(int num_objects, int height, int width, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "icp_refine_workspace_bytes: NULL output");
  PCNN_REQUIRE(num_objects >= 0 && height >= 1 && width >= 1, PCNN_EINVAL, "icp_refine_workspace_bytes: bad shape");
  *bytes = icp_ws_bytes(num_objects, height, width);
  return PCNN_OK;
}

extern "C" int pcnn_icp_refine_fwd(const float* live_vertices, const float* pred_vertices, const float* pred_normals,
                                   int num_objects, int height, int width, int pred_channels, float fx, float fy, float px,
                                   float py, float z_near, float z_far, float max_error, int iterations, double* update,
                                   float* stats, void* workspace, size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(num_objects >= 0 && height >= 8 && width >= 8, PCNN_EINVAL, "icp_refine: bad shape N=%d %dx%d", num_objects, height, width);
  PCNN_REQUIRE(pred_channels == 3 || pred_channels == 4, PCNN_EINVAL, "icp_refine: predicted maps carry 3 or 4 floats per pixel (got %d)", pred_channels);
  PCNN_REQUIRE(iterations >= 0 && max_error >= 0, PCNN_EINVAL, "icp_refine: iterations and max_error must be non-negative");
  PCNN_REQUIRE(num_objects <= 65535, PCNN_EINVAL, "icp_refine: at most 65535 objects per call");
  if (num_objects == 0) return PCNN_OK;
  PCNN_REQUIRE(live_vertices && pred_vertices && pred_normals && update, PCNN_ENULL, "icp_refine: NULL pointer");
  PCNN_REQUIRE(workspace && workspace_bytes >= icp_ws_bytes(num_objects, height, width), PCNN_EWORKSPACE,
               "icp_refine: workspace too small (%zu < %zu)", workspace_bytes, icp_ws_bytes(num_objects, height, width));
  hipStream_t stream = (hipStream_t)stream_;
  const long long P = (long long)height * width;
  const int nblocks = (int)((P + ICP_BLOCK - 1) / ICP_BLOCK);
  float* partial = static_cast<float*>(workspace);
  PCNN_LAUNCH(icp_init_kernel, dim3((12 * num_objects + 63) / 64), dim3(64), 0, stream, update, num_objects);
  for (int it = 0; it < iterations; it++) {
    PCNN_LAUNCH(icp_terms_kernel, dim3(nblocks, num_objects), dim3(ICP_BLOCK), 0, stream, live_vertices, pred_vertices, pred_normals,
                update, P, height, width, pred_channels, fx, fy, px, py, z_near, z_far, max_error, partial, nblocks);
    PCNN_LAUNCH(icp_solve_kernel, dim3(num_objects), dim3(64), 0, stream, partial, nblocks, update, stats, it, iterations);
  }
  return check_launch("icp_refine_fwd");
}
