// icp.hip — depth-based pose refinement, first slice: the projective point-to-plane ICP core
// behind Synthesizer::refinePose / solveICP (lib/synthesize/synthesize.cpp:2020-2026, :2052-2380; called from
// lib/fcn/test.py:1925-1933), i.e. df::icp (lib/kinect_fusion/src/optimization/icp.cpp:20-106) with its per-pixel
// kernel (src/optimization/icp.cu:25-136) and the masked depth -> vertex map step (synthesize.cpp:2139-2155 +
// src/image/backprojection.cu:10-27), plus the two host-side steps of solveICP around the iterations — the translation
// estimate (synthesize.cpp:2157-2225) and the SegICP scoring of the refined hypotheses (:2302-2343, PCL kd-tree in the
// reference) — as kernels. The predicted maps come from csrc/render.hip (OpenGL in the reference); the nlopt polish
// (poseWithOpt) is not reproduced.
//
// The reference runs, per iteration and per object: one kernel writing a 28-byte Jacobian/residual record for EVERY
// pixel of the frame (8.6 MB), cudaDeviceSynchronize, a thrust::transform_reduce over all of them, a second
// synchronise, a 6x6 solve on the host, and an H2D of the new pose — 8 to 50 times per object.
// Here an iteration is two launches and nothing leaves the device until the last one is done:
//   icp_terms_kernel   one thread per pixel of the predicted maps, all N objects of a frame in one grid (grid.y);
//                      the 29 sums a pixel contributes to (21 J^T J, 6 J^T r, inlier count, sum r^2) are reduced
//                      in LDS by a halving tree per 256-pixel block (blocks without a single contributing pixel —
//                      ~95 % of a frame — skip the tree) into one partial row per block;
//   icp_solve_kernel   one workgroup per object: partial rows added in f64 in a fixed order (8 segments x 29 columns),
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

// The content of the Sophus::SE3f icpKernel receives (`updatedPose`): unit quaternion (w, x, y, z) + translation, from the
// accumulated 3x4 transform (f64 in the solve, rounded to f32 here): rotation matrix -> quaternion by Eigen's published
// algorithm (trace branch / largest-diagonal branch), then Sophus's normalisation — coefficients / sqrt(squaredNorm), the
// 4-term reduction in Eigen's unrolled order over (x, y, z, w) — in f32 with correctly rounded sqrt and division. The
// same statement as icp_se3f_from_matrix of the CPU checker, which is held to the reference's kernel body bit for bit.
__device__ __forceinline__ void icp_se3f_from_matrix(const float* T, float* q /* wxyz */, float* t)
{
  const float m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8], m21 = T[9], m22 = T[10];
  float qx, qy, qz, qw;
  float tr = (m00 + m11) + m22;
  if (tr > 0.f) {
    tr = sqrt_rn(tr + 1.0f);
    qw = 0.5f * tr;
    tr = div_rn(0.5f, tr);
    qx = (m21 - m12) * tr;
    qy = (m02 - m20) * tr;
    qz = (m10 - m01) * tr;
  } else {
    int i = 0;
    if (m11 > m00) i = 1;
    if (m22 > (i == 0 ? m00 : m11)) i = 2;
    if (i == 0) {          // (i, j, k) = (0, 1, 2)
      tr = sqrt_rn(((m00 - m11) - m22) + 1.0f);
      qx = 0.5f * tr; tr = div_rn(0.5f, tr);
      qw = (m21 - m12) * tr; qy = (m10 + m01) * tr; qz = (m20 + m02) * tr;
    } else if (i == 1) {   // (1, 2, 0)
      tr = sqrt_rn(((m11 - m22) - m00) + 1.0f);
      qy = 0.5f * tr; tr = div_rn(0.5f, tr);
      qw = (m02 - m20) * tr; qz = (m21 + m12) * tr; qx = (m01 + m10) * tr;
    } else {               // (2, 0, 1)
      tr = sqrt_rn(((m22 - m00) - m11) + 1.0f);
      qz = 0.5f * tr; tr = div_rn(0.5f, tr);
      qw = (m10 - m01) * tr; qx = (m02 + m20) * tr; qy = (m12 + m21) * tr;
    }
  }
  const float n = sqrt_rn((qx * qx + qy * qy) + (qz * qz + qw * qw));
  q[0] = div_rn(qw, n); q[1] = div_rn(qx, n); q[2] = div_rn(qy, n); q[3] = div_rn(qz, n);
  t[0] = T[3]; t[1] = T[7]; t[2] = T[11];
}

// Sophus::SE3f * point = so3 * p + translation; so3 * p = Eigen's Quaternion::_transformVector: uv = 2 (q.vec x p);
// (p + w uv) + q.vec x uv
__device__ __forceinline__ void icp_se3f_apply(const float* q, const float* t, float x, float y, float z, float& ox, float& oy, float& oz)
{
  const float w = q[0], a = q[1], b = q[2], c = q[3];
  float ux = b * z - c * y, uy = c * x - a * z, uz = a * y - b * x;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = b * uz - c * uy, cy = c * ux - a * uz, cz = a * uy - b * ux;
  ox = ((x + w * ux) + cx) + t[0];
  oy = ((y + w * uy) + cy) + t[1];
  oz = ((z + w * uz) + cz) + t[2];
}

// icpKernel (icp.cu:25-136) + the per-block part of the reduction: the body's own order of tests, Eigen's published
// evaluation order for the 3-term reductions (t0 + (t1 + t2)) — see oracle/ref_shim/eigen_sophus_on_cpu.h for the list
__global__ __launch_bounds__(ICP_BLOCK) void icp_terms_kernel(
    const float* __restrict__ live, const float* __restrict__ pred_v, const float* __restrict__ pred_n,
    const double* __restrict__ state, long long P, int H, int W, int pc, float fx, float fy, float px, float py,
    float znear, float zfar, float max_error, float* __restrict__ partial, int nblocks)
{
  __shared__ float red[ICP_NSUM][ICP_BLOCK];
  __shared__ int s_any;
  __shared__ float s_pose[7];
  const int n = blockIdx.y, t = threadIdx.x;
  const long long p = (long long)blockIdx.x * ICP_BLOCK + t;
  if (t == 0) {
    s_any = 0;
    // the SE3f content once per block (2 correctly rounded square roots + 5 divisions: per thread it doubled the kernel)
    float T[12], q0[4], t0[3];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = (float)state[12 * n + i];
    icp_se3f_from_matrix(T, q0, t0);
#pragma unroll
    for (int i = 0; i < 4; i++) s_pose[i] = q0[i];
#pragma unroll
    for (int i = 0; i < 3; i++) s_pose[4 + i] = t0[i];
  }
  __syncthreads();
  float q[4], tt[3];
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = s_pose[i];
#pragma unroll
  for (int i = 0; i < 3; i++) tt[i] = s_pose[4 + i];
  float J[6], r = 0.f;
  bool ok = false;
  if (p < P) {
    const float* pv = pred_v + ((long long)n * P + p) * pc;
    const float* pn = pred_n + ((long long)n * P + p) * pc;
    const float* lv0 = live + (long long)n * P * 3;
    const float border = 2.f, ray_norm_dot_threshold = 0.1f;
    const float pvx = pv[0], pvy = pv[1], pvz = pv[2];
    if (!((pvz < znear) || pvz > zfar)) {                                       // :60 (a NaN passes, as in the body)
      float ux, uy, uz;
      icp_se3f_apply(q, tt, pvx, pvy, pvz, ux, uy, uz);                         // :67
      // :69 Poly3CameraModel::project with k = 0: the distortion factor ((1 + 0 r2) + 0 r4) + 0 r6 is exactly 1 for finite r2
      const float dhx = div_rn(ux, uz), dhy = div_rn(uy, uz);
      const float r2 = dhx * dhx + dhy * dhy, r4 = r2 * r2, r6 = r4 * r2;
      const float factor = ((1.f + 0.f * r2) + 0.f * r4) + 0.f * r6;
      const float projx = (factor * dhx) * fx + px, projy = (factor * dhy) * fy + py;
      const int u = (int)(projx + 0.5f), v = (int)(projy + 0.5f);               // :78-79 (v_cvt_i32_f32: NaN -> 0, saturating)
      if (!(((float)u <= border) || ((float)u >= (float)(unsigned)(W - 1) - border) || ((float)v <= border) || ((float)v >= (float)(unsigned)(H - 1) - border))) {
        const float* lv = lv0 + 3 * ((long long)v * W + u);
        const float lx = lv[0], ly = lv[1], ldepth = lv[2];
        if (!((ldepth < znear) || (ldepth > zfar))) {                           // :92
          const float sq = ux * ux + (uy * uy + uz * uz);
          float rx = ux, ry = uy, rz = uz;
          if (sq > 0.f) { const float nrm = sqrt_rn(sq); rx = div_rn(ux, nrm); ry = div_rn(uy, nrm); rz = div_rn(uz, nrm); }   // :100
          const float nx = pn[0], ny = pn[1], nz = pn[2];
          const float dotrn = rx * nx + (ry * ny + rz * nz);
          if (!(-dotrn < ray_norm_dot_threshold)) {                             // :104
            const float ex = lx - ux, ey = ly - uy, ez = ldepth - uz;
            const float error = nx * ex + (ny * ey + nz * ez);                  // :111
            if (!(fabsf(error) > max_error)) {                                  // :115
              const float w = div_rn(1.f, ldepth);
              const float wx = w * nx, wy = w * ny, wz = w * nz;
              J[0] = wx * 1.f + (wy * 0.f + wz * 0.f);
              J[1] = wx * 0.f + (wy * 1.f + wz * 0.f);
              J[2] = wx * 0.f + (wy * 0.f + wz * 1.f);
              J[3] = wx * 0.f + (wy * (-uz) + wz * uy);
              J[4] = wx * uz + (wy * 0.f + wz * (-ux));
              J[5] = wx * (-uy) + (wy * ux + wz * 0.f);
              r = w * error;
              ok = true;
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

// The canonical sum of the per-block partial rows (f32) in f64: ICP_NSEG contiguous segments of ceil(nblocks / ICP_NSEG)
// blocks, each added up in ascending block order, then the segment sums in ascending order. (One lane walking all 1200
// rows of a 480x640 frame one dependent load at a time cost 267 us per ICP iteration — 2/3 of the whole refinement;
// 8 segments x 8 loads in flight: a few us.)   256 threads: thread = (segment t >> 5, column t & 31), NQ <= 32 columns.
constexpr int ICP_NSEG = 8;

template <int NQ>
__device__ __forceinline__ void icp_segmented_sums(const float* __restrict__ rows, int nblocks, double* S /* shared [NQ] */,
                                                   double (*seg)[32] /* shared [ICP_NSEG][32] */)
{
  const int t = threadIdx.x, q = t & 31, sg = t >> 5;
  const int L = (nblocks + ICP_NSEG - 1) / ICP_NSEG;
  const int b0 = sg * L, b1 = min(nblocks, b0 + L);
  double acc = 0.0;
  if (q < NQ) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = rows[(long long)(b + u) * NQ + q];
#pragma unroll
      for (int u = 0; u < 8; u++) acc = acc + (double)v[u];
    }
    for (; b < b1; b++) acc = acc + (double)rows[(long long)b * NQ + q];
  }
  seg[sg][q] = acc;
  __syncthreads();
  if (t < NQ) {
    double s = seg[0][t];
#pragma unroll
    for (int k = 1; k < ICP_NSEG; k++) s = s + seg[k][t];
    S[t] = s;
  }
  __syncthreads();
}

// icp.cpp:58-100 on the device: sums -> 6x6 solve -> exp -> accumulated update
__global__ __launch_bounds__(256) void icp_solve_kernel(
    const float* __restrict__ partial, int nblocks, double* __restrict__ state, float* __restrict__ stats, int it, int iterations)
{
  __shared__ double S[ICP_NSUM];
  __shared__ double seg[ICP_NSEG][32];
  const int n = blockIdx.x, t = threadIdx.x;
  icp_segmented_sums<ICP_NSUM>(partial + (long long)n * nblocks * ICP_NSUM, nblocks, S, seg);
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

// ---- solveICP around the iterations: translation estimate and hypothesis selection (synthesize.cpp:2157-2262, :2302-2343) ----
constexpr int ICP_NCEN = 5;   // sum (d - m).x, .y, .z over the pixels that agree with the render, their count, valid pixels

// synthesize.cpp:2157-2207: pixels of the object with a depth reading and a rendered canonical vertex are the
// (depth point, model point) pairs of the rest of solveICP (mask = 1); those whose depth point lies within max_error of
// the rendered surface along its normal vote for the translation with (depth point - model point).
__global__ __launch_bounds__(ICP_BLOCK) void icp_center_kernel(
    const int* __restrict__ label, const float* __restrict__ live, const float* __restrict__ canon,
    const float* __restrict__ pred_v, const float* __restrict__ pred_n, int pc, long long P, int obj_id, float max_error,
    unsigned char* __restrict__ mask, float* __restrict__ partial)
{
  __shared__ float red[ICP_NCEN][ICP_BLOCK];
  const int t = threadIdx.x;
  const long long p = (long long)blockIdx.x * ICP_BLOCK + t;
  float c[ICP_NCEN] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (p < P) {
    bool valid = false;
    if (label[p] == obj_id) {
      const float dx = live[3 * p], dy = live[3 * p + 1], dz = live[3 * p + 2];
      if (dz > 0.f) {
        const float cx = canon[3 * p], vy = canon[3 * p + 1], vz = canon[3 * p + 2];
        const float vx = cx - roundf(cx);        // the model index rides in the integer part of x (synthesize.cpp:270-271, :2173)
        if (vx == vx && vy == vy && vz == vz) {
          valid = true;
          const float* pv = pred_v + p * pc;
          const float* pn = pred_n + p * pc;
          const float error = (pn[0] * (dx - pv[0]) + pn[1] * (dy - pv[1])) + pn[2] * (dz - pv[2]);
          if (fabsf(error) < max_error) {
            c[0] = dx - vx; c[1] = dy - vy; c[2] = dz - vz; c[3] = 1.f;
          }
          c[4] = 1.f;
        }
      }
    }
    mask[p] = valid ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < ICP_NCEN; q++) red[q][t] = c[q];
  __syncthreads();
  for (int s = ICP_BLOCK / 2; s >= 1; s >>= 1) {
    if (t < s) {
#pragma unroll
      for (int q = 0; q < ICP_NCEN; q++) red[q][t] = red[q][t] + red[q][t + s];
    }
    __syncthreads();
  }
  if (t < ICP_NCEN) partial[(long long)blockIdx.x * ICP_NCEN + t] = red[t][0];
}

__global__ __launch_bounds__(256) void icp_center_sum_kernel(const float* __restrict__ partial, int nblocks, double* __restrict__ sums)
{
  __shared__ double S[ICP_NCEN];
  __shared__ double seg[ICP_NSEG][32];
  icp_segmented_sums<ICP_NCEN>(partial, nblocks, S, seg);
  if (threadIdx.x < ICP_NCEN) sums[threadIdx.x] = S[threadIdx.x];
}

// synthesize.cpp:2302-2343 (the SegICP score): every model point, moved by the hypothesis, looks for its NEAREST depth
// point within `radius` (pcl::KdTreeFLANN::radiusSearch returns its hits sorted by distance; [0] is used) and marks it;
// the score is the number of distinct marked depth points. The reference builds a kd-tree per object on the host and
// marks from an OpenMP loop (a race on flags[]); the distinct count does not depend on the order. Here the depth points
// are still in image order, so the search is a window around the projection of the moved model point: a point within
// r of (X, Y, Z) projects within fx r (1 + |X/Z|) / (Z - r) pixels of it — an exact search, ties to the lower pixel index.
// At 0.7 m a 1 cm radius is a 41 x 41 pixel window (1 681 candidates; the first version scanned all of them: 0.5 ms per
// object); but the nearest point is almost always a pixel or two from the projection. So the window is shrunk first: a
// 5 x 5, then a 13 x 13 probe around the projection yields an upper bound b on the nearest squared distance, and every
// point at least that close lies in the window of radius sqrt(b) — typically 7 x 7. The final scan runs over that window
// in raster order from scratch (strict <, so ties still go to the lower pixel index): same result as the exhaustive search.
struct IcpWin { int x0, x1, y0, y1; };

__device__ __forceinline__ IcpWin icp_window(float qx, float qy, float qz, float rad, float fx, float fy, float px, float py, int H, int W,
                                             bool& none)
{
  IcpWin w = {0, W - 1, 0, H - 1};
  none = false;
  if (qz > 2.f * rad) {
    const float uc = qx / qz * fx + px, vc = qy / qz * fy + py;
    const float hw = fabsf(fx) * rad * (1.f + fabsf(qx / qz)) / (qz - rad) + 2.f;
    const float hh = fabsf(fy) * rad * (1.f + fabsf(qy / qz)) / (qz - rad) + 2.f;
    if (!(fabsf(uc) < 1e8f) || !(fabsf(vc) < 1e8f) || !(hw < 1e8f) || !(hh < 1e8f)) { none = true; return w; }   // nowhere near the image
    w.x0 = max(0, (int)floorf(uc - hw)); w.x1 = min(W - 1, (int)ceilf(uc + hw));
    w.y0 = max(0, (int)floorf(vc - hh)); w.y1 = min(H - 1, (int)ceilf(vc + hh));
  }
  return w;
}

// nearest masked depth point inside the window with d2 < best (strict), raster order; |dz| >= radius cannot qualify
// (then d2 >= dz^2 >= radius^2 also in rounded arithmetic: squares and sums of non-negatives are monotone)
constexpr int ICP_CH = 8;
__device__ __forceinline__ void icp_scan(const float* __restrict__ live, const unsigned char* __restrict__ mask, int W, IcpWin w,
                                         float qx, float qy, float qz, float radius, float& best, long long& bi)
{
  // ICP_CH candidates of a row at a time: their 4 ICP_CH loads are issued together, unconditionally (one candidate per trip
  // with its dependent loads — mask, z, then x and y, L2 hits each — made a thread with a full 41 x 41 window take 0.4 ms).
  // Measured alternatives: 16 per trip, or x / y fetched only after the depth test: both 6 % slower.
  for (int y = w.y0; y <= w.y1; y++) {
    const long long base = (long long)y * W;
    for (int x = w.x0; x <= w.x1; x += ICP_CH) {
      unsigned char mk[ICP_CH];
      float vx[ICP_CH], vy[ICP_CH], vz[ICP_CH];
#pragma unroll
      for (int u = 0; u < ICP_CH; u++) {
        const long long i = base + min(x + u, w.x1);
        mk[u] = mask[i];
        vx[u] = live[3 * i]; vy[u] = live[3 * i + 1]; vz[u] = live[3 * i + 2];
      }
#pragma unroll
      for (int u = 0; u < ICP_CH; u++) {
        if (x + u > w.x1 || !mk[u]) continue;
        const float ez = vz[u] - qz;
        if (!(fabsf(ez) < radius)) continue;
        const float ex = vx[u] - qx, ey = vy[u] - qy;
        const float d2 = (ex * ex + ey * ey) + ez * ez;
        if (d2 < best) { best = d2; bi = base + x + u; }
      }
    }
  }
}

__global__ __launch_bounds__(256) void icp_score_kernel(
    const float* __restrict__ live, const float* __restrict__ canon, const unsigned char* __restrict__ mask, int H, int W,
    const float* __restrict__ hyps, float fx, float fy, float px, float py, float radius, unsigned* __restrict__ flags,
    int nwords, int* __restrict__ hits)
{
  const long long P = (long long)H * W;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  if (p >= P || !mask[p]) return;
  const float* T = hyps + 12 * (size_t)m;
  const float cx = canon[3 * p];
  const float mx = cx - roundf(cx), my = canon[3 * p + 1], mz = canon[3 * p + 2];
  const float qx = ((T[0] * mx + T[1] * my) + T[2] * mz) + T[3];
  const float qy = ((T[4] * mx + T[5] * my) + T[6] * mz) + T[7];
  const float qz = ((T[8] * mx + T[9] * my) + T[10] * mz) + T[11];
  if (!(qx == qx) || !(qy == qy) || !(qz == qz)) return;
  bool none;
  IcpWin full = icp_window(qx, qy, qz, radius, fx, fy, px, py, H, W, none);
  if (none) return;
  const float r2 = radius * radius;
  if (qz > 2.f * radius) {
    // probes around the projection: an upper bound on the nearest squared distance, if anything is close at all
    const int uc = (int)rintf(qx / qz * fx + px), vc = (int)rintf(qy / qz * fy + py);
    float bound = r2;
    long long dummy = -1;
#pragma unroll
    for (int half = 2; half <= 6; half += 4) {
      if (bound < r2) break;
      IcpWin pw = {max(full.x0, uc - half), min(full.x1, uc + half), max(full.y0, vc - half), min(full.y1, vc + half)};
      icp_scan(live, mask, W, pw, qx, qy, qz, radius, bound, dummy);
    }
    if (bound < r2) {
      const float rb = sqrt_rn(bound) * 1.0001f + 1e-7f;      // >= the distance of the probe's best point
      bool n2;
      const IcpWin sw = icp_window(qx, qy, qz, fminf(rb, radius), fx, fy, px, py, H, W, n2);
      if (!n2) { full.x0 = max(full.x0, sw.x0); full.x1 = min(full.x1, sw.x1); full.y0 = max(full.y0, sw.y0); full.y1 = min(full.y1, sw.y1); }
    }
  }
  float best = r2;
  long long bi = -1;
  icp_scan(live, mask, W, full, qx, qy, qz, radius, best, bi);
  if (bi < 0) return;
  const unsigned bit = 1u << (bi & 31);
  const unsigned old = atomicOr(&flags[(size_t)m * nwords + (bi >> 5)], bit);
  if (!(old & bit)) atomicAdd(&hits[m], 1);
}

// ---- solveICP's polish (Synthesizer::poseWithOpt, synthesize.cpp:2529-2570; objective optEnergy :2476-2526) ---------------
// nlopt's Nelder-Mead over the 7 numbers of an update pose (quaternion wxyz + translation, box +-0.1 / +-0.01 / +-0.1 around
// the identity, maxeval = 50), minimising the mean distance between the moved predicted vertex and the depth point of the
// same pixel over the object's label pixels. In the reference every one of the 50 evaluations is a host loop over the label
// pixels. Here the whole optimisation is ONE launch of one 1024-thread workgroup: the simplex lives in LDS, an evaluation is a
// sweep of the label's bounding box (round-robin over the threads, halving tree) and nothing goes back to the host in between.
// The algorithm is the published one of nlopt's nldrmd.c (Box's bound handling, alpha 1, beta 0.5, gamma 2, delta 0.5,
// default initial step (ub - lb) / 4), restated — not its code, not its bits; the CPU checker restates the same statement
// and the two agree bit for bit.
constexpr int NM_N = 7;
constexpr int NM_LANES = 1024;

struct NmShared {
  double P[NM_N + 1][NM_N], c[NM_N], xr[NM_N], xe[NM_N], lb[NM_N], ub[NM_N];
  double f[NM_N + 1];        // (in LDS, not in registers: f[lo] / f[hi] are dynamically indexed — a private array went to scratch)
  float T[12];
  float part[NM_LANES];
  int cnt[NM_LANES];
  int box[4];
};

// optEnergy for the update x (LDS, 7 doubles): every thread returns the same value
__device__ float nm_energy(NmShared& sh, const double* x, const int* __restrict__ label, const float* __restrict__ live,
                           const float* __restrict__ pred_v, int pc, int W, int obj, float znear, float zfar)
{
  const int t = threadIdx.x;
  __syncthreads();                       // x is complete, the previous tree is consumed
  if (t == 0) {
    // the SE3f optEnergy builds (:2481-2484): Quaternionf(pose[0..3]) in f32, normalised by Sophus's constructor
    // (coefficients / sqrt(squaredNorm), 4-term reduction over (x, y, z, w) in Eigen's unrolled order); sh.T = (q wxyz, t)
    const float w = (float)x[0], a = (float)x[1], b = (float)x[2], c = (float)x[3];
    const float n = sqrt_rn((a * a + b * b) + (c * c + w * w));
    sh.T[0] = div_rn(w, n); sh.T[1] = div_rn(a, n); sh.T[2] = div_rn(b, n); sh.T[3] = div_rn(c, n);
    sh.T[4] = (float)x[4]; sh.T[5] = (float)x[5]; sh.T[6] = (float)x[6];
  }
  __syncthreads();
  float q[4], tt[3];
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = sh.T[i];
#pragma unroll
  for (int i = 0; i < 3; i++) tt[i] = sh.T[4 + i];
  const int bw = sh.box[1] - sh.box[0] + 1, bh = sh.box[3] - sh.box[2] + 1;
  float acc = 0.f;
  int c = 0;
  // a thread's pixels are j = t, t + 1024, ... of the box in raster order: (row, column) advance by (1024 / bw, 1024 % bw)
  // with a carry — no division per pixel (a 64-bit divide + modulo per pixel made an evaluation 18 us). 4 pixels per trip
  // with all their loads in flight together: with a single workgroup on the chip every dependent load is an L2 round trip.
  const int dq = NM_LANES / bw, dr = NM_LANES % bw;
  int yy = t / bw, xx = t % bw;
  constexpr int NB = 4;
  while (yy < bh) {
    int lab[NB], live_ok[NB];
    float pvv[NB][3], lv[NB][3];
#pragma unroll
    for (int u = 0; u < NB; u++) {
      live_ok[u] = yy < bh;
      const int yc = live_ok[u] ? yy : bh - 1;
      const long long p = (long long)(sh.box[2] + yc) * W + (sh.box[0] + xx);
      lab[u] = label[p];
      const float* pv = pred_v + p * pc;
      pvv[u][0] = pv[0]; pvv[u][1] = pv[1]; pvv[u][2] = pv[2];
      lv[u][0] = live[3 * p]; lv[u][1] = live[3 * p + 1]; lv[u][2] = live[3 * p + 2];
      xx += dr; yy += dq;
      if (xx >= bw) { xx -= bw; yy++; }
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
      if (!live_ok[u] || lab[u] != obj) continue;
      const float p0 = pvv[u][0], p1 = pvv[u][1], p2 = pvv[u][2];
      float qx, qy, qz;
      icp_se3f_apply(q, tt, p0, p1, p2, qx, qy, qz);                      // T_co * point, :2505
      const float vx = lv[u][0], vy = lv[u][1], vz = lv[u][2];
      if (qx == qx && qy == qy && qz == qz && vz > znear && vz < zfar && qz > znear && qz < zfar) {
        const float ex = qx - vx, ey = qy - vy, ez = qz - vz;
        acc = acc + sqrt_rn((ex * ex + ey * ey) + ez * ez);                 // :2517
        c++;
      }
    }
  }
  sh.part[t] = acc;
  sh.cnt[t] = c;
  __syncthreads();
  for (int s = NM_LANES / 2; s >= 1; s >>= 1) {
    if (t < s) { sh.part[t] = sh.part[t] + sh.part[t + s]; sh.cnt[t] = sh.cnt[t] + sh.cnt[t + s]; }
    __syncthreads();
  }
  const float e = sh.cnt[0] ? sh.part[0] / (float)sh.cnt[0] : 0.f;
  return e;
}

__global__ __launch_bounds__(NM_LANES) void icp_polish_kernel(
    const int* __restrict__ label, const float* __restrict__ live, const float* __restrict__ pred_v, int pc, int H, int W,
    int obj, float znear, float zfar, int maxeval, double* __restrict__ x_out, double* __restrict__ info)
{
  __shared__ NmShared sh;
  const int t = threadIdx.x;
  // bounding box of the object's label pixels
  if (t == 0) { sh.box[0] = W; sh.box[1] = -1; sh.box[2] = H; sh.box[3] = -1; }
  __syncthreads();
  {
    int x0 = W, x1 = -1, y0 = H, y1 = -1;
    const long long P = (long long)H * W;
    for (long long p = t; p < P; p += NM_LANES)
      if (label[p] == obj) {
        const int x = (int)(p % W), y = (int)(p / W);
        x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
      }
    if (x1 >= 0) { atomicMin(&sh.box[0], x0); atomicMax(&sh.box[1], x1); atomicMin(&sh.box[2], y0); atomicMax(&sh.box[3], y1); }
  }
  __syncthreads();
  const double x0v[NM_N] = {1, 0, 0, 0, 0, 0, 0};
  const double range[NM_N] = {0.1, 0.1, 0.1, 0.1, 0.01, 0.01, 0.1};
  if (sh.box[1] < sh.box[0]) {
    if (t < NM_N) x_out[t] = x0v[t];
    if (t == 0) { info[0] = 0.0; info[1] = 0.0; }
    return;
  }
  if (t == 0)
    for (int i = 0; i < NM_N; i++) { sh.lb[i] = x0v[i] - range[i]; sh.ub[i] = x0v[i] + range[i]; }
  __syncthreads();
  int evals = 0;
  double* f = sh.f;      // written by thread 0 only; every reader sits behind a barrier
#define NM_EVAL(X) ((double)nm_energy(sh, (X), label, live, pred_v, pc, W, obj, znear, zfar))
#define NM_SETF(K, V) do { if (t == 0) f[(K)] = (V); } while (0)
  for (int k = 0; k <= NM_N; k++) {
    if (t == 0) {
      for (int i = 0; i < NM_N; i++) sh.P[k][i] = x0v[i];
      if (k > 0) sh.P[k][k - 1] = x0v[k - 1] + (sh.ub[k - 1] - sh.lb[k - 1]) * 0.25;
    }
    const double fk = NM_EVAL(sh.P[k]);
    NM_SETF(k, fk);
    evals++;
  }
  while (evals < maxeval) {
    __syncthreads();
    int lo = 0, hi = 0, nh = -1;
    for (int k = 1; k <= NM_N; k++) {
      if (f[k] < f[lo]) lo = k;
      if (f[k] >= f[hi]) hi = k;
    }
    for (int k = 0; k <= NM_N; k++)
      if (k != hi && (nh < 0 || f[k] >= f[nh])) nh = k;
    const double flo = f[lo], fhi = f[hi], fnh = f[nh];
    __syncthreads();
    if (t == 0)
      for (int i = 0; i < NM_N; i++) {
        double sacc = 0.0;
        for (int k = 0; k <= NM_N; k++)
          if (k != hi) sacc = sacc + sh.P[k][i];
        sh.c[i] = sacc / (double)NM_N;
      }
#define NM_POINT(DST, COEF)                                               \
    if (t == 0)                                                           \
      for (int i = 0; i < NM_N; i++) {                                    \
        double v_ = sh.c[i] + (COEF) * (sh.c[i] - sh.P[hi][i]);           \
        if (v_ < sh.lb[i]) v_ = sh.lb[i];                                 \
        if (v_ > sh.ub[i]) v_ = sh.ub[i];                                 \
        (DST)[i] = v_;                                                    \
      }
#define NM_TAKE(SRC, FV)                                                  \
    do {                                                                  \
      if (t == 0) { for (int i = 0; i < NM_N; i++) sh.P[hi][i] = (SRC)[i]; f[hi] = (FV); } \
    } while (0)
    NM_POINT(sh.xr, 1.0);
    const double fr = NM_EVAL(sh.xr);
    evals++;
    if (fr < flo) {
      if (evals < maxeval) {
        NM_POINT(sh.xe, 2.0);
        const double fe = NM_EVAL(sh.xe);
        evals++;
        if (fe < fr) NM_TAKE(sh.xe, fe); else NM_TAKE(sh.xr, fr);
      } else {
        NM_TAKE(sh.xr, fr);
      }
    } else if (fr < fnh) {
      NM_TAKE(sh.xr, fr);
    } else {
      if (evals >= maxeval) { if (fr < fhi) NM_TAKE(sh.xr, fr); break; }
      const double coef = fr < fhi ? 0.5 : -0.5;
      NM_POINT(sh.xe, coef);
      const double fc = NM_EVAL(sh.xe);
      evals++;
      const double fref = fr < fhi ? fr : fhi;
      if (fc < fref) {
        NM_TAKE(sh.xe, fc);
      } else {
        for (int k = 0; k <= NM_N && evals < maxeval; k++) {
          if (k == lo) continue;
          __syncthreads();
          if (t == 0)
            for (int i = 0; i < NM_N; i++) sh.P[k][i] = sh.P[lo][i] + 0.5 * (sh.P[k][i] - sh.P[lo][i]);
          const double fk = NM_EVAL(sh.P[k]);
          NM_SETF(k, fk);
          evals++;
        }
      }
    }
  }
#undef NM_SETF
#undef NM_EVAL
#undef NM_POINT
#undef NM_TAKE
  __syncthreads();
  int lo = 0;
  for (int k = 1; k <= NM_N; k++)
    if (f[k] < f[lo]) lo = k;
  if (t < NM_N) x_out[t] = sh.P[lo][t];
  if (t == 0) { info[0] = f[lo]; info[1] = (double)evals; }
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
    PCNN_LAUNCH(icp_solve_kernel, dim3(num_objects), dim3(256), 0, stream, partial, nblocks, update, stats, it, iterations);
  }
  return check_launch("icp_refine_fwd");
}

extern "C" int pcnn_icp_center_workspace_bytes(int height, int width, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "icp_center_workspace_bytes: NULL output");
  PCNN_REQUIRE(height >= 1 && width >= 1, PCNN_EINVAL, "icp_center_workspace_bytes: bad shape");
  *bytes = align_up(sizeof(float) * (size_t)(((long long)height * width + ICP_BLOCK - 1) / ICP_BLOCK) * ICP_NCEN, 256);
  return PCNN_OK;
}

extern "C" int pcnn_icp_center_fwd(const int32_t* label, const float* live_vertices, const float* canonical,
                                   const float* pred_vertices, const float* pred_normals, int pred_channels, int height,
                                   int width, int obj_id, float max_error, double* sums, uint8_t* mask, void* workspace,
                                   size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(height >= 1 && width >= 1, PCNN_EINVAL, "icp_center: bad shape %dx%d", height, width);
  PCNN_REQUIRE(pred_channels == 3 || pred_channels == 4, PCNN_EINVAL, "icp_center: predicted maps carry 3 or 4 floats per pixel (got %d)", pred_channels);
  PCNN_REQUIRE(label && live_vertices && canonical && pred_vertices && pred_normals && sums && mask && workspace, PCNN_ENULL, "icp_center: NULL pointer");
  size_t need = 0;
  pcnn_icp_center_workspace_bytes(height, width, &need);
  PCNN_REQUIRE(workspace_bytes >= need, PCNN_EWORKSPACE, "icp_center: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t stream = (hipStream_t)stream_;
  const long long P = (long long)height * width;
  const int nblocks = (int)((P + ICP_BLOCK - 1) / ICP_BLOCK);
  float* partial = static_cast<float*>(workspace);
  PCNN_LAUNCH(icp_center_kernel, dim3(nblocks), dim3(ICP_BLOCK), 0, stream, label, live_vertices, canonical, pred_vertices,
              pred_normals, pred_channels, P, obj_id, max_error, mask, partial);
  PCNN_LAUNCH(icp_center_sum_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, sums);
  return check_launch("icp_center_fwd");
}

extern "C" int pcnn_icp_score_workspace_bytes(int num_hypotheses, int height, int width, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "icp_score_workspace_bytes: NULL output");
  PCNN_REQUIRE(num_hypotheses >= 0 && height >= 1 && width >= 1, PCNN_EINVAL, "icp_score_workspace_bytes: bad shape");
  *bytes = sizeof(unsigned) * (size_t)num_hypotheses * (size_t)(((long long)height * width + 31) / 32);
  return PCNN_OK;
}

extern "C" int pcnn_icp_score_fwd(const float* live_vertices, const float* canonical, const uint8_t* mask, int height, int width,
                                  const float* hypotheses, int num_hypotheses, float fx, float fy, float px, float py,
                                  float radius, int32_t* hits, void* workspace, size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(height >= 1 && width >= 1 && num_hypotheses >= 0, PCNN_EINVAL, "icp_score: bad shape %dx%d, %d hypotheses", height, width, num_hypotheses);
  PCNN_REQUIRE(num_hypotheses <= 65535, PCNN_EINVAL, "icp_score: at most 65535 hypotheses per call");
  PCNN_REQUIRE(radius > 0 && fx != 0 && fy != 0, PCNN_EINVAL, "icp_score: radius must be positive, focal lengths non-zero");
  if (num_hypotheses == 0) return PCNN_OK;
  PCNN_REQUIRE(live_vertices && canonical && mask && hypotheses && hits && workspace, PCNN_ENULL, "icp_score: NULL pointer");
  size_t need = 0;
  pcnn_icp_score_workspace_bytes(num_hypotheses, height, width, &need);
  PCNN_REQUIRE(workspace_bytes >= need, PCNN_EWORKSPACE, "icp_score: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t stream = (hipStream_t)stream_;
  const long long P = (long long)height * width;
  const int nwords = (int)((P + 31) / 32);
  int st = zero_async(workspace, need, stream, "icp_score");
  if (st == PCNN_OK) st = zero_async(hits, sizeof(int32_t) * (size_t)num_hypotheses, stream, "icp_score");
  if (st != PCNN_OK) return st;
  PCNN_LAUNCH(icp_score_kernel, dim3((unsigned)((P + 255) / 256), num_hypotheses), dim3(256), 0, stream, live_vertices, canonical, mask,
              height, width, hypotheses, fx, fy, px, py, radius, static_cast<unsigned*>(workspace), nwords, hits);
  return check_launch("icp_score_fwd");
}

extern "C" int pcnn_icp_polish_fwd(const int32_t* label, const float* live_vertices, const float* pred_vertices, int pred_channels,
                                   int height, int width, int obj_id, float z_near, float z_far, int max_evaluations, double* update,
                                   double* info, void* stream_)
{
  PCNN_REQUIRE(height >= 1 && width >= 1, PCNN_EINVAL, "icp_polish: bad shape %dx%d", height, width);
  PCNN_REQUIRE(pred_channels == 3 || pred_channels == 4, PCNN_EINVAL, "icp_polish: predicted maps carry 3 or 4 floats per pixel (got %d)", pred_channels);
  PCNN_REQUIRE(max_evaluations >= 8, PCNN_EINVAL, "icp_polish: the initial simplex alone takes 8 evaluations (got max_evaluations = %d)", max_evaluations);
  PCNN_REQUIRE(label && live_vertices && pred_vertices && update && info, PCNN_ENULL, "icp_polish: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  PCNN_LAUNCH(icp_polish_kernel, dim3(1), dim3(NM_LANES), 0, stream, label, live_vertices, pred_vertices, pred_channels, height, width,
              obj_id, z_near, z_far, max_evaluations, update, info);
  return check_launch("icp_polish_fwd");
}
