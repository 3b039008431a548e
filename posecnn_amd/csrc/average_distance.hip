// average_distance.hip — gfx950 PoseCNN pose loss (PLoss / SLoss) forward + analytic dq gradient
// (replaces TF1 ops "Averagedistance"/"AveragedistanceGrad",
//  lib/average_distance_loss/average_distance_loss_op.cc:253-314,
//  average_distance_loss_op_gpu.cu.cc:35-206 (AveragedistanceForward), :210-252
//  (sum_losses_gradients), :256-343 (launcher), :347-377 (backward)).
//
// The reference writes 54 floats of per-(roi, point) rotation scratch, 88 floats of per-point
// gradient scratch and a loss per point to global memory (~1.5 MB per ROI), then sums them with one
// thread per output channel. Here (DESIGN.md 3.5 has the measurements behind each choice):
//   adl_order   one workgroup: the rows that have a pose target, symmetric classes first (their workgroups are ~15x
//               longer than any other row's and must be dispatched first to spread over the CUs).
//   adl_terms   grid (point slabs, listed rows): rotations live in registers; for a symmetric class the
//               nearest-neighbour search runs over the gt-rotated model points staged through LDS in
//               1024-point tiles — trip minima over sixteen candidates (v_min tree), the reference's strict-'<'
//               walk only inside the winning trip; per point only the 5 non-zero terms
//               {loss, dq_s, dq_u, dq_v, dq_w} go to the workspace.
//   adl_sum     grid (rois): the canonical ascending-p sums (sum_losses_gradients :237-248) from
//               an LDS copy of the ROI's 5 x P terms; writes the whole 4C-wide gradient row.
//   adl_total   thrust::reduce over ROIs (:333-335), ascending.
#include <cfloat>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

// stage clocks for tools/adl_stamp_probe.hip (which defines ADL_STAMP before including this file); nothing in the library
#ifndef ADL_STAMP
#define ADL_STAMP(k)
#endif

constexpr int ADL_THREADS = 256;
constexpr int ADL_PPT = 1;            // adl_terms: points per thread (2 measured slower: see the symmetric scan's notes)
constexpr int ADL_QTILE = 1024;
constexpr int ADL_SUM_THREADS = 256;
constexpr int ADL_SUM_TILE_MAX = 3072;   // adl_sum_kernel: terms of one row staged per round, at most (62 KB of LDS for the five chains)
constexpr int ADL_SUM_PAD = 36;          // floats behind each chain's tile: the chain prefetches one trip of 16 past its end
// the staged tile: the whole row (to a multiple of 32 floats) when it fits
__host__ __device__ inline int adl_sum_tile(int P) { const int t = (P + 31) & ~31; return t < ADL_SUM_TILE_MAX ? t : ADL_SUM_TILE_MAX; }
constexpr int ADL_ROW_SLOTS = 512;   // grid.y of adl_terms_kernel at most; rows are strided over it

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
// One model point through the gt rotation (:156-158): the expression the tile fill, the in-trip walk and the final
// term all use, so the three see the same bits.
__device__ __forceinline__ void adl_rotate3(const float* rg, float a, float b, float c, float& x, float& y, float& z) {
  x = rg[0] * a + rg[1] * b + rg[2] * c;
  y = rg[3] * a + rg[4] * b + rg[5] * c;
  z = rg[6] * a + rg[7] * b + rg[8] * c;
}

// A row's class and its two quaternions in ONE round of loads: lane c reads class c's weight and both quaternions,
// a ballot picks the first positive weight (:52-60) and v_readlane hands its eight floats to every lane. The serial form —
// class first, then the quaternions at the address it gives — is one more dependent trip to memory in front of every
// workgroup, and this kernel's launch is a few rounds of exactly such trips (tools/probe_adl.py: 46 us with no symmetric
// row at all). More than 64 classes: the two-trip form. Must be called by all lanes of the wave; returns -1 for no target.
__device__ __forceinline__ int adl_row_header(const float* __restrict__ weight, const float* __restrict__ target,
                                              const float* __restrict__ prediction, int n, int C, float* qt, float* qp)
{
  const size_t row = (size_t)n * PCNN_POSE_CHANNELS * C;
  if (C <= 64) {
    const int c = lane_id() < C ? lane_id() : 0;
    const float* wr = weight + row + PCNN_POSE_CHANNELS * c;
    const float* tr = target + row + PCNN_POSE_CHANNELS * c;
    const float* pr = prediction + row + PCNN_POSE_CHANNELS * c;
    const float wv = wr[0];
    float t[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { t[i] = tr[i]; q[i] = pr[i]; }
    const unsigned long long m = __ballot(lane_id() < C && wv > 0);
    if (!m) return -1;
    const int cls = __ffsll((long long)m) - 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      qt[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t[i]), cls));
      qp[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q[i]), cls));
    }
    return cls;
  }
  const int cls = find_class(weight, n, C);
  if (cls < 0) return -1;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    qt[i] = target[row + PCNN_POSE_CHANNELS * cls + i];
    qp[i] = prediction[row + PCNN_POSE_CHANNELS * cls + i];
  }
  return cls;
}

// Rows with a pose target, symmetric classes first. counts = {symmetric rows, rows with a target}; entry k of the list is
// order[k] for k < counts[0] and order[R_cap - 1 - (k - counts[0])] after that: the symmetric rows are placed from the
// front of the buffer and the others from its back, both in ascending row order, so one pass over the rows places both
// groups without knowing their sizes (adl_order_entry below). adl_terms_kernel walks this list, so
//  * the long workgroups (a symmetric row's nearest-neighbour scan is ~35 us, any other row's work ~2) are dispatched FIRST
//    and round-robin over the CUs. Dispatched in row order they land wherever a slot happens to be free, some CU ends up
//    with seven of them and the launch waits for it: 180 us against 121 for the same rows listed symmetric-first
//    (tools/probe_adl.py, "spread" / "contiguous");
//  * rows without a target cost no workgroup at all.
constexpr int ADL_ORDER_THREADS = 1024;
__global__ __launch_bounds__(ADL_ORDER_THREADS) void adl_order_kernel(const float* __restrict__ weight,
                                                                       const float* __restrict__ symmetry, int* __restrict__ order,
                                                                       int* __restrict__ counts, int R_cap, int C,
                                                                       const int* __restrict__ num_rows_dev)
{
  __shared__ int s_wave[2][ADL_ORDER_THREADS / 64];
  __shared__ int s_run[2];   // rows placed so far, per group
  const int R = num_rows_dev ? min(R_cap, num_rows_dev[0]) : R_cap;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 2) s_run[tid] = 0;
  __syncthreads();
  for (int n0 = 0; n0 < R; n0 += ADL_ORDER_THREADS) {
    const int n = n0 + tid;
    // 0 = no target, 1 = symmetric class, 2 = any other class; the first positive weight names the class (:52-60).
    // Sixteen weights per round of loads (a loop with an early exit is one trip to memory per class: 23 us for 684 rows)
    int cls = -1;
    if (n < R) {
      const float* wr = weight + (size_t)n * PCNN_POSE_CHANNELS * C;
      for (int c0 = 0; c0 < C && cls < 0; c0 += 16) {
        float wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wv[i] = wr[PCNN_POSE_CHANNELS * min(c0 + i, C - 1)];
#pragma unroll
        for (int i = 15; i >= 0; i--)
          if (c0 + i < C && wv[i] > 0) cls = c0 + i;
      }
    }
    const int k = cls < 0 ? 0 : (symmetry[cls] > 0 ? 1 : 2);
    const unsigned long long b1 = __ballot(k == 1), b2 = __ballot(k == 2);
    if (lane == 0) { s_wave[0][wave] = __popcll(b1); s_wave[1][wave] = __popcll(b2); }
    __syncthreads();
    if (k) {
      const int g = k - 1;
      int pos = s_run[g] + __popcll((g ? b2 : b1) & lanemask_lt());
      for (int w = 0; w < wave; w++) pos += s_wave[g][w];
      order[g ? R_cap - 1 - pos : pos] = n;
    }
    __syncthreads();
    if (tid < 2) {
      int t = 0;
      for (int w = 0; w < ADL_ORDER_THREADS / 64; w++) t += s_wave[tid][w];
      s_run[tid] += t;
    }
    __syncthreads();
  }
  if (tid == 0) { counts[0] = s_run[0]; counts[1] = s_run[0] + s_run[1]; }
}
__device__ __forceinline__ int adl_order_entry(const int* __restrict__ order, const int* __restrict__ counts, int R_cap, int k)
{
  const int n_sym = counts[0];
  return order[k < n_sym ? k : R_cap - 1 - (k - n_sym)];
}

// the per-point tail of adl_terms: the loss term and the four quaternion derivatives of point p, matched to model point q
__device__ __forceinline__ void adl_point_terms(const float* rg, float s, float u, float v, float w, const float* pt,
                                                float x1, float y1, float z1, float qa, float qb, float qc,
                                                float margin, int R, int P, float* __restrict__ tn, int p)
{
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

// One row, the workgroup's ADL_THREADS * ADL_PPT points of it: thread t owns points p0 + t + ADL_THREADS i.
__device__ __forceinline__ void adl_terms_row(
    const float* __restrict__ prediction, const float* __restrict__ target,
    const float* __restrict__ weight, const float* __restrict__ point,
    const float* __restrict__ symmetry, float* __restrict__ terms, int C, int P,
    float margin, int R, int n, int p0, float* s_qx, float* s_qy, float* s_qz)
{
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  ADL_STAMP(10);
  float qt[4], qp[4];
  const int cls = adl_row_header(weight, target, prediction, n, C, qt, qp);
  ADL_STAMP(11);
  float* tn = terms + (size_t)n * 5 * P;
  if (cls < 0) return;   // no target: adl_sum writes this row's zeros without reading any terms
                         // (a train-mode buffer of 3024 rows holds ~470 with targets: 134 MB of zeros were written here
                         //  and read back there)
  float rg[9], ru[9];
  quat_rot(qt[0], qt[1], qt[2], qt[3], rg);
  const float s = qp[0], u = qp[1], v = qp[2], w = qp[3];
  quat_rot(s, u, v, w, ru);
  const float* pts = point + (size_t)cls * P * 3;
  const bool symmetric = symmetry[cls] > 0;
  float pt[ADL_PPT][3], x1[ADL_PPT], y1[ADL_PPT], z1[ADL_PPT];
  float qm[ADL_PPT][3];   // the matched point: the point itself unless the class is symmetric
#pragma unroll
  for (int i = 0; i < ADL_PPT; i++) {
    const int p = p0 + (int)threadIdx.x + ADL_THREADS * i;
    pt[i][0] = pt[i][1] = pt[i][2] = 0.f;
    if (p < P) { pt[i][0] = pts[p * 3]; pt[i][1] = pts[p * 3 + 1]; pt[i][2] = pts[p * 3 + 2]; }
    adl_rotate3(ru, pt[i][0], pt[i][1], pt[i][2], x1[i], y1[i], z1[i]);
    qm[i][0] = pt[i][0]; qm[i][1] = pt[i][1]; qm[i][2] = pt[i][2];
  }
  ADL_STAMP(12);

  if (symmetric) {
    // closest gt-rotated model point, strict '<' from FLT_MAX, first wins (:155-168).
    //
    // Two steps with the same answer as the reference's one-candidate-at-a-time walk:
    //  1. over trips of SIXTEEN consecutive candidates: m = the smallest distance of the trip (a v_min tree: exact, and a
    //     NaN distance is ignored exactly as 'NaN < dmin' ignores it); 'if (m < dmin) { dmin = m; tmin = trip; }'. The strict
    //     '<' keeps the FIRST trip that holds the overall minimum.
    //  2. inside that one trip, the reference's walk itself (strict '<' from FLT_MAX): the first candidate at the minimum.
    // Step 1 is where the time goes (P^2 = 6.9 M distances per symmetric RoI); what shaped it, all measured
    // (tools/probe_adl.py, tools/adl_stamp_probe.hip, tools/valu_rate_probe.hip):
    //  * the walk's compare -> select chain (a VALU write of VCC read by the next VALU instruction) runs at 12.6 cycles per
    //    instruction, 25 per candidate; a v_min tree has no chain and the compare is per trip;
    //  * a candidate is the same for every lane; handing it to 64 lanes through LDS costs the LDS pipe the full 64-lane
    //    bandwidth (a 128-bit "broadcast" read holds it 8 cycles, a trip's twelve reads 96) — but that is NOT the bound: the
    //    trip's 64 packed-f32 + 10 v_min instructions are (counters: the vector ALU busy 58 % of a launch of symmetric rows,
    //    the LDS a third). Two or four points per lane against each read (ADL_PPT; four with the waves splitting the
    //    candidates) measured 130 / 117 us against 100 for 63 symmetric rows: fewer, longer workgroups balance worse over
    //    the CUs. ADL_PPT stays 1;
    //  * packed f32 (two candidates per instruction) is worth its encoding here: a lone wave on its SIMD issues a v_pk_add
    //    every 5.3 cycles and a v_add every 4.7, so half the instructions is nearly half the time for the few workgroups
    //    of a launch's tail; with the SIMD full the two forms cost the same per distance.
    float dmin[ADL_PPT];
    int tmin[ADL_PPT];
    v2f xv[ADL_PPT], yv[ADL_PPT], zv[ADL_PPT];
#pragma unroll
    for (int i = 0; i < ADL_PPT; i++) {
      dmin[i] = FLT_MAX; tmin[i] = -1;
      xv[i] = (v2f){x1[i], x1[i]}; yv[i] = (v2f){y1[i], y1[i]}; zv[i] = (v2f){z1[i], z1[i]};
    }
    // the tile's model points are fetched one tile ahead, into registers: the next tile's trip to memory runs under this
    // tile's scan instead of in front of its own
    constexpr int NQ = ADL_QTILE / ADL_THREADS;
    float pa[NQ][3];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
      const int q = min((int)threadIdx.x + i * ADL_THREADS, P - 1);
      pa[i][0] = pts[q * 3]; pa[i][1] = pts[q * 3 + 1]; pa[i][2] = pts[q * 3 + 2];
    }
    for (int q0 = 0; q0 < P; q0 += ADL_QTILE) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NQ; i++) {
        const int j = threadIdx.x + i * ADL_THREADS;
        // past the last point: +inf coordinates -> distance +inf (or NaN), never '<' anything: whole trips, no tail
        float qx = __builtin_inff(), qy = qx, qz = qx;
        if (q0 + j < P) adl_rotate3(rg, pa[i][0], pa[i][1], pa[i][2], qx, qy, qz);
        s_qx[j] = qx; s_qy[j] = qy; s_qz[j] = qz;
      }
      __syncthreads();
      if (q0 == 0) ADL_STAMP(13);
      if (q0 + ADL_QTILE < P) {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
          const int q = min(q0 + ADL_QTILE + (int)threadIdx.x + i * ADL_THREADS, P - 1);
          pa[i][0] = pts[q * 3]; pa[i][1] = pts[q * 3 + 1]; pa[i][2] = pts[q * 3 + 2];
        }
      }
      const int lim = min(ADL_QTILE, P - q0);
      // one 128-bit LDS read per coordinate per four candidates, all twelve of a trip issued before the first use; two
      // candidates per packed-f32 instruction — v_pk_add / v_pk_mul round each half like the scalar form, so every
      // distance is ((ex ex) + (ey ey)) + (ez ez) bit for bit (:160-163)
      for (int j = 0; j < lim; j += 16) {
        v4f qx[4], qy[4], qz[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          qx[c] = *reinterpret_cast<const v4f*>(&s_qx[j + 4 * c]);
          qy[c] = *reinterpret_cast<const v4f*>(&s_qy[j + 4 * c]);
          qz[c] = *reinterpret_cast<const v4f*>(&s_qz[j + 4 * c]);
        }
#pragma unroll
        for (int i = 0; i < ADL_PPT; i++) {
          // (wave-uniform) all 64 of this wave's i-th points past the row's last one: the row's last workgroup
          if (p0 + (int)(threadIdx.x & ~63u) + ADL_THREADS * i >= P) continue;
          float m4[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const v2f ex0 = xv[i] - qx[c].xy, ey0 = yv[i] - qy[c].xy, ez0 = zv[i] - qz[c].xy;
            const v2f ex1 = xv[i] - qx[c].zw, ey1 = yv[i] - qy[c].zw, ez1 = zv[i] - qz[c].zw;
            const v2f d0 = (ex0 * ex0 + ey0 * ey0) + ez0 * ez0;
            const v2f d1 = (ex1 * ex1 + ey1 * ey1) + ez1 * ez1;
            m4[c] = __builtin_fminf(__builtin_fminf(d0.x, d0.y), __builtin_fminf(d1.x, d1.y));
          }
          const float m = __builtin_fminf(__builtin_fminf(m4[0], m4[1]), __builtin_fminf(m4[2], m4[3]));
          if (m < dmin[i]) { dmin[i] = m; tmin[i] = q0 + j; }
        }
      }
      if (q0 == 0) ADL_STAMP(14);
    }
    ADL_STAMP(15);
#pragma unroll
    for (int i = 0; i < ADL_PPT; i++) {
      if (tmin[i] < 0) continue;
      // step 2: the trip's sixteen candidates again, rotated by the expression that filled the tile (same bits); loads
      // eight at a time, all issued before the first use (clamped addresses: a candidate past P is skipped below)
      float dm = FLT_MAX;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float ca[8][3];
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int q = min(tmin[i] + 8 * h + c, P - 1);
          ca[c][0] = pts[q * 3]; ca[c][1] = pts[q * 3 + 1]; ca[c][2] = pts[q * 3 + 2];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int q = tmin[i] + 8 * h + c;
          float qx, qy, qz;
          adl_rotate3(rg, ca[c][0], ca[c][1], ca[c][2], qx, qy, qz);
          const float ex = x1[i] - qx, ey = y1[i] - qy, ez = z1[i] - qz;
          const float d = (ex * ex + ey * ey) + ez * ez;
          if (q < P && d < dm) { dm = d; qm[i][0] = ca[c][0]; qm[i][1] = ca[c][1]; qm[i][2] = ca[c][2]; }
        }
      }
    }
    ADL_STAMP(16);
  }
#pragma unroll
  for (int i = 0; i < ADL_PPT; i++) {
    const int p = p0 + (int)threadIdx.x + ADL_THREADS * i;
    if (p < P) adl_point_terms(rg, s, u, v, w, pt[i], x1[i], y1[i], z1[i], qm[i][0], qm[i][1], qm[i][2], margin, R, P, tn, p);
  }
  ADL_STAMP(17);
}

__global__ __launch_bounds__(ADL_THREADS) void adl_terms_kernel(
    const float* __restrict__ prediction, const float* __restrict__ target,
    const float* __restrict__ weight, const float* __restrict__ point,
    const float* __restrict__ symmetry, float* __restrict__ terms, int R_cap, int C, int P,
    float margin, const int* __restrict__ num_rows_dev, const int* __restrict__ order, const int* __restrict__ counts)
{
  __shared__ __attribute__((aligned(16))) float s_qx[ADL_QTILE], s_qy[ADL_QTILE], s_qz[ADL_QTILE];   // gt-rotated model points, coordinate-major
  const int p0 = blockIdx.x * (ADL_THREADS * ADL_PPT);
  // R = the op's row count: the buffers' row capacity, or (capacity-sized buffers of the sync-free
  // Hough op) the device-side count; rows past it do not exist for the loss
  const int R = num_rows_dev ? min(R_cap, num_rows_dev[0]) : R_cap;
  // The rows with a target (adl_order_kernel's list, symmetric classes first) are strided over grid.y (ADL_ROW_SLOTS at
  // most): workgroups are independent, so the row -> workgroup map is free
  const int n_rows = counts[1];
  for (int k = blockIdx.y; k < n_rows; k += gridDim.y)
    adl_terms_row(prediction, target, weight, point, symmetry, terms, C, P, margin, R, adl_order_entry(order, counts, R_cap, k), p0, s_qx, s_qy, s_qz);
}

// ascending-p sums, one workgroup per RoI: all ADL_SUM_THREADS lanes stage the row's [5][P] terms in LDS (one round of
// loads for P <= ADL_SUM_TILE), then lanes 0..4 of wave 0 own one chain each
__global__ __launch_bounds__(ADL_SUM_THREADS) void adl_sum_kernel(const float* __restrict__ terms,
                                                     const float* __restrict__ weight,
                                                     float* __restrict__ loss_batch,
                                                     float* __restrict__ bottom_diff, int C, int P,
                                                     int R_cap, const int* __restrict__ num_rows_dev)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float s_t[];  // tile of [5][TS]
  const int TILE = adl_sum_tile(P), TS = TILE + ADL_SUM_PAD;   // TS = 4 mod 32 banks: the five chains read different banks
  const int n = blockIdx.x, tid = threadIdx.x;
  const int CH = PCNN_POSE_CHANNELS * C;
  const int R = num_rows_dev ? min(R_cap, num_rows_dev[0]) : R_cap;
  ADL_STAMP(0);
  const int cls = n < R ? find_class(weight, n, C) : -1;
  ADL_STAMP(1);
  // zeros everywhere but the row's own class, whose four sums are written at the end (no address is written twice)
  for (int c = tid; c < CH; c += ADL_SUM_THREADS)
    if (cls < 0 || c / PCNN_POSE_CHANNELS != cls) bottom_diff[(size_t)n * CH + c] = 0.f;
  if (cls < 0) {         // every term of the row is +0 (never written by adl_terms): the sums are +0
    if (tid == 0) loss_batch[n] = 0.f;
    return;
  }
  const float* tn = terms + (size_t)n * 5 * P;
  const bool vec4 = (P & 3) == 0 && (reinterpret_cast<size_t>(terms) & 15) == 0;   // every chain of every row starts 16-byte aligned
  float acc = 0.f;
  for (int p0 = 0; p0 < P; p0 += TILE) {
    const int lim = min(TILE, P - p0);
    __syncthreads();
    if (vec4 && (lim & 3) == 0) {
      // every load of the round issued before the first LDS write (a tile is at most 3 x 256 float4 per chain): with the
      // plain loop below the compiler pairs each load with its LDS store, 51 trips to memory one after the other — 13.6 of
      // the row's 27 us (tools/adl_stamp_probe.hip)
      static_assert(ADL_SUM_TILE_MAX <= 3 * ADL_SUM_THREADS * 4, "adl_sum_kernel: three float4 per thread per chain");
      const int nq = lim >> 2;
      v4f r[5][3];
#pragma unroll
      for (int k = 0; k < 5; k++)
#pragma unroll
        for (int h = 0; h < 3; h++) {
          const int j4 = tid + h * ADL_SUM_THREADS;
          if (j4 < nq) r[k][h] = *reinterpret_cast<const v4f*>(tn + (size_t)k * P + p0 + 4 * j4);
        }
#pragma unroll
      for (int k = 0; k < 5; k++)
#pragma unroll
        for (int h = 0; h < 3; h++) {
          const int j4 = tid + h * ADL_SUM_THREADS;
          if (j4 < nq) *reinterpret_cast<v4f*>(s_t + k * TS + 4 * j4) = r[k][h];
        }
    } else {
      for (int k = 0; k < 5; k++)
        for (int j = tid; j < lim; j += ADL_SUM_THREADS) s_t[k * TS + j] = tn[(size_t)k * P + p0 + j];
    }
    __syncthreads();
    ADL_STAMP(2);
    if (tid < 5) {
      // the chain itself: one add per term in ascending p, nothing else on its path — sixteen terms per trip, the NEXT
      // trip's four 128-bit LDS reads issued before this trip's adds (the pad behind the tile keeps the last, unused,
      // prefetch inside the allocation). With one read per add the wave sat out an LDS round trip per term: 30 us for one
      // row (tools/probe_adl.py), most of it that
      const float* t = s_t + tid * TS;
      v4f a = *reinterpret_cast<const v4f*>(t), b = *reinterpret_cast<const v4f*>(t + 4);
      v4f c = *reinterpret_cast<const v4f*>(t + 8), d = *reinterpret_cast<const v4f*>(t + 12);
      int j = 0;
      for (; j + 16 <= lim; j += 16) {
        const v4f na = *reinterpret_cast<const v4f*>(t + j + 16), nb = *reinterpret_cast<const v4f*>(t + j + 20);
        const v4f nc = *reinterpret_cast<const v4f*>(t + j + 24), nd = *reinterpret_cast<const v4f*>(t + j + 28);
        acc += a.x; acc += a.y; acc += a.z; acc += a.w;
        acc += b.x; acc += b.y; acc += b.z; acc += b.w;
        acc += c.x; acc += c.y; acc += c.z; acc += c.w;
        acc += d.x; acc += d.y; acc += d.z; acc += d.w;
        a = na; b = nb; c = nc; d = nd;
      }
      for (; j < lim; j++) acc += t[j];
    }
  }
  ADL_STAMP(3);
  if (tid == 0) loss_batch[n] = acc;
  if (tid >= 1 && tid < 5) bottom_diff[(size_t)n * CH + PCNN_POSE_CHANNELS * cls + (tid - 1)] = acc;
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

// workspace: terms [R][5][P] | loss_batch [R] | order [R] + counts [2]
size_t adl_ws(int R, int P)
{
  const size_t r = (size_t)(R > 0 ? R : 1);
  return align_up(sizeof(float) * (size_t)R * 5 * P, 256) + align_up(sizeof(float) * r, 256) + align_up(sizeof(int) * (r + 2), 256);
}

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
  int* order = (int*)((char*)loss_batch + align_up(sizeof(float) * (size_t)R, 256));
  int* counts = order + R;
  PCNN_LAUNCH(adl_order_kernel, dim3(1), dim3(ADL_ORDER_THREADS), 0, stream, weight, symmetry, order, counts, R, C, num_rows_dev);
  PCNN_LAUNCH(adl_terms_kernel, dim3((P + ADL_THREADS * ADL_PPT - 1) / (ADL_THREADS * ADL_PPT), R < ADL_ROW_SLOTS ? R : ADL_ROW_SLOTS), dim3(ADL_THREADS), 0,
                     stream, prediction, target, weight, point, symmetry, terms, R, C, P, margin, num_rows_dev, order, counts);
  PCNN_LAUNCH(adl_sum_kernel, dim3(R), dim3(ADL_SUM_THREADS), sizeof(float) * 5 * (adl_sum_tile(P) + ADL_SUM_PAD), stream, terms, weight,
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
