// hough_voting.hip — gfx950 Hough voting for PoseCNN (replaces the TF1 op "Houghvotinggpu",
// lib/hough_voting_gpu_layer/hough_voting_gpu_op.cc:321-429 + hough_voting_gpu_op.cu.cc:615-797).
//
// This is NOT the reference's launch sequence. The reference runs, per image and serially,
// compute_arrays (atomic compaction) -> host filter -> compute_hough (one thread per Hough cell,
// brute force over every sampled class pixel, two passes) -> thrust::max_element per class on
// the host's clock -> compute_rois, with >=6 blocking host round trips per image. Here the whole
// batch is five asynchronous launches with no host involvement:
//
//   hv_hist     per-chunk class histograms of the label map (+ zero-fill of the outputs)
//   hv_scatter  deterministic ranks (ascending pixel index, as a serial run of
//               compute_arrays_kernel :174-187 would produce) from histogram prefixes + wave
//               ballots; every skip-th pixel of each class that passes label_threshold becomes a
//               48-byte record {x, y, thr(d), 1/|uv| ; u, v, |uv|, d ; ra, rb, g, mode}: everything the
//               inner loop of compute_hough_kernel :269-285 recomputes per (cell, pixel) pair, hoisted,
//               plus the two interval roots and the mode of the pixel's vote cone.
//   hv_vote     interval formulation on bands of 4 Hough rows (two waves per row): the records that can
//               reach a band are a contiguous range of the class' y-sorted list (64-ary search);
//               a record's vote cone cut by a row is one dx-interval -> +1 / -1 in the row's LDS
//               difference array; a wave-wide prefix sum yields the votes and the row maximum.
//               Votes are integers, so evaluation order is free. Only votes are produced; the
//               reference's second pass (mean depth, box extents, :296-331) is evaluated lazily,
//               at the cells that can reach an output.
//   hv_select   (threshold_vote <= 0) per class: first argmax over the row maxima
//               (thrust::max_element :752-762), then one wave recomputes that cell's depth sum
//               in canonical pixel order and its box extents.
//   hv_localmax + hv_gather (threshold_vote > 0): compute_max_indexes_kernel :335-383 in
//               ascending cell order, capacity MAX_ROI / batch.
//   hv_emit     compute_rois_kernel :386-576, rows in (image, maximum) order.
//
// Exactness: the vote predicate is evaluated either by a filter that provably agrees with the
// exact expression (|q~ - q| << 2e-5) or by the exact expression itself (IEEE div/sqrt, no
// contraction), so outputs equal the CPU oracle bit for bit.
#include "bilinear.h"

namespace {

using namespace pcnn;

constexpr int HV_CHUNK = 2048;      // label pixels per hist/scatter workgroup (4 waves x 8 x 64)
constexpr int HV_TILE = 32;         // Hough tile edge (cells)
constexpr int LM_CHUNK = 1024;      // consecutive Hough cells per hv_localmax workgroup
constexpr float HV_FILTER_EPS = 2e-5f;

struct __attribute__((aligned(16))) HvRec {
  float4 a;  // x, y, thr, rn1 (1/|uv|, or NaN when |uv| is outside the filter's safe range)
  float4 b;  // u, v, |uv|, d
  float4 c;  // ra, rb, g, mode: the record's vote cone cut by a row dy is the dx-interval with ends
             // dy*ra, dy*rb (see cone_interval); mode 0 = no closed form (evaluate cells one by one)
};

struct __attribute__((aligned(16))) HvMax {
  int cls;
  int idx;  // cy * W + cx
  float votes;
  float dist;
  float bh2;  // 2 * bb_height
  float bw2;  // 2 * bb_width
  int pad0, pad1;
};

struct HvLayout {
  int nchunk, ntx, nty, ntiles, reccap, cap, capmax, nlm;
  size_t off_hist, off_tot, off_slots, off_nslots, off_recoff, off_kmax, off_rec, off_tilemax, off_maxima,
      off_nmax, off_hs, off_chunkcnt, off_chunkcand, off_flags, off_order, off_rowstart, total;
};

HvLayout hv_layout(int B, int H, int W, int C, bool need_hs, int skip, int rois_per_image)
{
  HvLayout L;
  const size_t HW = (size_t)H * W;
  L.nchunk = (int)((HW + HV_CHUNK - 1) / HV_CHUNK);
  L.ntx = (W + HV_TILE - 1) / HV_TILE;
  L.nty = (H + HV_TILE - 1) / HV_TILE;
  L.ntiles = L.ntx * L.nty;
  L.reccap = (int)(HW / skip) + C + 1;
  // index_size, hough_voting_gpu_op.cu.cc:733 — or the caller's per-image capacity (posecnn_hip.h)
  L.cap = rois_per_image > 0 ? rois_per_image : PCNN_MAX_ROI / B;
  L.capmax = L.cap > 0 ? L.cap : 1;
  L.nlm = (int)(((size_t)(C - 1) * HW + LM_CHUNK - 1) / LM_CHUNK);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  L.off_hist = take(sizeof(int) * (size_t)B * L.nchunk * C);
  L.off_tot = take(sizeof(int) * (size_t)B * C);
  L.off_slots = take(sizeof(int) * (size_t)B * C);
  L.off_nslots = take(sizeof(int) * (size_t)B);
  L.off_recoff = take(sizeof(int) * (size_t)B * C);
  L.off_kmax = take(sizeof(int) * (size_t)B * C);   // f32 bits: largest vote window thr of a class' records
  L.off_order = take(sizeof(int) * ((size_t)B * (C - 1) + 4));   // hv_order: {-, live pairs, -, -} + the pairs, heaviest first
  L.off_rowstart = take(sizeof(int) * (size_t)B * (C - 1) * (H + 1));   // hv_order: per (image, slot): first record at or below row y
  L.off_rec = take(sizeof(HvRec) * (size_t)B * L.reccap);
  L.off_tilemax = take(sizeof(int2) * (size_t)B * (C - 1) * H);   // per Hough ROW: (max votes, first cell)
  L.off_maxima = take(sizeof(HvMax) * (size_t)B * L.capmax);
  L.off_nmax = take(sizeof(int) * (size_t)B);
  if (need_hs) {
    L.off_hs = take(sizeof(float) * (size_t)B * (C - 1) * HW);
    L.off_chunkcnt = take(sizeof(int) * (size_t)B * L.nlm);
    L.off_chunkcand = take(sizeof(HvMax) * (size_t)B * L.nlm * L.capmax);
    L.off_flags = take((size_t)B * (C - 1) * HW);
  } else {
    L.off_hs = L.off_chunkcnt = L.off_chunkcand = L.off_flags = 0;
  }
  L.total = o;
  return L;
}

// project_box, hough_voting_gpu_op.cu.cc:84-120 with factor 0.6 (:285, :317)
__device__ float project_box(const float* __restrict__ extents, int cls, float fx, float fy,
                             float px, float py, float distance)
{
  float xHalf = (float)((double)extents[cls * 3 + 0] * 0.5);
  float yHalf = (float)((double)extents[cls * 3 + 1] * 0.5);
  float zHalf = (float)((double)extents[cls * 3 + 2] * 0.5);
  float zf = zHalf + distance;
  float zb = -zHalf + distance;
  float minX = 1e8f, maxX = -1e8f, minY = 1e8f, maxY = -1e8f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float X = (i & 1) ? -xHalf : xHalf;
    float Y = (i & 2) ? -yHalf : yHalf;
    float Z = (i & 4) ? zb : zf;
    float x = fx * div_rn(X, Z) + px;
    float y = fy * div_rn(Y, Z) + py;
    minX = fminf(minX, x);
    minY = fminf(minY, y);
    maxX = fmaxf(maxX, x);
    maxY = fmaxf(maxY, y);
  }
  float width = maxX - minX + 1;
  float height = maxY - minY + 1;
  return fmaxf(width, height) * 0.6f;
}

// angle_distance(...) > inlierThreshold, hough_voting_gpu_op.cu.cc:32-42,283 — exact form.
__device__ __forceinline__ bool angle_pass_exact(float u, float v, float n1, float dx, float dy,
                                                 float inlier)
{
  float n2 = sqrt_rn(dx * dx + dy * dy);
  float dot = u * dx + v * dy;
  return div_rn(dot, n1 * n2) > inlier;
}

// Filtered form: q~ = dot * rsq(|d|^2) / |uv| differs from the exact quotient by < 1e-6 whenever
// rn1 is finite (|uv| in [1e-15, 1e15]); outside +-HV_FILTER_EPS of the threshold the decision
// is therefore already the exact one. NaN (rn1 poisoned, d = 0) falls through to the exact form.
__device__ __forceinline__ bool angle_pass(float u, float v, float n1, float rn1, float dx,
                                           float dy, float inlier)
{
  float dotf = __builtin_fmaf(u, dx, v * dy);
  float s2f = __builtin_fmaf(dx, dx, dy * dy);
  float qa = dotf * __builtin_amdgcn_rsqf(s2f) * rn1;
  if (qa > inlier + HV_FILTER_EPS) return true;
  if (qa < inlier - HV_FILTER_EPS) return false;
  return angle_pass_exact(u, v, n1, dx, dy, inlier);
}

// The vote predicate of a record (x, y, u, v) for the cell at (x + dx, y + dy) is
//   (u dx + v dy) / (n1 * sqrt(dx^2 + dy^2)) > c        (c = inlierThreshold, n1 = fl|uv|).
// For fixed dy != 0 the cells that satisfy it form ONE interval in dx (a convex cone cut by a line).
// Squaring:  A dx^2 + 2 u v dy dx + (v^2 - c^2 n1^2) dy^2 > 0  with  A = u^2 - c^2 n1^2, and
// u dx + v dy > 0. Its roots are dy * ra and dy * rb with
//   ra, rb = (-u v +- n1 sqrt(c^2 (u^2 + v^2 - c^2 n1^2))) / A      (discriminant always >= 0).
//   A < 0: the interval between the roots, if the dot product at their midpoint (= dy * g) is > 0;
//   A > 0: the half line beyond the larger root (u > 0) or below the smaller one (u < 0).
// Evaluated once per record in double, so the f32 ends dy*ra, dy*rb are within ~1e-4 cell of the
// real roots; hv_vote treats cells within 0.01 of an end as uncertain and decides those with the
// exact f32 predicate, so the interval form can never disagree with the per-cell definition.
__device__ float4 cone_coefficients(float u, float v, float n1, float rn1, float inlier)
{
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);  // mode 0
  if (!(rn1 == rn1) || !(inlier > 0.f && inlier < 1.f)) return c;
  const double U = u, V = v, N = n1, cc = (double)inlier;
  const double k = cc * cc;
  const double A = U * U - k * N * N;
  if (A == 0.0) return c;   // exactly degenerate (measure zero): cell by cell
  const double disc = k * (U * U + V * V - k * N * N);
  if (!(disc >= 0.0)) return c;
  const double D = N * sqrt(disc);
  // numerically stable root pair (no cancellation): q = -(UV + sgn(UV) D), roots q / A and C / q. When a
  // cone edge is (nearly) parallel to the rows A -> 0 and the first root runs off to infinity while the
  // second stays exact — round 1 gave such records up (|A| < 1e-4 N^2 -> cell-by-cell evaluation of the whole
  // window: ~200 exact predicates per (record, row), a lane-serial tail worth ~20 % of the kernel's
  // instructions). A root beyond +-1e6 is clamped there: |dy| >= 1 puts it >= 1e6 cells away, outside every
  // vote window (<= 65536), which is all hv_vote needs to know about it.
  const double UV = U * V, Cq = V * V - k * N * N;
  const double q = -(UV + (UV >= 0.0 ? D : -D));
  double ra = q / A, rb = q != 0.0 ? Cq / q : -UV / A;
  const double LIM = 1e6;
  ra = ra > LIM ? LIM : (ra < -LIM ? -LIM : ra);
  rb = rb > LIM ? LIM : (rb < -LIM ? -LIM : rb);
  if (!(ra == ra) || !(rb == rb)) return c;
  const double g = U * (ra + rb) * 0.5 + V;
  c.x = (float)ra;
  c.y = (float)rb;
  c.z = (float)g;
  c.w = A < 0.0 ? 1.f : (U > 0.0 ? 2.f : 3.f);
  return c;
}

struct ZeroJob {
  float* p[5];
  unsigned words[5];
};

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hv_hist_kernel(const int* __restrict__ label,
                                                      int* __restrict__ hist, int* __restrict__ kmax_g,
                                                      int HW, int C, int nchunk, ZeroJob zj)
{
  __shared__ int sh[PCNN_MAX_CLASSES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int chunk = blockIdx.x, n = blockIdx.y;
  if (tid < PCNN_MAX_CLASSES) sh[tid] = 0;

  // reset_outputs (hough_voting_gpu_op.cu.cc:579-588) folded into the first launch
  {
    const unsigned gtid = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + tid;
    const unsigned gsz = gridDim.x * gridDim.y * 256u;
#pragma unroll
    for (int a = 0; a < 5; a++)
      for (unsigned i = gtid; i < zj.words[a]; i += gsz) zj.p[a][i] = 0.f;
  }
  __syncthreads();

  const int base = chunk * HV_CHUNK;
  const int* lab = label + (size_t)n * HW;
  // (the chunk's labels first: with the load inside the ballot loop below each of the 8 rounds waited for its own)
  int lv[HV_CHUNK / 256];
#pragma unroll
  for (int k = 0; k < HV_CHUNK / 256; k++) {
    const int i = base + k * 256 + tid;
    lv[k] = lab[i < HW ? i : HW - 1];
  }
#pragma unroll
  for (int k = 0; k < HV_CHUNK / 256; k++) {
    int i = base + k * 256 + tid;
    int l = i < HW ? lv[k] : 0;
    bool valid = l > 0 && l < C;
    unsigned long long mask = __ballot(valid);
    while (mask) {
      int src = __ffsll((long long)mask) - 1;
      int c0 = __shfl(l, src);
      unsigned long long m = __ballot(valid && l == c0);
      if (lane == src) atomicAdd(&sh[c0], __popcll(m));
      mask &= ~m;
    }
  }
  __syncthreads();
  if (tid < C) hist[((size_t)n * nchunk + chunk) * C + tid] = sh[tid];
  if (chunk == 0 && tid < C) kmax_g[n * C + tid] = 0;   // max-reduced by hv_scatter (+0.0f)
}

// ---------------------------------------------------------------------------------------------
// Where the (u, v, log d) triple of a sampled pixel comes from: the full-resolution `vertex_pred`
// [B,H,W,3C] of the reference op, or (fused head) the 1/s-resolution field
// z [B,Hl,Wl,3C] + bias [3C] that `vertex_pred = deconv_k,s(z) + bias` would have been built from;
// then only the sampled pixels' own-class channels are ever interpolated (bilinear.h arithmetic,
// bit-identical to deconv_bilinear_kernel) and the 81 MB/frame tensor is never written or read.
struct HvVertexSrc {
  const float* full;
  const float* z;
  const float* bias;
  int Hl, Wl, k, s;
};

__global__ __launch_bounds__(256) void hv_scatter_kernel(
    const int* __restrict__ label, const HvVertexSrc vs,
    const float* __restrict__ extents, const float* __restrict__ meta,
    const int* __restrict__ hist, int* __restrict__ tot_g, int* __restrict__ slots_g,
    int* __restrict__ nslots_g, int* __restrict__ recoff_g, int* __restrict__ kmax_g,
    HvRec* __restrict__ rec, int HW, int W,
    int C, int nchunk, int skip, int label_thr, int num_meta, int reccap, float inlier)
{
  __shared__ int s_pre[PCNN_MAX_CLASSES], s_tot[PCNN_MAX_CLASSES], s_recoff[PCNN_MAX_CLASSES];
  __shared__ int s_wh[4][PCNN_MAX_CLASSES];
  __shared__ int s_kmax[PCNN_MAX_CLASSES];   // f32 bits: largest vote window of this block's records, per class
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, n = blockIdx.y;
  if (tid < PCNN_MAX_CLASSES) { s_pre[tid] = 0; s_tot[tid] = 0; s_kmax[tid] = 0; }
  __syncthreads();
  const int* h = hist + (size_t)n * nchunk * C;
  // (round 5) eight histogram entries per thread and round of loads: one entry per trip was a load, a wait and a branch —
  // 13 trips to L2 in sequence in front of every block of a 16-frame launch (nchunk * C = 3300 entries)
  for (int i0 = tid; i0 < nchunk * C; i0 += 8 * 256) {
    int hv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) hv[j] = h[min(i0 + j * 256, nchunk * C - 1)];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int idx = i0 + j * 256, v = hv[j];
      if (idx < nchunk * C && v) {
        int k = idx / C, c = idx - k * C;
        atomicAdd(&s_tot[c], v);
        if (k < chunk) atomicAdd(&s_pre[c], v);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    // class_indexes (hough_voting_gpu_op.cu.cc:653-663): classes with > labelThreshold pixels, ascending
    int off = 0, ns = 0;
    s_recoff[0] = -1;
    for (int c = 1; c < C; c++) {
      if (s_tot[c] > label_thr) {
        s_recoff[c] = off;
        off += (s_tot[c] + skip - 1) / skip;
        if (chunk == 0) slots_g[n * C + ns] = c;
        ns++;
      } else {
        s_recoff[c] = -1;
      }
    }
    if (chunk == 0) nslots_g[n] = ns;
  }
  __syncthreads();
  if (chunk == 0 && tid < C) {
    tot_g[n * C + tid] = s_tot[tid];
    recoff_g[n * C + tid] = s_recoff[tid];
  }

  // this wave's 512 consecutive pixels, 8 rounds of 64
  int lab[8];
  const int wbase = chunk * HV_CHUNK + wave * 512;
  const int* labp = label + (size_t)n * HW;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int i = wbase + r * 64 + lane;
    int l = i < HW ? labp[i] : 0;
    lab[r] = (l > 0 && l < C) ? l : 0;
  }
  // lane c counts class c inside this wave's span
  int wcnt = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int l = lab[r];
    unsigned long long mask = __ballot(l > 0);
    while (mask) {
      int src = __ffsll((long long)mask) - 1;
      int c0 = __shfl(l, src);
      unsigned long long m = __ballot(l == c0);
      if (lane == c0) wcnt += __popcll(m);
      mask &= ~m;
    }
  }
  s_wh[wave][lane] = wcnt;
  __syncthreads();
  // lane c: number of class-c pixels with a smaller index than this wave's first pixel
  int run = lane < C ? s_pre[lane] : 0;
  for (int w2 = 0; w2 < wave; w2++) run += s_wh[w2][lane];

  const float* md = meta + (size_t)n * num_meta;
  const float fx = md[0], px = md[2], fy = md[4], py = md[5];
  // Pass 1: ranks. The pixels that become records (every skip-th of a class, ~1 lane in 10) are only LISTED here —
  // (pixel | class << 24, rank) in LDS — and turned into records by pass 2 with every lane busy: the record arithmetic
  // (bilinear taps, the f64 exp, project_box, the cone's f64 roots) is ~2000 cycles per wave pass, and run in place it
  // was executed 32 times per block at a tenth of the lanes (hv_scatter 78 us of the sequence).
  __shared__ int s_list[HV_CHUNK][2];
  __shared__ int s_nlist;
  if (tid == 0) s_nlist = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int l = lab[r];
    int my_rank = -1;
    unsigned long long mask = __ballot(l > 0);
    while (mask) {
      int src = __ffsll((long long)mask) - 1;
      int c0 = __shfl(l, src);
      unsigned long long m = __ballot(l == c0);
      int basec = __shfl(run, c0);
      if (l == c0) my_rank = basec + __popcll(m & lanemask_lt());
      if (lane == c0) run += __popcll(m);
      mask &= ~m;
    }
    const bool take = l > 0 && s_recoff[l] >= 0 && (my_rank % skip) == 0;
    const unsigned long long tm = __ballot(take);
    if (tm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_nlist, __popcll(tm));
      base = __shfl(base, 0);
      if (take) {
        const int slot = base + __popcll(tm & lanemask_lt());
        s_list[slot][0] = (wbase + r * 64 + lane) | (l << 24);    // H * W < 2^24 (validate_common), classes < 64
        s_list[slot][1] = my_rank;
      }
    }
  }
  __syncthreads();
  // Pass 2: one record per thread (the list order is arbitrary: a record's place is its rank)
  const int nlist = s_nlist;
  for (int j = tid; j < nlist; j += 256) {
    const int packed = s_list[j][0], my_rank = s_list[j][1];
    const int i = packed & 0xffffff, l = packed >> 24;
    const int ro = s_recoff[l];
    {
      {
        int x = i % W, y = i / W;
        float u, v, logd;
        if (vs.full) {
          const float* vp = vs.full + ((size_t)n * HW + i) * (PCNN_VERTEX_CHANNELS * C) +
                            PCNN_VERTEX_CHANNELS * l;
          u = vp[0]; v = vp[1]; logd = vp[2];
        } else {
          const int VC = PCNN_VERTEX_CHANNELS * C, pad = (vs.k - vs.s) / 2;
          const Taps ty = make_taps(y, vs.k, vs.s, pad, vs.Hl);
          const Taps tx = make_taps(x, vs.k, vs.s, pad, vs.Wl);
          const float* zb = vs.z + (size_t)n * vs.Hl * vs.Wl * VC;
          const int c0 = PCNN_VERTEX_CHANNELS * l;
          // (round 5) the triple's twelve tap loads together: hv_scatter's ISA had 53 `s_waitcnt vmcnt(0)` for 62 loads
          const float b0 = vs.bias[c0], b1 = vs.bias[c0 + 1], b2 = vs.bias[c0 + 2];
          bilinear_at3(zb, ty, tx, vs.Wl, VC, c0, u, v, logd);
          u = u + b0; v = v + b1; logd = logd + b2;
        }
        float d = exp_f32(logd);
        float n1 = sqrt_rn(u * u + v * v);
        float thr = project_box(extents, l, fx, fy, px, py, d);
        float rn1 = (n1 >= 1e-15f && n1 <= 1e15f) ? div_rn(1.0f, n1) : __builtin_nanf("");
        HvRec R;
        R.a = make_float4((float)x, (float)y, thr, rn1);
        R.b = make_float4(u, v, n1, d);
        R.c = cone_coefficients(u, v, n1, rn1, inlier);
        rec[(size_t)n * reccap + ro + my_rank / skip] = R;
        // largest vote window of the class (hv_vote's band search): positive finite floats order like their bits
        if (thr == thr && thr > 0.f) atomicMax(&s_kmax[l], __float_as_int(fminf(thr, 1e6f)));
      }
    }
  }
  __syncthreads();
  if (tid < C && s_kmax[tid] > 0) atomicMax(&kmax_g[n * C + tid], s_kmax[tid]);   // one global atomic per class per block
}

// ---------------------------------------------------------------------------------------------
// hv_vote: one workgroup = (image, class slot, BAND of HV_BAND / HV_WPR Hough rows), HV_WPR waves per row.
//
// The round-1 kernel tiled the Hough space 32 x 32 and every one of the ~24 000 active tile blocks
// re-streamed its class' ~1500 records to cull them (1.1 GB of L2 reads per launch, 112 M (record,
// tile row) interval evaluations, 2 LDS atomics each on a 33-word row). Rows are the natural unit
// of the interval form — a record's vote cone cut by a Hough row is ONE interval — so:
//   * the records of a class are emitted in ascending pixel order, i.e. sorted by y: the records that
//     can reach a band of rows are a CONTIGUOUS range [lo, hi) of the list, found by a 64-ary search
//     on y with the class' largest vote window (`kmax`, a max-reduction in hv_scatter) — no cull
//     pass, and a band reads only its own range, once, through LDS for all its 8 rows;
//   * a (record, row) pair is evaluated once for the whole row (not once per 32-cell tile the window
//     overlaps): 24 M evaluations per launch instead of 112 M;
//   * each wave owns its row's difference array of W + 1 counters in LDS: interval -> +1 / -1, one
//     wave-wide prefix sum turns differences into votes, the row maximum (first cell among equals)
//     falls out of the same pass.
// Votes are integers: order-free, bit-identical to the per-cell definition (compute_hough_kernel
// :253-294); the exact IEEE predicate still decides the cells within the rounding uncertainty of an
// interval end and the records without a closed form.
constexpr int HV_BAND = 8;            // waves per workgroup
constexpr int HV_WPR = 2;             // waves per Hough row (1: 213 us, 2: 186 us, 4: 209 us per 16-frame launch; 16 waves x 2: 222 us)
constexpr int HV_RCHUNK = 256;        // records staged in LDS per round

__device__ __forceinline__ void diff_add(int* row, int c_lo, int c_hi)
{
  atomicAdd(row + c_lo, 1);
  atomicAdd(row + c_hi + 1, -1);
}

// Votes of one record for the Hough row `yrow`: columns [0, W) of the row's difference array `drow`.
__device__ __forceinline__ void vote_row(const float4 a, const float4 b, const float4 c, int yrow, int W,
                                         float inlier, int* drow)
{
  const int x = (int)a.x, y = (int)a.y;
  const int dyi = yrow - y;
  const float dy = (float)dyi;
  if (!(fabsf(dy) < a.z)) return;                         // |dy| < thr (.cu.cc:286-288)
  // columns with |dx| < thr:  |dx| <= kx,  kx = ceil(thr) - 1
  const float kxf = fminf(ceilf(a.z) - 1.f, 65536.f);
  const int kx = (int)kxf;
  const int w_lo = max(x - kx, 0), w_hi = min(x + kx, W - 1);   // vote window, clipped to the row
  if (w_lo > w_hi) return;
  const int mode = (int)c.w;
  // (round 6) a cone that opens away from this row: mode 1 is the interval BETWEEN the two roots and exists only on the side
  // of the pixel the direction points to (dy * g > 0, below) — for every other row it is L = 1 > R = -1, no sure cell and no
  // uncertain one. ~70 % of the records are mode 1 and half of their rows are on the wrong side; the records of a 64-record
  // slice are neighbours in the image and point the same way, so whole waves leave here instead of walking the
  // closed form to an empty result (88 M vector instructions per launch, the kernel's actual bound: bench.py `valu_frac`).
  if (mode == 1 && dyi != 0 && !(dy * c.z > 0.f)) return;
  bool per_cell = (mode == 0);
  // sure interval [lo, hi] of columns and two ranges [ul0, ul1], [ur0, ur1] of columns too close to
  // an interval end to trust the closed form: those are decided by the exact predicate. The f32
  // predicate can move an end by <= 5 ulp(q) / |dq/dx| = 6.9e-7 r^2 / |dy| cells (r = distance
  // pixel -> cell); the uncertain half-width is 0.01 + 2e-6 r^2 / |dy|.
  int lo = 1, hi = 0, ul0 = 1, ul1 = 0, ur0 = 1, ur1 = 0;
  if (!per_cell) {
    if (dyi == 0) {
      // q = sign(dx) * u / n1 for every dx != 0; dx = 0 is 0/0 = NaN -> never a vote
      const float sgn = b.x * a.w;
      if (sgn > inlier + 1e-5f) { lo = x + 1; hi = 0x3fffffff; }
      else if (sgn < -(inlier + 1e-5f)) { lo = -0x3fffffff; hi = x - 1; }
      else if (fabsf(sgn) < inlier - 1e-5f) { /* empty */ }
      else per_cell = true;
    } else {
      const float r1 = dy * c.x, r2 = dy * c.y;
      const float rl = fminf(r1, r2), rh = fmaxf(r1, r2);
      const float BIG = 3.0e9f;
      float L, R;  // open interval (L, R) in dx; +-BIG = unbounded
      if (mode == 1) {
        if (dy * c.z > 0.f) { L = rl; R = rh; } else { L = 1.f; R = -1.f; }
      } else if (mode == 2) { L = rh; R = BIG; }
      else { L = -BIG; R = rl; }
      // an end more than 1e5 cells away is outside every vote window (kx <= 65536): unbounded on that
      // side, or no cell at all — and no "uncertain" band around it
      const float FAR = 1.0e5f;
      if (L > FAR || R < -FAR) { L = 1.f; R = -1.f; }
      else { if (L < -FAR) L = -BIG; if (R > FAR) R = BIG; }
      if (L <= R) {
        const float ady = fabsf(dy), rdy = 2e-6f / ady;
        const float dL = L > -BIG ? 0.01f + rdy * (L * L + dy * dy) : 0.f;
        const float dR = R < BIG ? 0.01f + rdy * (R * R + dy * dy) : 0.f;
        const float CL = 70000.f;
        const float slo = fminf(fmaxf(floorf(L + dL) + 1.f, -CL), CL);   // first sure dx
        const float shi = fminf(fmaxf(ceilf(R - dR) - 1.f, -CL), CL);    // last sure dx
        lo = x + (int)slo;
        hi = x + (int)shi;
        if (L > -BIG) { ul0 = x + (int)fminf(fmaxf(ceilf(L - dL), -CL), CL); ul1 = lo - 1; }
        if (R < BIG) { ur0 = hi + 1; ur1 = x + (int)fminf(fmaxf(floorf(R + dR), -CL), CL); }
        if (lo > hi) {  // no sure cell: one exact range from the lowest to the highest candidate
          ul0 = L > -BIG ? ul0 : w_lo;
          ul1 = R < BIG ? ur1 : w_hi;
          ur0 = 1; ur1 = 0;
        }
      }
    }
  }
  if (per_cell) {
    for (int cxa = w_lo; cxa <= w_hi; cxa++)
      if (angle_pass_exact(b.x, b.y, b.z, (float)(cxa - x), dy, inlier)) diff_add(drow, cxa, cxa);
    return;
  }
  lo = max(lo, w_lo);
  hi = min(hi, w_hi);
  if (lo <= hi) diff_add(drow, lo, hi);
  // the (at most a few) cells near the two ends, ONE loop for both ranges: a wave takes as many trips as
  // its busiest lane, and most lanes have zero or one such cell
  ul0 = max(ul0, w_lo); ul1 = min(ul1, w_hi);
  ur0 = max(ur0, w_lo); ur1 = min(ur1, w_hi);
  int cxa = ul0 <= ul1 ? ul0 : ur0;
  int endc = ul0 <= ul1 ? ul1 : ur1;
  bool second = !(ul0 <= ul1);
  while (cxa <= endc) {
    if (angle_pass_exact(b.x, b.y, b.z, (float)(cxa - x), dy, inlier)) diff_add(drow, cxa, cxa);
    cxa++;
    if (cxa > endc && !second) { second = true; cxa = ur0; endc = ur1; }
  }
}

// Difference array of one Hough row -> votes, by ONE wave: lane owns a contiguous strip of the row, a wave-wide exclusive
// scan of the strip sums carries the votes in; the row maximum (most votes, lowest column among equals) comes out of the same
// pass. (round 6) The strip is an ODD number of cells (lane l starts at bank l * per: an even stride — 10 for W = 640 — puts
// four lanes on every bank, an odd one two, the minimum for 64 lanes on 32 banks); strips of up to 16 cells stay in registers
// between the two passes; and the votes go back to LDS only when the caller wants the Hough space itself (`hrow`,
// threshold_vote > 0): the default path needs the maximum alone. 25 us of the launch were this epilogue (ablation).
__device__ __forceinline__ void scan_row_votes(int* drow, int W, int yrow, float* hrow, int2* rowmax_out)
{
  const int lane = lane_id();
  const int per = ((W + 63) / 64) | 1;
  const int c0 = lane * per, c1 = min(c0 + per, W);
  int bv = -1, bi = 0x7fffffff;
  if (per <= 16) {
    int d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = (i < per && c0 + i < W) ? drow[c0 + i] : 0;
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += d[i];
    int pre = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(pre, off);
      if (lane >= off) pre += o;
    }
    int acc = pre - sum;   // votes carried into the strip
#pragma unroll
    for (int i = 0; i < 16; i++) {
      acc += d[i];
      d[i] = acc;
      if (i < per && c0 + i < W && acc > bv) { bv = acc; bi = yrow * W + c0 + i; }
    }
    if (hrow) {
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (i < per && c0 + i < W) drow[c0 + i] = d[i];
    }
  } else {
    int sum = 0;
    for (int cx = c0; cx < c1; cx++) sum += drow[cx];
    int pre = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(pre, off);
      if (lane >= off) pre += o;
    }
    int acc = pre - sum;
    for (int cx = c0; cx < c1; cx++) {
      acc += drow[cx];
      drow[cx] = acc;      // votes (same-wave LDS: in order)
      if (acc > bv) { bv = acc; bi = yrow * W + cx; }
    }
  }
  if (hrow) {
    __builtin_amdgcn_wave_barrier();
    for (int cx = lane; cx < W; cx += 64) hrow[cx] = (float)drow[cx];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int ov = __shfl_xor(bv, off), oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) *rowmax_out = make_int2(bv, bi);
}

// hv_order: what hv_vote needs to start a workgroup without searching or idling (round 6).
//   workgroup 0      the live (image, class slot) pairs of the batch, heaviest class first: a counting sort on
//                    floor(log2(records)) — the order inside a bucket is whatever the atomics give, which is fine: hv_vote
//                    only wants the long items dispatched before the short ones, every order gives the same votes;
//   workgroup 1 + p  pair p = (image, slot): rowstart[y] = index of the class' first record with pixel row >= y, for
//                    y = 0 .. H (the records are sorted by row): the range of records that can reach a band of Hough rows
//                    is two table reads instead of two 64-ary searches — two dependent trips to L2 and a workgroup
//                    barrier in front of every one of hv_vote's 9 000 workgroups (2.9 us each in the trace).
constexpr int HV_ORDER_THREADS = 256;
constexpr int HV_ORDER_SPLIT = 8;
__global__ __launch_bounds__(HV_ORDER_THREADS) void hv_order_kernel(const HvRec* __restrict__ rec, const int* __restrict__ nslots_g,
                                                                    const int* __restrict__ slots_g, const int* __restrict__ tot_g,
                                                                    const int* __restrict__ recoff_g, int* __restrict__ order,
                                                                    int* __restrict__ rowstart, int B, int C, int H, int skip, int reccap)
{
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    // HV_ORDER_SPLIT workgroups per pair, interleaved over the records: a class of 6 000 records is three trips to L2 per
    // thread instead of twenty-four one after the other (one workgroup per pair: 22 us for this kernel)
    const int pair = (blockIdx.x - 1) / HV_ORDER_SPLIT, sub = (blockIdx.x - 1) % HV_ORDER_SPLIT;
    const int n = pair / (C - 1), s = pair - n * (C - 1);
    if (s >= nslots_g[n]) return;
    const int cls = slots_g[n * C + s];
    const int m = (tot_g[n * C + cls] + skip - 1) / skip;
    const HvRec* r0 = rec + (size_t)n * reccap + recoff_g[n * C + cls];
    int* rs = rowstart + (size_t)pair * (H + 1);
    // record i starts the rows (y of record i - 1, y of record i]; "record m" = the end of the list starts every row left
    for (int i = sub * HV_ORDER_THREADS + tid; i <= m; i += HV_ORDER_THREADS * HV_ORDER_SPLIT) {
      const int y_prev = i > 0 ? (int)r0[i - 1].a.y : -1;
      const int y_cur = i < m ? (int)r0[i].a.y : H;
      for (int y = max(y_prev + 1, 0); y <= min(y_cur, H); y++) rs[y] = i;
    }
    return;
  }
  __shared__ int s_hist[32], s_base[32];
  if (tid < 32) s_hist[tid] = 0;
  __syncthreads();
  const int total = B * (C - 1);
  for (int i = tid; i < total; i += HV_ORDER_THREADS) {
    const int n = i / (C - 1), s = i - n * (C - 1);
    if (s < nslots_g[n]) {
      const int m = (tot_g[n * C + slots_g[n * C + s]] + skip - 1) / skip;
      atomicAdd(&s_hist[31 - __clz(max(m, 1))], 1);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 31; b >= 0; b--) { s_base[b] = run; run += s_hist[b]; }
    order[0] = 0;
    order[1] = run;    // live pairs
  }
  __syncthreads();
  for (int i = tid; i < total; i += HV_ORDER_THREADS) {
    const int n = i / (C - 1), s = i - n * (C - 1);
    if (s < nslots_g[n]) {
      const int m = (tot_g[n * C + slots_g[n * C + s]] + skip - 1) / skip;
      order[4 + atomicAdd(&s_base[31 - __clz(max(m, 1))], 1)] = i;
    }
  }
}

// hv_vote (round 6: heaviest class first, no empty dispatches in between). One workgroup = one ITEM = (image, class slot,
// band of `nrows` Hough rows), and workgroup i takes item i of hv_order's list: the hardware hands workgroups out in index
// order, so the bands of the class with the most records start first and the short items fill the end of the launch; the
// workgroups past the last live item (16 of 21 class slots are empty in a typical frame) leave at once, all of them AFTER
// the live ones. A trace of the round-5 launch (one workgroup per (band, slot, image) in grid order, wall_clock64 stamps:
// DESIGN §3.1) showed what its 197 us were: 9 120 live workgroups whose durations sum to 108 ms·wg — 105 us of a full
// chip —, 750-800 of 1 024 slots occupied while 31 200 empty workgroups were dispatched in between, and a 50 us TAIL of the
// one class with 5 113 records (50-60 us per band), which happened to belong to the last image.
// (A persistent grid pulling items off an atomic counter measured 269 us: one device-scope atomic per item, 10 000 per
// launch on one address, and the loop-carried state pushed the kernel into spills.)
__global__ __launch_bounds__(64 * HV_BAND, 8) void hv_vote_kernel(
    const HvRec* __restrict__ rec, const int* __restrict__ slots_g,
    const int* __restrict__ recoff_g, const int* __restrict__ kmax_g, const int* __restrict__ order,
    const int* __restrict__ rowstart, int2* __restrict__ rowmax,
    float* __restrict__ hs, int H, int W, int C, float inlier, int reccap, int need_hs, int wpr, int nbands)
{
  const int item = blockIdx.x;
  if (item >= order[1] * nbands) return;
  const int pair = order[4 + item / nbands], band = item % nbands;
  const int n = pair / (C - 1), s = pair - n * (C - 1);
  // `wpr` waves share a row (each takes every wpr-th 64-record slice of a chunk): the launch ends with its
  // longest workgroup — a band through the middle of a large object — so the records of a row are spread
  // over more lanes rather than the band over more rows
  const int nwaves = blockDim.x >> 6;
  const int nrows = nwaves / wpr;      // rows of this band (fewer for very wide images)
  const int cls = slots_g[n * C + s];
  const HvRec* r0 = rec + (size_t)n * reccap + recoff_g[n * C + cls];

  extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [HV_BAND][W + 1] difference arrays
  // two chunk buffers: chunk i + 1 travels L2 -> registers while chunk i is voted and lands in the OTHER buffer: one
  // workgroup barrier per chunk, and no trip to memory is waited for with nothing else to do
  __shared__ float4 sA[2][HV_RCHUNK], sB[2][HV_RCHUNK], sC[2][HV_RCHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W1 = W + 1;
  const int y0 = band * nrows;
  const int row = wave % nrows, part = wave / nrows;
  const int yrow = y0 + row;
  int* drow = s_dyn + row * W1;

  // records that can reach the band: pixel rows y0 - kmax .. y0 + nrows - 1 + kmax, kmax = the class' largest window
  // half-size ceil(thr) - 1 (|dy| < thr  <=>  |dy| <= ceil(thr) - 1); hv_order's table turns the two rows into indices
  const float tmax = __int_as_float(kmax_g[n * C + cls]);   // max thr of the class' records (>= 0, finite)
  const int kmax = (int)fminf(ceilf(tmax) - 1.f, 65536.f);
  const int* rs = rowstart + (size_t)pair * (H + 1);
  const int lo = rs[min(max(y0 - kmax, 0), H)], hi = rs[min(max(y0 + nrows + kmax, 0), H)];
  if (lo >= hi) {
    // no record reaches the band (a third of the live items): every cell has 0 votes, the row maximum is its first cell
    if (tid < nrows && y0 + tid < H) rowmax[((size_t)n * (C - 1) + s) * H + y0 + tid] = make_int2(0, (y0 + tid) * W);
    if (need_hs) {
      float* h0 = hs + ((size_t)n * (C - 1) + s) * ((size_t)H * W) + (size_t)y0 * W;
      const int cells = min(nrows, H - y0) * W;
      for (int i = tid; i < cells; i += 64 * nwaves) h0[i] = 0.f;
    }
    return;
  }
  for (int i = tid; i < nrows * W1; i += 64 * nwaves) s_dyn[i] = 0;   // (the first chunk's barrier comes before any vote)

  // thread t < CH carries record t of the chunk in flight (CH = HV_RCHUNK, or the workgroup's size when a very wide
  // image leaves it fewer threads than that)
  const int CH = min(HV_RCHUNK, 64 * nwaves);
  float4 pa, pb, pc;
  if (tid < CH && lo + tid < hi) { pa = r0[lo + tid].a; pb = r0[lo + tid].b; pc = r0[lo + tid].c; }
  int buf = 0;
  for (int b0 = lo; b0 < hi; b0 += CH, buf ^= 1) {
    const int cnt = min(CH, hi - b0);
    if (tid < cnt) { sA[buf][tid] = pa; sB[buf][tid] = pb; sC[buf][tid] = pc; }
    __syncthreads();                     // (also: every wave has left the votes of chunk i - 1, which read the other buffer)
    const int kn = b0 + CH + tid;
    if (tid < CH && kn < hi) { pa = r0[kn].a; pb = r0[kn].b; pc = r0[kn].c; }
    if (yrow < H)
      for (int k = part * 64 + lane; k < cnt; k += 64 * wpr) vote_row(sA[buf][k], sB[buf][k], sC[buf][k], yrow, W, inlier, drow);
  }
  __syncthreads();                       // the row's other waves are done adding

  if (yrow >= H || part != 0) return;
  scan_row_votes(drow, W, yrow, need_hs ? hs + ((size_t)n * (C - 1) + s) * ((size_t)H * W) + (size_t)yrow * W : nullptr,
                 rowmax + ((size_t)n * (C - 1) + s) * H + yrow);
}

constexpr int WCD_CAP = 1024;  // floats of LDS per wave for wave_cell_data

// One wave: hough_data of a single cell — second half of compute_hough_kernel (:296-331) plus the
// depth sum of the first half (:269-294) in canonical (ascending pixel) order.
__device__ void wave_cell_data(const HvRec* __restrict__ r0, int m, int cx, int cy, int cls,
                               const float* __restrict__ extents, float fx, float fy, float px,
                               float py, float inlier, float* s_buf, float& votes_out,
                               float& dist, float& bh2, float& bw2)
{
  const int lane = lane_id();
  const float cxf = (float)cx, cyf = (float)cy;
  float sumd = 0.f;
  int cnt = 0, fill = 0;
  // 4 x 64 records per trip: the 8 loads are issued before any use, so a trip costs one L2 round
  // trip; ballots are consumed in ascending record order (k, then lane) = canonical pixel order
  for (int b0 = 0; b0 < m; b0 += 256) {
    float4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ri = min(b0 + k * 64 + lane, m - 1);
      a[k] = r0[ri].a;
      b[k] = r0[ri].b;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ri = b0 + k * 64 + lane;
      const float dx = cxf - a[k].x, dy = cyf - a[k].y;
      const float d = b[k].w;
      const bool pass = ri < m && fabsf(dx) < a[k].z && fabsf(dy) < a[k].z &&
                        angle_pass_exact(b[k].x, b[k].y, b[k].z, dx, dy, inlier);
      // voters' depths go to this wave's LDS strip in record order; lane 0 adds them up one by
      // one (the reference's `distance += d` is a sequential f32 sum). A shuffle per voter costs
      // ~10x more than an LDS read here.
      const unsigned long long mask = __ballot(pass);
      if (pass) s_buf[fill + __popcll(mask & lanemask_lt())] = d;
      fill += __popcll(mask);
      __builtin_amdgcn_wave_barrier();  // same-wave LDS: in order in hardware; keep the compiler from reordering
      if (fill > WCD_CAP - 64) {
        if (lane == 0)
          for (int i = 0; i < fill; i++) sumd += s_buf[i];
        cnt += fill;
        fill = 0;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (lane == 0)
    for (int i = 0; i < fill; i++) sumd += s_buf[i];
  cnt += fill;
  sumd = __shfl(sumd, 0);
  votes_out = (float)cnt;
  dist = 0.f; bh2 = 0.f; bw2 = 0.f;
  if (cnt > 0) {
    dist = div_rn(sumd, (float)cnt);
    float thr = project_box(extents, cls, fx, fy, px, py, dist);
    float bw = -1.f, bh = -1.f;
    for (int b0 = 0; b0 < m; b0 += 256) {
      float4 a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int ri = min(b0 + k * 64 + lane, m - 1);
        a[k] = r0[ri].a;
        b[k] = r0[ri].b;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int ri = b0 + k * 64 + lane;
        const float dx = cxf - a[k].x, dy = cyf - a[k].y;
        const float ax = fabsf(dx), ay = fabsf(dy);
        if (ri < m && ax < thr && ay < thr && angle_pass_exact(b[k].x, b[k].y, b[k].z, dx, dy, inlier)) {
          bw = fmaxf(bw, ax);
          bh = fmaxf(bh, ay);
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      bw = fmaxf(bw, __shfl_xor(bw, off));
      bh = fmaxf(bh, __shfl_xor(bh, off));
    }
    bh2 = 2 * bh;
    bw2 = 2 * bw;
  }
}

// ---------------------------------------------------------------------------------------------
// 1024 threads (round 4; 256 before): the two sweeps over a class' ~1500-3000 records are trips of one record per thread
// with two barriers each — latency, not work — so four times the threads is a quarter of the trips (74 -> ~35 us).
constexpr int HV_SEL_NT = 1024, HV_SEL_NW = HV_SEL_NT / 64;
__global__ __launch_bounds__(HV_SEL_NT) void hv_select_kernel(
    const HvRec* __restrict__ rec, const int* __restrict__ nslots_g,
    const int* __restrict__ slots_g, const int* __restrict__ tot_g,
    const int* __restrict__ recoff_g, const int2* __restrict__ tilemax,
    const float* __restrict__ extents, const float* __restrict__ meta, HvMax* __restrict__ maxima,
    int* __restrict__ nmax_g, int W, int C, int skip, float inlier, int ntiles, int reccap,
    int cap, int capmax, int num_meta)
{
  const int s = blockIdx.x, n = blockIdx.y;
  const int ns = nslots_g[n];
  if (s == 0 && threadIdx.x == 0) nmax_g[n] = ns < cap ? ns : cap;
  if (s >= ns || s >= cap) return;
  __shared__ int s_rv[HV_SEL_NW], s_ri[HV_SEL_NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int2* tm = tilemax + ((size_t)n * (C - 1) + s) * ntiles;
  int bv = -1, bi = 0x7fffffff;
  for (int t = tid; t < ntiles; t += HV_SEL_NT) {
    int2 e = tm[t];
    if (e.x > bv || (e.x == bv && e.y < bi)) { bv = e.x; bi = e.y; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int ov = __shfl_xor(bv, off), oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_rv[wave] = bv; s_ri[wave] = bi; }
  __syncthreads();
  for (int w2 = 0; w2 < HV_SEL_NW; w2++)
    if (s_rv[w2] > bv || (s_rv[w2] == bv && s_ri[w2] < bi)) { bv = s_rv[w2]; bi = s_ri[w2]; }

  // hough_data of the winning cell (second half of compute_hough_kernel :296-331 and the depth sum
  // of the first half :269-294). All threads evaluate the exact vote predicate; the depths of
  // the voters are compacted IN PIXEL ORDER into LDS and summed by one lane, because the reference
  // accumulates `distance += d` sequentially and float addition does not reassociate.
  constexpr int SEL_CAP = 4 * HV_SEL_NT;
  __shared__ __attribute__((aligned(16))) float s_d[SEL_CAP];
  __shared__ int s_wc[HV_SEL_NW];
  __shared__ float s_res[4];
  __shared__ float s_bw[HV_SEL_NW], s_bh[HV_SEL_NW];
  const int cls = slots_g[n * C + s];
  const int m = (tot_g[n * C + cls] + skip - 1) / skip;
  const HvRec* r0 = rec + (size_t)n * reccap + recoff_g[n * C + cls];
  const float* md = meta + (size_t)n * num_meta;
  const float fx = md[0], px = md[2], fy = md[4], py = md[5];
  const float cxf = (float)(bi % W), cyf = (float)(bi / W);
  float sumd = 0.f;  // meaningful in thread 0
  int total = 0, fill = 0;
  for (int b0 = 0; b0 < m; b0 += HV_SEL_NT) {
    const int ri = b0 + tid;
    bool pass = false;
    float d = 0.f;
    if (ri < m) {
      const float4 a = r0[ri].a, b = r0[ri].b;
      const float dx = cxf - a.x, dy = cyf - a.y;
      d = b.w;
      pass = fabsf(dx) < a.z && fabsf(dy) < a.z && angle_pass_exact(b.x, b.y, b.z, dx, dy, inlier);
    }
    const unsigned long long mask = __ballot(pass);
    if (lane == 0) s_wc[wave] = __popcll(mask);
    __syncthreads();
    int pos = fill + __popcll(mask & lanemask_lt());
    for (int w2 = 0; w2 < wave; w2++) pos += s_wc[w2];
    if (pass) s_d[pos] = d;
    for (int w2 = 0; w2 < HV_SEL_NW; w2++) fill += s_wc[w2];
    __syncthreads();
    if (fill > SEL_CAP - HV_SEL_NT || b0 + HV_SEL_NT >= m) {
      if (tid == 0) {
        // the ordered sum, 16 depths per trip: four independent 128-bit LDS reads, then the 16 additions in pixel order
        // (one ds_read_b32 + wait per addend made this loop ~45 of the launch's 88 us)
        int i = 0;
        for (; i + 16 <= fill; i += 16) {
          const float4 q0 = *reinterpret_cast<const float4*>(&s_d[i]), q1 = *reinterpret_cast<const float4*>(&s_d[i + 4]);
          const float4 q2 = *reinterpret_cast<const float4*>(&s_d[i + 8]), q3 = *reinterpret_cast<const float4*>(&s_d[i + 12]);
          sumd += q0.x; sumd += q0.y; sumd += q0.z; sumd += q0.w;
          sumd += q1.x; sumd += q1.y; sumd += q1.z; sumd += q1.w;
          sumd += q2.x; sumd += q2.y; sumd += q2.z; sumd += q2.w;
          sumd += q3.x; sumd += q3.y; sumd += q3.z; sumd += q3.w;
        }
        for (; i < fill; i++) sumd += s_d[i];
      }
      total += fill;
      fill = 0;
      __syncthreads();
    }
  }
  float dist = 0.f, bh2 = 0.f, bw2 = 0.f;
  if (tid == 0) s_res[0] = total > 0 ? div_rn(sumd, (float)total) : 0.f;
  __syncthreads();
  if (total > 0) {
    dist = s_res[0];
    const float thr = project_box(extents, cls, fx, fy, px, py, dist);
    float bw = -1.f, bh = -1.f;
    for (int ri = tid; ri < m; ri += HV_SEL_NT) {
      const float4 a = r0[ri].a, b = r0[ri].b;
      const float dx = cxf - a.x, dy = cyf - a.y;
      if (angle_pass_exact(b.x, b.y, b.z, dx, dy, inlier)) {
        const float ax = fabsf(dx), ay = fabsf(dy);
        if (ax < thr && ay < thr) { bw = fmaxf(bw, ax); bh = fmaxf(bh, ay); }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      bw = fmaxf(bw, __shfl_xor(bw, off));
      bh = fmaxf(bh, __shfl_xor(bh, off));
    }
    if (lane == 0) { s_bw[wave] = bw; s_bh[wave] = bh; }
    __syncthreads();
    bw = s_bw[0]; bh = s_bh[0];
    for (int w2 = 1; w2 < HV_SEL_NW; w2++) { bw = fmaxf(bw, s_bw[w2]); bh = fmaxf(bh, s_bh[w2]); }
    bh2 = 2 * bh;
    bw2 = 2 * bw;
  }
  if (tid == 0) {
    HvMax e;
    e.cls = cls; e.idx = bi; e.votes = (float)total; e.dist = dist; e.bh2 = bh2; e.bw2 = bw2;
    e.pad0 = e.pad1 = 0;
    maxima[(size_t)n * capmax + s] = e;
  }
}

// ---------------------------------------------------------------------------------------------
// threshold_vote > 0, step 1: the conditions of compute_max_indexes_kernel (:351-367) that need
// only the votes — v > threshold and no strictly greater cell in the 7x7 neighbourhood — per
// 32x32 tile with a 3-cell halo staged in LDS; tiles whose maximum (known from hv_vote) is not above
// the threshold are cleared without reading the Hough space.
__global__ __launch_bounds__(256) void hv_lmflag_kernel(const int* __restrict__ nslots_g,
                                                        const int2* __restrict__ tilemax,
                                                        const float* __restrict__ hs,
                                                        unsigned char* __restrict__ flags, int H, int W,
                                                        int C, float vote_thr, int ntx, int ntiles)
{
  const int n = blockIdx.z, s = blockIdx.y, tile = blockIdx.x;
  if (s >= nslots_g[n]) return;
  constexpr int HALO = 3, TS = HV_TILE + 2 * HALO;
  __shared__ float s_h[TS][TS + 1];
  const int tid = threadIdx.x;
  const int tx0 = (tile % ntx) * HV_TILE, ty0 = (tile / ntx) * HV_TILE;
  const size_t base = ((size_t)n * (C - 1) + s) * ((size_t)H * W);
  // `tilemax` holds per-ROW maxima: the tile is live when one of its rows is (a superset of the exact test)
  bool live = false;
  {
    const int2* rm = tilemax + ((size_t)n * (C - 1) + s) * H;
    const int yy = ty0 + (tid & 31);
    const bool mine = tid < HV_TILE && yy < H && (float)rm[yy].x > vote_thr;
    live = __syncthreads_or(mine ? 1 : 0) != 0;
  }
  if (live) {
    for (int i = tid; i < TS * TS; i += 256) {
      const int ly = i / TS, lx = i - ly * TS;
      const int y = ty0 + ly - HALO, x = tx0 + lx - HALO;
      s_h[ly][lx] = (x >= 0 && x < W && y >= 0 && y < H) ? hs[base + (size_t)y * W + x] : -1.f;  // votes >= 0
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int lx = tid & 31, ly = (tid >> 5) + 8 * j;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) continue;
    unsigned char f = 0;
    if (live) {
      const float v = s_h[ly + HALO][lx + HALO];
      if (v > vote_thr) {
        bool greater = false;
#pragma unroll
        for (int dy = -HALO; dy <= HALO; dy++)
#pragma unroll
          for (int dx = -HALO; dx <= HALO; dx++) greater |= s_h[ly + HALO + dy][lx + HALO + dx] > v;
        f = greater ? 0 : 1;
      }
    }
    flags[base + (size_t)y * W + x] = f;
  }
}

// ---------------------------------------------------------------------------------------------
// threshold_vote > 0: compute_max_indexes_kernel (:335-383), one thread per Hough cell.
__global__ __launch_bounds__(256) void hv_localmax_kernel(
    const HvRec* __restrict__ rec, const int* __restrict__ nslots_g,
    const int* __restrict__ slots_g, const int* __restrict__ tot_g,
    const int* __restrict__ recoff_g, const float* __restrict__ hs,
    const unsigned char* __restrict__ flags, const float* __restrict__ extents,
    const float* __restrict__ meta, int* __restrict__ chunkcnt, HvMax* __restrict__ chunkcand, int H, int W, int C, int skip,
    float inlier, float vote_thr, float per_thr, int reccap, int capmax, int cap, int nlm,
    int num_meta)
{
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int HW = H * W;
  const int ns = nslots_g[n];
  const long long ncell = (long long)ns * HW;
  const long long cbase = (long long)chunk * LM_CHUNK;
  if (cbase >= ncell) return;
  __shared__ int s_wc[HV_SEL_NW];
  __shared__ int s_base;
  __shared__ float s_wcd[4][WCD_CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const float* md = meta + (size_t)n * num_meta;
  const float fx = md[0], px = md[2], fy = md[4], py = md[5];
  HvMax* out = chunkcand + ((size_t)n * nlm + chunk) * capmax;

  __shared__ int s_cs[256], s_ccell[256];
  __shared__ float s_cv[256];
  __shared__ HvMax s_res[4];
  __shared__ int s_acc[4];
  for (int r = 0; r < LM_CHUNK / 256 && s_base < cap; r++) {
    long long g = cbase + r * 256 + tid;
    int s = 0, cell = 0;
    float v = 0.f;
    bool hs_cand = false;  // v > threshold and 7x7 local maximum, from hv_lmflag_kernel
    if (g < ncell) {
      s = (int)(g / HW);
      cell = (int)(g - (long long)s * HW);
      const size_t gi = ((size_t)n * (C - 1) + s) * HW + cell;
      hs_cand = flags[gi] != 0;
      if (hs_cand) v = hs[gi];
    }
    // most 256-cell rounds hold no candidate: one coalesced byte read, then out
    if (!__syncthreads_or(hs_cand ? 1 : 0)) continue;
    // compact the round's candidates (ascending cell order) into LDS
    const unsigned long long mask = __ballot(hs_cand);
    if (lane == 0) s_wc[wave] = __popcll(mask);
    __syncthreads();
    int pos = __popcll(mask & lanemask_lt());
    for (int w2 = 0; w2 < wave; w2++) pos += s_wc[w2];
    const int nc = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    if (hs_cand) { s_cs[pos] = s; s_ccell[pos] = cell; s_cv[pos] = v; }
    __syncthreads();
    // hough_data of the candidates, 4 at a time (one wave each, depth sums in canonical order),
    // accepted in ascending order. Only the first `cap` accepted maxima of an image can ever be
    // emitted (index_size, .cu.cc:377-379,773-774), so a chunk stops as soon as it holds `cap`: a
    // plateau of equal votes (every cell a "no strictly greater neighbour" maximum) would otherwise
    // cost one full pass over the class' pixels per plateau cell.
    for (int i0 = 0; i0 < nc && s_base < cap; i0 += 4) {
      const int ci = i0 + wave;
      if (ci < nc) {
        const int cs = s_cs[ci], ccell = s_ccell[ci];
        const float cv = s_cv[ci];
        const int cls = slots_g[n * C + cs];
        const int m = (tot_g[n * C + cls] + skip - 1) / skip;
        const HvRec* r0 = rec + (size_t)n * reccap + recoff_g[n * C + cls];
        float votes, dist, bh2, bw2;
        wave_cell_data(r0, m, ccell % W, ccell / W, cls, extents, fx, fy, px, py, inlier, s_wcd[wave], votes, dist, bh2, bw2);
        if (lane == 0) {
          const bool ok = bh2 > 0 && bw2 > 0 && !(div_rn(cv, bh2 * bw2) < per_thr);
          s_acc[wave] = ok ? 1 : 0;
          HvMax e;
          e.cls = cls; e.idx = ccell; e.votes = cv; e.dist = dist; e.bh2 = bh2; e.bw2 = bw2;
          e.pad0 = e.pad1 = 0;
          s_res[wave] = e;
        }
      } else if (lane == 0) {
        s_acc[wave] = 0;
      }
      __syncthreads();
      if (tid == 0) {
        int b = s_base;
        for (int w2 = 0; w2 < 4; w2++)
          if (s_acc[w2]) { if (b < cap) out[b] = s_res[w2]; b++; }
        s_base = b;
      }
      __syncthreads();
    }
  }
  if (tid == 0) chunkcnt[(size_t)n * nlm + chunk] = s_base;
}

// first `cap` candidates of an image in ascending (slot, cell) order
__global__ __launch_bounds__(256) void hv_gather_kernel(const int* __restrict__ nslots_g,
                                                        const int* __restrict__ chunkcnt,
                                                        const HvMax* __restrict__ chunkcand,
                                                        HvMax* __restrict__ maxima,
                                                        int* __restrict__ nmax_g, int HW, int nlm,
                                                        int cap, int capmax)
{
  const int n = blockIdx.x, tid = threadIdx.x;
  const long long ncell = (long long)nslots_g[n] * HW;
  const int nact = (int)((ncell + LM_CHUNK - 1) / LM_CHUNK);
  const int per = (nact + 255) / 256;
  __shared__ int s_sum[256];
  const int c_lo = tid * per, c_hi = min(nact, c_lo + per);
  int sum = 0;
  for (int c = c_lo; c < c_hi; c++) sum += chunkcnt[(size_t)n * nlm + c];
  s_sum[tid] = sum;
  __syncthreads();
  int before = 0;
  for (int t = 0; t < tid; t++) before += s_sum[t];
  for (int c = c_lo; c < c_hi && before < cap; c++) {
    int k = chunkcnt[(size_t)n * nlm + c];
    for (int j = 0; j < k && before + j < cap; j++)
      maxima[(size_t)n * capmax + before + j] = chunkcand[((size_t)n * nlm + c) * capmax + j];
    before += k;
  }
  if (tid == 0) {
    int total = 0;
    for (int t = 0; t < 256; t++) total += s_sum[t];
    nmax_g[n] = total < cap ? total : cap;
  }
}

// ---------------------------------------------------------------------------------------------
// IoU (:73-82) and compute_box_overlap (:123-172; Eigen Quaternionf(w,x,y,z).toRotationMatrix())
__device__ float box_iou(const float* a, const float* b)
{
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return div_rn(interS, Sa + Sb - interS);
}

__device__ float compute_box_overlap(int cls, const float* __restrict__ extents, float fx, float fy,
                                     float px, float py, const float* __restrict__ pose,
                                     const float* box)
{
  float xHalf = (float)((double)extents[cls * 3 + 0] * 0.5);
  float yHalf = (float)((double)extents[cls * 3 + 1] * 0.5);
  float zHalf = (float)((double)extents[cls * 3 + 2] * 0.5);
  float qw = pose[6], qx = pose[7], qy = pose[8], qz = pose[9];
  float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  float txx = tx * qx, txy = ty * qx, txz = tz * qx;
  float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  float R00 = 1 - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
  float R10 = txy + twz, R11 = 1 - (txx + tzz), R12 = tyz - twx;
  float R20 = txz - twy, R21 = tyz + twx, R22 = 1 - (txx + tyy);
  float x1 = 1e8f, x2 = -1e8f, y1 = 1e8f, y2 = -1e8f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float bx = (i & 1) ? -xHalf : xHalf;
    float by = (i & 2) ? -yHalf : yHalf;
    float bz = (i & 4) ? -zHalf : zHalf;
    float X = (R00 * bx + R01 * by + R02 * bz) + pose[10];
    float Y = (R10 * bx + R11 * by + R12 * bz) + pose[11];
    float Z = (R20 * bx + R21 * by + R22 * bz) + pose[12];
    float x = fx * div_rn(X, Z) + px;
    float y = fy * div_rn(Y, Z) + py;
    x1 = fminf(x1, x);
    y1 = fminf(y1, y);
    x2 = fmaxf(x2, x);
    y2 = fmaxf(y2, y);
  }
  float box_gt[4] = {x1, y1, x2, y2};
  return box_iou(box, box_gt);
}

// compute_rois_kernel (:386-576). One WAVE per (image, maximum) (round 4; one thread per maximum before: ~210 dependent
// scattered stores and a serial walk over the ground-truth rows per thread, 39 us for two waves' worth of work):
// lane j < 9 writes row j of the maximum (the box, or its j-th jitter) and that row's pose / domain entries, the lanes
// walk the ground-truth rows 64 at a time (the FIRST matching row in ascending order wins, as in the serial loop),
// lanes 0..35 write the 9 x 4 target / weight entries. Same expressions per value, so the same bits.
__global__ __launch_bounds__(256) void hv_emit_kernel(
    const HvMax* __restrict__ maxima, const int* __restrict__ nmax_g,
    const float* __restrict__ extents, const float* __restrict__ meta,
    const float* __restrict__ gt, float* __restrict__ top_box, float* __restrict__ top_pose,
    float* __restrict__ top_target, float* __restrict__ top_weight, int* __restrict__ top_domain,
    int* __restrict__ num_rois, int B, int W, int C, int cap, int capmax, int num_meta, int num_gt,
    int is_train)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && wave == 0) {        // the row count (every launch, also with cap == 0)
    int acc = 0;
    if (cap > 0)
      for (int b0 = 0; b0 < B; b0 += 64) {
        int v = b0 + lane < B ? nmax_g[b0 + lane] : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc += v;
      }
    if (lane == 0) {
      const int rows = acc * (is_train ? 9 : 1);
      num_rois[0] = rows == 0 ? 1 : rows;  // dummy row, hough_voting_gpu_op.cc:381-383
      num_rois[1] = rows;
    }
  }
  if (cap <= 0) return;
  const int item = blockIdx.x * 4 + wave;
  const int n = item / cap, k = item - n * cap;
  if (n >= B || k >= nmax_g[n]) return;
  // rows of the images in front
  int off_n = 0;
  for (int b0 = 0; b0 < n; b0 += 64) {
    int v = b0 + lane < n ? nmax_g[b0 + lane] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    off_n += v;
  }
  const HvMax e = maxima[(size_t)n * capmax + k];
  const int rows_per = is_train ? 9 : 1;
  const int roi_index = (off_n + k) * rows_per;
  const float* md = meta + (size_t)n * num_meta;
  const float fx = md[0], px = md[2], fy = md[4], py = md[5];
  const int x = e.idx % W, y = e.idx / W;
  const int cls = e.cls;
  const float rx = div_rn((float)x - px, fx);
  const float ry = div_rn((float)y - py, fy);
  const float scale = 0.05f;
  const double kk = 0.5 + (double)scale;
  float box[4];
  box[0] = (float)((double)x - (double)e.bw2 * kk);
  box[1] = (float)((double)y - (double)e.bh2 * kk);
  box[2] = (float)((double)x + (double)e.bw2 * kk);
  box[3] = (float)((double)y + (double)e.bh2 * kk);
  if (lane < rows_per) {
    // row `lane`: 0 = the box itself, 1..8 = its jitters (:540-574)
    const float x1 = box[0], y1 = box[1], x2 = box[2], y2 = box[3];
    const float ww = x2 - x1, hh = y2 - y1;
    float r2 = x1, r3 = y1, r4 = x2, r5 = y2;
    if (lane > 0) {
      const int j = lane - 1;
      const int sxj = (j == 4 || j == 6) ? 0 : ((j == 1 || j == 3 || j == 7) ? 1 : -1);     // {-1,+1,-1,+1, 0,-1, 0,+1}
      const int syj = (j == 5 || j == 7) ? 0 : ((j == 2 || j == 3 || j == 6) ? 1 : -1);     // {-1,-1,+1,+1,-1, 0,+1, 0}
      if (sxj < 0) r2 = (float)((double)x1 - 0.05 * (double)ww);
      if (sxj > 0) r2 = (float)((double)x1 + 0.05 * (double)ww);
      if (syj < 0) r3 = (float)((double)y1 - 0.05 * (double)hh);
      if (syj > 0) r3 = (float)((double)y1 + 0.05 * (double)hh);
      r4 = r2 + ww;
      r5 = r3 + hh;
    }
    float* r = top_box + (size_t)(roi_index + lane) * 7;
    r[0] = (float)n;
    r[1] = (float)cls;
    r[2] = r2;
    r[3] = r3;
    r[4] = r4;
    r[5] = r5;
    r[6] = e.votes;
    float* p = top_pose + (size_t)(roi_index + lane) * 7;
    p[0] = 1; p[1] = 0; p[2] = 0; p[3] = 0;
    p[4] = rx * e.dist;
    p[5] = ry * e.dist;
    p[6] = e.dist;
    if (is_train) top_domain[roi_index + lane] = (num_gt == 0) ? 1 : 0;
  }
  if (!is_train) return;
  // the first ground-truth row (ascending) of this image and class whose projected box overlaps by more than 0.2 (:440-466)
  int found = -1;
  for (int i0 = 0; i0 < num_gt && found < 0; i0 += 64) {
    const int i = i0 + lane;
    bool hit = false;
    if (i < num_gt && cls == (int)gt[i * 13 + 1] && n == (int)gt[i * 13 + 0]) {
      const float overlap = compute_box_overlap(cls, extents, fx, fy, px, py, gt + i * 13, box);
      hit = (double)overlap > 0.2;
    }
    const unsigned long long hm = __ballot(hit);
    if (hm) found = i0 + (__ffsll((long long)hm) - 1);
  }
  if (found >= 0 && lane < 36) {
    const int j = lane >> 2, q = lane & 3;
    top_target[(size_t)(roi_index + j) * 4 * C + 4 * cls + q] = gt[found * 13 + 6 + q];
    top_weight[(size_t)(roi_index + j) * 4 * C + 4 * cls + q] = 1.f;
  }
}

int validate_common(int B, int H, int W, int C, int skip)
{
  PCNN_REQUIRE(B >= 1, PCNN_EINVAL, "hough_voting: label must be 3-dimensional with batch >= 1 (got %d)", B);
  PCNN_REQUIRE(H >= 1 && W >= 1, PCNN_EINVAL, "hough_voting: bad image size %dx%d", H, W);
  PCNN_REQUIRE((long long)H * W < (1ll << 24), PCNN_EINVAL, "hough_voting: image too large (%dx%d)", H, W);
  PCNN_REQUIRE(W <= 12000, PCNN_EINVAL, "hough_voting: images wider than 12000 are not supported (got %d): a Hough "
               "row lives in LDS", W);
  PCNN_REQUIRE(C >= 2 && C <= PCNN_MAX_CLASSES, PCNN_EINVAL,
               "hough_voting: num_classes must be in [2, %d] (got %d)", PCNN_MAX_CLASSES, C);
  PCNN_REQUIRE(skip >= 1, PCNN_EINVAL, "hough_voting: skip_pixels must be >= 1 (got %d)", skip);
  return PCNN_OK;
}

}  // namespace

extern "C" int pcnn_hough_voting_workspace_bytes(int batch, int height, int width,
                                                 int num_classes, float threshold_vote,
                                                 int skip_pixels, int rois_per_image, size_t* bytes)
{
  PCNN_REQUIRE(bytes != nullptr, PCNN_ENULL, "hough_voting_workspace_bytes: bytes is NULL");
  int st = validate_common(batch, height, width, num_classes, skip_pixels);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(rois_per_image >= 0, PCNN_EINVAL, "hough_voting: rois_per_image < 0");
  *bytes = hv_layout(batch, height, width, num_classes, threshold_vote > 0, skip_pixels, rois_per_image).total;
  return PCNN_OK;
}

// Diagnostics: byte offsets of the intermediate buffers inside the workspace, so tests can check
// every Hough cell (not only the maxima) against the oracle. offsets[0..7] = hough space
// (f32 [B][C-1][H*W], only when threshold_vote > 0; else SIZE_MAX), records, class totals,
// slot classes, slot counts, record offsets, tile maxima, record capacity per image (a count).
extern "C" int pcnn_hough_voting_debug_layout(int batch, int height, int width, int num_classes,
                                              float threshold_vote, int skip_pixels,
                                              int rois_per_image, size_t* offsets)
{
  PCNN_REQUIRE(offsets != nullptr, PCNN_ENULL, "hough_voting_debug_layout: offsets is NULL");
  int st = validate_common(batch, height, width, num_classes, skip_pixels);
  if (st != PCNN_OK) return st;
  const bool need_hs = threshold_vote > 0;
  const HvLayout L = hv_layout(batch, height, width, num_classes, need_hs, skip_pixels, rois_per_image);
  offsets[0] = need_hs ? L.off_hs : (size_t)-1;
  offsets[1] = L.off_rec;
  offsets[2] = L.off_tot;
  offsets[3] = L.off_slots;
  offsets[4] = L.off_nslots;
  offsets[5] = L.off_recoff;
  offsets[6] = L.off_tilemax;
  offsets[7] = (size_t)L.reccap;
  return PCNN_OK;
}

namespace {
int hough_fwd_impl(const int32_t* label, const HvVertexSrc vs,
                                     const float* extents, const float* meta, const float* gt,
                                     int B, int H, int W, int C, int num_meta, int num_gt,
                                     int is_train, float vote_thr, float per_thr, int skip,
                                     float inlier, int label_thr, int rois_per_image,
                                     int rows_capacity, float* top_box, float* top_pose,
                                     float* top_target, float* top_weight, int32_t* top_domain,
                                     int32_t* num_rois, void* workspace, size_t workspace_bytes,
                                     void* stream_)
{
  int st = validate_common(B, H, W, C, skip);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(rois_per_image >= 0, PCNN_EINVAL, "hough_voting: rois_per_image < 0");
  PCNN_REQUIRE(is_train >= 0, PCNN_EINVAL, "hough_voting: Need is_train >= 0, got %d", is_train);
  PCNN_REQUIRE(num_meta >= 6, PCNN_EINVAL, "hough_voting: meta_data needs >= 6 values per image (got %d)", num_meta);
  PCNN_REQUIRE(num_gt >= 0, PCNN_EINVAL, "hough_voting: num_gt < 0");
  PCNN_REQUIRE(label && (vs.full || (vs.z && vs.bias)) && extents && meta, PCNN_ENULL, "hough_voting: NULL input");
  PCNN_REQUIRE(gt || num_gt == 0, PCNN_ENULL, "hough_voting: gt is NULL but num_gt = %d", num_gt);
  PCNN_REQUIRE(top_box && top_pose && top_target && top_weight && top_domain && num_rois,
               PCNN_ENULL, "hough_voting: NULL output");
  const bool need_hs = vote_thr > 0;
  const HvLayout L = hv_layout(B, H, W, C, need_hs, skip, rois_per_image);
  PCNN_REQUIRE(rows_capacity >= 1 && (long long)rows_capacity >= (long long)B * L.cap * (is_train ? 9 : 1),
               PCNN_EINVAL, "hough_voting: outputs hold %d rows, batch %d x %d maxima x %d rows needs %lld",
               rows_capacity, B, L.cap, is_train ? 9 : 1, (long long)B * L.cap * (is_train ? 9 : 1));
  PCNN_REQUIRE(workspace && aligned16(workspace), PCNN_EWORKSPACE,
               "hough_voting: workspace NULL or not 16-byte aligned");
  PCNN_REQUIRE(workspace_bytes >= L.total, PCNN_EWORKSPACE,
               "hough_voting: workspace too small (%zu < %zu)", workspace_bytes, L.total);

  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  int* hist = (int*)(ws + L.off_hist);
  int* tot = (int*)(ws + L.off_tot);
  int* slots = (int*)(ws + L.off_slots);
  int* nslots = (int*)(ws + L.off_nslots);
  int* recoff = (int*)(ws + L.off_recoff);
  HvRec* rec = (HvRec*)(ws + L.off_rec);
  int2* tilemax = (int2*)(ws + L.off_tilemax);
  HvMax* maxima = (HvMax*)(ws + L.off_maxima);
  int* nmax = (int*)(ws + L.off_nmax);
  float* hs = need_hs ? (float*)(ws + L.off_hs) : nullptr;
  int* chunkcnt = need_hs ? (int*)(ws + L.off_chunkcnt) : nullptr;
  HvMax* chunkcand = need_hs ? (HvMax*)(ws + L.off_chunkcand) : nullptr;
  const int HW = H * W;

  ZeroJob zj;
  zj.p[0] = top_box;    zj.words[0] = (unsigned)rows_capacity * 7;
  zj.p[1] = top_pose;   zj.words[1] = (unsigned)rows_capacity * 7;
  zj.p[2] = top_target; zj.words[2] = (unsigned)rows_capacity * 4 * C;
  zj.p[3] = top_weight; zj.words[3] = (unsigned)rows_capacity * 4 * C;
  zj.p[4] = (float*)top_domain; zj.words[4] = (unsigned)rows_capacity;

  int* kmax = (int*)(ws + L.off_kmax);
  PCNN_LAUNCH(hv_hist_kernel, dim3(L.nchunk, B), dim3(256), 0, stream, label, hist, kmax, HW, C,
                     L.nchunk, zj);
  PCNN_LAUNCH(hv_scatter_kernel, dim3(L.nchunk, B), dim3(256), 0, stream, label, vs,
                     extents, meta, hist, tot, slots, nslots, recoff, kmax, rec, HW, W, C, L.nchunk,
                     skip, label_thr, num_meta, L.reccap, inlier);
  // rows per band: HV_BAND / HV_WPR, fewer when the difference arrays of W + 1 counters would not fit 48 KB of LDS
  const int wpr = HV_WPR;
  int band_rows = HV_BAND / wpr;
  while (band_rows > 1 && sizeof(int) * band_rows * (size_t)(W + 1) > 48 * 1024) band_rows--;
  const int nbands = (H + band_rows - 1) / band_rows;
  int* order = (int*)(ws + L.off_order);
  int* rowstart = (int*)(ws + L.off_rowstart);
  PCNN_LAUNCH(hv_order_kernel, dim3(1 + B * (C - 1) * HV_ORDER_SPLIT), dim3(HV_ORDER_THREADS), 0, stream, rec, nslots, slots, tot, recoff, order, rowstart,
              B, C, H, skip, L.reccap);
  const long long max_items = (long long)B * (C - 1) * nbands;
  PCNN_REQUIRE(max_items < (1ll << 31), PCNN_EINVAL, "hough_voting: B * (C - 1) * bands = %lld overflows the grid", max_items);
  PCNN_LAUNCH(hv_vote_kernel, dim3((unsigned)max_items), dim3(64 * band_rows * wpr), sizeof(int) * band_rows * (size_t)(W + 1), stream,
              rec, slots, recoff, kmax, order, rowstart, tilemax, hs, H, W, C, inlier, L.reccap, need_hs ? 1 : 0, wpr, nbands);
  if (!need_hs) {
    PCNN_LAUNCH(hv_select_kernel, dim3(C - 1, B), dim3(HV_SEL_NT), 0, stream, rec, nslots, slots,
                       tot, recoff, tilemax, extents, meta, maxima, nmax, W, C, skip, inlier,
                       H, L.reccap, L.cap, L.capmax, num_meta);
  } else {
    unsigned char* flags = (unsigned char*)(ws + L.off_flags);
    PCNN_LAUNCH(hv_lmflag_kernel, dim3(L.ntiles, C - 1, B), dim3(256), 0, stream, nslots, tilemax, hs,
                flags, H, W, C, vote_thr, L.ntx, L.ntiles);
    PCNN_LAUNCH(hv_localmax_kernel, dim3(L.nlm, B), dim3(256), 0, stream, rec, nslots,
                       slots, tot, recoff, hs, flags, extents, meta, chunkcnt, chunkcand, H, W, C, skip,
                       inlier, vote_thr, per_thr, L.reccap, L.capmax, L.cap, L.nlm, num_meta);
    PCNN_LAUNCH(hv_gather_kernel, dim3(B), dim3(256), 0, stream, nslots, chunkcnt,
                       chunkcand, maxima, nmax, HW, L.nlm, L.cap, L.capmax);
  }
  const int emit_items = B * (L.cap > 0 ? L.cap : 0);      // one wave per (image, maximum slot), four to a workgroup
  PCNN_LAUNCH(hv_emit_kernel, dim3(emit_items > 4 ? (emit_items + 3) / 4 : 1), dim3(256),
              0, stream, maxima, nmax, extents, meta,
                     gt, top_box, top_pose, top_target, top_weight, top_domain, num_rois, B, W, C,
                     L.cap, L.capmax, num_meta, num_gt, is_train);
  return check_launch("hough_voting_fwd");
}
}  // namespace

extern "C" int pcnn_hough_voting_fwd(const int32_t* label, const float* vertex,
                                     const float* extents, const float* meta, const float* gt,
                                     int B, int H, int W, int C, int num_meta, int num_gt,
                                     int is_train, float vote_thr, float per_thr, int skip,
                                     float inlier, int label_thr, int rois_per_image,
                                     int rows_capacity, float* top_box, float* top_pose,
                                     float* top_target, float* top_weight, int32_t* top_domain,
                                     int32_t* num_rois, void* workspace, size_t workspace_bytes,
                                     void* stream_)
{
  HvVertexSrc vs = {vertex, nullptr, nullptr, 0, 0, 0, 0};
  return hough_fwd_impl(label, vs, extents, meta, gt, B, H, W, C, num_meta, num_gt, is_train,
                        vote_thr, per_thr, skip, inlier, label_thr, rois_per_image, rows_capacity,
                        top_box, top_pose, top_target,
                        top_weight, top_domain, num_rois, workspace, workspace_bytes, stream_);
}

extern "C" int pcnn_hough_voting_lowres_fwd(const int32_t* label, const float* z, const float* bias,
                                            int kernel, int stride, const float* extents,
                                            const float* meta, const float* gt, int B, int H, int W,
                                            int C, int num_meta, int num_gt, int is_train,
                                            float vote_thr, float per_thr, int skip, float inlier,
                                            int label_thr, int rois_per_image, int rows_capacity,
                                            float* top_box, float* top_pose,
                                            float* top_target, float* top_weight,
                                            int32_t* top_domain, int32_t* num_rois, void* workspace,
                                            size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(z && bias, PCNN_ENULL, "hough_voting_lowres: NULL vertex field or bias");
  PCNN_REQUIRE(stride >= 1 && kernel >= stride && (kernel - stride) % 2 == 0 && kernel <= 4 * stride,
               PCNN_EINVAL,
               "hough_voting_lowres: need stride >= 1, stride <= kernel <= 4*stride and (kernel - stride) even (got k=%d s=%d)",
               kernel, stride);
  PCNN_REQUIRE(H >= 1 && W >= 1 && H % stride == 0 && W % stride == 0, PCNN_EINVAL,
               "hough_voting_lowres: label map %dx%d is not a multiple of the stride %d", H, W, stride);
  HvVertexSrc vs = {nullptr, z, bias, H / stride, W / stride, kernel, stride};
  return hough_fwd_impl(label, vs, extents, meta, gt, B, H, W, C, num_meta, num_gt, is_train,
                        vote_thr, per_thr, skip, inlier, label_thr, rois_per_image, rows_capacity,
                        top_box, top_pose, top_target,
                        top_weight, top_domain, num_rois, workspace, workspace_bytes, stream_);
}

extern "C" int pcnn_hough_voting_bwd(float* grad_label, float* grad_vertex, int B, int H, int W,
                                     int C, void* stream_)
{
  PCNN_REQUIRE(grad_label && grad_vertex, PCNN_ENULL, "hough_voting_bwd: NULL output");
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, PCNN_EINVAL, "hough_voting_bwd: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  // set_gradients, hough_voting_gpu_op.cu.cc:608-612
  int st = zero_async(grad_label, sizeof(float) * (size_t)B * H * W, stream, "hough_voting_bwd");
  if (st == PCNN_OK)
    st = zero_async(grad_vertex, sizeof(float) * (size_t)B * H * W * PCNN_VERTEX_CHANNELS * C, stream, "hough_voting_bwd");
  if (st != PCNN_OK) return st;
  return PCNN_OK;
}
