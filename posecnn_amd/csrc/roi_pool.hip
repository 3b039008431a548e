// roi_pool.hip — gfx950 ROI max pooling, NHWC (replaces TF1 ops "RoiPool"/"RoiPoolGrad",
// lib/roi_pooling_layer/roi_pooling_op.cc:306-347,384-461 + roi_pooling_op_gpu.cu.cc:20-254).
//
// Forward: separable, staged through LDS. Features are NHWC, so a 32-channel chunk of one spatial
// position is one 128 B line. A workgroup owns one (roi, bin row ph, channel chunk): phase 1 reduces
// every feature column the ROI covers over the bin row's h range into LDS as (max value, first h
// that attains it) — all 256 threads issue distinct 16 B loads, each feature line is read once per
// bin row instead of once per bin that touches it; phase 2 lets thread (pw, c) finish the bin with a
// horizontal scan of the staged columns. The reference walks each bin h-major with a strict `>`
// (roi_pooling_op_gpu.cu.cc:77-93), i.e. the first maximum in (h, w) order wins; "max value, then
// smallest h, then smallest w" over the staged columns selects the same element, so values and
// argmax stay bit-identical. ROIs wider than the LDS window fall back to reading columns from
// global memory inside phase 2 (same routine, no staging).
// The fused `add2` entry pools conv5_3 and conv4_3 and adds them (vgg16_convs.py:177-187) without
// writing either pooled tensor or any argmax (inference does not consume them); it keeps the
// register-only bin-per-workgroup form, which measures faster for a value-only max (see below).
//
// Backward: tile-binned ordered gather. A workgroup owns (image, row h, 8 consecutive w, 256
// channels); wave 0 compacts — in ascending ROI order, by ballot — the ROIs of that image whose
// rectangle touches the tile into an LDS list, then every thread walks the short list for its
// channel. The reference gathers over *all* ROIs per output element (:135-229); here the
// O(B*H*W*C*R) test collapses to one scan of the ROI table per tile while the summation order per
// element (roi, ph, pw ascending) — and therefore every bit of the result — is unchanged.
#include <cfloat>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int RP_THREADS = 256;
constexpr int RP_CHUNK = 32;        // channels per workgroup (one 128 B line per position)
constexpr int RP_LDS_WORDS = 4096;  // 16 KB window => 8 workgroups/CU keep all 32 wave slots busy

struct RoiGeom {
  int batch, cls, sw, sh, ew, eh;
  float bin_h, bin_w;
};

// roi_pooling_op_gpu.cu.cc:45-62
__device__ __forceinline__ RoiGeom roi_geometry(const float* __restrict__ roi, float scale, int PH,
                                                int PW)
{
  RoiGeom g;
  g.batch = (int)roi[0];
  g.cls = (int)roi[1];
  g.sw = (int)roundf(roi[2] * scale);
  g.sh = (int)roundf(roi[3] * scale);
  g.ew = (int)roundf(roi[4] * scale);
  g.eh = (int)roundf(roi[5] * scale);
  g.bin_h = div_rn((float)max(g.eh - g.sh + 1, 1), (float)PH);
  g.bin_w = div_rn((float)max(g.ew - g.sw + 1, 1), (float)PW);
  return g;
}

// feature range [lo, hi) of pooled bin p along one axis, clipped to the map (:64-75)
__device__ __forceinline__ void bin_span(float bin, int p, int start, int limit, int& lo, int& hi)
{
  lo = min(max((int)floorf((float)p * bin) + start, 0), limit);
  hi = min(max((int)ceilf((float)(p + 1) * bin) + start, 0), limit);
}

// vertical reduction of one feature column, one channel: max over h in [hlo, hhi) and the first h
// attaining it (-1 if nothing compares greater than -FLT_MAX, e.g. NaN columns)
__device__ __forceinline__ void column_max(const float* __restrict__ img, int W, int C, int ch,
                                           int hlo, int hhi, int w, float& best, int& row)
{
  best = -FLT_MAX;
  row = -1;
  const float* p = img + ((size_t)hlo * W + w) * C + ch;
  for (int h = hlo; h < hhi; ++h, p += (size_t)W * C) {
    const float v = *p;
    if (v > best) { best = v; row = h; }
  }
}

// phase 1: columns [w0, w0+ncols) x channels [c0, c0+cc) -> LDS (value, optionally first row)
template <bool VEC>
__device__ __forceinline__ void stage_columns(const float* __restrict__ img, int W, int C, int c0,
                                              int cc, int hlo, int hhi, int w0, int ncols,
                                              float* __restrict__ sval, int* __restrict__ srow)
{
  if (VEC) {
    const int groups = cc >> 2;
    for (int i = threadIdx.x; i < ncols * groups; i += RP_THREADS) {
      const int g = i % groups, col = i / groups;
      float4 mv = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
      int4 mh = make_int4(-1, -1, -1, -1);
      const float* p = img + ((size_t)hlo * W + w0 + col) * C + c0 + 4 * g;
      for (int h = hlo; h < hhi; ++h, p += (size_t)W * C) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        if (v.x > mv.x) { mv.x = v.x; mh.x = h; }
        if (v.y > mv.y) { mv.y = v.y; mh.y = h; }
        if (v.z > mv.z) { mv.z = v.z; mh.z = h; }
        if (v.w > mv.w) { mv.w = v.w; mh.w = h; }
      }
      *reinterpret_cast<float4*>(sval + col * cc + 4 * g) = mv;
      if (srow) *reinterpret_cast<int4*>(srow + col * cc + 4 * g) = mh;
    }
  } else {
    for (int i = threadIdx.x; i < ncols * cc; i += RP_THREADS) {
      const int c = i % cc, col = i / cc;
      float mv;
      int mh;
      column_max(img, W, C, c0 + c, hlo, hhi, w0 + col, mv, mh);
      sval[col * cc + c] = mv;
      if (srow) srow[col * cc + c] = mh;
    }
  }
}

// grid: (roi, ph, channel chunk), chunk fastest. pool_channel: one chunk holding channel roi_cls.
template <bool VEC>
__global__ __launch_bounds__(RP_THREADS) void roi_pool_fwd_staged(
    const float* __restrict__ data, const float* __restrict__ rois, float* __restrict__ top,
    int* __restrict__ argmax, int B, int H, int W, int C, int roi_cols, int PH, int PW, float scale,
    int pool_channel, int nchunks)
{
  __shared__ __attribute__((aligned(16))) float smem[RP_LDS_WORDS];
  const int chunk = blockIdx.x % nchunks;
  const int ph = (blockIdx.x / nchunks) % PH;
  const int n = blockIdx.x / (nchunks * PH);
  const RoiGeom g = roi_geometry(rois + (size_t)n * roi_cols, scale, PH, PW);
  const int c0 = pool_channel ? g.cls : chunk * RP_CHUNK;
  const int cc = pool_channel ? 1 : min(RP_CHUNK, C - c0);
  const int out_c = pool_channel ? 1 : C;  // channels per output bin
  const int oc0 = pool_channel ? 0 : c0;
  const size_t row0 = ((size_t)n * PH + ph) * PW;
  // CHECK_GE/LT in the CPU op (roi_pooling_op.cc:146-147); the reference GPU op would read out of bounds
  const bool valid = g.batch >= 0 && g.batch < B && c0 >= 0 && c0 < C;
  int hlo, hhi, w0, w1, tmp;
  bin_span(g.bin_h, ph, g.sh, H, hlo, hhi);
  bin_span(g.bin_w, 0, g.sw, W, w0, tmp);
  bin_span(g.bin_w, PW - 1, g.sw, W, tmp, w1);
  const int ncols = max(w1 - w0, 0);
  const bool row_empty = hhi <= hlo;
  const float* img = data + (size_t)(valid ? g.batch : 0) * H * W * C;
  const int per_col = argmax ? 2 : 1;
  const bool staged = valid && !row_empty && (long long)ncols * cc * per_col <= RP_LDS_WORDS;
  float* sval = smem;
  int* srow = argmax ? reinterpret_cast<int*>(smem + ncols * cc) : nullptr;
  if (staged) {
    stage_columns<VEC>(img, W, C, c0, cc, hlo, hhi, w0, ncols, sval, srow);
    __syncthreads();
  }
  for (int j = threadIdx.x; j < PW * cc; j += RP_THREADS) {
    const int c = j % cc, pw = j / cc;
    int wlo, whi;
    bin_span(g.bin_w, pw, g.sw, W, wlo, whi);
    const bool empty = row_empty || whi <= wlo;
    float best = (empty || !valid) ? 0.f : -FLT_MAX;
    int brow = -1, bcol = -1;
    if (valid && !empty) {
      for (int w = wlo; w < whi; ++w) {
        float v;
        int r = -1;
        if (staged) {
          v = sval[(w - w0) * cc + c];
          if (srow) r = srow[(w - w0) * cc + c];
        } else {
          column_max(img, W, C, c0 + c, hlo, hhi, w, v, r);
        }
        if (v > best || (v == best && r < brow)) { best = v; brow = r; bcol = w; }
      }
    }
    const size_t o = (row0 + pw) * out_c + oc0 + c;
    top[o] = best;
    if (argmax) argmax[o] = brow < 0 ? -1 : (brow * W + bcol) * C + c0 + c;
  }
}

// pool_score = roi_pool(conv5_3, 1/16) + roi_pool(conv4_3, 1/8), vgg16_convs.py:177-187.
// One workgroup per (roi, ph, pw) bin, thread t owns channels [4t, 4t+3]: registers only. The
// LDS-staged form above was measured for this op too (tools/bench_roi_pool.py, 16 frames x 468
// rows): 169 us with 256 threads, 137 us with one wave per bin row, against 130 us for this
// kernel (397 vs 165 us inside the bench step) — without an argmax to resolve, the work per bin
// row is too thin to pay for the barrier and the second pass, so the fused op stays register-only.
__global__ __launch_bounds__(128) void roi_pool_add2_perbin(
    const float* __restrict__ data_a, int Ha, int Wa, float scale_a,
    const float* __restrict__ data_b, int Hb, int Wb, float scale_b,
    const float* __restrict__ rois, float* __restrict__ out, int B, int C, int roi_cols, int PH,
    int PW, const int* __restrict__ num_rows_dev)
{
  const int bin = blockIdx.x;
  const int pw = bin % PW, ph = (bin / PW) % PH, n = bin / (PW * PH);
  const float* roi = rois + (size_t)n * roi_cols;
  const RoiGeom ga = roi_geometry(roi, scale_a, PH, PW);
  const RoiGeom gb = roi_geometry(roi, scale_b, PH, PW);
  int ha0, ha1, hb0, hb1, wa0, wa1, wb0, wb1;
  bin_span(ga.bin_h, ph, ga.sh, Ha, ha0, ha1);
  bin_span(gb.bin_h, ph, gb.sh, Hb, hb0, hb1);
  bin_span(ga.bin_w, pw, ga.sw, Wa, wa0, wa1);
  bin_span(gb.bin_w, pw, gb.sw, Wb, wb0, wb1);
  const bool ok = ga.batch >= 0 && ga.batch < B && (num_rows_dev == nullptr || n < num_rows_dev[0]);
  const bool ea = ha1 <= ha0 || wa1 <= wa0, eb = hb1 <= hb0 || wb1 <= wb0;
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    float4 ma = make_float4(0, 0, 0, 0), mb = ma;
    if (ok) {
      const float* ia = data_a + (size_t)ga.batch * Ha * Wa * C;
      const float* ib = data_b + (size_t)gb.batch * Hb * Wb * C;
      const float inita = ea ? 0.f : -FLT_MAX, initb = eb ? 0.f : -FLT_MAX;
      ma = make_float4(inita, inita, inita, inita);
      mb = make_float4(initb, initb, initb, initb);
      for (int h = ha0; h < ha1; ++h)
        for (int w = wa0; w < wa1; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(ia + (h * Wa + w) * C + c);
          ma.x = v.x > ma.x ? v.x : ma.x; ma.y = v.y > ma.y ? v.y : ma.y;
          ma.z = v.z > ma.z ? v.z : ma.z; ma.w = v.w > ma.w ? v.w : ma.w;
        }
      for (int h = hb0; h < hb1; ++h)
        for (int w = wb0; w < wb1; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(ib + (h * Wb + w) * C + c);
          mb.x = v.x > mb.x ? v.x : mb.x; mb.y = v.y > mb.y ? v.y : mb.y;
          mb.z = v.z > mb.z ? v.z : mb.z; mb.w = v.w > mb.w ? v.w : mb.w;
        }
    }
    *reinterpret_cast<float4*>(out + (size_t)bin * C + c) =
        make_float4(ma.x + mb.x, ma.y + mb.y, ma.z + mb.z, ma.w + mb.w);
  }
}

// The same op, one workgroup per BIN ROW (roi, ph) instead of per bin (round 4). On the capacity-sized row buffer behind
// the sync-free Hough layer (3024 rows of which ~700 exist) the per-bin grid was 148 000 workgroups of 128 threads, three
// quarters of them writing 2 KB of zeros each: 193 us for 303 MB, 1.6 TB/s. Here a workgroup covers the PW bins of a row
// (14 KB of output), a row past the device-side count costs one wide zero fill — or nothing at all with KEEP_DEAD, the
// variant the network uses: fc6 (csrc/fc_mfma.hip, csrc/fc_skinny.hip) masks rows at or past the same count itself, so
// `pool_score`'s rows past it are never read. Same per-value expressions as the per-bin kernel -> same bits.
constexpr int RP_XCDS = 8;     // workgroup w is dispatched to XCD w % 8 (MI355X: 8 XCDs, 32 CUs and one 4 MB L2 each)
constexpr int RP_GROUP = 9;    // consecutive rois that share an XCD
template <bool KEEP_DEAD>
__global__ __launch_bounds__(256) void roi_pool_add2_rows(
    const float* __restrict__ data_a, int Ha, int Wa, float scale_a,
    const float* __restrict__ data_b, int Hb, int Wb, float scale_b,
    const float* __restrict__ rois, float* __restrict__ out, int B, int C, int roi_cols, int PH,
    int PW, const int* __restrict__ num_rows_dev, int R)
{
  // Workgroup -> (roi, bin row), XCD-aware. Consecutive rows of the RoI buffer overlap: the Hough layer's training mode
  // emits a box and its 8 jitters as 9 consecutive rows (hough_voting_gpu_op.cu.cc:440-466), and test-mode neighbours are at
  // least boxes of the same image. Workgroup w runs on XCD w % 8, each with its own L2; with w = roi * PH + ph the same bin
  // row of the 9 jittered boxes landed on 8 different XCDs and every XCD fetched the cells for itself (FETCH_SIZE 490 MB per
  // launch for 196 MB of feature maps). Here RP_GROUP consecutive rois x one bin row are consecutive workgroups of ONE XCD.
  const int w = blockIdx.x, xcd = w % RP_XCDS, q = w / RP_XCDS;
  const int k = (q / RP_GROUP) * RP_XCDS + xcd;          // chunk = (roi group, bin row)
  const int ph = k % PH, n = (k / PH) * RP_GROUP + q % RP_GROUP;
  if (n >= R) return;                                    // (padding of the last group / of the chunk count to 8)
  const bool live = num_rows_dev == nullptr || n < num_rows_dev[0];
  if (KEEP_DEAD && !live) return;
  const float* roi = rois + (size_t)n * roi_cols;
  const int nq = C >> 2;
  float* orow = out + ((size_t)n * PH + ph) * PW * C;
  const RoiGeom ga = roi_geometry(roi, scale_a, PH, PW);
  const bool ok = live && ga.batch >= 0 && ga.batch < B;
  if (!ok) {
    for (int i = threadIdx.x; i < PW * nq; i += 256) *reinterpret_cast<float4*>(orow + (size_t)i * 4) = make_float4(0, 0, 0, 0);
    return;
  }
  const RoiGeom gb = roi_geometry(roi, scale_b, PH, PW);
  int ha0, ha1, hb0, hb1;
  bin_span(ga.bin_h, ph, ga.sh, Ha, ha0, ha1);
  bin_span(gb.bin_h, ph, gb.sh, Hb, hb0, hb1);
  const float* ia = data_a + (size_t)ga.batch * Ha * Wa * C;
  const float* ib = data_b + (size_t)gb.batch * Hb * Wb * C;
  for (int i = threadIdx.x; i < PW * nq; i += 256) {
    const int pw = i / nq, c = (i - pw * nq) * 4;
    int wa0, wa1, wb0, wb1;
    bin_span(ga.bin_w, pw, ga.sw, Wa, wa0, wa1);
    bin_span(gb.bin_w, pw, gb.sw, Wb, wb0, wb1);
    const bool ea = ha1 <= ha0 || wa1 <= wa0, eb = hb1 <= hb0 || wb1 <= wb0;
    const float inita = ea ? 0.f : -FLT_MAX, initb = eb ? 0.f : -FLT_MAX;
    float4 ma = make_float4(inita, inita, inita, inita), mb = make_float4(initb, initb, initb, initb);
    // a bin row's cells eight at a time: the loads of a trip are independent and leave together; columns past the bin
    // re-read its last cell (max is idempotent: same value as the cell-by-cell walk, which paid one memory round trip per
    // cell — ~20 per bin at pool4's 1/8 resolution, the whole cost of this kernel; round 5)
#define RP_SCAN(M, BASE, WW, H0, H1, W0, W1)                                                          \
    for (int h = (H0); h < (H1); ++h)                                                                 \
      for (int w0 = (W0); w0 < (W1); w0 += 8) {                                                       \
        float4 v[8];                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; j++) {                                               \
          const int w = w0 + j < (W1) ? w0 + j : (W1) - 1;                                            \
          v[j] = *reinterpret_cast<const float4*>((BASE) + (h * (WW) + w) * C + c);                   \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; j++) {                                               \
          M.x = v[j].x > M.x ? v[j].x : M.x; M.y = v[j].y > M.y ? v[j].y : M.y;                       \
          M.z = v[j].z > M.z ? v[j].z : M.z; M.w = v[j].w > M.w ? v[j].w : M.w;                       \
        }                                                                                             \
      }
    RP_SCAN(ma, ia, Wa, ha0, ha1, wa0, wa1)
    RP_SCAN(mb, ib, Wb, hb0, hb1, wb0, wb1)
#undef RP_SCAN
    *reinterpret_cast<float4*>(orow + (size_t)pw * C + c) = make_float4(ma.x + mb.x, ma.y + mb.y, ma.z + mb.z, ma.w + mb.w);
  }
}

constexpr int RB_TILE_W = 8;
constexpr int RB_LIST = 512;

struct RoiEntry {
  int idx, cls, sw, sh, ew, eh;
  float bin_h, bin_w;
};

// grid.x: (image, h, tile of RB_TILE_W columns); grid.y: blocks of 256 channels
__global__ __launch_bounds__(256) void roi_pool_bwd_binned(
    const float* __restrict__ top_diff, const float* __restrict__ rois,
    const int* __restrict__ argmax, float* __restrict__ bottom_diff, int H, int W, int C, int R,
    int roi_cols, int PH, int PW, float scale, int pool_channel)
{
  __shared__ RoiEntry s_list[RB_LIST];
  __shared__ int s_count, s_cursor;
  const int tiles_w = (W + RB_TILE_W - 1) / RB_TILE_W;
  const int tw = blockIdx.x % tiles_w;
  const int h = (blockIdx.x / tiles_w) % H;
  const int n = blockIdx.x / (tiles_w * H);
  const int w_lo = tw * RB_TILE_W, w_hi = min(W, w_lo + RB_TILE_W);
  const int c = blockIdx.y * 256 + threadIdx.x;
  const bool lane_ok = c < C;
  const int lane = threadIdx.x & 63;
  float acc[RB_TILE_W];
#pragma unroll
  for (int j = 0; j < RB_TILE_W; ++j) acc[j] = 0.f;

  int cursor = 0;
  while (cursor < R) {
    if (threadIdx.x < 64) {
      // ordered compaction of the ROIs of image n that touch this tile; ascending roi index is the
      // reference's summation order (:150)
      int count = 0, r = cursor;
      while (r < R && count <= RB_LIST - 64) {
        const int i = r + lane;
        bool hit = false;
        RoiEntry e;
        if (i < R) {
          const RoiGeom g = roi_geometry(rois + (size_t)i * roi_cols, scale, PH, PW);
          // :161-166 — the element must lie inside the (unclipped) ROI rectangle
          hit = g.batch == n && h >= g.sh && h <= g.eh && w_lo <= g.ew && w_hi - 1 >= g.sw;
          e.idx = i; e.cls = g.cls; e.sw = g.sw; e.sh = g.sh; e.ew = g.ew; e.eh = g.eh;
          e.bin_h = g.bin_h; e.bin_w = g.bin_w;
        }
        const unsigned long long mask = __ballot(hit);
        if (hit) s_list[count + __popcll(mask & ((1ull << lane) - 1ull))] = e;
        count += __popcll(mask);
        r += 64;
      }
      if (lane == 0) { s_count = count; s_cursor = r; }
    }
    __syncthreads();
    const int count = s_count;
    cursor = s_cursor;
    for (int k = 0; k < count; ++k) {
      const RoiEntry e = s_list[k];
      if (!lane_ok || (pool_channel && c != e.cls)) continue;
      // bins whose span can contain row h / column w (:180-191)
      int phs = (int)floorf(div_rn((float)(h - e.sh), e.bin_h));
      int phe = (int)ceilf(div_rn((float)(h - e.sh + 1), e.bin_h));
      phs = min(max(phs, 0), PH);
      phe = min(max(phe, 0), PH);
      const size_t base = (size_t)e.idx * PH * PW;
#pragma unroll
      for (int j = 0; j < RB_TILE_W; ++j) {
        const int w = w_lo + j;
        if (w >= w_hi || w < e.sw || w > e.ew) continue;
        int pws = (int)floorf(div_rn((float)(w - e.sw), e.bin_w));
        int pwe = (int)ceilf(div_rn((float)(w - e.sw + 1), e.bin_w));
        pws = min(max(pws, 0), PW);
        pwe = min(max(pwe, 0), PW);
        const int want = (h * W + w) * C + c;
        for (int ph = phs; ph < phe; ++ph)
          for (int pw = pws; pw < pwe; ++pw) {
            const size_t o = pool_channel ? base + ph * PW + pw : (base + ph * PW + pw) * C + c;
            if (argmax[o] == want) acc[j] += top_diff[o];
          }
      }
    }
    __syncthreads();
  }
  if (lane_ok) {
    float* dst = bottom_diff + (((size_t)n * H + h) * W + w_lo) * C + c;
#pragma unroll
    for (int j = 0; j < RB_TILE_W; ++j)
      if (w_lo + j < w_hi) dst[(size_t)j * C] = acc[j];
  }
}

int validate(int B, int H, int W, int C, int R, int roi_cols, int PH, int PW, float scale,
             int pool_channel)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, PCNN_EINVAL, "roi_pool: data must be 4-dimensional (got %dx%dx%dx%d)", B, H, W, C);
  PCNN_REQUIRE(R >= 0, PCNN_EINVAL, "roi_pool: rois must be 2-dimensional (num_rois %d)", R);
  PCNN_REQUIRE(roi_cols >= 6, PCNN_EINVAL, "roi_pool: rois need >= 6 columns (batch, cls, x1, y1, x2, y2), got %d", roi_cols);
  // attribute checks, roi_pooling_op.cc:60-69 (`>= 0` there; 0 makes the output empty and divides by zero)
  PCNN_REQUIRE(PH >= 1, PCNN_EINVAL, "roi_pool: Need pooled_height >= 1, got %d", PH);
  PCNN_REQUIRE(PW >= 1, PCNN_EINVAL, "roi_pool: Need pooled_width >= 1, got %d", PW);
  PCNN_REQUIRE(pool_channel == 0 || pool_channel == 1, PCNN_EINVAL, "roi_pool: pool_channel must be 0 or 1");
  PCNN_REQUIRE((long long)H * W * C < (1ll << 31), PCNN_EINVAL, "roi_pool: image too large for int32 argmax");
  (void)scale;
  return PCNN_OK;
}

}  // namespace

extern "C" int pcnn_roi_pool_fwd(const float* data, const float* rois, int B, int H, int W, int C,
                                 int R, int roi_cols, int PH, int PW, float scale,
                                 int pool_channel, float* top, int32_t* argmax, void* stream_)
{
  int st = validate(B, H, W, C, R, roi_cols, PH, PW, scale, pool_channel);
  if (st != PCNN_OK) return st;
  if (R == 0) return PCNN_OK;
  PCNN_REQUIRE(data && rois && top, PCNN_ENULL, "roi_pool: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int nchunks = pool_channel ? 1 : (C + RP_CHUNK - 1) / RP_CHUNK;
  const long long blocks = (long long)R * PH * nchunks;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "roi_pool: too many bin rows (%lld)", blocks);
  if (!pool_channel && (C % 4 == 0) && aligned16(data)) {
    PCNN_LAUNCH(roi_pool_fwd_staged<true>, dim3((unsigned)blocks), dim3(RP_THREADS), 0, stream, data,
                rois, top, argmax, B, H, W, C, roi_cols, PH, PW, scale, pool_channel, nchunks);
  } else {
    PCNN_LAUNCH(roi_pool_fwd_staged<false>, dim3((unsigned)blocks), dim3(RP_THREADS), 0, stream, data,
                rois, top, argmax, B, H, W, C, roi_cols, PH, PW, scale, pool_channel, nchunks);
  }
  return check_launch("roi_pool_fwd");
}

static int roi_pool_add2_impl(const float* data_a, int Ha, int Wa, float scale_a,
                              const float* data_b, int Hb, int Wb, float scale_b,
                              const float* rois, int B, int C, int R, int roi_cols, int PH,
                              int PW, const int32_t* num_rows_dev, float* out, void* stream_, bool keep_dead)
{
  int st = validate(B, Ha, Wa, C, R, roi_cols, PH, PW, scale_a, 0);
  if (st != PCNN_OK) return st;
  st = validate(B, Hb, Wb, C, R, roi_cols, PH, PW, scale_b, 0);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(C % 4 == 0, PCNN_EINVAL, "roi_pool_add2: channels must be a multiple of 4 (got %d)", C);
  if (R == 0) return PCNN_OK;
  PCNN_REQUIRE(data_a && data_b && rois && out, PCNN_ENULL, "roi_pool_add2: NULL pointer");
  PCNN_REQUIRE(aligned16(data_a) && aligned16(data_b) && aligned16(out), PCNN_EINVAL,
               "roi_pool_add2: tensors must be 16-byte aligned");
  PCNN_REQUIRE(!keep_dead || num_rows_dev, PCNN_ENULL, "roi_pool_add2_live: needs the device-side row count");
  hipStream_t stream = (hipStream_t)stream_;
  // (roi groups x bin rows) chunks, padded to a multiple of the XCD count, RP_GROUP workgroups each
  const long long chunks = (((long long)(R + RP_GROUP - 1) / RP_GROUP * PH + RP_XCDS - 1) / RP_XCDS) * RP_XCDS;
  const long long blocks = chunks * RP_GROUP;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "roi_pool_add2: too many bin rows (%lld)", blocks);
  if (keep_dead)
    PCNN_LAUNCH(roi_pool_add2_rows<true>, dim3((unsigned)blocks), dim3(256), 0, stream, data_a, Ha, Wa, scale_a, data_b, Hb,
                Wb, scale_b, rois, out, B, C, roi_cols, PH, PW, num_rows_dev, R);
  else
    PCNN_LAUNCH(roi_pool_add2_rows<false>, dim3((unsigned)blocks), dim3(256), 0, stream, data_a, Ha, Wa, scale_a, data_b, Hb,
                Wb, scale_b, rois, out, B, C, roi_cols, PH, PW, num_rows_dev, R);
  return check_launch("roi_pool_add2_fwd");
}

extern "C" int pcnn_roi_pool_add2_fwd(const float* data_a, int Ha, int Wa, float scale_a,
                                      const float* data_b, int Hb, int Wb, float scale_b,
                                      const float* rois, int B, int C, int R, int roi_cols, int PH,
                                      int PW, const int32_t* num_rows_dev, float* out, void* stream_)
{
  return roi_pool_add2_impl(data_a, Ha, Wa, scale_a, data_b, Hb, Wb, scale_b, rois, B, C, R, roi_cols, PH, PW, num_rows_dev,
                            out, stream_, false);
}

extern "C" int pcnn_roi_pool_add2_live_fwd(const float* data_a, int Ha, int Wa, float scale_a,
                                           const float* data_b, int Hb, int Wb, float scale_b,
                                           const float* rois, int B, int C, int R, int roi_cols, int PH,
                                           int PW, const int32_t* num_rows_dev, float* out, void* stream_)
{
  return roi_pool_add2_impl(data_a, Ha, Wa, scale_a, data_b, Hb, Wb, scale_b, rois, B, C, R, roi_cols, PH, PW, num_rows_dev,
                            out, stream_, true);
}

extern "C" int pcnn_roi_pool_bwd(const float* top_diff, const float* rois, const int32_t* argmax,
                                 int B, int H, int W, int C, int R, int roi_cols, int PH, int PW,
                                 float scale, int pool_channel, float* bottom_diff, void* stream_)
{
  int st = validate(B, H, W, C, R, roi_cols, PH, PW, scale, pool_channel);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(bottom_diff, PCNN_ENULL, "roi_pool_bwd: NULL output");
  PCNN_REQUIRE(R == 0 || (top_diff && rois && argmax), PCNN_ENULL, "roi_pool_bwd: NULL input");
  hipStream_t stream = (hipStream_t)stream_;
  const long long tiles = (long long)B * H * ((W + RB_TILE_W - 1) / RB_TILE_W);
  PCNN_REQUIRE(tiles < (1ll << 31) && (C + 255) / 256 < 65536, PCNN_EINVAL,
               "roi_pool_bwd: feature map too large for one launch");
  PCNN_LAUNCH(roi_pool_bwd_binned, dim3((unsigned)tiles, (unsigned)((C + 255) / 256)), dim3(256), 0,
              stream, top_diff, rois, argmax, bottom_diff, H, W, C, R, roi_cols, PH, PW, scale,
              pool_channel);
  return check_launch("roi_pool_bwd");
}
