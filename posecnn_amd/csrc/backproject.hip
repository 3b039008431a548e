// backproject.hip — gfx950 2-D -> 3-D voxel back-projection (replaces TF1 ops
// "Backproject"/"BackprojectGrad", lib/backprojecting_layer/backprojecting_op.cc:295-383,
// backprojecting_op_gpu.cu.cc:17-126 (forward), :159-217 (backward)).
//
// The op is HBM-write bound: 2*Cd + Cl floats leave per voxel (10 GB per frame at G=256, Cd=64,
// Cl=22) while most voxels are off-surface and read almost nothing. The reference spends one
// thread per (voxel, channel) and redoes the projection + (2k+1)^2 depth scan in each, and lets
// the c==0 thread write all Cl label floats serially. Here a thread owns 4 consecutive channels of
// one voxel (dwordx4 loads/stores; 16 lanes cover a 64-channel voxel, a wave covers 4 voxels), the
// window scan is shared by those 4 channels, and the label tensor is produced by its own
// flattened (voxel, class) pass so that its writes are coalesced too.
#include <cstdlib>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

struct Proj {
  int px, py;
  float Z1;
};

// backprojecting_op_gpu.cu.cc:43-59
__device__ __forceinline__ Proj project_voxel(const float* __restrict__ md, int d, int h, int w)
{
  float X = d * md[42] + md[45];
  float Y = h * md[43] + md[46];
  float Z = w * md[44] + md[47];
  float X1 = md[18] * X + md[19] * Y + md[20] * Z + md[21];
  float Y1 = md[22] * X + md[23] * Y + md[24] * Z + md[25];
  float Z1 = md[26] * X + md[27] * Y + md[28] * Z + md[29];
  float x1 = md[0] * X1 + md[1] * Y1 + md[2] * Z1;
  float x2 = md[3] * X1 + md[4] * Y1 + md[5] * Z1;
  float x3 = md[6] * X1 + md[7] * Y1 + md[8] * Z1;
  Proj p;
  p.px = round_to_int_sat(div_rn(x1, x3));
  p.py = round_to_int_sat(div_rn(x2, x3));
  p.Z1 = Z1;
  return p;
}

struct Window {
  int xlo, xhi, ylo, yhi;
};

__device__ __forceinline__ Window clip_window(const Proj& p, int ksize, int H, int W)
{
  long long xlo = (long long)p.px - ksize, xhi = (long long)p.px + ksize;
  long long ylo = (long long)p.py - ksize, yhi = (long long)p.py + ksize;
  Window w;
  w.xlo = (int)(xlo < 0 ? 0 : (xlo > W ? W : xlo));
  w.ylo = (int)(ylo < 0 ? 0 : (ylo > H ? H : ylo));
  w.xhi = (int)(xhi > W - 1 ? W - 1 : (xhi < -1 ? -1 : xhi));
  w.yhi = (int)(yhi > H - 1 ? H - 1 : (yhi < -1 ? -1 : yhi));
  return w;
}

// data + flag. VEC channels per thread (4 when Cd % 4 == 0, else 1).
template <int VEC>
__global__ __launch_bounds__(256) void backproject_data_kernel(
    const float* __restrict__ data, const float* __restrict__ depth, const float* __restrict__ meta,
    float* __restrict__ top_data, float* __restrict__ top_flag, long long total, int H, int W,
    int Cd, int num_meta, int G, int ksize, float threshold)
{
  const int cpv = Cd / VEC;  // threads per voxel
  for (long long index = (long long)blockIdx.x * 256 + threadIdx.x; index < total;
       index += (long long)gridDim.x * 256) {
    long long t = index;
    const int c = (int)(t % cpv) * VEC; t /= cpv;
    const long long vox = t;
    const int w = (int)(t % G); t /= G;
    const int h = (int)(t % G); t /= G;
    const int d = (int)(t % G); t /= G;
    const int n = (int)t;
    const float* md = meta + (size_t)n * num_meta;
    const Proj p = project_voxel(md, d, h, w);
    const Window win = clip_window(p, ksize, H, W);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) acc[i] = 0.f;
    int count = 0;
    for (int x = win.xlo; x <= win.xhi; x++)
      for (int y = win.ylo; y <= win.yhi; y++) {
        const long long index_pixel = (long long)n * H * W + (long long)y * W + x;
        const float dep = depth[index_pixel];
        if (fabsf(dep - p.Z1) < threshold) {
          count++;
          if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(data + index_pixel * Cd + c);
            acc[0] += v.x; acc[1 % VEC] += v.y; acc[2 % VEC] += v.z; acc[3 % VEC] += v.w;
          } else {
            acc[0] += data[index_pixel * Cd + c];
          }
        }
      }
    float flag = 0.f;
    if (count > 0) {
      const float cf = (float)count;
#pragma unroll
      for (int i = 0; i < VEC; i++) acc[i] = div_rn(acc[i], cf);
      flag = 1.f;
    }
    if (VEC == 4) {
      *reinterpret_cast<float4*>(top_data + vox * Cd + c) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
      *reinterpret_cast<float4*>(top_flag + vox * Cd + c) = make_float4(flag, flag, flag, flag);
    } else {
      top_data[vox * Cd + c] = acc[0];
      top_flag[vox * Cd + c] = flag;
    }
  }
}

// label: one thread per (voxel, class)
__global__ __launch_bounds__(256) void backproject_label_kernel(
    const float* __restrict__ label, const float* __restrict__ depth, const float* __restrict__ meta,
    const float* __restrict__ label_3d, float* __restrict__ top_label, long long total, int H,
    int W, int Cl, int num_meta, int G, int ksize, float threshold)
{
  for (long long index = (long long)blockIdx.x * 256 + threadIdx.x; index < total;
       index += (long long)gridDim.x * 256) {
    long long t = index;
    const int cl = (int)(t % Cl); t /= Cl;
    const int w = (int)(t % G); t /= G;
    const int h = (int)(t % G); t /= G;
    const int d = (int)(t % G); t /= G;
    const int n = (int)t;
    const float* md = meta + (size_t)n * num_meta;
    const Proj p = project_voxel(md, d, h, w);
    const Window win = clip_window(p, ksize, H, W);
    float acc = 0.f;
    int count = 0;
    for (int x = win.xlo; x <= win.xhi; x++)
      for (int y = win.ylo; y <= win.yhi; y++) {
        const long long index_pixel = (long long)n * H * W + (long long)y * W + x;
        if (fabsf(depth[index_pixel] - p.Z1) < threshold) {
          count++;
          acc += label[index_pixel * Cl + cl];
        }
      }
    top_label[index] = count == 0 ? label_3d[index] : div_rn(acc, (float)count);
  }
}

// Depth range of every window the forward pass can ask about (round 5). For a window CENTRE (xc, yc) with
// xc in [-k, W-1+k], yc in [-k, H-1+k] — every centre whose (2k+1)^2 window meets the image — the minimum and the
// maximum depth over the window's pixels inside the image, NaNs ignored (a NaN depth never matches:
// backprojecting_op_gpu.cu.cc:75 compares fabs(depth - Z1) < threshold). Layout [B][H+2k][W+2k] float2, index = centre + k.
// A voxel whose Z1 lies outside [min - threshold, max + threshold] of its window cannot match a single pixel and
// skips the scan (85 % of the voxels of the test scene; the test is made with the scan's own subtraction, see
// backproject_fused_kernel). Block = 64 x 16 centres; the (64+2k) x (16+2k) depth tile goes through LDS, rows
// first (min / max over 2k+1 columns), then columns.
__global__ __launch_bounds__(256) void backproject_window_range_kernel(
    const float* __restrict__ depth, float2* __restrict__ wrange, int H, int W, int ksize)
{
  constexpr int TW = 64, TH = 16, KMAX = 3;
  __shared__ float tile[TH + 2 * KMAX][TW + 2 * KMAX];
  __shared__ float2 rowr[TH + 2 * KMAX][TW];
  const int S = 2 * ksize + 1, Wc = W + 2 * ksize, Hc = H + 2 * ksize;
  const int cx0 = blockIdx.x * TW, cy0 = blockIdx.y * TH, n = blockIdx.z;
  const float* dn = depth + (long long)n * H * W;
  const float qnan = __int_as_float(0x7fc00000);
  const int rw = TW + 2 * ksize, rh = TH + 2 * ksize;
  // centre index c covers pixels c - 2k .. c (pixel = centre index - k -+ k)
  for (int i = threadIdx.x; i < rw * rh; i += 256) {
    const int ry = i / rw, rx = i - ry * rw;
    const int x = cx0 - 2 * ksize + rx, y = cy0 - 2 * ksize + ry;
    tile[ry][rx] = (x >= 0 && x < W && y >= 0 && y < H) ? dn[(long long)y * W + x] : qnan;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TW * rh; i += 256) {
    const int ry = i / TW, cx = i - ry * TW;
    float mn = qnan, mx = qnan;
    for (int j = 0; j < S; j++) {
      const float v = tile[ry][cx + j];
      mn = fminf(mn, v);   // (minNum / maxNum: the non-NaN operand wins)
      mx = fmaxf(mx, v);
    }
    rowr[ry][cx] = make_float2(mn, mx);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TW * TH; i += 256) {
    const int cy = i / TW, cx = i - cy * TW;
    if (cx0 + cx >= Wc || cy0 + cy >= Hc) continue;
    float mn = qnan, mx = qnan;
    for (int j = 0; j < S; j++) {
      const float2 r = rowr[cy + j][cx];
      mn = fminf(mn, r.x);
      mx = fmaxf(mx, r.y);
    }
    wrange[((long long)n * Hc + cy0 + cy) * Wc + cx0 + cx] = make_float2(mn, mx);
  }
}

template <bool NT>
__device__ __forceinline__ void bp_store4(float4* p, const float4 v)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  if (NT) __builtin_nontemporal_store((v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
  else *p = v;
}

// Fused forward for windows of at most 49 pixels (k <= 3 — the reference's k = 3, vgg16.py:131-132). A wave owns 64
// consecutive voxels (32 KB of data + flag and 64 Cl label floats, all contiguous):
//   A  lane = voxel: ONE projection per voxel; with the window-range table (above) one 8-byte load decides whether
//      the voxel can match anything at all; only then the window is scanned — a column of 2k+1 loads in flight at a
//      time — and the matching pixels are LISTED in LDS in the reference's summation order (x outer, y inner,
//      backprojecting_op_gpu.cu.cc:66-92) as 16-bit offsets from the window origin: 64 x 56 entries per wave;
//   all-miss wave (most of the grid): the outputs are a 32-KB run of zeros and a straight copy of label_3d, written
//      as independent 16-byte stores with nothing to wait for in between;
//   B  otherwise: lane = (voxel of a group of 64/LPV, channel quad): a voxel's list is read 8 entries per LDS
//      instruction, its 8 pixel rows are requested together and added in list order (`acc += v`, the reference's
//      order); every store instruction writes 64/LPV voxels x Cd floats contiguously (data and flag);
//   C  lane = flattened (voxel, class) of the wave's 64 x Cl contiguous label outputs, the label_3d values of 8
//      trips requested up front, the gathers 8 at a time from the same lists.
// History. Round 4: every lane walked its window one dependent load at a time before it learned that 85 % of the
// voxels miss, and fetched label_3d one trip at a time behind a queue of stores — a wave lived ~170 us for 43 KB of
// output (0.27 of the HBM peak at G = 256). Round 5, first version: range table + batched loads + the all-miss path,
// the matches as a 64-bit mask per voxel walked bit by bit in every lane of phases B and C (ffs, clear, a division by
// the window side, offset arithmetic: ~50 vector instructions per matching pixel and lane): all-miss scenes at 0.70-0.86
// of the peak, the parity scene (14 % of the voxels hit) only 0.27 -> 0.36 — its hit waves were bound by exactly that
// instruction stream, not by the gathers' latency (4 / 8 / 16 gathers in flight per lane: 4.06 / 3.89 / 4.51 ms).
template <int LPV, bool NT>  // lanes per voxel in phase B = min(Cd, 64) / 4; NT: non-temporal stores
__global__ __launch_bounds__(256) void backproject_fused_kernel(
    const float* __restrict__ data, const float* __restrict__ label, const float* __restrict__ depth,
    const float* __restrict__ meta, const float* __restrict__ label_3d, const float2* __restrict__ wrange,
    float* __restrict__ top_data, float* __restrict__ top_label, float* __restrict__ top_flag, long long nvox,
    int H, int W, int Cd, int Cl, int num_meta, int G, int ksize, float threshold, int lab_vec)
{
  constexpr int LIST = 56;   // list slots per voxel: 49 rounded up to whole 16-byte LDS reads
  typedef uint4 __attribute__((may_alias)) uint4_a;   // (the lists are written as 16-bit entries and read 8 at a time)
  __shared__ __attribute__((aligned(16))) unsigned short s_rel[4][64][LIST];   // matched pixels: dy * W + dx from the window origin
  __shared__ unsigned char s_hit[4][64];                                       // the wave's hit voxels, ascending
  __shared__ int s_cnt[4][64], s_org[4][64];                                   // matches; window origin py * W + px (may be negative)
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long nwave = (nvox + 63) / 64;
  const int S = 2 * ksize + 1;
  const unsigned G2 = (unsigned)G * (unsigned)G, G3u = G2 * (unsigned)G;   // (G <= 1024: the launcher checks)
  const long long G3 = (long long)G3u;
  const unsigned cl_magic = (unsigned)((0x100000000ull + (unsigned)Cl - 1) / (unsigned)Cl);   // i / Cl = umulhi(i, magic) for i < 2^32 / Cl
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned short* my_rel = &s_rel[wib][lane][0];
  for (long long wv = (long long)blockIdx.x * 4 + wib; wv < nwave; wv += (long long)gridDim.x * 4) {
    const long long vbase = wv * 64;
    const int nv = (int)(nvox - vbase < 64 ? nvox - vbase : 64);
    // ---- A: one voxel per lane
    int cnt = 0;
    if (lane < nv) {
      const long long n0 = vbase / G3;                         // (wave-uniform: scalar unit)
      unsigned r = (unsigned)(vbase - n0 * G3) + (unsigned)lane;
      const unsigned dn_ = r / G3u;                            // a wave may straddle images
      r -= dn_ * G3u;
      const int n = (int)n0 + (int)dn_;
      const unsigned d = r / G2, r2 = r - d * G2, h = r2 / (unsigned)G, w = r2 - h * (unsigned)G;
      const float* md = meta + (size_t)n * num_meta;
      const Proj p = project_voxel(md, (int)d, (int)h, (int)w);
      const Window win = clip_window(p, ksize, H, W);
      bool cand = win.xlo <= win.xhi && win.ylo <= win.yhi;
      if (cand && wrange) {
        // (non-empty window <=> centre in [-k, W-1+k] x [-k, H-1+k]: inside the table)
        const float2 rg = wrange[((long long)n * (H + 2 * ksize) + (p.py + ksize)) * (W + 2 * ksize) + (p.px + ksize)];
        // f32 subtraction is monotone: depth >= min  =>  depth - Z1 >= min - Z1 >= threshold  =>  no pixel matches;
        // depth <= max  =>  depth - Z1 <= max - Z1 <= -threshold  =>  likewise. (NaN range or NaN Z1: both tests
        // fail and the window is scanned.)
        cand = !((rg.x - p.Z1) >= threshold || (rg.y - p.Z1) <= -threshold);
      }
      if (cand) {
        const int px = p.px - ksize, py = p.py - ksize;        // window origin (a candidate's projection is near the image: no overflow)
        s_org[wib][lane] = py * W + px;
        const float* dn = depth + (long long)n * H * W;
        for (int xi = 0; xi < S; xi++) {
          const int x = px + xi;
          if (x < 0 || x >= W) continue;
          float dv[7];
#pragma unroll
          for (int yi = 0; yi < 7; yi++) {   // the column's 2k+1 <= 7 loads in flight together
            const int y = py + yi;
            const bool ok = yi < S && y >= 0 && y < H;
            dv[yi] = dn[ok ? (long long)y * W + x : (long long)x];
          }
#pragma unroll
          for (int yi = 0; yi < 7; yi++) {
            const int y = py + yi;
            const bool ok = yi < S && y >= 0 && y < H;
            if (ok && fabsf(dv[yi] - p.Z1) < threshold) my_rel[cnt++] = (unsigned short)(yi * W + xi);   // (7 W < 65536: the launcher checks)
          }
        }
      }
    }
    s_cnt[wib][lane] = cnt;
    __builtin_amdgcn_wave_barrier();   // (the lists are wave-private: LDS operations of one wave complete in order)
    float4* wdata = reinterpret_cast<float4*>(top_data + vbase * Cd);
    float4* wflag = reinterpret_cast<float4*>(top_flag + vbase * Cd);
    const long long lbase = vbase * Cl;
    if (__ballot(cnt != 0) == 0) {
      // ---- the whole wave misses: zeros + label_3d straight through
      const int n4 = nv * (Cd / 4);
#pragma unroll 4
      for (int i = lane; i < n4; i += 64) {
        bp_store4<NT>(wdata + i, zero4);
        bp_store4<NT>(wflag + i, zero4);
      }
      if (lab_vec && nv == 64) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f* src = reinterpret_cast<const v4f*>(label_3d + lbase);
        v4f* dst = reinterpret_cast<v4f*>(top_label + lbase);
        const int l4 = 16 * Cl;
        for (int i0 = 0; i0 < l4; i0 += 64 * 8) {
          v4f t[8];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const int i = i0 + 64 * j + lane;
            if (i < l4) t[j] = NT ? __builtin_nontemporal_load(src + i) : src[i];
          }
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const int i = i0 + 64 * j + lane;
            if (i < l4) { if (NT) __builtin_nontemporal_store(t[j], dst + i); else dst[i] = t[j]; }
          }
        }
      } else {
        for (int i = lane; i < nv * Cl; i += 64) top_label[lbase + i] = label_3d[lbase + i];
      }
      continue;
    }
    // ---- B: data + flag. The voxels that hit are listed (ascending) so that every gather instruction works for
    // 64/LPV voxels that have something to gather — about half of a hit wave's voxels miss, and walked in voxel
    // order their lane groups would idle through the other half's trips; the misses get their zeros first.
    constexpr int VPI = 64 / LPV;  // voxels per iteration
    const int sub = lane / LPV, cq = lane % LPV;
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const unsigned long long hitmask = __ballot(cnt != 0);
    const int nhit = __popcll(hitmask);
    if (cnt != 0) s_hit[wib][__popcll(hitmask & ((1ull << lane) - 1ull))] = (unsigned char)lane;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = 0; it < 64 / VPI; it++) {
      const int vl = it * VPI + sub;
      if (vl < nv && !((hitmask >> vl) & 1ull)) {
        const long long vox = vbase + vl;
        for (int ch = cq * 4; ch < Cd; ch += LPV * 4) {
          bp_store4<NT>(reinterpret_cast<float4*>(top_data + vox * Cd + ch), zero4);
          bp_store4<NT>(reinterpret_cast<float4*>(top_flag + vox * Cd + ch), zero4);
        }
      }
    }
#pragma unroll 1
    for (int h0 = 0; h0 < nhit; h0 += VPI) {
      if (h0 + sub >= nhit) continue;
      const int vl = s_hit[wib][h0 + sub];
      const long long vox = vbase + vl;
      const int c = s_cnt[wib][vl];
      const int org = s_org[wib][vl];
      const float* dbase = data + (vox / G3) * H * W * (long long)Cd;
      const float cf = (float)c;
      for (int ch = cq * 4; ch < Cd; ch += LPV * 4) {
        float4 acc = zero4;
#pragma unroll 1
        for (int k0 = 0; k0 < c; k0 += 8) {   // 8 list entries per LDS read, their pixel rows requested together
          const uint4 r8 = *reinterpret_cast<const uint4_a*>(&s_rel[wib][vl][k0]);
          const unsigned rr[4] = {r8.x, r8.y, r8.z, r8.w};
          float4 v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const int rel = (int)((rr[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            v[j] = *reinterpret_cast<const float4*>(dbase + (org + (k0 + j < c ? rel : (int)(rr[0] & 0xffffu))) * Cd + ch);   // (H W Cd < 2^31: the launcher checks)
          }
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (k0 + j < c) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
        acc.x = div_rn(acc.x, cf); acc.y = div_rn(acc.y, cf); acc.z = div_rn(acc.z, cf); acc.w = div_rn(acc.w, cf);
        bp_store4<NT>(reinterpret_cast<float4*>(top_data + vox * Cd + ch), acc);
        bp_store4<NT>(reinterpret_cast<float4*>(top_flag + vox * Cd + ch), one4);
      }
    }
    // ---- C: labels, 64 * Cl contiguous outputs of this wave
    const int nl = nv * Cl;
    if (Cl >= 4) {
      // lane = (voxel, class quad): a voxel's Cl classes are ceil(Cl / 4) dword-aligned 16-byte pieces (the last one
      // anchored at the row's END, so that nothing is read past a pixel's classes: it overlaps the one before and
      // delivers only the classes that one does not), 64 / nq voxels per step: 7 steps for 22 classes where the
      // scalar form took 22, each a chain of list trips.
      typedef float v4f __attribute__((ext_vector_type(4)));
      typedef v4f __attribute__((aligned(4), may_alias)) v4f_u;   // 88-byte rows: dword-aligned multi-dword accesses
      const int nq = (Cl + 3) >> 2, vps = 64 / nq;               // voxels per step
      const int vs = lane / nq, q = lane - vs * nq;
      const int cb = q == nq - 1 ? Cl - 4 : 4 * q;               // first class of this lane's piece
      const int first_new = 4 * q - cb;                          // elements below this index belong to the previous piece
#pragma unroll 1
      for (int v0 = 0; v0 < nv; v0 += vps) {
        const int vl = v0 + vs;
        const bool live = vs < vps && vl < nv;
        const int c = live ? s_cnt[wib][vl] : -1;
        v4f out = (v4f){0.f, 0.f, 0.f, 0.f};
        if (c == 0) out = *reinterpret_cast<const v4f_u*>(label_3d + lbase + (long long)vl * Cl + cb);
        if (c > 0) {
          const int org = s_org[wib][vl];
          const float* lb = label + ((vbase + vl) / G3) * H * W * (long long)Cl + cb;
          v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
          for (int k0 = 0; k0 < c; k0 += 8) {
            const uint4 r8 = *reinterpret_cast<const uint4_a*>(&s_rel[wib][vl][k0]);
            const unsigned rr[4] = {r8.x, r8.y, r8.z, r8.w};
            v4f v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int rel = (int)((rr[j >> 1] >> (16 * (j & 1))) & 0xffffu);
              v[j] = *reinterpret_cast<const v4f_u*>(lb + (org + (k0 + j < c ? rel : (int)(rr[0] & 0xffffu))) * Cl);
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
              if (k0 + j < c) acc += v[j];
          }
          const float cf = (float)c;
#pragma unroll
          for (int e = 0; e < 4; e++) out[e] = div_rn(acc[e], cf);
        }
        if (c >= 0) {
          float* dst = top_label + lbase + (long long)vl * Cl + cb;
          if (first_new == 0) *reinterpret_cast<v4f_u*>(dst) = out;
          else {
#pragma unroll
            for (int e = 0; e < 4; e++)
              if (e >= first_new) dst[e] = out[e];
          }
        }
      }
    } else {
      // fewer than 4 classes: one output per lane and trip, the label_3d values of 8 trips requested up front
      for (int i0 = 0; i0 < nl; i0 += 64 * 8) {
        float l3[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int i = i0 + 64 * j + lane;
          l3[j] = label_3d[lbase + (i < nl ? i : nl - 1)];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int i = i0 + 64 * j + lane;
          if (i0 + 64 * j >= nl) break;             // (wave-uniform)
          if (i >= nl) continue;
          const int vl = (int)__umulhi((unsigned)i, cl_magic), cl = i - vl * Cl;
          const int c = s_cnt[wib][vl];
          float out = l3[j];
          if (c) {
            const int org = s_org[wib][vl];
            const float* lb = label + ((vbase + vl) / G3) * H * W * (long long)Cl + cl;
            float acc = 0.f;
            for (int k = 0; k < c; k++) acc += lb[(org + (int)s_rel[wib][vl][k]) * Cl];
            out = div_rn(acc, (float)c);
          }
          top_label[lbase + i] = out;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the lists are rewritten by the next trip's phase A
  }
}

// BackprojectBackward, backprojecting_op_gpu.cu.cc:159-217
__global__ __launch_bounds__(256) void backproject_bwd_kernel(
    const float* __restrict__ top_diff, const float* __restrict__ depth,
    const float* __restrict__ meta, float* __restrict__ bottom_diff, long long total, int H, int W,
    int Cd, int num_meta, int G)
{
  for (long long index = (long long)blockIdx.x * 256 + threadIdx.x; index < total;
       index += (long long)gridDim.x * 256) {
    long long t = index;
    const int c = (int)(t % Cd); t /= Cd;
    const long long pix = t;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int n = (int)t;
    const float* md = meta + (size_t)n * num_meta;
    const float dep = depth[pix];
    float RX = md[9] * w + md[10] * h + md[11];
    float RY = md[12] * w + md[13] * h + md[14];
    float RZ = md[15] * w + md[16] * h + md[17];
    float X = dep * RX, Y = dep * RY, Z = dep * RZ;
    float X1 = md[30] * X + md[31] * Y + md[32] * Z + md[33];
    float Y1 = md[34] * X + md[35] * Y + md[36] * Z + md[37];
    float Z1 = md[38] * X + md[39] * Y + md[40] * Z + md[41];
    int vd = round_to_int_sat(div_rn(X1 - md[45], md[42]));
    int vh = round_to_int_sat(div_rn(Y1 - md[46], md[43]));
    int vw = round_to_int_sat(div_rn(Z1 - md[47], md[44]));
    float g = 0.f;
    if (vd >= 0 && vd < G && vh >= 0 && vh < G && vw >= 0 && vw < G)
      g = top_diff[((((long long)n * G + vd) * G + vh) * G + vw) * Cd + c];
    bottom_diff[index] = g;
  }
}

int validate(int B, int H, int W, int Cd, int num_meta, int G)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cd >= 1, PCNN_EINVAL, "backproject: data must be 4-dimensional (got %dx%dx%dx%d)", B, H, W, Cd);
  PCNN_REQUIRE(num_meta >= 48, PCNN_EINVAL, "backproject: meta data needs 48 values per image (got %d)", num_meta);
  PCNN_REQUIRE(G >= 1, PCNN_EINVAL, "backproject: Need grid_size >= 1, got %d", G);
  return PCNN_OK;
}

inline int grid_for(long long total)
{
  long long b = (total + 255) / 256;
  return (int)(b < 256 * 64 ? b : 256 * 64);
}

}  // namespace

extern "C" int pcnn_backproject_workspace_bytes(int B, int H, int W, int ksize, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "backproject_workspace_bytes: NULL pointer");
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && ksize >= 0, PCNN_EINVAL, "backproject_workspace_bytes: bad shape %dx%dx%d, kernel_size %d", B, H, W, ksize);
  // the window-range table: one (min, max) pair per window centre that meets the image; windows of more than 64
  // pixels (k > 3) take the per-channel kernels, which do not use it
  *bytes = ksize <= 3 ? (size_t)B * (size_t)(H + 2 * ksize) * (size_t)(W + 2 * ksize) * sizeof(float2) : 0;
  return PCNN_OK;
}

static int backproject_fwd_impl(const float* data, const float* label, const float* depth,
                                const float* meta, const float* label_3d, int B, int H, int W,
                                int Cd, int Cl, int num_meta, int G, int ksize, float threshold,
                                float* top_data, float* top_label, float* top_flag, void* ws, size_t ws_bytes,
                                void* stream_)
{
  int st = validate(B, H, W, Cd, num_meta, G);
  if (st != PCNN_OK) return st;
  // attribute checks, backprojecting_op.cc:303-320
  PCNN_REQUIRE(ksize >= 0, PCNN_EINVAL, "backproject: Need kernel_size >= 0, got %d", ksize);
  PCNN_REQUIRE(threshold >= 0, PCNN_EINVAL, "backproject: Need threshold >= 0, got %g", (double)threshold);
  PCNN_REQUIRE(Cl >= 1, PCNN_EINVAL, "backproject: label must be 4-dimensional (num_classes %d)", Cl);
  PCNN_REQUIRE(data && label && depth && meta && label_3d && top_data && top_label && top_flag,
               PCNN_ENULL, "backproject: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const long long nvox = (long long)B * G * G * G;
  const bool vec = (Cd % 4 == 0) && aligned16(data) && aligned16(top_data) && aligned16(top_flag);
  const int lpv = Cd >= 64 ? 16 : Cd / 4;
  // Cl <= 256: phase C of the fused kernel maps ceil(Cl / 4) class quads of a voxel onto the 64 lanes of a wave (64 / nq voxels
  // per step) — more quads than lanes would make that step count 0 (the loop then never advances); wider label
  // rows take the per-channel kernels below
  const bool fused = vec && Cl <= 256 && ksize <= 3 && G <= 1024 && 7ll * W < 65536 && ((long long)H + 8) * W * (Cd > Cl ? Cd : Cl) < (1ll << 31) && (Cd % 64 == 0 || Cd == 32 || Cd == 16 || Cd == 8 || Cd == 4);
  if (fused) {
    float2* wrange = nullptr;
    if (ws) {
      size_t need = 0;
      pcnn_backproject_workspace_bytes(B, H, W, ksize, &need);
      PCNN_REQUIRE(ws_bytes >= need, PCNN_EWORKSPACE, "backproject: workspace too small (%zu < %zu bytes)", ws_bytes, need);
      PCNN_REQUIRE(aligned16(ws), PCNN_EINVAL, "backproject: workspace must be 16-byte aligned");
      wrange = static_cast<float2*>(ws);
      PCNN_LAUNCH(backproject_window_range_kernel, dim3((unsigned)((W + 2 * ksize + 63) / 64), (unsigned)((H + 2 * ksize + 15) / 16), (unsigned)B),
                  dim3(256), 0, stream, depth, wrange, H, W, ksize);
    }
    const long long nwave = (nvox + 63) / 64;
    const long long blocks = (nwave + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 64 ? blocks : 256 * 64));
    const int lab_vec = aligned16(label_3d) && aligned16(top_label) ? 1 : 0;
    static const bool nt = [] { const char* e = getenv("PCNN_BP_NT"); return !(e && e[0] == '0'); }();   // A/B switch (default: non-temporal stores)
#define BP_GO2(L, N) PCNN_LAUNCH((backproject_fused_kernel<L, N>), grid, dim3(256), 0, stream, data, label, depth, meta, label_3d, wrange, \
                                 top_data, top_label, top_flag, nvox, H, W, Cd, Cl, num_meta, G, ksize, threshold, lab_vec)
#define BP_GO(L) do { if (nt) BP_GO2(L, true); else BP_GO2(L, false); } while (0)
    if (lpv == 16) BP_GO(16); else if (lpv == 8) BP_GO(8); else if (lpv == 4) BP_GO(4); else if (lpv == 2) BP_GO(2); else BP_GO(1);
#undef BP_GO
#undef BP_GO2
    return pcnn::check_launch("backproject_fwd");
  }
  if (vec) {
    const long long total = nvox * (Cd / 4);
    PCNN_LAUNCH(backproject_data_kernel<4>, dim3(grid_for(total)), dim3(256), 0, stream, data,
                       depth, meta, top_data, top_flag, total, H, W, Cd, num_meta, G, ksize, threshold);
  } else {
    const long long total = nvox * Cd;
    PCNN_LAUNCH(backproject_data_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, data,
                       depth, meta, top_data, top_flag, total, H, W, Cd, num_meta, G, ksize, threshold);
  }
  const long long ltotal = nvox * Cl;
  PCNN_LAUNCH(backproject_label_kernel, dim3(grid_for(ltotal)), dim3(256), 0, stream, label, depth,
                     meta, label_3d, top_label, ltotal, H, W, Cl, num_meta, G, ksize, threshold);
  return pcnn::check_launch("backproject_fwd");
}

extern "C" int pcnn_backproject_fwd(const float* data, const float* label, const float* depth,
                                    const float* meta, const float* label_3d, int B, int H, int W,
                                    int Cd, int Cl, int num_meta, int G, int ksize, float threshold,
                                    float* top_data, float* top_label, float* top_flag,
                                    void* stream_)
{
  return backproject_fwd_impl(data, label, depth, meta, label_3d, B, H, W, Cd, Cl, num_meta, G, ksize, threshold, top_data,
                              top_label, top_flag, nullptr, 0, stream_);
}

extern "C" int pcnn_backproject_ws_fwd(const float* data, const float* label, const float* depth,
                                       const float* meta, const float* label_3d, int B, int H, int W,
                                       int Cd, int Cl, int num_meta, int G, int ksize, float threshold,
                                       float* top_data, float* top_label, float* top_flag,
                                       void* workspace, size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(workspace || workspace_bytes == 0, PCNN_ENULL, "backproject: NULL workspace with %zu bytes", workspace_bytes);
  return backproject_fwd_impl(data, label, depth, meta, label_3d, B, H, W, Cd, Cl, num_meta, G, ksize, threshold, top_data,
                              top_label, top_flag, workspace_bytes ? workspace : nullptr, workspace_bytes, stream_);
}

extern "C" int pcnn_backproject_bwd(const float* top_diff, const float* depth, const float* meta,
                                    int B, int H, int W, int Cd, int num_meta, int G,
                                    float* bottom_diff, void* stream_)
{
  int st = validate(B, H, W, Cd, num_meta, G);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(top_diff && depth && meta && bottom_diff, PCNN_ENULL, "backproject_bwd: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * H * W * Cd;
  PCNN_LAUNCH(backproject_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, top_diff, depth,
                     meta, bottom_diff, total, H, W, Cd, num_meta, G);
  return pcnn::check_launch("backproject_bwd");
}
