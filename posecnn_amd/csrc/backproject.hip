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

// Fused forward for windows of at most 64 pixels ((2k+1)^2 <= 64, i.e. k <= 3 — the reference's
// k = 3, vgg16.py:131-132). A wave owns 64 consecutive voxels:
//   A  lane = voxel: ONE projection + ONE window scan per voxel, kept as a 64-bit mask of the
//      matching window pixels (bit = column-major position inside the unclipped window, which is
//      the reference's summation order) — the kernels above redo this in every channel's thread;
//   B  lane = (voxel of a group of 64/LPV, channel quad): masks travel by shuffle, the set bits are
//      walked in ascending order (`acc += v`, same order as the nested loops) and every store
//      instruction writes 64/LPV voxels x Cd floats contiguously (data and flag);
//   C  lane = flattened (voxel, class) of the wave's 64 x Cl contiguous label outputs.
template <int LPV>  // lanes per voxel in phase B = min(Cd, 64) / 4
__global__ __launch_bounds__(256) void backproject_fused_kernel(
    const float* __restrict__ data, const float* __restrict__ label, const float* __restrict__ depth,
    const float* __restrict__ meta, const float* __restrict__ label_3d, float* __restrict__ top_data,
    float* __restrict__ top_label, float* __restrict__ top_flag, long long nvox, int H, int W, int Cd,
    int Cl, int num_meta, int G, int ksize, float threshold)
{
  const int lane = threadIdx.x & 63;
  const long long nwave = (nvox + 63) / 64;
  const int S = 2 * ksize + 1;
  const long long G3 = (long long)G * G * G;
  for (long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < nwave; wv += (long long)gridDim.x * 4) {
    const long long vbase = wv * 64;
    // ---- A: one voxel per lane
    unsigned long long mask = 0;
    int px = 0, py = 0;  // window origin (column, row) of this lane's voxel
    {
      const long long vox = vbase + lane;
      if (vox < nvox) {
        long long t = vox;
        const int w = (int)(t % G); t /= G;
        const int h = (int)(t % G); t /= G;
        const int d = (int)(t % G); t /= G;
        const int n = (int)t;
        const float* md = meta + (size_t)n * num_meta;
        const Proj p = project_voxel(md, d, h, w);
        const Window win = clip_window(p, ksize, H, W);
        // window origin; clamped so that garbage projections (saturated to INT_MIN/MAX, empty
        // window, never dereferenced) cannot overflow the subtraction
        px = (int)max(-(1ll << 30), min(1ll << 30, (long long)p.px - ksize));
        py = (int)max(-(1ll << 30), min(1ll << 30, (long long)p.py - ksize));
        const float* dn = depth + (long long)n * H * W;
        for (int x = win.xlo; x <= win.xhi; x++)
          for (int y = win.ylo; y <= win.yhi; y++)
            if (fabsf(dn[(long long)y * W + x] - p.Z1) < threshold)
              mask |= 1ull << ((x - px) * S + (y - py));
      }
    }
    const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
    // ---- B: data + flag
    constexpr int VPI = 64 / LPV;  // voxels per iteration
    const int sub = lane / LPV, cq = lane % LPV;
    for (int it = 0; it < 64 / VPI; it++) {
      const int vl = it * VPI + sub;
      const long long vox = vbase + vl;
      unsigned long long m = ((unsigned long long)(unsigned)__shfl((int)mhi, vl) << 32) | (unsigned)__shfl((int)mlo, vl);
      const int vx = __shfl(px, vl), vy = __shfl(py, vl);
      if (vox >= nvox) continue;
      const long long n = vox / G3;
      const float* dbase = data + n * H * W * (long long)Cd;
      const float cnt = (float)__popcll(m);
      for (int c = cq * 4; c < Cd; c += LPV * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long mm = m;
        while (mm) {
          const int b = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          const int x = vx + b / S, y = vy + b % S;
          const float4 v = *reinterpret_cast<const float4*>(dbase + ((long long)y * W + x) * Cd + c);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float flag = 0.f;
        if (m) {
          acc.x = div_rn(acc.x, cnt); acc.y = div_rn(acc.y, cnt); acc.z = div_rn(acc.z, cnt); acc.w = div_rn(acc.w, cnt);
          flag = 1.f;
        }
        *reinterpret_cast<float4*>(top_data + vox * Cd + c) = acc;
        *reinterpret_cast<float4*>(top_flag + vox * Cd + c) = make_float4(flag, flag, flag, flag);
      }
    }
    // ---- C: labels, 64 * Cl contiguous outputs of this wave
    const long long lbase = vbase * Cl;
    for (int i = lane; i < 64 * Cl; i += 64) {
      const int vl = i / Cl, cl = i - vl * Cl;
      // (every lane reaches the shuffles: 64 * Cl is a multiple of 64)
      unsigned long long m = ((unsigned long long)(unsigned)__shfl((int)mhi, vl) << 32) | (unsigned)__shfl((int)mlo, vl);
      const int vx = __shfl(px, vl), vy = __shfl(py, vl);
      const long long vox = vbase + vl;
      if (vox >= nvox) continue;
      const long long n = vox / G3;
      const float* lb = label + n * H * W * (long long)Cl;
      float acc = 0.f;
      const float cnt = (float)__popcll(m);
      unsigned long long mm = m;
      while (mm) {
        const int b = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int x = vx + b / S, y = vy + b % S;
        acc += lb[((long long)y * W + x) * Cl + cl];
      }
      top_label[lbase + i] = m ? div_rn(acc, cnt) : label_3d[lbase + i];
    }
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

extern "C" int pcnn_backproject_fwd(const float* data, const float* label, const float* depth,
                                    const float* meta, const float* label_3d, int B, int H, int W,
                                    int Cd, int Cl, int num_meta, int G, int ksize, float threshold,
                                    float* top_data, float* top_label, float* top_flag,
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
  const bool fused = vec && (2 * ksize + 1) * (2 * ksize + 1) <= 64 && (Cd % 64 == 0 || Cd == 32 || Cd == 16 || Cd == 8 || Cd == 4);
  if (fused) {
    const long long nwave = (nvox + 63) / 64;
    const long long blocks = (nwave + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 64 ? blocks : 256 * 64));
#define BP_GO(L) PCNN_LAUNCH(backproject_fused_kernel<L>, grid, dim3(256), 0, stream, data, label, depth, meta, label_3d, \
                             top_data, top_label, top_flag, nvox, H, W, Cd, Cl, num_meta, G, ksize, threshold)
    if (lpv == 16) BP_GO(16); else if (lpv == 8) BP_GO(8); else if (lpv == 4) BP_GO(4); else if (lpv == 2) BP_GO(2); else BP_GO(1);
#undef BP_GO
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
