// wino_mfma.hip — the Winograd-domain contractions of the VGG16 trunk on the gfx950 matrix cores,
// with the output transform fused in (`Network.conv` for every 3x3 / stride 1 / SAME layer with
// Cin, Cout multiples of 64: lib/networks/network.py:159-187, lib/networks/vgg16_convs.py:37-52 and
// the depth tower :54-67). Replaces the library batched GEMM + wino43_output_kernel pair of round 1
// and its Cin = 64-only fused kernel.
//
//   y = [ReLU](A^T (sum_ci V[k][t][ci] U[k][ci][co]) A + bias[co])   [2x2 max-pooled]
//
// F(4x4,3x3): 36 planes k = 6 xi + nu, each a GEMM [T x Cin] . [Cin x Cout] in exact f32 on
// v_mfma_f32_16x16x4_f32 (no xf32/TF32 on gfx950; 157 TFLOP/s peak). Design for the chip, not for a
// generic GEMM:
//
//  * Workgroup = 64 tiles x 64 output channels, 8 waves (2 per SIMD so that one wave's LDS / global
//    / VALU work hides under the other's MFMAs). Wave (wm, wn) owns tiles [32 wm, 32 wm + 32) x
//    channels [16 wn, 16 wn + 16): two 16x16 accumulator blocks.
//  * The transform-domain product M NEVER exists, not even in registers as a whole: planes are
//    processed column by column (nu outer, xi inner). Once the 6 planes of column nu are summed
//    over all of Cin, t = A^T M[:, nu] (4 values) is folded into the 16 outputs Y += t (x) A[nu, :]
//    and the 6 accumulators are recycled. 22 live values per (tile, channel) instead of 36 is what
//    lets a 64 x 64 block fit the register file (360 KB of the CU's 512 KB) — 4x the tile of the
//    round-1 kernel, i.e. 2.5x fewer operand bytes per flop through L2 -> LDS (16 B/clk/CU).
//  * Operands go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no
//    ds_write pass), double buffered, ONE barrier per 64-deep K stage: the 4 DMA instructions a
//    wave issues for stage s+1 fly under the 32 MFMAs of stage s. The LDS image is the DMA's
//    lane-linear one (rows of 64 floats, unpadded); bank conflicts are avoided by an XOR swizzle of
//    the 16-byte chunk index with the row (applied to the SOURCE address of the DMA and to the
//    ds_read_b128 address — the same involution on both sides). One b128 read feeds 4 MFMAs (K is
//    consumed in the order 16 j + 4 (lane >> 4) + i for both operands).
//  * Epilogue: bias + ReLU (+ 2x2 max-pool of the 4x4 tile) in registers, then through LDS so that
//    every store instruction writes whole 256-byte channel rows (the C/D layout alone would give
//    64-byte segments).
//  * blockIdx -> (tile block, channel block) is XCD-aware: a tile block's V rows are shared by the
//    Cout / 64 workgroups that run back to back on ONE XCD (same L2), tile blocks are dealt round
//    robin to the 8 XCDs.
//  * `groups`: G independent filter sets over G equal slices of the tile range — the colour and the
//    depth tower of an RGB-D network (identical shapes, different weights) run as ONE launch with
//    twice the workgroups.
#include "pcnn_device.h"

namespace {

using namespace pcnn;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WM_BT = 64;     // tiles per workgroup
constexpr int WM_BC = 64;     // output channels per workgroup
constexpr int WM_KC = 64;     // K (input channels) per pipeline stage
constexpr int WM_LD = 64;     // LDS row (floats): unpadded, XOR-swizzled 16-byte chunks
constexpr int WM_THREADS = 512;

// 16 bytes per lane, global -> LDS, no register round trip. LDS destination = wave-uniform base +
// 16 * lane (so the image is lane-linear); the per-lane SOURCE address carries the swizzle.
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Output transform, one column of the 6x6 transform domain at a time:
//   t = A^T m   with A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6_col(const float* m, float* t)
{
  const float s = m[1] + m[2], d = m[1] - m[2], S = m[3] + m[4], D = m[3] - m[4];
  t[0] = (m[0] + s) + S;
  t[1] = d + 2.f * D;
  t[2] = s + 4.f * S;
  t[3] = (d + 8.f * D) + m[5];
}

// POOL: 0 = y [.,H,W,Cout]; 1 = only max_pool_2x2(y) (written to y); 2 = both (y and ypool)
template <int POOL>
__global__ __launch_bounds__(WM_THREADS, 2) void wino43_mfma_kernel(
    const float* __restrict__ v, const float* __restrict__ ut, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ ypool, int H, int W, int Cin, int Cout, int Ht, int Wt,
    long long T, long long tiles_per_group, int relu, int nbt, int ncb)
{
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 64 * WM_LD];   // sA[2][64][68] | sB[2][64][68]
  float(*sA)[64][WM_LD] = reinterpret_cast<float(*)[64][WM_LD]>(smem);
  float(*sB)[64][WM_LD] = reinterpret_cast<float(*)[64][WM_LD]>(smem + 2 * 64 * WM_LD);

  // XCD-aware block map: XCD x = blockIdx % 8 takes tile blocks tb == x (mod 8); on an XCD the
  // channel blocks of a tile block are consecutive
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int cb = q % ncb;
  const int tb = (q / ncb) * 8 + x;
  if (tb >= nbt) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, lk = lane >> 4;
  // tile blocks never straddle a group: group g owns tiles [g tpg, (g+1) tpg) in nbg blocks of 64
  const int nbg = (int)((tiles_per_group + WM_BT - 1) / WM_BT);
  const int grp = tb / nbg;
  const long long t0 = (long long)grp * tiles_per_group + (long long)(tb - grp * nbg) * WM_BT;
  const long long tend = (long long)(grp + 1) * tiles_per_group;   // first tile that is not this block's business
  const float* utg = ut + (size_t)grp * 36 * Cout * Cin;
  const int NK = Cin / WM_KC;

  // staging: a stage = 64 rows x 256 B of V and of U^T = 2 x 16 DMA instructions of 4 rows each;
  // wave w issues instructions 2w and 2w + 1 of both. Lane l of instruction ii lands at row
  // 4 ii + (l >> 4), physical chunk l & 15, and therefore fetches logical chunk (l & 15) ^ (row & 15).
  const int dr0 = 8 * wave + lk, dr1 = dr0 + 4;                 // the lane's rows in its two instructions
  const int dc0 = (lr ^ (dr0 & 15)) * 4, dc1 = (lr ^ (dr1 & 15)) * 4;
  const long long tlast = tend - 1;
  const long long ta0 = t0 + dr0 < tend ? t0 + dr0 : tlast, ta1 = t0 + dr1 < tend ? t0 + dr1 : tlast;   // rows past the end: any finite data, never stored
  const float* va0 = v + ta0 * Cin + dc0;
  const float* va1 = v + ta1 * Cin + dc1;
  const float* ub0 = utg + (size_t)(cb * WM_BC + dr0) * Cin + dc0;
  const float* ub1 = utg + (size_t)(cb * WM_BC + dr1) * Cin + dc1;
  const long long vplane = T * Cin;
  const long long uplane = (long long)Cout * Cin;
  const int ldsw = 8 * wave * WM_LD;                            // this wave's first row (floats) inside a block

  v4f acc[6][2];
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i][0] = acc[i][1] = (v4f){0.f, 0.f, 0.f, 0.f};
  float yo[2][4][16];   // [block][tile row i of the C/D layout][4 a + e]
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int o = 0; o < 16; o++) yo[b][i][o] = 0.f;

#define WM_DMA(BUF, VO, UO)                                                  \
  do {                                                                       \
    glds16(va0 + (VO), &sA[BUF][0][0] + ldsw);                               \
    glds16(va1 + (VO), &sA[BUF][0][0] + ldsw + 4 * WM_LD);                   \
    glds16(ub0 + (UO), &sB[BUF][0][0] + ldsw);                               \
    glds16(ub1 + (UO), &sB[BUF][0][0] + ldsw + 4 * WM_LD);                   \
  } while (0)

  // stage 0 = plane k = 0 (xi = 0, nu = 0), kc = 0
  WM_DMA(0, 0, 0);
  __syncthreads();
  int cur = 0;

  // ds_read side of the swizzle: row R = (..) + lr, logical chunk 4 j + lk -> physical (4 j + lk) ^ lr
  const int ra0 = (32 * wm + lr) * WM_LD, ra1 = ra0 + 16 * WM_LD, rb = (16 * wn + lr) * WM_LD;

  // One pipeline stage on accumulator set XI. `nk`/`nkc` = plane and K slice of the NEXT stage.
#define WM_STAGE(XI)                                                                                  \
  do {                                                                                                \
    int nkc = kc + 1, nxi = (XI), nnu = nu;                                                           \
    if (nkc == NK) { nkc = 0; nxi = (XI) + 1; if (nxi == 6) { nxi = 0; nnu = nu + 1; } }               \
    if (nnu < 6) {                                                                                    \
      const long long nk = 6 * nxi + nnu;                                                             \
      WM_DMA(cur ^ 1, nk * vplane + nkc * WM_KC, nk * uplane + nkc * WM_KC);                           \
    }                                                                                                 \
    const float* pa = &sA[cur][0][0];                                                                 \
    const float* pb = &sB[cur][0][0];                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                   \
      const int ch = ((4 * j + lk) ^ lr) * 4;                                                         \
      const v4f fb = *reinterpret_cast<const v4f*>(pb + rb + ch);                                     \
      const v4f fa0 = *reinterpret_cast<const v4f*>(pa + ra0 + ch);                                   \
      const v4f fa1 = *reinterpret_cast<const v4f*>(pa + ra1 + ch);                                   \
      _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                 \
        acc[XI][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[i], fb[i], acc[XI][0], 0, 0, 0);         \
        acc[XI][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[i], fb[i], acc[XI][1], 0, 0, 0);         \
      }                                                                                               \
    }                                                                                                 \
    __syncthreads();                                                                                  \
    cur ^= 1;                                                                                         \
  } while (0)

  for (int nu = 0; nu < 6; nu++) {
    for (int kc = 0; kc < NK; kc++) WM_STAGE(0);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(1);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(2);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(3);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(4);
    for (int kc = 0; kc < NK; kc++) WM_STAGE(5);
    // column nu of the transform domain is complete: t = A^T M[:, nu]; Y[a][e] += t[a] * A[nu][e]
    // with A[nu][:] = (1,0,0,0) (1,1,1,1) (1,-1,1,-1) (1,2,4,8) (1,-2,4,-8) (0,0,0,1)
    const float c0 = nu == 5 ? 0.f : 1.f;
    const float c1 = nu == 1 ? 1.f : nu == 2 ? -1.f : nu == 3 ? 2.f : nu == 4 ? -2.f : 0.f;
    const float c2 = (nu == 1 || nu == 2) ? 1.f : (nu == 3 || nu == 4) ? 4.f : 0.f;
    const float c3 = nu == 1 ? 1.f : nu == 2 ? -1.f : nu == 3 ? 8.f : nu == 4 ? -8.f : nu == 5 ? 1.f : 0.f;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float m[6], t[4];
#pragma unroll
        for (int xi = 0; xi < 6; xi++) m[xi] = acc[xi][b][i];
        at6_col(m, t);
#pragma unroll
        for (int a = 0; a < 4; a++) {
          yo[b][i][4 * a + 0] += t[a] * c0;
          yo[b][i][4 * a + 1] += t[a] * c1;
          yo[b][i][4 * a + 2] += t[a] * c2;
          yo[b][i][4 * a + 3] += t[a] * c3;
        }
      }
#pragma unroll
    for (int xi = 0; xi < 6; xi++) acc[xi][0] = acc[xi][1] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
#undef WM_STAGE
#undef WM_DMA

  // ---- epilogue -------------------------------------------------------------------------------
  // lane holds, per block b, tiles 32 wm + 16 b + 4 lk + i (i = 0..3) x channel 16 wn + lr
  const int col = 16 * wn + lr;
  const int co = cb * WM_BC + col;
  const float bv = bias[(size_t)grp * Cout + co];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int o = 0; o < 16; o++) {
        float val = yo[b][i][o] + bv;
        if (relu) val = val > 0.f ? val : 0.f;
        yo[b][i][o] = val;
      }

  float* sY = smem;   // [64 tiles][4][64 channels] floats = 64 KB (the staging buffers are free now)
  const int HtWt = Ht * Wt;
  if (POOL != 1) {
    // four passes, one output row a of the 4x4 tiles each: [tile][e][channel]
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int tl = 32 * wm + 16 * b + 4 * lk + i;
#pragma unroll
          for (int e = 0; e < 4; e++) sY[(tl * 4 + e) * 64 + col] = yo[b][i][4 * a + e];
        }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int idx = tid + WM_THREADS * r;   // 4096 float4: (tile, e, c4)
        const int tl = idx >> 6, e = (idx >> 4) & 3, c4 = (idx & 15) * 4;
        const long long t = t0 + tl;
        if (t < tend) {
          const int bimg = (int)(t / HtWt);
          const int rem = (int)(t - (long long)bimg * HtWt);
          const int ty = rem / Wt, tx = rem - ty * Wt;
          const int oy = 4 * ty + a, ox = 4 * tx + e;
          if (oy < H && ox < W)
            *reinterpret_cast<v4f*>(y + (((long long)bimg * H + oy) * W + ox) * Cout + cb * WM_BC + c4) =
                *reinterpret_cast<const v4f*>(&sY[(tl * 4 + e) * 64 + c4]);
        }
      }
      __syncthreads();
    }
  }
  if (POOL != 0) {
    // pooled 2x2 windows of the 4x4 tile: [tile][2 a' + e'][channel]
    float* yp = POOL == 1 ? y : ypool;
    const int Hp = H / 2, Wp = W / 2;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int tl = 32 * wm + 16 * b + 4 * lk + i;
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
          for (int e2 = 0; e2 < 2; e2++) {
            float p = yo[b][i][4 * (2 * a2) + 2 * e2];
            const float p1 = yo[b][i][4 * (2 * a2) + 2 * e2 + 1], p2 = yo[b][i][4 * (2 * a2 + 1) + 2 * e2],
                        p3 = yo[b][i][4 * (2 * a2 + 1) + 2 * e2 + 1];
            p = p1 > p ? p1 : p;
            p = p2 > p ? p2 : p;
            p = p3 > p ? p3 : p;
            sY[(tl * 4 + 2 * a2 + e2) * 64 + col] = p;
          }
      }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int idx = tid + WM_THREADS * r;
      const int tl = idx >> 6, w4 = (idx >> 4) & 3, c4 = (idx & 15) * 4;
      const long long t = t0 + tl;
      if (t < tend) {
        const int bimg = (int)(t / HtWt);
        const int rem = (int)(t - (long long)bimg * HtWt);
        const int ty = rem / Wt, tx = rem - ty * Wt;
        const int py = 2 * ty + (w4 >> 1), px = 2 * tx + (w4 & 1);
        if (py < Hp && px < Wp)
          *reinterpret_cast<v4f*>(yp + (((long long)bimg * Hp + py) * Wp + px) * Cout + cb * WM_BC + c4) =
              *reinterpret_cast<const v4f*>(&sY[(tl * 4 + w4) * 64 + c4]);
      }
    }
  }
}

}  // namespace

extern "C" int pcnn_winograd43_conv_fwd(const float* v, const float* ut, const float* bias, int B, int H,
                                        int W, int Cin, int Cout, int groups, int relu, int pool,
                                        float* y, float* y_pool, void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1, PCNN_EINVAL, "winograd43_conv: bad shape %dx%dx%d", B, H, W);
  PCNN_REQUIRE(Cin >= 64 && Cin % 64 == 0, PCNN_EINVAL, "winograd43_conv: input channels must be a multiple of 64 (got %d)", Cin);
  PCNN_REQUIRE(Cout >= 64 && Cout % 64 == 0, PCNN_EINVAL, "winograd43_conv: output channels must be a multiple of 64 (got %d)", Cout);
  PCNN_REQUIRE(pool >= 0 && pool <= 2, PCNN_EINVAL, "winograd43_conv: pool must be 0 (none), 1 (pooled only) or 2 (both)");
  PCNN_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), PCNN_EINVAL, "winograd43_conv: pooling needs even height/width");
  PCNN_REQUIRE(groups >= 1 && B % groups == 0, PCNN_EINVAL, "winograd43_conv: batch %d is not a multiple of groups %d", B, groups);
  PCNN_REQUIRE(v && ut && bias && y && (pool != 2 || y_pool), PCNN_ENULL, "winograd43_conv: NULL pointer");
  PCNN_REQUIRE(aligned16(v) && aligned16(ut) && aligned16(y) && (pool != 2 || aligned16(y_pool)), PCNN_EINVAL,
               "winograd43_conv: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const long long T = (long long)B * Ht * Wt;
  const long long tpg = T / groups;
  const long long nbt = (long long)groups * ((tpg + WM_BT - 1) / WM_BT);
  const int ncb = Cout / WM_BC;
  const long long blocks = ((nbt + 7) / 8) * 8 * ncb;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "winograd43_conv: grid too large");
#define WM_GO(P) PCNN_LAUNCH((wino43_mfma_kernel<P>), dim3((unsigned)blocks), dim3(WM_THREADS), 0, stream, v, ut, bias, y, y_pool, \
                             H, W, Cin, Cout, Ht, Wt, T, tpg, relu, (int)nbt, ncb)
  if (pool == 0) WM_GO(0);
  else if (pool == 1) WM_GO(1);
  else WM_GO(2);
#undef WM_GO
  return check_launch("winograd43_conv_fwd");
}
