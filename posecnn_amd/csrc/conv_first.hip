// conv_first.hip — the first layer of each VGG tower (conv1_1 / conv1_1_p, vgg16_convs.py:36,53:
// 3x3, stride 1, SAME, 3 -> 64 channels) with its bias_add + ReLU (network.py:181-187) fused.
//
// With 3 input channels the layer is no GEMM worth the name: K = 27, so the implicit-GEMM library
// kernel runs at 33 TFLOP/s and a separate bias/ReLU pass re-reads and re-writes the 78.6 MB/frame
// output (0.52 + 0.53 ms per 16 frames, tools/bench_layers.py). The layer is bound by WRITING its
// output once: 4*H*W*64 B per frame, 1.26 GB per 16 frames at 640x480. This kernel does exactly that.
//
// Work split (wave64): lane = (pixel slot 0..3) x (channel quad 0..15). A workgroup owns a strip of
// CF_ROWS rows x CF_SEG columns: its (CF_ROWS+2) x (CF_SEG+2) x 3 input window sits in LDS (zero
// filled outside the image) and is read as 4-address broadcasts; a lane keeps the 27 x 4 weights of
// its channel quad in VGPRs for the whole strip (loaded once, as 27 float4 from the TF-layout
// [ky,kx,ci,co] filter) and walks pixels slot, slot+4, ...; every store instruction writes
// 4 pixels x 256 B = 1 KB contiguous.
// Arithmetic: acc = fma(w, x, acc) over (ky, kx, ci) ascending from 0, then + bias, then ReLU.
#include <cstdlib>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr int CF_SEG = 128;   // output columns per workgroup
constexpr int CF_ROWS = 16;   // output rows per workgroup
constexpr int CF_CIN = 3;
constexpr int CF_ROWF = (CF_SEG + 2) * CF_CIN;

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void conv3x3_c3_bias_relu_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int H, int W, int Cout, int relu, int nseg, int nstrip)
{
  __shared__ float s_in[CF_ROWS + 2][CF_ROWF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = blockIdx.x % nseg;
  const int strip = (blockIdx.x / nseg) % nstrip;
  const int b = blockIdx.x / (nseg * nstrip);
  const int cg = blockIdx.y;                 // group of 64 output channels
  const int ox0 = seg * CF_SEG, oy0 = strip * CF_ROWS;
  const int npx = min(CF_SEG, W - ox0), nrow = min(CF_ROWS, H - oy0);

  // input window -> LDS: rows oy0-1..oy0+nrow, columns ox0-1..ox0+npx, 3 channels each
  const int rowf = (npx + 2) * CF_CIN;
  for (int i = tid; i < (nrow + 2) * rowf; i += 256) {
    const int r = i / rowf, j = i - r * rowf;
    const int iy = oy0 - 1 + r, ix = ox0 - 1 + j / CF_CIN;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
      v = x[(((size_t)b * H + iy) * W + ix) * CF_CIN + (j % CF_CIN)];
    s_in[r][j] = v;
  }

  // this lane's channel quad: filter [ky][kx][ci][co] -> wq[t] = w[t][c0..c0+3]
  const int quad = lane & 15, slot = lane >> 4;
  const int c0 = cg * 64 + quad * 4;
  f4 wq[27];
#pragma unroll
  for (int t = 0; t < 27; t++) wq[t] = *reinterpret_cast<const f4*>(w + (size_t)t * Cout + c0);
  const f4 bq = *reinterpret_cast<const f4*>(bias + c0);
  __syncthreads();

  // wave w owns columns [w*32, w*32+32) of the strip, 4 pixels at a time, row after row
  for (int r = 0; r < nrow; r++) {
    float* yrow = y + (((size_t)b * H + oy0 + r) * W + ox0) * Cout + c0;
#pragma unroll 2
    for (int it = 0; it < CF_SEG / 4 / 4; it++) {
      const int px = wave * (CF_SEG / 4) + it * 4 + slot;
      if (px < npx) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const float* win = &s_in[r + ky][px * CF_CIN];   // columns px-1..px+1 = 9 floats
#pragma unroll
          for (int j = 0; j < 9; j++) {
            const float v = win[j];
            const f4 vv = {v, v, v, v};
            acc = __builtin_elementwise_fma(wq[ky * 9 + j], vv, acc);
          }
        }
        acc = acc + bq;
        if (relu) {
          acc.x = acc.x > 0.f ? acc.x : 0.f;
          acc.y = acc.y > 0.f ? acc.y : 0.f;
          acc.z = acc.z > 0.f ? acc.z : 0.f;
          acc.w = acc.w > 0.f ? acc.w : 0.f;
        }
        *reinterpret_cast<f4*>(yrow + (size_t)px * Cout) = acc;
      }
    }
  }
}

// conv1_1 + bias + ReLU feeding conv1_2's Winograd F(4x4,3x3) input transform directly: the
// 78.6 MB/frame activation between the two layers is neither written nor read back. A workgroup owns
// one row of FW_TILES 4x4 output tiles (4: 30 KB of LDS, five workgroups per CU; 8 tiles measured 6 % slower): phase 1 computes the (4+2) x (4*FW_TILES+2) pixel patch of
// relu(conv1_1) those tiles need into LDS (zeros outside the image: conv1_2's SAME padding), phase 2
// applies B^T d B to each tile's 6x6 patch with exactly the expressions of wino43_input_kernel
// (csrc/winograd.hip), so V is bit-identical to the unfused pair.
constexpr int FW_TILES = 4;
constexpr int FW_COLS = 4 * FW_TILES + 2;   // 18 patch columns
constexpr int FW_INF = (FW_COLS + 6) * CF_CIN;   // 20 window columns + 4 of slack for the last (partial) pixel quad of a row

__device__ __forceinline__ void fw_bt6(const f4* d, f4* r)
{
  r[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
  const f4 a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1];
  r[1] = a + b;
  r[2] = a - b;
  const f4 c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
  r[3] = c + e;
  r[4] = c - e;
  r[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
}

// RAW: the frames arrive as the sensor delivers them — colour uint8 BGR [Bc,H,W,3] and / or depth uint16 [Bd,H,W] — and
// the network's input blobs (lib/fcn/test.py:56-74: BGR - PIXEL_MEANS; clip(depth / 2000, 0, 1) * 255 tiled to 3 channels
// - PIXEL_MEANS, numpy's float32 -= float64 semantics) are formed while the input window is staged into LDS, value for value
// what posecnn_amd.fcn._get_image_blob builds on the host: V is bit-identical to the blob path, 4x (colour) / 6x (depth)
// fewer bytes cross PCIe and the 3.7 MB per frame f32 blobs never exist. Images [0, Bc) are colour (filter set 0),
// [Bc, Bc + Bd) depth (the next filter set).
struct RawFrames {
  const unsigned char* color;
  const unsigned short* depth;
  int n_color;
  double mean[3];
};

template <bool RAW>
__global__ __launch_bounds__(256) void conv3x3_c3_wino43_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ v, int H, int W, int Cout, int relu, int Ht, int Wt, int nseg,
    long long plane, int imgs_per_group, RawFrames raw)
{
  __shared__ float s_in[8][FW_INF];
  __shared__ __attribute__((aligned(16))) float s_y[6][FW_COLS][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = blockIdx.x % nseg;
  const int ty = (blockIdx.x / nseg) % Ht;
  const int b = blockIdx.x / (nseg * Ht);
  const int cg = blockIdx.y;
  const int tx0 = seg * FW_TILES;
  const int py0 = 4 * ty - 1, px0 = 4 * tx0 - 1;   // image coordinates of patch pixel (0, 0)
  // filter set of this image (groups: the colour and the depth tower in one launch)
  const bool is_depth = RAW && (raw.color == nullptr || b >= raw.n_color);
  const int grp = RAW ? ((raw.color != nullptr && is_depth) ? 1 : 0) : b / imgs_per_group;
  w += (size_t)grp * 27 * Cout;
  bias += (size_t)grp * Cout;

  // input window: rows py0-1 .. py0+6, columns px0-1 .. px0+22 (FW_COLS + 6 = 24 columns incl. the slack of the last pixel quad)
  for (int i = tid; i < 8 * FW_INF; i += 256) {
    const int r = i / FW_INF, j = i - r * FW_INF;
    const int iy = py0 - 1 + r, ix = px0 - 1 + j / CF_CIN, ch = j % CF_CIN;
    float val = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      if (!RAW) {
        val = x[(((size_t)b * H + iy) * W + ix) * CF_CIN + ch];
      } else if (!is_depth) {
        val = (float)((double)raw.color[(((size_t)b * H + iy) * W + ix) * CF_CIN + ch] - raw.mean[ch]);
      } else {
        const int bd = b - (raw.color ? raw.n_color : 0);
        const float t = fminf(fmaxf(div_rn((float)raw.depth[((size_t)bd * H + iy) * W + ix], 2000.f), 0.f), 1.f) * 255.f;
        val = (float)((double)t - raw.mean[ch]);
      }
    }
    s_in[r][j] = val;
  }
  {
    // phase 1: the 6 x FW_COLS patch as quads of horizontally adjacent pixels x PAIRS of output channels: a lane keeps
    // 27 x 2 filter taps (54 VGPRs; the earlier pixel-pair x channel-quad form kept 27 x 4 = 108 and ran at 3 waves
    // per SIMD, latency-bound with 43 % of its wave cycles in s_waitcnt), reads the 18 window floats a quad shares per
    // filter row (4.5 LDS reads per pixel instead of 6; all 32 lanes of a half-wave read the same address: broadcast)
    // and runs four independent FMA chains. Same per-pixel order (ky, kx, ci ascending) -> same bits.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int half = lane >> 5, cp = lane & 31;
    const int c0 = cg * 64 + cp * 2;
    f2 wq[27];
#pragma unroll
    for (int t = 0; t < 27; t++) wq[t] = *reinterpret_cast<const f2*>(w + (size_t)t * Cout + c0);
    const f2 bq = *reinterpret_cast<const f2*>(bias + c0);
    __syncthreads();
    constexpr int NQUAD = (FW_COLS + 3) / 4;   // 5 per patch row, the last one half empty
    for (int it = 0; it < (6 * NQUAD + 7) / 8; it++) {
      const int p = it * 8 + wave * 2 + half;
      if (p < 6 * NQUAD) {
        const int r = p / NQUAD, cx = 4 * (p - r * NQUAD);
        const int yy = py0 + r, xx = px0 + cx;
        f2 acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = (f2){0.f, 0.f};
        const bool rowok = yy >= 0 && yy < H;
        if (rowok) {
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            const float* win = &s_in[r + ky][cx * CF_CIN];   // 18 floats: columns cx-1 .. cx+4
            float wv[18];
#pragma unroll
            for (int j = 0; j < 18; j++) wv[j] = win[j];
#pragma unroll
            for (int j = 0; j < 9; j++)
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const f2 v = {wv[j + 3 * q], wv[j + 3 * q]};
                acc[q] = __builtin_elementwise_fma(wq[ky * 9 + j], v, acc[q]);
              }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f2 a = acc[q] + bq;
          if (relu) {
            a.x = a.x > 0.f ? a.x : 0.f;
            a.y = a.y > 0.f ? a.y : 0.f;
          }
          if (!(rowok && xx + q >= 0 && xx + q < W)) a = (f2){0.f, 0.f};   // outside the image: conv1_2's zero padding
          if (cx + q < FW_COLS) *reinterpret_cast<f2*>(&s_y[r][cx + q][cp * 2]) = a;
        }
      }
    }
  }
  __syncthreads();
  // phase 2: thread = (tile, channel quad, pair of transform rows). Three waves share a tile's patch:
  // wave p produces rows 2p, 2p+1 of B^T d B — the same expression trees as fw_bt6 / wino43_input_kernel,
  // only the rows it owns — so all of the block's transform work is spread over 192 threads instead of
  // sitting in 64 of them while the rest idle (the block is short: occupancy, not arithmetic, is what the
  // kernel is short of).
  if (wave < 3) {
    const int t = (tid & 63) >> 4, q = tid & 15, pr = wave;
    const int tx = tx0 + t;
    if (tx < Wt) {
      f4 tmp[2][6];
#pragma unroll
      for (int s2 = 0; s2 < 6; s2++) {
        f4 d[6];
#pragma unroll
        for (int r = 0; r < 6; r++) d[r] = *reinterpret_cast<const f4*>(&s_y[r][4 * t + s2][q * 4]);
        if (pr == 0) {
          tmp[0][s2] = (4.f * d[0] - 5.f * d[2]) + d[4];
          const f4 a = d[4] - 4.f * d[2], b2 = d[3] - 4.f * d[1];
          tmp[1][s2] = a + b2;
        } else if (pr == 1) {
          const f4 a = d[4] - 4.f * d[2], b2 = d[3] - 4.f * d[1];
          tmp[0][s2] = a - b2;
          const f4 c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
          tmp[1][s2] = c + e;
        } else {
          const f4 c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
          tmp[0][s2] = c - e;
          tmp[1][s2] = (4.f * d[1] - 5.f * d[3]) + d[5];
        }
      }
      const long long tile = ((long long)b * Ht + ty) * Wt + tx;
      float* vo = v + tile * Cout + cg * 64 + q * 4;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        f4 o[6];
        fw_bt6(tmp[i], o);
#pragma unroll
        for (int j = 0; j < 6; j++) *reinterpret_cast<f4*>(vo + (6 * (2 * pr + i) + j) * plane) = o[j];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// conv1_1 -> conv1_2 (-> pool1) in ONE kernel (round 4; VERDICT r3 "Next" #4). Cin = Cout = 64: the whole contraction of
// conv1_2 is one K stage and one channel block, so nothing of it has to leave the CU: the unfused pair writes V
// (2.25 x 78.6 MB per frame) with conv3x3_c3_wino43_kernel and reads it back with wino43_mfma_kernel — 11.3 GB of HBM
// traffic per 2 x 16 frames and the 1.7 ms kernel that is bound by it. Here a workgroup owns a 4 x 4 block of F(4x4,3x3)
// tiles (16 x 16 output pixels) of one image:
//   phase 1  relu(conv1_1 + bias) on the 18 x 18 pixel patch the block's tiles need, into LDS (83 KB; zeros outside the
//            image = conv1_2's SAME padding) — as 16 x 16 x 28 products on the matrix cores (see the phase's comment);
//   phase 2  in six groups of 6 transform planes (row xi of B^T d B), producer / consumer: a PRODUCER thread = (tile, channel
//            quad) builds the 6 planes of its tile from the patch with the expression trees of fw_bt6 / wino43_input_kernel
//            and parks them in LDS as the A operand (16 tiles x 64 channels per plane, XOR-swizzled 16-byte chunks, two
//            group buffers); one step later each CONSUMER wave contracts those planes for its 16 output channels on
//            v_mfma_f32_16x16x4_f32 — B (U^T rows, L2-resident: the filter bank is 590 KB) comes straight from global
//            memory into registers two planes ahead, A from LDS one plane ahead; MFMA
//            order and K mapping are wino43_mfma_kernel's (k = 16 g + 4 (lane >> 4) + i, g then i ascending, C = 0
//            first), so every accumulator holds the same bits. The producers' vector work runs under the consumers'
//            matrix work of the previous group (other waves of the same SIMDs), one barrier per group;
//   epilogue all 36 accumulators of a (tile, channel) element are in registers (16 tiles per workgroup: 144 VGPRs), so the
//            output transform runs once at the end with exactly the column-by-column sequence of the MFMA kernel's fold
//            (at6_col, then Y += t (x) A[nu, :] for nu = 0..5), + bias, ReLU, 2 x 2 max — through LDS, so that the pooled
//            8 x 8 x 64 block leaves as 2-KB contiguous rows.
// One workgroup per CU (LDS 138 KB), 8 waves. Bit-identical to conv3x3_c3_winograd43 + winograd43_conv(pool = 1)
// (tests/test_gpu_round4.py). H and W must be multiples of 16.
// tools/conv12_probe.hip defines CONV12_PROBE: wall-clock stamps of one workgroup's phases (not in the library build)
#ifdef CONV12_PROBE
__device__ unsigned long long g_conv12_probe[64];
#define F12_TS(I) do { if (blockIdx.x == CONV12_PROBE && threadIdx.x == 0) g_conv12_probe[I] = wall_clock64(); } while (0)
#define F12_TSW(I, WV) do { if (blockIdx.x == CONV12_PROBE && threadIdx.x == 64 * (WV)) g_conv12_probe[I] = wall_clock64(); } while (0)
#else
#define F12_TS(I)
#define F12_TSW(I, WV)
#endif
// probe-only ablations (tools/conv12_probe.hip): drop the B operand fetches / the A operand fetches / the producers' transforms
#ifndef F12_ABL_NO_B
#define F12_ABL_NO_B 0
#endif
#ifndef F12_ABL_NO_A
#define F12_ABL_NO_A 0
#endif
#ifndef F12_ABL_NO_T
#define F12_ABL_NO_T 0
#endif
constexpr int F12_P = 18;                       // patch rows / columns of a 4 x 4 tile block
constexpr int F12_NQ = 5;                       // pixel quads per patch row (20 columns computed, 18 kept)
constexpr int F12_INF = 24 * CF_CIN;            // window columns px0-1 .. px0+22 (22 used)

typedef float v4f12 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void f12_at6_col(const float* m, float* t)
{
  const float s = m[1] + m[2], d = m[1] - m[2], S = m[3] + m[4], D = m[3] - m[4];
  t[0] = (m[0] + s) + S;
  t[1] = __builtin_fmaf(2.f, D, d);
  t[2] = __builtin_fmaf(4.f, S, s);
  t[3] = __builtin_fmaf(8.f, D, d) + m[5];
}

// 512 threads: waves 0-3 are the CONSUMERS (each owns 16 output channels: the 36 accumulators of its elements, the matrix
// work, the output transform), waves 4-7 the PRODUCERS of the A operand (the transform planes); all eight share phase 1.
// Measured and dropped (round 4): workgroups of 4 consecutive blocks with the NEXT block's frame window requested by the
// producer waves when phase 2 starts (the window load is a bare 2 us round trip in front of every block). The window
// phase fell to 1.1 us, but the block loop cost the kernel its register slack (216 -> 256 VGPRs, scratch reloads that wait
// — vmcnt counts in order — for the very loads they were meant to hide): 3.39 ms against 3.13 alone, and the step the same
// within the run-to-run spread (785.0 vs 783.0 frames/s, A / B / A / B on one box).
template <bool RAW>
__global__ __launch_bounds__(512, 1) void conv12_wino43_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ ut2, const float* __restrict__ b2, float* __restrict__ ypool, int H, int W, int nbx,
    int nby, int imgs_per_group, int relu1, int relu2, RawFrames raw, int ut_frag)
{
  __shared__ float s_in[F12_P + 2][F12_INF];
  __shared__ __attribute__((aligned(16))) float s_y[F12_P][F12_P][64];
  __shared__ __attribute__((aligned(16))) float s_v[2][6 * 16 * 64];   // two groups of 6 planes x 16 tiles x 64 channels; epilogue: 8 x 8 x 64
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, b = blockIdx.x / (nbx * nby);
  const int py0 = 16 * by - 1, px0 = 16 * bx - 1;   // image coordinates of patch pixel (0, 0)
  const bool is_depth = RAW && (raw.color == nullptr || b >= raw.n_color);
  const int grp = RAW ? ((raw.color != nullptr && is_depth) ? 1 : 0) : b / imgs_per_group;
  w1 += (size_t)grp * 27 * 64;
  b1 += (size_t)grp * 64;
  ut2 += (size_t)grp * 36 * 64 * 64;
  b2 += (size_t)grp * 64;

  // Round 5: what a block needs from global memory besides its frame window — conv1_1's taps and bias (14 + 2 registers),
  // conv1_2's bias — is REQUESTED here, in front of the window, and the window's (up to) three elements per thread are
  // requested together. With one workgroup per CU nothing else covers a global round trip; in round 4 the window loop was
  // load - wait - store three times over, and the taps followed behind it: five serial L2 / HBM latencies in front of
  // every block. tools/conv12_probe: window phase 2.2 -> 1.0 us, block 22.1 -> 20.4 us, launch 3.22 -> 3.04 ms. (Also
  // requesting the first plane pair's B operands up here measured 3.08 ms: not kept.) No arithmetic changes.
  const int n16 = lane & 15, kk = lane >> 4, chalf = wave & 1;
  float wb[7][2], bia[2];
  int tapoff[7];
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int slot = 4 * j + kk, t = slot - 1;        // slot 0: the zero slot
    const int ky = t / 9, rem = t - 9 * ky;           // rem = 3 kx + ci
    tapoff[j] = slot == 0 ? 0 : ky * F12_INF + rem;
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) wb[j][c2] = w1[(size_t)(t < 0 ? 0 : t) * 64 + 32 * chalf + 16 * c2 + n16];   // (slot 0 is zeroed behind the window loads: no use of a loaded value up here)
  }
#pragma unroll
  for (int c2 = 0; c2 < 2; c2++) bia[c2] = b1[32 * chalf + 16 * c2 + n16];
  const float bv2 = b2[16 * (wave & 3) + (lane & 15)];   // conv1_2's bias of a consumer lane's channel (epilogue)
  F12_TS(0);
  // input window: rows py0-1 .. py0+18, columns px0-1 .. px0+22. A thread's (up to) three elements are requested
  // together and parked in LDS afterwards: the loop this replaces (round 4) loaded, waited and stored one element per
  // trip — three serial round trips to memory in front of every block (the 2-3 us "window" phase of the probe).
  {
    constexpr int NWIN = ((F12_P + 2) * F12_INF + 511) / 512;
    float wv_[NWIN];
#pragma unroll
    for (int e = 0; e < NWIN; e++) {
      const int i = tid + 512 * e;
      const int r = i / F12_INF, j = i - r * F12_INF;
      const int iy = py0 - 1 + r, ix = px0 - 1 + j / CF_CIN, ch = j % CF_CIN;
      const bool ok = i < (F12_P + 2) * F12_INF && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const size_t pix = ok ? ((size_t)b * H + iy) * W + ix : (size_t)b * H * W;   // (outside: any valid address, value dropped — no branch around the load)
      float val;
      if (!RAW) {
        val = x[pix * CF_CIN + (ok ? ch : 0)];
      } else if (!is_depth) {
        val = (float)((double)raw.color[pix * CF_CIN + (ok ? ch : 0)] - raw.mean[ch]);
      } else {
        const int bd = b - (raw.color ? raw.n_color : 0);
        const size_t pixd = ok ? ((size_t)bd * H + iy) * W + ix : (size_t)bd * H * W;
        const float t = fminf(fmaxf(div_rn((float)raw.depth[pixd], 2000.f), 0.f), 1.f) * 255.f;
        val = (float)((double)t - raw.mean[ch]);
      }
      if (!ok) val = 0.f;
      wv_[e] = val;
    }
#pragma unroll
    for (int e = 0; e < NWIN; e++) {
      const int i = tid + 512 * e;
      if (i < (F12_P + 2) * F12_INF) (&s_in[0][0])[i] = wv_[e];
    }
  }
  {
    // phase 1: relu(conv1_1 + bias) on the 18 x 18 patch, ON THE MATRIX CORES. On gfx950 an fp32 MFMA and the vector ALU of
    // a SIMD never run in the same cycle (tools/mfma_bare.hip: every VALU instruction between two MFMAs costs its 3.5
    // cycles of matrix-pipe time), and v_mfma_f32_16x16x4_f32 retires 32 FMAs per cycle where the packed-FMA formulation of
    // this phase (648 v_pk_fma_f32 per thread, conv3x3_c3_wino43_kernel's phase 1 on a taller patch) measured 16: the
    // K = 27 contraction is a poor GEMM and still twice as fast there. M = 16 consecutive patch pixels (the patch as a list
    // of 324, row-major: pixel q sits at s_y + 64 q), N = 16 channels, K = 28 = one zero slot + the 27 taps in their
    // (ky, kx, ci) order: an f32 MFMA is a k-ordered fmaf chain, so every output is fma(x26, w26, ... fma(x0, w0,
    // fma(0, 0, +0))) — the per-pixel chain of the vector version, bit for bit (the zero slot comes FIRST: +0 + 0 * 0
    // is the +0 that chain starts from). Work unit = (pixel group, channel half): 42 units over 8 waves; a wave's channel
    // half is fixed (unit = wave + 8 i), so it keeps 14 B-operand registers; the A operand is one ds_read_b32 per K step
    // (window address = pixel base + the lane's tap offset).
    if (kk == 0) wb[0][0] = wb[0][1] = 0.f;           // the zero slot (K slot 0 of the lanes with kk == 0)
    const bool interior = py0 >= 0 && py0 + F12_P <= H && px0 >= 0 && px0 + F12_P <= W;   // (uniform) every patch pixel inside the image
    const float* s_in_f = &s_in[0][0];
    float* s_y_f = &s_y[0][0][0];
    __syncthreads();
    F12_TS(1);
    // the A operand of a unit: 7 window values per lane (pixel 16 g + n16, K slots 4 j + kk)
#define F12_LOAD_A(DST, G)                                                                            \
    {                                                                                                 \
      const int q_ = 16 * (G) + n16, qc_ = q_ < F12_P * F12_P ? q_ : F12_P * F12_P - 1;   /* (the last group is ragged: 324 = 20 x 16 + 4) */ \
      const int r_ = qc_ / F12_P, c_ = qc_ - F12_P * r_;                                              \
      const int pixb_ = r_ * F12_INF + c_ * CF_CIN;                                                   \
      _Pragma("unroll") for (int j = 0; j < 7; j++) DST[j] = s_in_f[pixb_ + tapoff[j]];               \
      if (kk == 0) DST[0] = 0.f;   /* the zero slot (0 * 0, whatever the window holds) */               \
    }
    constexpr int NU = 2 * ((F12_P * F12_P + 15) / 16);
    float av[7], an[7];
    F12_LOAD_A(av, wave >> 1)
    for (int u = wave; u < NU; u += 8) {
      const int g = u >> 1;
      if (u + 8 < NU) F12_LOAD_A(an, (u + 8) >> 1)   // the next unit's operand flies under this unit's MFMAs
      v4f12 acc[2];
#pragma unroll
      for (int c2 = 0; c2 < 2; c2++) acc[c2] = (v4f12){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 7; j++)
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) acc[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wb[j][c2], acc[c2], 0, 0, 0);
      // lane (channel n16, kk) holds pixels 16 g + 4 kk + i, i = 0..3
      const int qo0 = 16 * g + 4 * kk;
      float* dst = s_y_f + qo0 * 64 + 32 * chalf + n16;
      if (interior && g < F12_P * F12_P / 16) {   // (uniform) nothing to mask
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int c2 = 0; c2 < 2; c2++) {
            float a = acc[c2][i] + bia[c2];
            if (relu1) a = a > 0.f ? a : 0.f;
            dst[i * 64 + 16 * c2] = a;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int qo = qo0 + i;
          const int ro = qo / F12_P, co = qo - F12_P * ro;
          const int yy = py0 + ro, xx = px0 + co;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
          for (int c2 = 0; c2 < 2; c2++) {
            float a = acc[c2][i] + bia[c2];
            if (relu1) a = a > 0.f ? a : 0.f;
            if (!ok) a = 0.f;   // outside the image: conv1_2's zero padding
            if (qo < F12_P * F12_P) dst[i * 64 + 16 * c2] = a;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 7; j++) av[j] = an[j];
    }
#undef F12_LOAD_A
  }
  __syncthreads();

  F12_TS(2);
  // ---- phase 2: six groups of six planes (row xi of B^T d B), software pipelined by one group -------------------------
  // step s: the producers build group s into s_v[s & 1] while the consumers contract group s - 1 out of s_v[(s - 1) & 1]
  const bool consumer = wave < 4;
  const int lr = lane & 15, lk = lane >> 4;
  const int ptid = tid & 255;
  const int tile = ptid >> 4, cq = ptid & 15;          // producer item: (tile, channel quad)
  const int tyy = tile >> 2, txx = tile & 3;
  v4f12 acc[36];
  const float* ub = ut2 + (size_t)(16 * (wave & 3) + lr) * 64 + 4 * lk;     // a consumer lane's U^T row and K chunk inside a plane
  // Consumer schedule: two planes at a time, their MFMAs interleaved — back-to-back MFMAs into ONE accumulator issue at half
  // rate (measured with tools/conv12_probe.hip: 67 cycles per dependent v_mfma_f32_16x16x4_f32, i.e. a plane at a time ran
  // the contraction at 47 %) — and the operands roll through one register set: behind the MFMAs of K group g the A operand
  // (LDS) and B operand (global / L2) of the NEXT pair's K group g are requested into the registers just read, so every
  // operand has a pair's worth of MFMAs (32 x 32 cycles) to arrive. sched_barrier pins that order: left to itself the
  // compiler sinks each read to just in front of its first use and the matrix pipe waits out every latency.
  f4 a0[4], a1[4], q0[4], q1[4];
  const unsigned uoff = (unsigned)((16 * (wave & 3) + lr) * 64 + 4 * lk);
  // ut_frag: the filter bank re-laid fragment-major — [plane][consumer wave][K group][lane][4] — so that one load
  // instruction of a wave covers 1 KB of contiguous memory (8 whole cache lines) instead of 16 half lines 256 B apart
  const unsigned foff = (unsigned)(((wave & 3) * 4) * 256 + lane * 4);
#define F12_LOADB(D0, D1, K, G)                                                                                  \
  if (ut_frag) {                                                                                                 \
    const float* pk_ = ut2 + (size_t)(K) * 4096 + 256 * (G);                                                     \
    D0 = *reinterpret_cast<const f4*>(pk_ + foff);                                                               \
    D1 = *reinterpret_cast<const f4*>(pk_ + 4096 + foff);                                                        \
  } else {                                                                                                       \
    const float* pk_ = ut2 + (size_t)(K) * 4096 + 16 * (G);                                                      \
    D0 = *reinterpret_cast<const f4*>(pk_ + uoff);                                                               \
    D1 = *reinterpret_cast<const f4*>(pk_ + 4096 + uoff);                                                        \
  }
#define F12_LOADA(D0, D1, BUF, J, G)                                                                             \
  { D0 = *reinterpret_cast<const f4*>(&s_v[BUF][((J) * 16 + lr) * 64 + (((4 * (G) + lk) ^ lr) * 4)]);            \
    D1 = *reinterpret_cast<const f4*>(&s_v[BUF][(((J) + 1) * 16 + lr) * 64 + (((4 * (G) + lk) ^ lr) * 4)]); }
  // (round 5: the arithmetic runs on the two 64-bit halves of every 128-bit patch value — v_pk_add / v_pk_mul with negate
  //  modifiers; written on 4-wide vectors the compiler split every subtraction into four scalar ones: 480 v_sub_f32 — same
  //  expression trees per element, as fw_bt6 / wino43_input_kernel)
#define F12_PRODUCE(XI, BUF)                                                                                    \
  {                                                                                                             \
    typedef float f2p __attribute__((ext_vector_type(2)));                                                      \
    f2p ta[2][6];                                                                                               \
    _Pragma("unroll") for (int hh = 0; hh < 2; hh++) {                                                          \
      f4 d[3][6];                                                                                               \
      _Pragma("unroll") for (int s3 = 0; s3 < 3; s3++)                                                          \
        _Pragma("unroll") for (int r = 0; r < 6; r++)                                                           \
          if (((XI) == 0 && (r == 0 || r == 2 || r == 4)) || ((XI) == 5 && (r == 1 || r == 3 || r == 5)) ||     \
              ((XI) >= 1 && (XI) <= 4 && r >= 1 && r <= 4))                                                     \
            d[s3][r] = *reinterpret_cast<const f4*>(&s_y[4 * tyy + r][4 * txx + 3 * hh + s3][cq * 4]);          \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      _Pragma("unroll") for (int s3 = 0; s3 < 3; s3++) {                                                        \
        const int s2 = 3 * hh + s3;                                                                             \
        _Pragma("unroll") for (int hf = 0; hf < 2; hf++) {                                                      \
          f2p e_[6];                                                                                            \
          _Pragma("unroll") for (int r = 0; r < 6; r++) e_[r] = hf ? d[s3][r].zw : d[s3][r].xy;                 \
          if ((XI) == 0) ta[hf][s2] = (4.f * e_[0] - 5.f * e_[2]) + e_[4];                                      \
          else if ((XI) == 5) ta[hf][s2] = (4.f * e_[1] - 5.f * e_[3]) + e_[5];                                 \
          else if ((XI) == 1 || (XI) == 2) {                                                                    \
            const f2p a = e_[4] - 4.f * e_[2], b_ = e_[3] - 4.f * e_[1];                                        \
            ta[hf][s2] = (XI) == 1 ? a + b_ : a - b_;                                                           \
          } else {                                                                                              \
            const f2p c = e_[4] - e_[2], e = 2.f * (e_[3] - e_[1]);                                             \
            ta[hf][s2] = (XI) == 3 ? c + e : c - e;                                                             \
          }                                                                                                     \
        }                                                                                                       \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
    f2p oa[2][6];                                                                                               \
    _Pragma("unroll") for (int hf = 0; hf < 2; hf++) {                                                          \
      const f2p* t_ = ta[hf];                                                                                   \
      oa[hf][0] = (4.f * t_[0] - 5.f * t_[2]) + t_[4];                                                          \
      { const f2p a = t_[4] - 4.f * t_[2], b_ = t_[3] - 4.f * t_[1]; oa[hf][1] = a + b_; oa[hf][2] = a - b_; }  \
      { const f2p c = t_[4] - t_[2], e = 2.f * (t_[3] - t_[1]); oa[hf][3] = c + e; oa[hf][4] = c - e; }         \
      oa[hf][5] = (4.f * t_[1] - 5.f * t_[3]) + t_[5];                                                          \
    }                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 6; j++)                                                               \
      *reinterpret_cast<f4*>(&s_v[BUF][(j * 16 + tile) * 64 + ((cq ^ tile) * 4)]) =                             \
          (f4){oa[0][j].x, oa[0][j].y, oa[1][j].x, oa[1][j].y};                                                 \
  }
  // planes 6 XI + J and 6 XI + J + 1 out of buffer BUF. On entry q0 / q1 hold their B operands; a0 / a1 their A operands
  // unless J == 0 (first pair behind the group's barrier: fetched here). NK = first plane of the next pair (its B operands
  // are fetched behind the MFMAs) or -1.
#define F12_PAIR(XI, J, BUF, NK)                                                                                \
  {                                                                                                             \
    if ((J) == 0) { _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++) F12_LOADA(a0[g_], a1[g_], BUF, 0, g_) }    \
    v4f12 ca_ = (v4f12){0.f, 0.f, 0.f, 0.f}, cb_ = ca_;                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++) {                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {                                                        \
        ca_ = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g_][i_], q0[g_][i_], ca_, 0, 0, 0);                       \
        cb_ = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g_][i_], q1[g_][i_], cb_, 0, 0, 0);                       \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if ((J) < 4 && !F12_ABL_NO_A) { F12_LOADA(a0[g_], a1[g_], BUF, (J) + 2, g_) }                             \
      if ((NK) >= 0 && (NK) < 36 && !F12_ABL_NO_B) { F12_LOADB(q0[g_], q1[g_], (NK), g_) }                      \
    }                                                                                                           \
    acc[6 * (XI) + (J)] = ca_;                                                                                  \
    acc[6 * (XI) + (J) + 1] = cb_;                                                                              \
  }
#define F12_CONSUME(XI, BUF)                                                                                    \
  F12_PAIR(XI, 0, BUF, 6 * (XI) + 2) F12_PAIR(XI, 2, BUF, 6 * (XI) + 4) F12_PAIR(XI, 4, BUF, 6 * (XI) + 6)
#define F12_STEP(S)                                                                                             \
  if (consumer) { if ((S) >= 1) { F12_CONSUME((S) - 1, ((S) - 1) & 1) } }                                       \
  else { if ((S) <= 5 && !F12_ABL_NO_T) { F12_PRODUCE((S) <= 5 ? (S) : 5, (S) & 1) } }                          \
  F12_TS(10 + 2 * (S)); F12_TSW(11 + 2 * (S), 4);                                                                \
  __syncthreads();
  if (consumer) { _Pragma("unroll") for (int g = 0; g < 4; g++) F12_LOADB(q0[g], q1[g], 0, g) }
  F12_STEP(0) F12_STEP(1) F12_STEP(2) F12_STEP(3) F12_STEP(4) F12_STEP(5) F12_STEP(6)
#undef F12_STEP
#undef F12_CONSUME
#undef F12_PAIR
#undef F12_PRODUCE
#undef F12_LOADA
#undef F12_LOADB

  F12_TS(3);
  // ---- epilogue: output transform (the MFMA kernel's fold, column by column), bias, ReLU, 2 x 2 max -----------------
  // a consumer lane holds tiles 4 lk + i (i = 0..3: tile row lk, tile column i) x channel 16 wave + lr
  float* s_o = &s_v[0][0];                                // [8 pooled rows][8 pooled columns][64]
  if (consumer) {
    // Two of a lane's four tiles per pass, on packed f32 (round 5): the accumulators of tiles i, i + 1 are adjacent registers, so
    // the column transforms and the 96 rank-1 FMAs of a pass are v_pk_add / v_pk_mul / v_pk_fma on register pairs as they lie —
    // element for element the scalar sequence of wino43_mfma_kernel's fold (one rounding per operation, explicit FMAs), half
    // the instructions (the output transform was 770 vector instructions per thread, paid in matrix time: 2.4 of a block's 20 us).
    typedef float f2e __attribute__((ext_vector_type(2)));
    const int co = 16 * wave + lr;
    const float bv = bv2;
#pragma unroll
    for (int ih = 0; ih < 2; ih++) {
      f2e yo[16];
#pragma unroll
      for (int o = 0; o < 16; o++) yo[o] = (f2e){0.f, 0.f};
#pragma unroll
      for (int nu = 0; nu < 6; nu++) {
        const float c0 = nu == 5 ? 0.f : 1.f;
        const float c1 = nu == 1 ? 1.f : nu == 2 ? -1.f : nu == 3 ? 2.f : nu == 4 ? -2.f : 0.f;
        const float c2 = (nu == 1 || nu == 2) ? 1.f : (nu == 3 || nu == 4) ? 4.f : 0.f;
        const float c3 = nu == 1 ? 1.f : nu == 2 ? -1.f : nu == 3 ? 8.f : nu == 4 ? -8.f : nu == 5 ? 1.f : 0.f;
        f2e m_[6], t_[4];
#pragma unroll
        for (int x_ = 0; x_ < 6; x_++) m_[x_] = (f2e){acc[6 * x_ + nu][2 * ih], acc[6 * x_ + nu][2 * ih + 1]};
        {   // f12_at6_col, two elements at a time
          const f2e s = m_[1] + m_[2], d = m_[1] - m_[2], S = m_[3] + m_[4], D = m_[3] - m_[4];
          t_[0] = (m_[0] + s) + S;
          t_[1] = __builtin_elementwise_fma((f2e){2.f, 2.f}, D, d);
          t_[2] = __builtin_elementwise_fma((f2e){4.f, 4.f}, S, s);
          t_[3] = __builtin_elementwise_fma((f2e){8.f, 8.f}, D, d) + m_[5];
        }
#pragma unroll
        for (int a_ = 0; a_ < 4; a_++) {
          yo[4 * a_ + 0] = __builtin_elementwise_fma(t_[a_], (f2e){c0, c0}, yo[4 * a_ + 0]);
          yo[4 * a_ + 1] = __builtin_elementwise_fma(t_[a_], (f2e){c1, c1}, yo[4 * a_ + 1]);
          yo[4 * a_ + 2] = __builtin_elementwise_fma(t_[a_], (f2e){c2, c2}, yo[4 * a_ + 2]);
          yo[4 * a_ + 3] = __builtin_elementwise_fma(t_[a_], (f2e){c3, c3}, yo[4 * a_ + 3]);
        }
      }
#pragma unroll
      for (int o = 0; o < 16; o++) {
        f2e val = yo[o] + (f2e){bv, bv};
        if (relu2) { val.x = val.x > 0.f ? val.x : 0.f; val.y = val.y > 0.f ? val.y : 0.f; }
        yo[o] = val;
      }
#pragma unroll
      for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
        for (int e2 = 0; e2 < 2; e2++) {
          f2e p = yo[4 * (2 * a2) + 2 * e2];
          const f2e p1 = yo[4 * (2 * a2) + 2 * e2 + 1], p2 = yo[4 * (2 * a2 + 1) + 2 * e2], p3 = yo[4 * (2 * a2 + 1) + 2 * e2 + 1];
          p.x = p1.x > p.x ? p1.x : p.x; p.y = p1.y > p.y ? p1.y : p.y;
          p.x = p2.x > p.x ? p2.x : p.x; p.y = p2.y > p.y ? p2.y : p.y;
          p.x = p3.x > p.x ? p3.x : p.x; p.y = p3.y > p.y ? p3.y : p.y;
          s_o[((2 * lk + a2) * 8 + (2 * (2 * ih) + e2)) * 64 + co] = p.x;          // tile (row lk, column i) -> pooled (2 lk + a2, 2 i + e2)
          s_o[((2 * lk + a2) * 8 + (2 * (2 * ih + 1) + e2)) * 64 + co] = p.y;
        }
    }
  }
  F12_TS(4);
  __syncthreads();
  const int Hp = H >> 1, Wp = W >> 1;
  for (int i = tid; i < 8 * 8 * 16; i += 512) {
    const int c4 = (i & 15) * 4, pxl = (i >> 4) & 7, pyl = i >> 7;
    *reinterpret_cast<f4*>(ypool + (((size_t)b * Hp + 8 * by + pyl) * Wp + 8 * bx + pxl) * 64 + c4) =
        *reinterpret_cast<const f4*>(&s_o[(pyl * 8 + pxl) * 64 + c4]);
  }
  F12_TS(5);
}

}  // namespace

extern "C" int pcnn_conv3x3_c3_winograd43_fwd(const float* x, const float* weights, const float* bias,
                                              int B, int H, int W, int Cout, int groups, int relu, float* v,
                                              void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1, PCNN_EINVAL, "conv3x3_c3_winograd43: bad shape %dx%dx%d", B, H, W);
  PCNN_REQUIRE(groups >= 1 && B % groups == 0, PCNN_EINVAL, "conv3x3_c3_winograd43: batch %d is not a multiple of groups %d", B, groups);
  PCNN_REQUIRE(Cout >= 64 && Cout % 64 == 0, PCNN_EINVAL,
               "conv3x3_c3_winograd43: output channels must be a multiple of 64 (got %d)", Cout);
  PCNN_REQUIRE(x && weights && bias && v, PCNN_ENULL, "conv3x3_c3_winograd43: NULL pointer");
  PCNN_REQUIRE(aligned16(v) && aligned16(weights) && aligned16(bias), PCNN_EINVAL,
               "conv3x3_c3_winograd43: weights, bias and output must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const int nseg = (Wt + FW_TILES - 1) / FW_TILES;
  const long long blocks = (long long)B * Ht * nseg;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "conv3x3_c3_winograd43: grid too large");
  const long long plane = (long long)B * Ht * Wt * Cout;
  const RawFrames none = {nullptr, nullptr, 0, {0.0, 0.0, 0.0}};
  PCNN_LAUNCH((conv3x3_c3_wino43_kernel<false>), dim3((unsigned)blocks, Cout / 64), dim3(256), 0, stream, x,
              weights, bias, v, H, W, Cout, relu, Ht, Wt, nseg, plane, B / groups, none);
  return check_launch("conv3x3_c3_winograd43_fwd");
}

extern "C" int pcnn_conv3x3_c3_winograd43_raw_fwd(const uint8_t* color_bgr, int num_color, const uint16_t* depth, int num_depth,
                                                  const double* pixel_means, const float* weights, const float* bias, int H,
                                                  int W, int Cout, int relu, float* v, void* stream_)
{
  PCNN_REQUIRE(num_color >= 0 && num_depth >= 0 && num_color + num_depth >= 1 && H >= 1 && W >= 1, PCNN_EINVAL,
               "conv3x3_c3_winograd43_raw: bad shape (%d colour + %d depth frames of %dx%d)", num_color, num_depth, H, W);
  PCNN_REQUIRE(Cout >= 64 && Cout % 64 == 0, PCNN_EINVAL,
               "conv3x3_c3_winograd43_raw: output channels must be a multiple of 64 (got %d)", Cout);
  PCNN_REQUIRE((num_color == 0 || color_bgr) && (num_depth == 0 || depth) && pixel_means && weights && bias && v, PCNN_ENULL,
               "conv3x3_c3_winograd43_raw: NULL pointer");
  PCNN_REQUIRE(aligned16(v) && aligned16(weights) && aligned16(bias), PCNN_EINVAL,
               "conv3x3_c3_winograd43_raw: weights, bias and output must be 16-byte aligned");
  PCNN_REQUIRE(pixel_means[0] - pixel_means[0] == 0.0 && pixel_means[1] - pixel_means[1] == 0.0 && pixel_means[2] - pixel_means[2] == 0.0,
               PCNN_EINVAL, "conv3x3_c3_winograd43_raw: pixel_means must be finite");
  PCNN_REQUIRE((reinterpret_cast<uintptr_t>(depth) & 1u) == 0, PCNN_EINVAL, "conv3x3_c3_winograd43_raw: depth must be 2-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_color + num_depth;
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const int nseg = (Wt + FW_TILES - 1) / FW_TILES;
  const long long blocks = (long long)B * Ht * nseg;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "conv3x3_c3_winograd43_raw: grid too large");
  const long long plane = (long long)B * Ht * Wt * Cout;
  const RawFrames raw = {num_color ? color_bgr : nullptr, num_depth ? depth : nullptr, num_color,
                         {pixel_means[0], pixel_means[1], pixel_means[2]}};       // (host pointer: read here, passed by value)
  PCNN_LAUNCH((conv3x3_c3_wino43_kernel<true>), dim3((unsigned)blocks, Cout / 64), dim3(256), 0, stream, (const float*)nullptr,
              weights, bias, v, H, W, Cout, relu, Ht, Wt, nseg, plane, 1, raw);
  return check_launch("conv3x3_c3_winograd43_raw_fwd");
}

extern "C" int pcnn_conv3x3_c3_fwd(const float* x, const float* weights, const float* bias, int B,
                                   int H, int W, int Cout, int relu, float* y, void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1, PCNN_EINVAL, "conv3x3_c3: bad shape %dx%dx%d", B, H, W);
  PCNN_REQUIRE(Cout >= 64 && Cout % 64 == 0, PCNN_EINVAL,
               "conv3x3_c3: output channels must be a multiple of 64 (got %d)", Cout);
  PCNN_REQUIRE(x && weights && bias && y, PCNN_ENULL, "conv3x3_c3: NULL pointer");
  PCNN_REQUIRE(aligned16(y) && aligned16(weights) && aligned16(bias), PCNN_EINVAL,
               "conv3x3_c3: weights, bias and output must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int nseg = (W + CF_SEG - 1) / CF_SEG;
  const int nstrip = (H + CF_ROWS - 1) / CF_ROWS;
  const long long blocks = (long long)B * nstrip * nseg;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "conv3x3_c3: grid too large");
  PCNN_LAUNCH(conv3x3_c3_bias_relu_kernel, dim3((unsigned)blocks, Cout / 64), dim3(256), 0, stream,
              x, weights, bias, y, H, W, Cout, relu, nseg, nstrip);
  return check_launch("conv3x3_c3_fwd");
}

static int conv12_check(int B, int H, int W, const void* w1, const void* b1, const void* ut2, const void* b2, const void* y)
{
  PCNN_REQUIRE(B >= 1 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0, PCNN_EINVAL,
               "conv1_1_conv1_2_fused: the frame must be a multiple of 16 in both directions (got %dx%d)", W, H);
  PCNN_REQUIRE(w1 && b1 && ut2 && b2 && y, PCNN_ENULL, "conv1_1_conv1_2_fused: NULL pointer");
  PCNN_REQUIRE(aligned16(w1) && aligned16(b1) && aligned16(ut2) && aligned16(b2) && aligned16(y), PCNN_EINVAL,
               "conv1_1_conv1_2_fused: weights, biases and output must be 16-byte aligned");
  PCNN_REQUIRE((long long)B * (H / 16) * (W / 16) < (1ll << 31), PCNN_EINVAL, "conv1_1_conv1_2_fused: grid too large");
  return PCNN_OK;
}

// (Round 4 also built a persistent half-channel pipeline of this layer pair — 4.52 vs 3.15 ms, bound by the same filter-bank
// re-reads, its static walk losing the dynamic balance of 38 400 short workgroups; it left the library in round 5:
// tools/variants/conv12_wino43_pipelined_kernel.inc)
// (Round 5 built the two-workgroups-per-CU form of this kernel — four waves, two channel halves of 32, 78 KB of LDS: bit-
// identical, 3 % faster alone, 2 % slower in the three-stream step and at one frame; tools/variants/conv12_wino43_pair_kernel.inc.)
extern "C" int pcnn_conv1_1_conv1_2_fused_fwd(const float* x, const float* w1, const float* b1, const float* ut2, int ut2_layout,
                                              const float* b2, int B, int H, int W, int groups, int relu1, int relu2,
                                              float* y_pool, void* stream_)
{
  PCNN_REQUIRE(ut2_layout == 0 || ut2_layout == 1, PCNN_EINVAL,
               "conv1_1_conv1_2_fused: ut2_layout must be 0 (U^T [36][64][64]) or 1 (fragment-major)");
  int st = conv12_check(B, H, W, w1, b1, ut2, b2, y_pool);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(x, PCNN_ENULL, "conv1_1_conv1_2_fused: NULL input");
  PCNN_REQUIRE(groups >= 1 && B % groups == 0, PCNN_EINVAL, "conv1_1_conv1_2_fused: batch %d is not a multiple of groups %d", B, groups);
  hipStream_t stream = (hipStream_t)stream_;
  const RawFrames none = {nullptr, nullptr, 0, {0.0, 0.0, 0.0}};
  PCNN_REQUIRE(groups <= 2, PCNN_EINVAL, "conv1_1_conv1_2_fused: at most two filter sets (got %d)", groups);
  const long long blocks = (long long)B * (H / 16) * (W / 16);
  PCNN_LAUNCH((conv12_wino43_fused_kernel<false>), dim3((unsigned)blocks), dim3(512), 0, stream, x, w1, b1, ut2,
              b2, y_pool, H, W, W / 16, H / 16, B / groups, relu1, relu2, none, ut2_layout);
  return check_launch("conv1_1_conv1_2_fused_fwd");
}

extern "C" int pcnn_conv1_1_conv1_2_fused_raw_fwd(const uint8_t* color_bgr, int num_color, const uint16_t* depth, int num_depth,
                                                  const double* pixel_means, const float* w1, const float* b1, const float* ut2,
                                                  int ut2_layout, const float* b2, int H, int W, int relu1, int relu2, float* y_pool,
                                                  void* stream_)
{
  PCNN_REQUIRE(ut2_layout == 0 || ut2_layout == 1, PCNN_EINVAL,
               "conv1_1_conv1_2_fused_raw: ut2_layout must be 0 (U^T [36][64][64]) or 1 (fragment-major)");
  PCNN_REQUIRE(num_color >= 0 && num_depth >= 0 && num_color + num_depth >= 1, PCNN_EINVAL,
               "conv1_1_conv1_2_fused_raw: bad frame counts (%d colour, %d depth)", num_color, num_depth);
  int st = conv12_check(num_color + num_depth, H, W, w1, b1, ut2, b2, y_pool);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE((num_color == 0 || color_bgr) && (num_depth == 0 || depth) && pixel_means, PCNN_ENULL, "conv1_1_conv1_2_fused_raw: NULL pointer");
  PCNN_REQUIRE(pixel_means[0] - pixel_means[0] == 0.0 && pixel_means[1] - pixel_means[1] == 0.0 && pixel_means[2] - pixel_means[2] == 0.0,
               PCNN_EINVAL, "conv1_1_conv1_2_fused_raw: pixel_means must be finite");
  PCNN_REQUIRE((reinterpret_cast<uintptr_t>(depth) & 1u) == 0, PCNN_EINVAL, "conv1_1_conv1_2_fused_raw: depth must be 2-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_color + num_depth;
  const RawFrames raw = {num_color ? color_bgr : nullptr, num_depth ? depth : nullptr, num_color,
                         {pixel_means[0], pixel_means[1], pixel_means[2]}};
  const long long blocks = (long long)B * (H / 16) * (W / 16);
  PCNN_LAUNCH((conv12_wino43_fused_kernel<true>), dim3((unsigned)blocks), dim3(512), 0, stream, (const float*)nullptr,
              w1, b1, ut2, b2, y_pool, H, W, W / 16, H / 16, 1, relu1, relu2, raw, ut2_layout);
  return check_launch("conv1_1_conv1_2_fused_raw_fwd");
}
