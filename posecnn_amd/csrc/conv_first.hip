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

}  // namespace

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
