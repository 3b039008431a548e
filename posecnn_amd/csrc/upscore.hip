// upscore.hip — gfx950 kernels for PoseCNN's fixed bilinear "deconv" layers and the label-head
// epilogue (lib/networks/network.py:141-157 make_deconv_filter, :207-222 deconv;
// lib/networks/vgg16_convs.py:128-146,152-163).
//
// The reference runs tf.nn.conv2d_transpose with a dense [k,k,C,C] filter whose only non-zero
// taps are weights[:, :, i, i] = outer(bilinear, bilinear): 25 GMAC per frame of multiplications
// by exact zeros. (Handing the same layer to MIOpen as a depthwise transposed convolution costs
// 45 ms per call on MI355X — 83 % of the whole pipeline in the first profile, profiles/r01.)
// The layer is a per-channel 2-tap-per-axis interpolation, i.e. a pure HBM stream:
//
//   pcnn_deconv_bilinear_fwd        out = deconv(in) [+ add1] [+ add2] [+ bias] [ReLU]
//   pcnn_upscore_softmax_argmax_fwd out-of-core label head: score = [ReLU](deconv(z) + bias),
//                                   prob = softmax(score), label = first argmax(prob); the
//                                   full-resolution score never touches HBM unless asked for.
//
// Canonical arithmetic (there is no bit-truth for cuDNN's conv2d_transpose): per output element
// acc = 0; for input rows ascending, input cols ascending: acc += (fy*fx) * in — products of the
// bilinear taps are exact in f32 (multiples of 1/f^2), one rounding per multiply and per add,
// then + add1, + add2, + bias in that order (the CPU checker restates it).
#include "bilinear.h"

namespace {

using namespace pcnn;

template <int V>
struct Vec;
template <>
struct Vec<4> { typedef float4 T; };
template <>
struct Vec<2> { typedef float2 T; };
template <>
struct Vec<1> { typedef float T; };

template <int V>
__device__ __forceinline__ void load_v(const float* p, float* r)
{
  typename Vec<V>::T v = *reinterpret_cast<const typename Vec<V>::T*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < V; i++) r[i] = f[i];
}
template <int V>
__device__ __forceinline__ void store_v(float* p, const float* r)
{
  typename Vec<V>::T v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < V; i++) f[i] = r[i];
  *reinterpret_cast<typename Vec<V>::T*>(p) = v;
}

// A workgroup owns DC_SEG consecutive output pixels of one output row (all channels): the row taps
// are wave-uniform, the column taps sit in a small LDS table, and the npx*C output floats leave as
// one contiguous run of V-wide stores. Inputs are 1/s^2 of the output and stay L2-resident.
constexpr int DC_SEG = 64;

template <int V>
__global__ __launch_bounds__(256) void deconv_bilinear_kernel(
    const float* __restrict__ in, const float* __restrict__ add1, const float* __restrict__ add2,
    const float* __restrict__ bias, float* __restrict__ out, int H, int W, int C, int k, int s,
    int relu, int nseg)
{
  __shared__ int s_i0[DC_SEG], s_n[DC_SEG];
  __shared__ float s_w[DC_SEG][4];
  const int pad = (k - s) / 2;
  const int cv = C / V;
  const int Ho = H * s, Wo = W * s;
  const int tid = threadIdx.x;
  const int seg = blockIdx.x % nseg;
  const int oy = (blockIdx.x / nseg) % Ho;
  const int b = blockIdx.x / (nseg * Ho);
  const int ox0 = seg * DC_SEG;
  const int npx = min(DC_SEG, Wo - ox0);
  if (tid < npx) {
    const Taps t = make_taps(ox0 + tid, k, s, pad, W);
    s_i0[tid] = t.i0;
    s_n[tid] = t.n;
#pragma unroll
    for (int j = 0; j < 4; j++) s_w[tid][j] = t.w[j];
  }
  const Taps ty = make_taps(oy, k, s, pad, H);
  __syncthreads();
  const float* inb = in + (size_t)b * H * W * C;
  const size_t obase = (((size_t)b * Ho + oy) * Wo + ox0) * C;
  const int total = npx * cv;
  for (int idx = tid; idx < total; idx += 256) {
    const int px = idx / cv;
    const int c = (idx - px * cv) * V;
    const int i0 = s_i0[px], nx = s_n[px];
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; i++) acc[i] = 0.f;
    const size_t o = obase + (size_t)px * C + c;
    if (ty.n >= 1 && ty.n <= 2 && nx >= 1 && nx <= 2) {
      // (round 5) k <= 2s: at most 2 x 2 taps. Their loads and the two addends' are issued together — absent taps read a
      // present one's address and are skipped in the sum (same terms, same order, same bits). One guarded load per tap was one
      // trip to memory per tap: 19 waits for 19 loads in this kernel's ISA, 0.34 of the HBM rate for a pure stream.
      float tv[4][V], a1v[V], a2v[V];
#pragma unroll
      for (int jy = 0; jy < 2; jy++)
#pragma unroll
        for (int jx = 0; jx < 2; jx++) {
          const int yy = ty.i0 + (jy < ty.n ? jy : ty.n - 1), xx = i0 + (jx < nx ? jx : nx - 1);
          load_v<V>(inb + ((size_t)yy * W + xx) * C + c, tv[jy * 2 + jx]);
        }
      if (add1) load_v<V>(add1 + o, a1v);
      if (add2) load_v<V>(add2 + o, a2v);
#pragma unroll
      for (int jy = 0; jy < 2; jy++)
#pragma unroll
        for (int jx = 0; jx < 2; jx++)
          if (jy < ty.n && jx < nx) {
            const float w = ty.w[jy] * s_w[px][jx];
#pragma unroll
            for (int i = 0; i < V; i++) acc[i] = acc[i] + w * tv[jy * 2 + jx][i];
          }
      if (add1) {
#pragma unroll
        for (int i = 0; i < V; i++) acc[i] = acc[i] + a1v[i];
      }
      if (add2) {
#pragma unroll
        for (int i = 0; i < V; i++) acc[i] = acc[i] + a2v[i];
      }
    } else {
#pragma unroll
      for (int jy = 0; jy < 4; jy++) {
        if (jy < ty.n) {
          const float* row = inb + (size_t)(ty.i0 + jy) * W * C + c;
#pragma unroll
          for (int jx = 0; jx < 4; jx++) {
            if (jx < nx) {
              const float w = ty.w[jy] * s_w[px][jx];
              float v[V];
              load_v<V>(row + (size_t)(i0 + jx) * C, v);
#pragma unroll
              for (int i = 0; i < V; i++) acc[i] = acc[i] + w * v[i];
            }
          }
        }
      }
      if (add1) {
        float v[V];
        load_v<V>(add1 + o, v);
#pragma unroll
        for (int i = 0; i < V; i++) acc[i] = acc[i] + v[i];
      }
      if (add2) {
        float v[V];
        load_v<V>(add2 + o, v);
#pragma unroll
        for (int i = 0; i < V; i++) acc[i] = acc[i] + v[i];
      }
    }
    if (bias) {
#pragma unroll
      for (int i = 0; i < V; i++) acc[i] = acc[i] + bias[c + i];
    }
    if (relu) {
#pragma unroll
      for (int i = 0; i < V; i++) acc[i] = acc[i] > 0.f ? acc[i] : 0.f;  // tf.nn.relu: max(x, 0)
    }
    store_v<V>(out + o, acc);
  }
}

// Gradient of the fixed bilinear deconv w.r.t. its input (what TF's conv2d_transpose gradient
// returns for the make_deconv_filter weights): the transposed interpolation, as a gather so that
// the sum order is fixed — output rows ascending, output columns ascending, acc = acc + (wy*wx)*g.
// One thread owns V channels of one input cell and reads its k x k output footprint (each output
// element is shared by <= ceil(k/s)^2 neighbouring cells, served by L2).
template <int V>
__global__ __launch_bounds__(256) void deconv_bilinear_bwd_kernel(const float* __restrict__ g,
                                                                  float* __restrict__ gin, int H,
                                                                  int W, int C, int k, int s,
                                                                  long long total)
{
  const int pad = (k - s) / 2;
  const int cv = C / V;
  const int Ho = H * s, Wo = W * s;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * V;
    long long p = idx / cv;
    const int j = (int)(p % W);
    p /= W;
    const int i = (int)(p % H);
    const int b = (int)(p / H);
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; q++) acc[q] = 0.f;
    for (int ty = 0; ty < k; ty++) {
      const int oy = s * i + ty - pad;
      if (oy < 0 || oy >= Ho) continue;
      const float wy = bilinear_tap(ty, k);
      const float* grow = g + (((size_t)b * Ho + oy) * Wo) * C + c;
      for (int tx = 0; tx < k; tx++) {
        const int ox = s * j + tx - pad;
        if (ox < 0 || ox >= Wo) continue;
        const float w = wy * bilinear_tap(tx, k);
        float v[V];
        load_v<V>(grow + (size_t)ox * C, v);
#pragma unroll
        for (int q = 0; q < V; q++) acc[q] = acc[q] + w * v[q];
      }
    }
    store_v<V>(gin + idx * V, acc);
  }
}

// y = [ReLU](x + bias[c]) over NHWC rows, in place or out of place: the bias_add + relu pair of
// Network.conv (network.py:181-187) as one pass instead of two framework kernels.
template <int V>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y, long long total, int C,
                                                       int relu)
{
  const int cv = C / V;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * V;
    float v[V];
    load_v<V>(x + idx * V, v);
#pragma unroll
    for (int i = 0; i < V; i++) {
      float t = v[i] + bias[c + i];
      v[i] = relu ? (t > 0.f ? t : 0.f) : t;
    }
    store_v<V>(y + idx * V, v);
  }
}

// pool = max_pool_2x2(ReLU(x + bias)) straight from the raw convolution output: the conv -> pool
// pairs of the VGG trunk (vgg16_convs.py:36-49) without writing and re-reading the activated
// full-resolution tensor. max and (+bias, ReLU) commute exactly — x1 >= x2 implies
// fl(x1 + b) >= fl(x2 + b) — so the four raw values are reduced first and one add + ReLU follows;
// the bits equal max_pool(bias_act(x)) for finite inputs. Window order (y, x), (y, x+1), (y+1, x),
// (y+1, x+1).
template <int V>
__global__ __launch_bounds__(256) void bias_relu_pool2_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ y, long long total,
                                                              int Ho, int Wo, int C, int relu)
{
  const int cv = C / V;
  const long long rowstride = (long long)2 * Wo * C;  // one input row
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % cv) * V;
    const long long p = idx / cv;           // pooled pixel (b, oy, ox)
    const int ox = (int)(p % Wo);
    const long long bo = p / Wo;            // b * Ho + oy
    const float* s0 = x + (bo * 2) * rowstride + (long long)(2 * ox) * C + c;
    float a[V], t[V];
    load_v<V>(s0, a);
    load_v<V>(s0 + C, t);
#pragma unroll
    for (int i = 0; i < V; i++) a[i] = t[i] > a[i] ? t[i] : a[i];
    load_v<V>(s0 + rowstride, t);
#pragma unroll
    for (int i = 0; i < V; i++) a[i] = t[i] > a[i] ? t[i] : a[i];
    load_v<V>(s0 + rowstride + C, t);
#pragma unroll
    for (int i = 0; i < V; i++) {
      float m = t[i] > a[i] ? t[i] : a[i];
      m = m + bias[c + i];
      a[i] = relu ? (m > 0.f ? m : 0.f) : m;
    }
    store_v<V>(y + idx * V, a);
  }
}

// one segment's npx * C floats from LDS to a contiguous run of global memory
__device__ __forceinline__ void up_copy_out(float* __restrict__ o, const float* s_out, int nfl, int tid, int nthreads, int seg_floats)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  if (((nfl | seg_floats) & 3) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
    // the segment's run of npx * C floats starts 16-byte aligned: dwordx4 stores (4x fewer store instructions)
    // Non-temporal: 432 MB per 16 frames of probabilities (and as much again of label weights) that nothing in the
    // step reads back — kept out of the L2 / MALL lines the feature maps live in. Measured in the step: this kernel
    // 224 -> 212 us, everything else unchanged (two runs each way).
    for (int i = tid * 4; i < nfl; i += nthreads * 4)
      __builtin_nontemporal_store(*reinterpret_cast<const v4f*>(s_out + i), reinterpret_cast<v4f*>(o + i));
  } else {
    for (int i = tid; i < nfl; i += nthreads) o[i] = s_out[i];
  }
}

// The hard-example label weights of the segment (TF1 op "Hardlabel", hard_label_op_gpu.cu.cc:17-29, fed prob_normalized
// and gt_label_2d by vgg16_convs.py:148-149) while its probabilities are still in LDS: out[pixel][c] = 1 for the pixel's
// ground-truth class g if g > 0 or prob[g] < threshold, 0 everywhere else. As its own launch (pcnn_hard_label_fwd) the op
// is 432 MB of stores per 16 frames behind a kernel that is bound by its arithmetic: 93 us that fit under the label
// head's own 169. Must be called by every thread of the workgroup; s_out holds the probabilities [npx][C] on entry.
__device__ __forceinline__ void up_hard_label_rows(float* s_out, const int* __restrict__ gt, float* __restrict__ hard,
                                                   long long pix0, int tid, int npx, int C, float threshold, int nthreads)
{
  int hot = -1;
  if (tid < npx) {
    const int g = gt[pix0 + tid];
    // labels outside [-1, C) index out of bounds in the reference — ignored here, as in hard_label.hip
    if (g >= 0 && g < C && (g > 0 || s_out[tid * C + g] < threshold)) hot = g;
  }
  __syncthreads();   // the probabilities have been read (and copied out)
  if (tid < npx) {
    // the row is zeros with at most one 1: wide zero stores, then the one
    float* row = s_out + tid * C;
    if ((C & 1) == 0) {
      typedef float f2z __attribute__((ext_vector_type(2)));
      for (int c = 0; c < C; c += 2) *reinterpret_cast<f2z*>(row + c) = (f2z){0.f, 0.f};   // (tid * C is even)
    } else {
      for (int c = 0; c < C; c++) row[c] = 0.f;
    }
    if (hot >= 0) row[hot] = 1.f;
  }
  __syncthreads();
  up_copy_out(hard + pix0 * C, s_out, npx * C, tid, nthreads, nthreads * C);
}

// Label head epilogue. A workgroup owns SEG consecutive output pixels of one output row; the
// (<= 2) x (SEG/s + 2) low-resolution cells it needs sit in LDS; every thread owns one pixel:
// C interpolated scores in registers -> softmax -> argmax; prob / score rows are parked in LDS and
// leave as one contiguous run of dword stores.
constexpr int UP_SEG = 128;

template <int CMAX>
__global__ __launch_bounds__(UP_SEG) void upscore_softmax_argmax_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ score_out,
    float* __restrict__ prob, int* __restrict__ label, int H, int W, int C, int k, int s, int relu,
    int nseg, int s_out_off, const int* __restrict__ gt, float* __restrict__ hard, float threshold)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = (k - s) / 2;
  const int Ho = H * s, Wo = W * s;
  const int seg = blockIdx.x % nseg;
  const int oy = (blockIdx.x / nseg) % Ho;
  const int b = blockIdx.x / (nseg * Ho);
  const int ox0 = seg * UP_SEG;
  const int npx = min(UP_SEG, Wo - ox0);
  const int tid = threadIdx.x;

  const Taps ty = make_taps(oy, k, s, pad, H);
  // input columns touched by this segment
  const Taps tfirst = make_taps(ox0, k, s, pad, W);
  const Taps tlast = make_taps(ox0 + npx - 1, k, s, pad, W);
  const int cx0 = tfirst.i0;
  const int ncx = tlast.i0 + (tlast.n > 0 ? tlast.n : 1) - cx0;
  float* s_z = smem;                       // [ty.n][ncx][C]
  float* s_out = smem + s_out_off;         // [UP_SEG][C]
  // a row's ncx cells x C channels are one contiguous run of z: straight copies, no index arithmetic per element
  const int rowlen = ncx * C;
  {
    // four loads per thread issued before the first LDS write (see the compile-time-C kernel below)
    const int tot = ty.n * rowlen;
    const float* src0 = z + (((long long)b * H + ty.i0) * W + cx0) * C;
    for (int e0 = tid; e0 < tot; e0 += 4 * UP_SEG) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int e = min(e0 + j * UP_SEG, tot - 1);
        const int ry = e / rowlen, i = e - ry * rowlen;
        v[j] = src0[(long long)ry * W * C + i];
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (e0 + j * UP_SEG < tot) s_z[e0 + j * UP_SEG] = v[j];
    }
  }
  __syncthreads();

  const int ox = ox0 + tid;
  float e[CMAX];
  int best = 0;
  if (tid < npx) {
    const Taps tx = make_taps(ox, k, s, pad, W);
#pragma unroll
    for (int c = 0; c < CMAX; c++) e[c] = 0.f;
#pragma unroll
    for (int jy = 0; jy < 4; jy++)
      if (jy < ty.n) {
#pragma unroll
        for (int jx = 0; jx < 4; jx++)
          if (jx < tx.n) {
            const float w = ty.w[jy] * tx.w[jx];
            const float* zc = s_z + (jy * ncx + (tx.i0 + jx - cx0)) * C;
#pragma unroll
            for (int c = 0; c < CMAX; c++)
              if (c < C) e[c] = e[c] + w * zc[c];
          }
      }
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) {
        float v = e[c] + bias[c];
        if (relu) v = v > 0.f ? v : 0.f;
        e[c] = v;
      }
  }
  if (score_out) {
    if (tid < npx)
#pragma unroll
      for (int c = 0; c < CMAX; c++)
        if (c < C) s_out[tid * C + c] = e[c];
    __syncthreads();
    float* o = score_out + (((long long)b * Ho + oy) * Wo + ox0) * C;
    for (int i = tid; i < npx * C; i += UP_SEG) o[i] = s_out[i];
    __syncthreads();
  }
  if (tid < npx) {
    // softmax_high_dimension (network.py:474-488) + argmax_2d (:432-434)
    float m = e[0];
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) m = fmaxf(m, e[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) { e[c] = exp_softmax_f32(e[c] - m); sum += e[c]; }
    float bestp = div_rn(e[0], sum);
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) {
        const float p = div_rn(e[c], sum);
        e[c] = p;
        if (p > bestp) { bestp = p; best = c; }
      }
    label[((long long)b * Ho + oy) * Wo + ox] = best;
  }
  if (prob || hard) {
    if (tid < npx)
#pragma unroll
      for (int c = 0; c < CMAX; c++)
        if (c < C) s_out[tid * C + c] = e[c];
    __syncthreads();
    const long long pix0 = ((long long)b * Ho + oy) * Wo + ox0;
    if (prob) up_copy_out(prob + pix0 * C, s_out, npx * C, tid, UP_SEG, UP_SEG * C);
    if (hard) up_hard_label_rows(s_out, gt, hard, pix0, tid, npx, C, threshold, UP_SEG);
  }
}

// ---- the same epilogue for a class count known at compile time (even: 22 YCB-Video, 14 / 16 LINEMOD) --------------
// The generic kernel above guards every channel operation with `c < C` and indexes its score array in a runtime loop
// (the 22 divisions ran out of scratch memory): 2 223 VALU instructions per pixel, VALU-bound at 0.22 of the HBM rate.
// With C a template parameter everything unrolls, the scores stay in registers as float2 pairs, LDS is read 8 bytes at
// a time and the interpolation, the bias add and the exponential's polynomial are packed-f32 instructions
// (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction, no contraction). Same expression trees, same
// order, same bits as the generic kernel and the CPU checker. Measured (16 frames, tools/bench_ops.py --ops upscore;
// counters tools/pmc_label_head.sh): 2 223 -> 1 188 VALU instructions per pixel-wave changed NOTHING (258 us) — nor did
// replacing the per-thread double-precision tap evaluation by a 16-entry LDS table; what did (258 -> 228 us) was
// unrolling the tap loops so that the tap arrays are not indexed dynamically: they had been living in scratch memory
// (52 bytes per lane), a chain of private-segment loads in front of every pixel. Now 36 % of the wave cycles sit in
// s_waitcnt and 34 % wait for issue with 5.5 waves per SIMD (LDS-limited): latency-bound at 0.25 of the HBM rate.
typedef float f2 __attribute__((ext_vector_type(2)));

// exp_softmax_f32 (pcnn_device.h) on two values at once; branch-free: the sub-normal scaling step multiplies by 1.0f
// where it does not apply (exact), a NaN argument is passed through by a final select
__device__ __forceinline__ f2 exp_softmax_f32x2(f2 x0)
{
  f2 x;
  x.x = fminf(fmaxf(x0.x, -104.f), 88.f);
  x.y = fminf(fmaxf(x0.y, -104.f), 88.f);
  const f2 t = x * 1.44269502f;
  f2 kf;
  kf.x = __builtin_rintf(t.x);
  kf.y = __builtin_rintf(t.y);
  f2 r = x - kf * 0.693145752f;
  r = r - kf * 1.42860677e-06f;
  f2 p = (f2){1.98412698e-04f, 1.98412698e-04f};
  p = p * r + 1.38888889e-03f;
  p = p * r + 8.33333377e-03f;
  p = p * r + 4.16666679e-02f;
  p = p * r + 1.66666672e-01f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  const int k0 = (int)kf.x, k1 = (int)kf.y;
  const bool s0 = k0 < -126, s1 = k1 < -126;
  f2 sc, m2;
  sc.x = __int_as_float((k0 + (s0 ? 64 + 127 : 127)) << 23);
  sc.y = __int_as_float((k1 + (s1 ? 64 + 127 : 127)) << 23);
  m2.x = s0 ? 5.42101086e-20f : 1.0f;   // 2^-64
  m2.y = s1 ? 5.42101086e-20f : 1.0f;
  f2 y = (p * sc) * m2;
  if (x0.x != x0.x) y.x = x0.x;
  if (x0.y != x0.y) y.y = x0.y;
  return y;
}

template <int C>
__global__ __launch_bounds__(UP_SEG) void upscore_softmax_argmax_fixed_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ score_out,
    float* __restrict__ prob, int* __restrict__ label, int H, int W, int k, int s, int relu, int nseg, int s_out_off,
    const int* __restrict__ gt, float* __restrict__ hard, float threshold)
{
  static_assert(C % 2 == 0 && C >= 2, "even class counts");
  constexpr int C2 = C / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = (k - s) / 2;
  const int Ho = H * s, Wo = W * s;
  const int seg = blockIdx.x % nseg;
  const int oy = (blockIdx.x / nseg) % Ho;
  const int b = blockIdx.x / (nseg * Ho);
  const int ox0 = seg * UP_SEG;
  const int npx = min(UP_SEG, Wo - ox0);
  const int tid = threadIdx.x;

  // the k tap weights once per workgroup (evaluated in double like make_deconv_filter), then integer arithmetic + lookups
  __shared__ float s_tab[64];
  if (tid < k) s_tab[tid] = bilinear_tap(tid, k);
  __syncthreads();
  const Taps ty = make_taps_tab(oy, k, s, pad, H, s_tab);
  const Taps tfirst = make_taps_tab(ox0, k, s, pad, W, s_tab);
  const Taps tlast = make_taps_tab(ox0 + npx - 1, k, s, pad, W, s_tab);
  const int cx0 = tfirst.i0;
  const int ncx = tlast.i0 + (tlast.n > 0 ? tlast.n : 1) - cx0;
  float* s_z = smem;                       // [ty.n][ncx][C]
  float* s_out = smem + s_out_off;         // [UP_SEG][C]
  const int rowlen = ncx * C;              // even: copied as float2
  {
    // the segment's ty.n rows of ncx cells: FOUR loads per thread issued before the first LDS write (a row-by-row copy loop
    // is a load, a wait and a store per trip — four trips to memory in sequence in front of every workgroup's first barrier)
    const int nrow2 = rowlen / 2, tot = ty.n * nrow2;
    const float* src0 = z + (((long long)b * H + ty.i0) * W + cx0) * C;
    f2* dst = reinterpret_cast<f2*>(s_z);
    for (int e0 = tid; e0 < tot; e0 += 4 * UP_SEG) {
      f2 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int e = min(e0 + j * UP_SEG, tot - 1);
        const int ry = e / nrow2, i = e - ry * nrow2;
        v[j] = *reinterpret_cast<const f2*>(src0 + (long long)ry * W * C + 2 * i);
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (e0 + j * UP_SEG < tot) dst[e0 + j * UP_SEG] = v[j];
    }
  }
  __syncthreads();

  const int ox = ox0 + tid;
  f2 e[C2];
  int best = 0;
  if (tid < npx) {
    const Taps tx = make_taps_tab(ox, k, s, pad, W, s_tab);
#pragma unroll
    for (int c = 0; c < C2; c++) e[c] = (f2){0.f, 0.f};
    // (unrolled over the <= 4 x 4 taps with guards: runtime loop bounds would index the tap arrays dynamically, which
    //  puts them in scratch memory — a chain of ~1 us private-segment loads in front of every pixel)
#pragma unroll
    for (int jy = 0; jy < 4; jy++)
      if (jy < ty.n) {
#pragma unroll
        for (int jx = 0; jx < 4; jx++)
          if (jx < tx.n) {
            const float w = ty.w[jy] * tx.w[jx];
            const f2* zc = reinterpret_cast<const f2*>(s_z + (jy * ncx + (tx.i0 + jx - cx0)) * C);
#pragma unroll
            for (int c = 0; c < C2; c++) e[c] = e[c] + w * zc[c];
          }
      }
    const f2* b2 = reinterpret_cast<const f2*>(bias);
#pragma unroll
    for (int c = 0; c < C2; c++) {
      f2 v = e[c] + b2[c];
      if (relu) {
        v.x = v.x > 0.f ? v.x : 0.f;
        v.y = v.y > 0.f ? v.y : 0.f;
      }
      e[c] = v;
    }
  }
  if (score_out) {
    if (tid < npx) {
      f2* so = reinterpret_cast<f2*>(s_out + tid * C);
#pragma unroll
      for (int c = 0; c < C2; c++) so[c] = e[c];
    }
    __syncthreads();
    float* o = score_out + (((long long)b * Ho + oy) * Wo + ox0) * C;
    for (int i = tid; i < npx * C; i += UP_SEG) o[i] = s_out[i];
    __syncthreads();
  }
  if (tid < npx) {
    float m = e[0].x;
#pragma unroll
    for (int c = 0; c < C2; c++) { m = fmaxf(m, e[c].x); m = fmaxf(m, e[c].y); }
    float sum = 0.f;
    const f2 m2 = (f2){m, m};
#pragma unroll
    for (int c = 0; c < C2; c++) {
      e[c] = exp_softmax_f32x2(e[c] - m2);
      sum += e[c].x;
      sum += e[c].y;
    }
    float bestp = div_rn(e[0].x, sum);
#pragma unroll
    for (int c = 0; c < C2; c++) {
      const float p0 = div_rn(e[c].x, sum), p1 = div_rn(e[c].y, sum);
      e[c] = (f2){p0, p1};
      if (p0 > bestp) { bestp = p0; best = 2 * c; }
      if (p1 > bestp) { bestp = p1; best = 2 * c + 1; }
    }
    label[((long long)b * Ho + oy) * Wo + ox] = best;
  }
  if (prob || hard) {
    if (tid < npx) {
      f2* so = reinterpret_cast<f2*>(s_out + tid * C);
#pragma unroll
      for (int c = 0; c < C2; c++) so[c] = e[c];
    }
    __syncthreads();
    const long long pix0 = ((long long)b * Ho + oy) * Wo + ox0;
    if (prob) up_copy_out(prob + pix0 * C, s_out, npx * C, tid, UP_SEG, UP_SEG * C);
    if (hard) up_hard_label_rows(s_out, gt, hard, pix0, tid, npx, C, threshold, UP_SEG);
  }
}

int validate(int B, int H, int W, int C, int k, int s)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, PCNN_EINVAL, "deconv: bad shape %dx%dx%dx%d", B, H, W, C);
  PCNN_REQUIRE(s >= 1 && k >= s && (k - s) % 2 == 0 && k <= 4 * s, PCNN_EINVAL,
               "deconv: need stride >= 1, stride <= kernel <= 4*stride and (kernel - stride) even (got k=%d s=%d)", k, s);
  PCNN_REQUIRE((long long)B * H * s * W * s * C < (1ll << 40), PCNN_EINVAL, "deconv: output too large");
  return PCNN_OK;
}

}  // namespace

extern "C" int pcnn_deconv_bilinear_fwd(const float* in, int B, int H, int W, int C, int k, int s,
                                        const float* add1, const float* add2, const float* bias,
                                        int relu, float* out, void* stream_)
{
  int st = validate(B, H, W, C, k, s);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(in && out, PCNN_ENULL, "deconv: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const bool al = aligned16(in) && aligned16(out) && (!add1 || aligned16(add1)) && (!add2 || aligned16(add2));
  const int Wo = W * s, Ho = H * s;
  const int nseg = (Wo + DC_SEG - 1) / DC_SEG;
  const long long blocks = (long long)B * Ho * nseg;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "deconv: grid too large");
  if (al && C % 4 == 0)
    PCNN_LAUNCH(deconv_bilinear_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, in, add1, add2, bias, out, H, W, C, k, s, relu, nseg);
  else if (al && C % 2 == 0)
    PCNN_LAUNCH(deconv_bilinear_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, in, add1, add2, bias, out, H, W, C, k, s, relu, nseg);
  else
    PCNN_LAUNCH(deconv_bilinear_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, in, add1, add2, bias, out, H, W, C, k, s, relu, nseg);
  return check_launch("deconv_bilinear_fwd");
}

extern "C" int pcnn_deconv_bilinear_bwd(const float* grad_out, int B, int H, int W, int C, int k,
                                        int s, float* grad_in, void* stream_)
{
  int st = validate(B, H, W, C, k, s);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(grad_out && grad_in, PCNN_ENULL, "deconv_bwd: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  auto grid = [](long long total) { long long b = (total + 255) / 256; return (unsigned)(b < 256 * 64 ? b : 256 * 64); };
  const long long n = (long long)B * H * W * C;
  if (C % 4 == 0 && aligned16(grad_out) && aligned16(grad_in))
    PCNN_LAUNCH(deconv_bilinear_bwd_kernel<4>, dim3(grid(n / 4)), dim3(256), 0, stream, grad_out, grad_in, H, W, C, k, s, n / 4);
  else
    PCNN_LAUNCH(deconv_bilinear_bwd_kernel<1>, dim3(grid(n)), dim3(256), 0, stream, grad_out, grad_in, H, W, C, k, s, n);
  return check_launch("deconv_bilinear_bwd");
}

extern "C" int pcnn_bias_act_fwd(const float* x, const float* bias, int64_t num_pixels, int channels,
                                 int relu, float* y, void* stream_)
{
  PCNN_REQUIRE(num_pixels >= 0 && channels >= 1, PCNN_EINVAL, "bias_act: bad shape");
  if (num_pixels == 0) return PCNN_OK;
  PCNN_REQUIRE(x && bias && y, PCNN_ENULL, "bias_act: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  auto grid = [](long long total) { long long b = (total + 255) / 256; return (unsigned)(b < 256 * 32 ? b : 256 * 32); };
  const long long n = (long long)num_pixels * channels;
  if (channels % 4 == 0 && aligned16(x) && aligned16(y))
    PCNN_LAUNCH(bias_act_kernel<4>, dim3(grid(n / 4)), dim3(256), 0, stream, x, bias, y, n / 4, channels, relu);
  else
    PCNN_LAUNCH(bias_act_kernel<1>, dim3(grid(n)), dim3(256), 0, stream, x, bias, y, n, channels, relu);
  return check_launch("bias_act_fwd");
}

extern "C" int pcnn_bias_relu_pool2_fwd(const float* x, const float* bias, int B, int H, int W, int C,
                                       int relu, float* y, void* stream_)
{
  PCNN_REQUIRE(B >= 1 && H >= 2 && W >= 2 && C >= 1 && H % 2 == 0 && W % 2 == 0, PCNN_EINVAL,
               "bias_relu_pool2: need even height/width (got %dx%dx%dx%d)", B, H, W, C);
  PCNN_REQUIRE(x && bias && y, PCNN_ENULL, "bias_relu_pool2: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int Ho = H / 2, Wo = W / 2;
  auto grid = [](long long total) { long long b = (total + 255) / 256; return (unsigned)(b < 256 * 32 ? b : 256 * 32); };
  const long long n = (long long)B * Ho * Wo * C;
  if (C % 4 == 0 && aligned16(x) && aligned16(y))
    PCNN_LAUNCH(bias_relu_pool2_kernel<4>, dim3(grid(n / 4)), dim3(256), 0, stream, x, bias, y, n / 4, Ho, Wo, C, relu);
  else
    PCNN_LAUNCH(bias_relu_pool2_kernel<1>, dim3(grid(n)), dim3(256), 0, stream, x, bias, y, n, Ho, Wo, C, relu);
  return check_launch("bias_relu_pool2_fwd");
}

namespace {
int upscore_softmax_argmax_impl(const char* who, const float* z, const float* bias, int B, int H, int W, int C, int k, int s,
                                int relu, float* score_out, float* prob, int32_t* label, const int32_t* gt, float threshold,
                                float* hard, void* stream_)
{
  int st = validate(B, H, W, C, k, s);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(C <= PCNN_MAX_CLASSES, PCNN_EINVAL, "%s: num_classes must be <= %d (got %d)", who, PCNN_MAX_CLASSES, C);
  PCNN_REQUIRE(k <= 64, PCNN_EINVAL, "%s: kernel size must be <= 64 (got %d)", who, k);
  PCNN_REQUIRE(z && bias && label, PCNN_ENULL, "%s: NULL pointer", who);
  hipStream_t stream = (hipStream_t)stream_;
  const int Wo = W * s, Ho = H * s;
  const int nseg = (Wo + UP_SEG - 1) / UP_SEG;
  const long long blocks = (long long)B * Ho * nseg;
  PCNN_REQUIRE(blocks < (1ll << 31), PCNN_EINVAL, "%s: grid too large", who);
  const int rows = (k + s - 1) / s;                      // input rows per output row
  const int ncx_max = (UP_SEG - 1) / s + rows + 1;       // input columns per segment
  const int s_out_off = (rows * ncx_max * C + 3) / 4 * 4;
  const size_t sh = sizeof(float) * ((size_t)s_out_off + (size_t)UP_SEG * C);
  PCNN_REQUIRE(sh <= 160 * 1024, PCNN_EINVAL, "%s: tile does not fit LDS", who);
  // class counts known at compile time (even; 8-byte aligned operands): the unrolled packed-f32 kernel
  const bool al8 = ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(bias)) & 7) == 0 && s_out_off % 2 == 0;
#define UP_FIXED(CC) PCNN_LAUNCH(upscore_softmax_argmax_fixed_kernel<CC>, dim3((unsigned)blocks), dim3(UP_SEG), sh, stream, z, bias, score_out, prob, label, H, W, k, s, relu, nseg, s_out_off, gt, hard, threshold)
  if (al8 && C == 22) UP_FIXED(22);
  else if (al8 && C == 14) UP_FIXED(14);
  else if (al8 && C == 16) UP_FIXED(16);
#undef UP_FIXED
  else if (C <= 24)
    PCNN_LAUNCH(upscore_softmax_argmax_kernel<24>, dim3((unsigned)blocks), dim3(UP_SEG), sh, stream, z, bias, score_out, prob, label, H, W, C, k, s, relu, nseg, s_out_off, gt, hard, threshold);
  else
    PCNN_LAUNCH(upscore_softmax_argmax_kernel<64>, dim3((unsigned)blocks), dim3(UP_SEG), sh, stream, z, bias, score_out, prob, label, H, W, C, k, s, relu, nseg, s_out_off, gt, hard, threshold);
  return check_launch(who);
}
}  // namespace

extern "C" int pcnn_upscore_softmax_argmax_fwd(const float* z, const float* bias, int B, int H,
                                               int W, int C, int k, int s, int relu,
                                               float* score_out, float* prob, int32_t* label,
                                               void* stream_)
{
  return upscore_softmax_argmax_impl("upscore_softmax_argmax_fwd", z, bias, B, H, W, C, k, s, relu, score_out, prob, label,
                                     nullptr, 0.f, nullptr, stream_);
}

extern "C" int pcnn_upscore_softmax_argmax_hard_fwd(const float* z, const float* bias, int B, int H,
                                                    int W, int C, int k, int s, int relu,
                                                    float* score_out, float* prob, int32_t* label,
                                                    const int32_t* gt, float threshold, float* hard, void* stream_)
{
  // attribute check of the Hardlabel op, hard_label_op.cc:150-155
  PCNN_REQUIRE(threshold > 0, PCNN_EINVAL, "upscore_softmax_argmax_hard: Need threshold > 0, got %g", (double)threshold);
  PCNN_REQUIRE(gt && hard, PCNN_ENULL, "upscore_softmax_argmax_hard: NULL pointer");
  return upscore_softmax_argmax_impl("upscore_softmax_argmax_hard_fwd", z, bias, B, H, W, C, k, s, relu, score_out, prob, label,
                                     gt, threshold, hard, stream_);
}
