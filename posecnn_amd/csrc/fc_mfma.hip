// fc_mfma.hip — the fully connected layers of the pose branch (`Network.fc`, lib/networks/network.py:392-422:
// fc6 [R, 7*7*512] -> 4096 and fc7 4096 -> 4096 of lib/networks/vgg16_convs.py:188-192, relu_layer) on the
// gfx950 matrix cores, over a CAPACITY-SIZED row buffer with the row count on the device.
//
// Why not the library GEMM: the sync-free Hough layer hands the pose branch `rows_capacity` ROI rows (3024
// for 16 frames x 21 classes x 9 training rows) of which only `*num_rows_dev` exist (~680 in the bench). A
// library GEMM needs M on the host — a device->host sync per batch — or computes all 3024 rows (4.8 ms of
// fp32 MFMA per batch, measured). This kernel reads the count on the device: row blocks at or past it store
// zeros and exit before touching an operand.
//
//   y[m, n] = [ReLU](sum_k x[m, k] wt[n, k] + bias[n])   m < min(M_cap, *num_rows_dev), else 0
//
// Same machinery as csrc/wino_mfma.hip, minus the Winograd planes: 64 x 64 block per 8-wave workgroup,
// v_mfma_f32_16x16x4_f32 (exact f32), global -> LDS DMA into a 2-stage ring (64 KB: TWO workgroups per
// CU, which like the trunk's pairs drift out of phase and cover each other's barrier and DMA waits —
// fc6 at 468 rows 1.38 -> 1.15 ms against the 3-stage ring with one workgroup per CU) with XOR-swizzled
// 16-byte chunks, one barrier per 64-deep K stage, K loop software pipelined by half a stage, epilogue
// through LDS for 256-byte row stores, XCD-aware block map (the column blocks that share a row block's x
// rows run on one XCD). `wt` is the weight matrix TRANSPOSED ([N][K], K contiguous) so that both operands
// are K-major rows for the DMA.
#include <algorithm>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int FC_LD = 64;
constexpr int FC_NBUF = 2;

// 16 bytes per lane, global -> LDS: SGPR base + 32-bit VGPR byte offset, LDS base in M0 (spelled out in asm
// like wino_mfma.hip's glds16_s: through the builtin the compiler builds 64-bit VGPR addresses per load).
__device__ __forceinline__ void fc_glds16_s(const char* sbase, unsigned voff, unsigned lds_addr)
{
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// Split-K factor, decided ON THE DEVICE from the live row count (same value in every workgroup of the
// launch and in the reduction kernel): with few live rows there are few (row block, column block)
// workgroups — fc6 at 75 rows: 128 for 512 slots, each streaming its 6.4 MB weight slab through a
// 2-deep ring, i.e. bound by HBM latency (0.92 ms for 411 MB). Splitting K over up to `smax` workgroups
// multiplies the loads in flight; the partial products meet in a fixed-order reduction (deterministic).
// (round 6) A second reason to split: a launch a little larger than the chip's 512 workgroup slots. fc6 / fc7 of the train
// batch: 684 live rows = 11 row blocks x 64 column blocks = 704 workgroups = 1.375 rounds — the last 192 run alone on
// their CUs at ~65 % of the pair's rate while 320 slots idle (the same shape of loss as the trunk's conv5_x, traced in
// profiles/r06_wino_mfma_trace.txt). Two half-length splits are 1 408 workgroups = 2.75 rounds, i.e. 3 half rounds
// instead of 2 whole ones. The smallest S whose rounds-per-split figure ceil(live S / 512) / S beats S = 1 by 15 % (the
// price of the reduction pass) is taken. Measured (tools/bench_fc_rows.py, same box A / B): fc6 at 684 rows 1.42 -> 1.355 ms
// (99 -> 104 TFLOP/s; the kernel does 131 at 1 500 rows = three exact rounds, so most of the quantisation loss is still
// there: half-length workgroups pay their prologue and 17-us epilogue twice), at 468 rows 0.88 -> 0.87.
// `wscap` = rows the workspace holds over all splits; a split needs S x roundup64(count) of them.
__device__ __forceinline__ int fc_split(int count, int ncb, int NK, int smax, int wscap)
{
  if (smax <= 1) return 1;
  const int rows64 = ((count + 63) >> 6) << 6;
  const int live = (rows64 >> 6) * ncb;
  const int s_fit = min(min(smax, wscap / max(rows64, 64)), NK / 8);   // at least 8 stages (512 of K) per split
  if (s_fit <= 1) return 1;
  if (live <= 512) return max(min(s_fit, 1024 / max(live, 1)), 1);   // few workgroups: up to two rounds' worth of them
  if (NK < 128) return 1;   // fc7 (64 stages per workgroup): half-length workgroups + the reduction measured 0.22 -> 0.25 ms at 684 rows
  int best = 1;
  float best_cost = (float)((live + 511) / 512);
  for (int S = 2; S <= min(s_fit, 4); S++) {
    const float cost = (float)((live * S + 511) / 512) / (float)S;
    if (cost < 0.85f * best_cost) { best = S; best_cost = cost / 0.85f; }   // (a later S must beat the one taken outright)
  }
  return best;
}
// rows between two splits' partial outputs in the workspace
__device__ __forceinline__ int fc_split_stride(int count) { return ((count + 63) >> 6) << 6; }

__global__ __launch_bounds__(512, 4) void fc_rows_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ addend, float* __restrict__ y, int K, int N, int Mcap, int relu,
    const int* __restrict__ num_rows_dev, int nbm, int ncb, int tall, float* __restrict__ part,
    int smax, int wscap, int ldy, int nvalid, float* __restrict__ y2, int split_col, float* __restrict__ y_b, int relu_b)
{
  // split_col / y_b (pcnn_fc_rows_split_fwd): output columns [split_col, N) go to a SECOND tensor y_b [Mcap][N - split_col] with
  // their own ReLU flag — two layers that read the same rows (score_conv4 and score_conv4_vertex on conv4_3) as one product.
  // ldy / nvalid (pcnn_fc_rows_cols_fwd): y has `ldy` floats per row and only output columns < nvalid exist (the weight rows
  // past them are zero padding up to the kernel's 64-column blocks); y2, if given, receives tanh(y). Plain fc_rows: ldy =
  // nvalid = N, y2 = NULL.
  __shared__ __attribute__((aligned(16))) float smem[FC_NBUF * 128 * FC_LD];   // sA[3][64][64] | sB[3][64][64]
  float* sAp = smem;
  float* sBp = smem + FC_NBUF * 64 * FC_LD;

  // XCD-aware block map: the weight matrix is the big operand (fc6: 411 MB) and every row block needs
  // all of a column block's rows of it, so XCD x takes the column blocks cb == x (mod 8) and runs their
  // row blocks back to back: W^T streams from HBM once, the (few) live x rows are re-read from L2 / MALL
  // `tall` (x is the big operand: the 1x1 head convolutions, 76 800 rows x 64..128 columns): the other
  // way round — XCD x takes the row blocks tb == x (mod 8) and runs their column blocks back to back.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tb = tall ? (q / ncb) * 8 + xcd : q % nbm;
  const int cb = tall ? q % ncb : (q / nbm) * 8 + xcd;
  if (cb >= ncb || tb >= nbm) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS bases and M0 stay on the SALU
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int m0 = tb * 64;
  const int count = num_rows_dev ? min(Mcap, num_rows_dev[0]) : Mcap;
  const int ks = blockIdx.y;
  const int S = fc_split(count, ncb, K / 64, smax, wscap);
  if (ks >= S) return;
  if (m0 >= count) {
    // rows that do not exist: zeros, no operand traffic
    if (ks > 0) return;
    for (int i = tid; i < 64 * 16; i += 512) {
      const int r = m0 + (i >> 4);
      const int c4_ = cb * 64 + (i & 15) * 4;
      if (r < Mcap && c4_ < nvalid) {
        if (y_b != nullptr && c4_ >= split_col) { *reinterpret_cast<v4f*>(y_b + (size_t)r * (N - split_col) + (c4_ - split_col)) = (v4f){0.f, 0.f, 0.f, 0.f}; continue; }
        *reinterpret_cast<v4f*>(y + (size_t)r * ldy + c4_) = (v4f){0.f, 0.f, 0.f, 0.f};
        if (y2) *reinterpret_cast<v4f*>(y2 + (size_t)r * ldy + c4_) = (v4f){0.f, 0.f, 0.f, 0.f};   // tanh(0)
      }
    }
    return;
  }
  const int kfirst = (int)((long long)(K / 64) * ks / S), klast = (int)((long long)(K / 64) * (ks + 1) / S);   // this split's stages
  const int NK = klast - kfirst;
  const int mlast = count - 1;

  const int dr0 = 8 * wave + lk, dr1 = dr0 + 4;
  const int ra_ = min(m0 + dr0, mlast), rb_ = min(m0 + dr1, mlast);   // rows past the count: any finite data, zeroed at the end
  const unsigned va0 = (unsigned)(((size_t)ra_ * K + (lr ^ (dr0 & 15)) * 4) * 4);
  const unsigned va1 = (unsigned)(((size_t)rb_ * K + (lr ^ (dr1 & 15)) * 4) * 4);
  const unsigned ub0 = (unsigned)(((size_t)(cb * 64 + dr0) * K + (lr ^ (dr0 & 15)) * 4) * 4);
  const unsigned ub1 = (unsigned)(((size_t)(cb * 64 + dr1) * K + (lr ^ (dr1 & 15)) * 4) * 4);
  const char* xbase = reinterpret_cast<const char*>(x);
  const char* wbase = reinterpret_cast<const char*>(wt);
  const int ldsw = 8 * wave * FC_LD;
  const unsigned lds_base_ = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  const unsigned lds_a0 = lds_base_ + (unsigned)ldsw * 4u;                               // wave-uniform: M0 of the DMA
  const unsigned lds_b0 = lds_base_ + (unsigned)(FC_NBUF * 64 * FC_LD + ldsw) * 4u;

#define FC_DMA(BUF, KO)                                                                  \
  do {                                                                                   \
    const char* xs_ = xbase + (size_t)(KO) * 4;                                          \
    const char* ws_ = wbase + (size_t)(KO) * 4;                                          \
    const unsigned la_ = lds_a0 + (unsigned)(BUF) * (64 * FC_LD * 4);                    \
    const unsigned lb_ = lds_b0 + (unsigned)(BUF) * (64 * FC_LD * 4);                    \
    fc_glds16_s(xs_, va0, la_);                                                          \
    fc_glds16_s(xs_, va1, la_ + 4 * FC_LD * 4);                                          \
    fc_glds16_s(ws_, ub0, lb_);                                                          \
    fc_glds16_s(ws_, ub1, lb_ + 4 * FC_LD * 4);                                          \
  } while (0)

  v4f acc0 = (v4f){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  int pk = kfirst * 64;   // K offset (floats) of the prefetch pointer; parks on the split's last stage
  const int kpark = klast * 64 - 64;
  FC_DMA(0, pk); pk = min(pk + 64, kpark);
  if (FC_NBUF == 3) { FC_DMA(1, pk); pk = min(pk + 64, kpark); }
  int cur = 0;

  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  const int ra0 = (32 * wm + lr) * FC_LD, rb = (16 * wn + lr) * FC_LD;
  unsigned adA[4], adB[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned chb = (unsigned)(((4 * j + lk) ^ lr) * 16);
    adA[j] = lds0 + (unsigned)ra0 * 4u + chb;
    adB[j] = lds0 + (unsigned)(FC_NBUF * 64 * FC_LD + rb) * 4u + chb;
  }
  v4f xa0[2], xa1[2], xb[2], ya0[2], ya1[2], yb[2];
#pragma unroll
  for (int g = 0; g < 2; g++) xa0[g] = xa1[g] = xb[g] = ya0[g] = ya1[g] = yb[g] = (v4f){0.f, 0.f, 0.f, 0.f};
#define FC_DSREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR))
#define FC_READ(SA0, SA1, SB, G0)                                             \
  _Pragma("unroll") for (int g_ = 0; g_ < 2; g_++) {                          \
    const unsigned aa_ = adA[(G0) + g_] + curo, ab_ = adB[(G0) + g_] + curo;  \
    FC_DSREAD(SB[g_], ab_, 0);                                                \
    FC_DSREAD(SA0[g_], aa_, 0);                                               \
    FC_DSREAD(SA1[g_], aa_, 4096);                                            \
  }
#define FC_MFMA1(SA0, SA1, SB, G)                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {                                          \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(SA0[G][i_], SB[G][i_], acc0, 0, 0, 0);        \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(SA1[G][i_], SB[G][i_], acc1, 0, 0, 0);        \
  }

  // stage s: own DMAs landed (the next stage's stay in flight) -> barrier -> reads X(s) -> MFMAs Y(s-1)
  // + DMA of stage s+2 -> X landed -> MFMAs X(s), reads Y(s) in between. Y of "stage -1" is zeros.
  for (int s = 0; s < NK; s++) {
    if (FC_NBUF == 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned curo = (unsigned)cur * (64 * FC_LD * 4);
    FC_READ(xa0, xa1, xb, 0);
    __builtin_amdgcn_sched_barrier(0);
    FC_MFMA1(ya0, ya1, yb, 0) FC_MFMA1(ya0, ya1, yb, 1)
    {
      const int nb = FC_NBUF == 3 ? (cur >= 1 ? cur - 1 : 2) : (cur ^ 1);
      FC_DMA(nb, pk);
      pk = min(pk + 64, kpark);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    FC_MFMA1(xa0, xa1, xb, 0)
    __builtin_amdgcn_sched_barrier(0);
    FC_READ(ya0, ya1, yb, 2);
    __builtin_amdgcn_sched_barrier(0);
    FC_MFMA1(xa0, xa1, xb, 1)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur = FC_NBUF == 3 ? (cur == 2 ? 0 : cur + 1) : (cur ^ 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  FC_MFMA1(ya0, ya1, yb, 0) FC_MFMA1(ya0, ya1, yb, 1)
  // Every wave's outstanding DMA must have LANDED before anybody recycles the ring as the epilogue's staging buffer.
  // The DMA instructions are inline asm, invisible to the compiler's s_waitcnt insertion, so __syncthreads() alone
  // is a bare s_barrier here (checked in the ISA): a parked prefetch of ANOTHER wave could land after this wave had
  // written its output rows into the same LDS — output rows replaced by operand rows, a few times in a thousand
  // launches on an idle GPU, a few per hundred with a second stream competing for memory (round 3,
  // tools/debug_streams.py). The wait has to be explicit and in front of the barrier.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // ... and every LDS read is done: the buffers can be recycled
#undef FC_DMA
#undef FC_READ
#undef FC_DSREAD
#undef FC_MFMA1

  // epilogue: lane holds rows 32 wm + 16 b + 4 lk + i (b = 0: acc0, 1: acc1) x column 16 wn + lr
  const int col = 16 * wn + lr;
  const float bv = S > 1 ? 0.f : bias[cb * 64 + col];
  float* sY = smem;   // [64 rows][64 columns]
#pragma unroll
  for (int i = 0; i < 4; i++) {
    sY[(32 * wm + 4 * lk + i) * 64 + col] = acc0[i] + bv;
    sY[(32 * wm + 16 + 4 * lk + i) * 64 + col] = acc1[i] + bv;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int idx = tid + 512 * r;   // 1024 float4: (row, c4)
    const int row = idx >> 4, c4 = (idx & 15) * 4;
    const int m = m0 + row;
    if (S > 1) {   // partial product of this K range; bias, addend, ReLU and the zero rows belong to the reduction
      if (m < count)
        *reinterpret_cast<v4f*>(part + ((size_t)ks * fc_split_stride(count) + m) * N + cb * 64 + c4) = *reinterpret_cast<const v4f*>(&sY[row * 64 + c4]);
      continue;
    }
    if (m < Mcap && cb * 64 + c4 < nvalid) {
      const int col0 = cb * 64 + c4;
      const bool second = y_b != nullptr && col0 >= split_col;
      v4f val = (v4f){0.f, 0.f, 0.f, 0.f};
      if (m < count) {
        val = *reinterpret_cast<const v4f*>(&sY[row * 64 + c4]);
        if (addend) val += *reinterpret_cast<const v4f*>(addend + (size_t)m * N + col0);
        if (second ? relu_b : relu) {
#pragma unroll
          for (int e = 0; e < 4; e++) val[e] = val[e] > 0.f ? val[e] : 0.f;
        }
      }
      if (second) { *reinterpret_cast<v4f*>(y_b + (size_t)m * (N - split_col) + (col0 - split_col)) = val; continue; }
      *reinterpret_cast<v4f*>(y + (size_t)m * ldy + col0) = val;
      if (y2) {
        v4f t;
#pragma unroll
        for (int e = 0; e < 4; e++) t[e] = tanhf(val[e]);
        *reinterpret_cast<v4f*>(y2 + (size_t)m * ldy + cb * 64 + c4) = t;
      }
    }
  }
}

// y[m] = [ReLU](sum_{ks < S} part[ks][m] + bias + addend[m]) in ascending ks order; rows between the count
// and the end of its 64-row block are zeroed (whole blocks past the count: by the product kernel)
__global__ __launch_bounds__(256) void fc_rows_reduce_kernel(
    const float* __restrict__ part, const float* __restrict__ bias, const float* __restrict__ addend,
    float* __restrict__ y, int K, int N, int Mcap, int relu, const int* __restrict__ num_rows_dev, int ncb,
    int smax, int wscap)
{
  const int count = num_rows_dev ? min(Mcap, num_rows_dev[0]) : Mcap;
  const int S = fc_split(count, ncb, K / 64, smax, wscap);
  if (S <= 1) return;
  const int n4 = N >> 2;
  const long long total = (long long)min(Mcap, ((count + 63) >> 6) << 6) * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / n4), c4 = (int)(i - (long long)m * n4) * 4;
    v4f val = (v4f){0.f, 0.f, 0.f, 0.f};
    if (m < count) {
      for (int ks = 0; ks < S; ks++) val += *reinterpret_cast<const v4f*>(part + ((size_t)ks * fc_split_stride(count) + m) * N + c4);
      val += *reinterpret_cast<const v4f*>(bias + c4);
      if (addend) val += *reinterpret_cast<const v4f*>(addend + (size_t)m * N + c4);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; e++) val[e] = val[e] > 0.f ? val[e] : 0.f;
      }
    }
    *reinterpret_cast<v4f*>(y + (size_t)m * N + c4) = val;
  }
}

constexpr int FC_SMAX = 8;        // K splits at most
constexpr int FC_WS_ROWS = 512;   // rows the split-K workspace holds (more live rows never split)

bool fc_can_split(int rows_capacity, int in_features, int out_features)
{
  return rows_capacity <= out_features && in_features >= 1024;   // not the tall (1x1 conv) shape; a K worth splitting
}

}  // namespace

extern "C" int pcnn_fc_rows_workspace_bytes(int rows_capacity, int in_features, int out_features, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "fc_rows_workspace_bytes: NULL output");
  PCNN_REQUIRE(rows_capacity >= 0 && in_features >= 1 && out_features >= 1, PCNN_EINVAL, "fc_rows_workspace_bytes: bad shape");
  const int ws_rows = rows_capacity < FC_WS_ROWS ? (rows_capacity + 63) / 64 * 64 : FC_WS_ROWS;
  *bytes = fc_can_split(rows_capacity, in_features, out_features) ? sizeof(float) * (size_t)FC_SMAX * ws_rows * out_features : 0;
  return PCNN_OK;
}

extern "C" int pcnn_fc_rows_fwd(const float* x, const float* wt, const float* bias, int rows_capacity,
                                int in_features, int out_features, int relu, const int32_t* num_rows_dev,
                                const float* addend, float* y, void* workspace, size_t workspace_bytes,
                                void* stream_)
{
  PCNN_REQUIRE(rows_capacity >= 0, PCNN_EINVAL, "fc_rows: negative row capacity");
  PCNN_REQUIRE(in_features >= 128 && in_features % 64 == 0, PCNN_EINVAL,
               "fc_rows: in_features must be a multiple of 64, >= 128 (got %d)", in_features);
  PCNN_REQUIRE(out_features >= 64 && out_features % 64 == 0, PCNN_EINVAL,
               "fc_rows: out_features must be a multiple of 64 (got %d)", out_features);
  if (rows_capacity == 0) return PCNN_OK;
  PCNN_REQUIRE(x && wt && bias && y, PCNN_ENULL, "fc_rows: NULL pointer");
  PCNN_REQUIRE(aligned16(x) && aligned16(wt) && aligned16(y) && aligned16(addend) && aligned16(bias), PCNN_EINVAL,
               "fc_rows: pointers (x, wt, bias, addend, y) must be 16-byte aligned");
  PCNN_REQUIRE((long long)rows_capacity * in_features < (1ll << 30) && (long long)out_features * in_features < (1ll << 30),
               PCNN_EINVAL, "fc_rows: operand larger than the 32-bit byte offsets of the kernel");
  hipStream_t stream = (hipStream_t)stream_;
  const int nbm = (rows_capacity + 63) / 64, ncb = out_features / 64;
  const int tall = (long long)rows_capacity > (long long)out_features;   // which operand is the big one
  const long long blocks = tall ? (long long)((nbm + 7) / 8) * 8 * ncb : (long long)((ncb + 7) / 8) * 8 * nbm;
  // split-K for launches with few live rows (decided on the device): needs the caller's workspace
  const int ws_rows = rows_capacity < FC_WS_ROWS ? (rows_capacity + 63) / 64 * 64 : FC_WS_ROWS;
  const size_t need = sizeof(float) * (size_t)FC_SMAX * ws_rows * out_features;
  // grid.y = the most splits a launch may use; every split of every block is a workgroup that must at least
  // be launched to find out that it has nothing to do: a big capacity (train mode: 48 x 64 blocks) gets 2 — what
  // the balance split above can use — and a small one up to FC_SMAX
  // (a short-K layer — fc7: 64 stages — never takes the balance split: no second grid row, no reduction launch for it)
  const long long floor_s = in_features / 64 >= 128 ? 2 : 1;
  const int smax_cap = (int)std::min<long long>(FC_SMAX, std::max<long long>(floor_s, 4096 / ((long long)nbm * ncb)));
  const int smax = (workspace && workspace_bytes >= need && fc_can_split(rows_capacity, in_features, out_features) && aligned16(workspace)) ? smax_cap : 1;
  float* part = smax > 1 ? static_cast<float*>(workspace) : nullptr;
  const int wscap = FC_SMAX * ws_rows;   // rows of partial outputs the workspace holds
  PCNN_LAUNCH(fc_rows_mfma_kernel, dim3((unsigned)blocks, smax), dim3(512), 0, stream, x, wt, bias, addend, y, in_features,
              out_features, rows_capacity, relu, num_rows_dev, nbm, ncb, tall, part, smax, wscap, out_features, out_features, (float*)nullptr, 0, (float*)nullptr, 0);
  if (smax > 1) {
    const long long items = (long long)std::min(rows_capacity, wscap / 2) * (out_features / 4);   // (a split launch has at most wscap / 2 live rows)
    PCNN_LAUNCH(fc_rows_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, part, bias, addend, y,
                in_features, out_features, rows_capacity, relu, num_rows_dev, ncb, smax, wscap);
  }
  return check_launch("fc_rows_fwd");
}

// The same product for a layer whose width is no multiple of the kernel's 64-column blocks (fc8: 4 C = 88 outputs of 4096
// inputs, lib/networks/vgg16_convs.py:192-193, at more rows than csrc/fc_skinny.hip takes): `wt` / `bias` are PADDED with
// zero rows / entries to `out_padded` (a multiple of 64), y and y_tanh are [rows_capacity][out_features] with out_features
// % 4 == 0; activation 0 none, 1 ReLU, 2 tanh — then y holds the linear output and y_tanh its tanh (fc8 and poses_tanh
// from one launch). No split-K (the layer's K is 4096: 64 stages per workgroup). Replaces the library GEMM + tanh.
extern "C" int pcnn_fc_rows_cols_fwd(const float* x, const float* wt, const float* bias, int rows_capacity,
                                     int in_features, int out_padded, int out_features, int activation,
                                     const int32_t* num_rows_dev, float* y, float* y_tanh, void* stream_)
{
  PCNN_REQUIRE(rows_capacity >= 0, PCNN_EINVAL, "fc_rows_cols: negative row capacity");
  PCNN_REQUIRE(in_features >= 128 && in_features % 64 == 0, PCNN_EINVAL,
               "fc_rows_cols: in_features must be a multiple of 64, >= 128 (got %d)", in_features);
  PCNN_REQUIRE(out_padded >= 64 && out_padded % 64 == 0, PCNN_EINVAL, "fc_rows_cols: out_padded must be a multiple of 64 (got %d)", out_padded);
  PCNN_REQUIRE(out_features >= 4 && out_features % 4 == 0 && out_features <= out_padded && out_features > out_padded - 64, PCNN_EINVAL,
               "fc_rows_cols: out_features must be a multiple of 4 inside the last 64-column block of out_padded (got %d of %d)", out_features, out_padded);
  PCNN_REQUIRE(activation >= 0 && activation <= 2, PCNN_EINVAL, "fc_rows_cols: activation 0 (none), 1 (ReLU) or 2 (tanh)");
  if (rows_capacity == 0) return PCNN_OK;
  PCNN_REQUIRE(x && wt && bias && y && (activation != 2 || y_tanh), PCNN_ENULL, "fc_rows_cols: NULL pointer");
  PCNN_REQUIRE(aligned16(x) && aligned16(wt) && aligned16(y) && aligned16(y_tanh) && aligned16(bias), PCNN_EINVAL,
               "fc_rows_cols: pointers (x, wt, bias, y, y_tanh) must be 16-byte aligned");
  PCNN_REQUIRE((long long)rows_capacity * in_features < (1ll << 30) && (long long)out_padded * in_features < (1ll << 30),
               PCNN_EINVAL, "fc_rows_cols: operand larger than the 32-bit byte offsets of the kernel");
  hipStream_t stream = (hipStream_t)stream_;
  const int nbm = (rows_capacity + 63) / 64, ncb = out_padded / 64;
  const int tall = (long long)rows_capacity > (long long)out_padded;
  const long long blocks = tall ? (long long)((nbm + 7) / 8) * 8 * ncb : (long long)((ncb + 7) / 8) * 8 * nbm;
  PCNN_LAUNCH(fc_rows_mfma_kernel, dim3((unsigned)blocks, 1), dim3(512), 0, stream, x, wt, bias, (const float*)nullptr, y, in_features,
              out_padded, rows_capacity, activation == 1 ? 1 : 0, num_rows_dev, nbm, ncb, tall, (float*)nullptr, 1, 0, out_features, out_features,
              activation == 2 ? y_tanh : (float*)nullptr, 0, (float*)nullptr, 0);
  return check_launch("fc_rows_cols_fwd");
}

// Two layers on the same rows as ONE product (round 5: score_conv4 [ReLU] and score_conv4_vertex [none] both read conv4_3,
// likewise on conv5_3 — vgg16_convs.py:128-133,151-157): wt [out_a + out_b][in_features] = the two filters one after the other,
// y_a [rows][out_a], y_b [rows][out_b], a ReLU flag each. out_a, out_b multiples of 64. Same per-output arithmetic as two
// pcnn_fc_rows_fwd calls (the K loop does not know which layer a column belongs to): same bits, one launch less.
extern "C" int pcnn_fc_rows_split_fwd(const float* x, const float* wt, const float* bias, int rows_capacity, int in_features,
                                      int out_a, int out_b, int relu_a, int relu_b, const int32_t* num_rows_dev,
                                      float* y_a, float* y_b, void* stream_)
{
  PCNN_REQUIRE(rows_capacity >= 0, PCNN_EINVAL, "fc_rows_split: negative row capacity");
  PCNN_REQUIRE(in_features >= 128 && in_features % 64 == 0, PCNN_EINVAL,
               "fc_rows_split: in_features must be a multiple of 64, >= 128 (got %d)", in_features);
  PCNN_REQUIRE(out_a >= 64 && out_a % 64 == 0 && out_b >= 64 && out_b % 64 == 0, PCNN_EINVAL,
               "fc_rows_split: both widths must be multiples of 64 (got %d, %d)", out_a, out_b);
  if (rows_capacity == 0) return PCNN_OK;
  PCNN_REQUIRE(x && wt && bias && y_a && y_b, PCNN_ENULL, "fc_rows_split: NULL pointer");
  PCNN_REQUIRE(aligned16(x) && aligned16(wt) && aligned16(y_a) && aligned16(y_b) && aligned16(bias), PCNN_EINVAL,
               "fc_rows_split: pointers (x, wt, bias, y_a, y_b) must be 16-byte aligned");
  const int N = out_a + out_b;
  PCNN_REQUIRE((long long)rows_capacity * in_features < (1ll << 30) && (long long)N * in_features < (1ll << 30),
               PCNN_EINVAL, "fc_rows_split: operand larger than the 32-bit byte offsets of the kernel");
  hipStream_t stream = (hipStream_t)stream_;
  const int nbm = (rows_capacity + 63) / 64, ncb = N / 64;
  const int tall = (long long)rows_capacity > (long long)N;
  const long long blocks = tall ? (long long)((nbm + 7) / 8) * 8 * ncb : (long long)((ncb + 7) / 8) * 8 * nbm;
  PCNN_LAUNCH(fc_rows_mfma_kernel, dim3((unsigned)blocks, 1), dim3(512), 0, stream, x, wt, bias, (const float*)nullptr, y_a, in_features,
              N, rows_capacity, relu_a ? 1 : 0, num_rows_dev, nbm, ncb, tall, (float*)nullptr, 1, 0, out_a, N, (float*)nullptr, out_a, y_b,
              relu_b ? 1 : 0);
  return check_launch("fc_rows_split_fwd");
}
