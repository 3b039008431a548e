// fc_skinny.hip — fc6 / fc7 / fc8 (`Network.fc`, lib/networks/network.py:392-422; vgg16_convs.py:188-192) when
// the pose branch has only a handful of ROI rows: the single-frame loop of lib/fcn/test.py:1867-1888 (one image,
// <= 21 detections; BASELINE configs[1]).
//
// At <= 32 rows the layer is a weight STREAM, not a GEMM: fc6 reads 411 MB of weights to produce 5 x 4096 numbers.
// csrc/fc_mfma.hip (64 x 64 blocks through an LDS ring, built for hundreds of rows) reaches 3.2 TB/s there with
// split-K plus a separate reduction launch. This kernel is built for the stream:
//
//   * no LDS, no barriers in the main loop: a wave owns (32 output columns) x (a slice of K) and pulls its weight
//     rows straight from HBM into registers (global_load_dwordx4 ... nt — the stream must not evict the
//     activations from L2), two unrolled groups of 4 K steps in flight (software-pipelined in registers);
//   * the rows ride along as the A operand of v_mfma_f32_16x16x4_f32 (<= 2 blocks of 16 rows, the second skipped
//     when the device-side count is <= 16): the K reduction happens inside the matrix core, so there is no
//     cross-lane reduction and the useless rows cost nothing that matters (21 us of MFMA time for fc6 at full rate,
//     under an 85 us stream). x (<= 32 x K floats) is L2-resident and shared through L1 by the 4 waves of a
//     workgroup, which work on the same K slice;
//   * split-K over grid.y fills the chip (2048+ waves, ~20 KB in flight each); the partial products meet in a
//     fixed-order sum performed by the LAST workgroup of each column group (ticket counter; partials exchanged
//     through agent-scope atomic stores / loads; the counter returns to zero) — deterministic, one launch;
//   * epilogue in that last workgroup: + bias, then none / ReLU / tanh (fc8 -> poses_tanh, vgg16_convs.py:192-193:
//     the linear output is kept as well), rows at or past the device-side count are written as zeros.
//
//   y[m, n] = act(sum_k x[m, k] wt[n, k] + bias[n])    m < min(Mcap, *num_rows_dev), else 0
#include <algorithm>
#include <cstdlib>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int SK_NBW = 2;          // 16-column blocks per wave
constexpr int SK_WAVES = 4;        // waves per workgroup (same K slice, consecutive column blocks)
constexpr int SK_COLS = 16 * SK_NBW * SK_WAVES;   // 128 output columns per workgroup

// (a non-temporal hint on this load measured 2.0 TB/s against 3+ without: a wave's load instruction covers 64 of a
//  line's 128 bytes, the next K step the other 64 — streamed lines were fetched from HBM twice)
__device__ __forceinline__ v4f ldg_w(const float* p)
{
  return *reinterpret_cast<const v4f*>(p);
}

// The K loop of one wave over its slice [ks0, ks1) of 16-float steps, for ML live row blocks: two groups of SK_U steps
// in flight (registers), MFMAs of group g under the loads of group g + 1. A clamped step (past the slice end) is loaded
// again and its MFMAs are skipped — at most SK_U - 1 redundant loads per wave instead of a remainder loop.
template <int ML, int SK_U>
__device__ __forceinline__ void sk_loop(const float* const* xp, const float* const* wp, int ks0, int ks1, v4f (*acc)[SK_NBW])
{
  v4f a[2][SK_U][ML], b[2][SK_U][SK_NBW];
  const int ngroups = (ks1 - ks0 + SK_U - 1) / SK_U;
  if (ngroups <= 0) return;
#define SK_LOAD(BUF, G)                                                                     \
  _Pragma("unroll") for (int u = 0; u < SK_U; u++) {                                         \
    const int st = min(ks0 + (G) * SK_U + u, ks1 - 1);                                       \
    _Pragma("unroll") for (int j = 0; j < SK_NBW; j++) b[BUF][u][j] = ldg_w(wp[j] + 16 * st); \
    _Pragma("unroll") for (int mb = 0; mb < ML; mb++) a[BUF][u][mb] = *reinterpret_cast<const v4f*>(xp[mb] + 16 * st); \
  }
#define SK_MATH(BUF, G)                                                                     \
  _Pragma("unroll") for (int u = 0; u < SK_U; u++) {                                         \
    if (ks0 + (G) * SK_U + u < ks1) {   /* wave-uniform */                                   \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                          \
        _Pragma("unroll") for (int j = 0; j < SK_NBW; j++)                                   \
          _Pragma("unroll") for (int mb = 0; mb < ML; mb++)                                  \
            acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[BUF][u][mb][i], b[BUF][u][j][i], acc[mb][j], 0, 0, 0); \
    }                                                                                       \
  }
  SK_LOAD(0, 0);
  int g = 0;
  for (; g + 2 <= ngroups - 1; g += 2) {
    SK_LOAD(1, g + 1);
    SK_MATH(0, g);
    SK_LOAD(0, g + 2);
    SK_MATH(1, g + 1);
  }
  // 1 or 2 groups left: g (in buffer 0) and possibly g + 1
  if (g + 1 < ngroups) {
    SK_LOAD(1, g + 1);
    SK_MATH(0, g);
    SK_MATH(1, g + 1);
  } else {
    SK_MATH(0, g);
  }
#undef SK_LOAD
#undef SK_MATH
}

// MB = 16-row blocks of x the buffer can hold (1 or 2). act: 0 none, 1 ReLU, 2 tanh (y2 = tanh(y), y = linear)
// FENCED (debug, PCNN_FC_SKINNY_FENCED=1 in the environment): the textbook exchange — plain stores, an agent-scope
// release fence before an acq_rel ticket, an acquire fence in the reducer — in place of the fence-free sc1 exchange
// below. Twice as slow (every fence writes back / invalidates the XCD's L2); it exists so that the two can be diffed
// (tests/test_gpu_round3.py::test_fc_skinny_fence_free_exchange_equals_the_fenced_one).
template <int MB, int SK_U, bool FENCED>
__global__ __launch_bounds__(64 * SK_WAVES) void fc_skinny_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ y2, int K, int N, int Mcap, int act, const int* __restrict__ num_rows_dev,
    float* __restrict__ part, int* __restrict__ counters, int steps_per_slice)
{
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int count = num_rows_dev ? min(Mcap, max(num_rows_dev[0], 0)) : Mcap;
  const int S = gridDim.y, s = blockIdx.y;
  const int Npad = gridDim.x * SK_COLS;
  const int n0 = blockIdx.x * SK_COLS + wave * (16 * SK_NBW);
  const int KS = K >> 4;
  const int ks0 = min(KS, s * steps_per_slice), ks1 = min(KS, ks0 + steps_per_slice);
  const bool two = MB > 1 && count > 16;     // wave-uniform: the second row block exists

  // per-lane row pointers. Rows past the LIVE count are clamped to the last live row (their products are never
  // stored): the lanes of the dead rows then share one address, and a frame with 5 detections in a 21-row buffer
  // pulls 5 rows of x through L2 instead of 16
  const int last_row = max(count, 1) - 1;
  const float* xp[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++) xp[mb] = x + (size_t)min(16 * mb + r, last_row) * K + 4 * q;
  const float* wp[SK_NBW];
#pragma unroll
  for (int j = 0; j < SK_NBW; j++) wp[j] = wt + (size_t)min(n0 + 16 * j + r, N - 1) * K + 4 * q;

  v4f acc[MB][SK_NBW];
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int j = 0; j < SK_NBW; j++) acc[mb][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  if (count > 0) {
    // the loop is specialised on the LIVE row blocks: a frame with <= 16 detections in a 21-row buffer runs the
    // one-block loop (branch-free bodies, half the operand registers in use)
    if (MB > 1 && two) sk_loop<MB, SK_U>(xp, wp, ks0, ks1, acc);
    else sk_loop<1, SK_U>(xp, wp, ks0, ks1, acc);
  }

  // partial products: lane holds D[m = 4 q + e][n = r] of each block
  if (count > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
      if (mb > 0 && !two) break;
#pragma unroll
      for (int j = 0; j < SK_NBW; j++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if constexpr (FENCED) part[((size_t)s * (16 * MB) + 16 * mb + 4 * q + e) * Npad + n0 + 16 * j + r] = acc[mb][j][e];
          else __hip_atomic_store(&part[((size_t)s * (16 * MB) + 16 * mb + 4 * q + e) * Npad + n0 + 16 * j + r], acc[mb][j][e],
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ticket: the last workgroup of this column group sums the S partials in ascending order
  // Coherence across the 8 XCDs (one L2 each) without cache-wide operations. A release fence at agent scope writes
  // the XCD's whole L2 back, an acquire invalidates it: one __threadfence() per thread (first version) meant ~2000
  // L2 invalidations during the launch and HALVED the stream rate (fc6 205 us); one release per workgroup still cost
  // ~0.3 us each (119 us). The partials are therefore written with agent-scope atomic stores (sc1: written through to
  // the coherence point) and read back with agent-scope atomic loads (sc1: never served from a stale line); the
  // wait + barrier orders every wave's stores before the ticket. No fence, no invalidate.
  // ASSUMPTIONS: (1) gfx942 / gfx950 cache policy — an agent-scope atomic store is written through to the
  // coherence point and an agent-scope atomic load is never served from a stale L2 line (the Makefile refuses any other
  // --offload-arch); (2) the inline-asm wait is a compiler barrier ("memory" clobber), so no store moves below it;
  // (3) the counters are zero on entry: the last workgroup puts its counter back to zero, every launch on a stream
  // therefore leaves them as it found them; the launcher zero-fills them if a launch is refused, and after a device fault
  // (the only way a launch ends early) the HIP context is gone with them.
  __shared__ int s_last;
  if constexpr (FENCED) __threadfence();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int old = FENCED ? __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                           : __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == S - 1;
    if (old == S - 1) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the counter as we found it
  }
  __syncthreads();
  if (!s_last) return;
  if constexpr (FENCED) __threadfence();
  // 256 threads over (column, row parity): thread handles column c = tid & 127, rows m = h, h + 2, ...
  const int c = tid & (SK_COLS - 1), h = tid / SK_COLS;
  const int n = blockIdx.x * SK_COLS + c;
  if (n >= N) return;
  const float bv = bias[n];
  const int rows = 16 * MB;
  // two rows at a time, the partials fetched 16 per row in one batch (32 independent loads in flight per thread: the
  // partials come from other XCDs' workgroups, i.e. from memory, ~1 us each if taken one by one); the SUM stays
  // sequential in ascending slice order
  for (int m0 = h; m0 < Mcap; m0 += 4) {
    float sum[2] = {0.f, 0.f};
    const int m1 = m0 + 2;
    const bool l0 = m0 < count, l1 = m1 < count && m1 < Mcap;
    if (l0) {
      for (int ks0 = 0; ks0 < S; ks0 += 16) {
        float p0[16], p1[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int ks = min(ks0 + u, S - 1);
          p0[u] = __hip_atomic_load(&part[((size_t)ks * rows + m0) * Npad + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          p1[u] = l1 ? __hip_atomic_load(&part[((size_t)ks * rows + m1) * Npad + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; u++)
          if (ks0 + u < S) { sum[0] += p0[u]; sum[1] += p1[u]; }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int m = m0 + 2 * t;
      if (m >= Mcap) break;
      float v = 0.f, v2 = 0.f;
      if (m < count) {
        v = sum[t] + bv;
        if (act == 1) v = v > 0.f ? v : 0.f;
        if (act == 2) v2 = tanhf(v);
      }
      y[(size_t)m * N + n] = v;
      if (act == 2 && y2) y2[(size_t)m * N + n] = v2;
    }
  }
}

// split: enough waves to fill the chip with deep load queues, at least 16 K steps (256 floats) per slice
int sk_splits(int K, int N)
{
  const int KS = K / 16;
  const int groups = (N + SK_COLS - 1) / SK_COLS;
  // ONE 4-wave workgroup per CU: the launch is a single round of equal workgroups, so any count that is not a
  // multiple of the 256 CUs leaves CUs idle or gives some of them two (measured on fc6: 256 workgroups 82 us,
  // 192 -> 100, 320 -> 131, 512 -> 90, 1024 -> 98, 128 -> 131; 256 workgroups of 8 waves: 88)
  int S = std::max(1, 256 / groups);
  S = std::min(S, std::max(1, KS / 16));
  return std::min(S, 64);
}

}  // namespace

extern "C" int pcnn_fc_skinny_workspace_bytes(int rows_capacity, int in_features, int out_features, size_t* bytes,
                                              int* num_counters)
{
  PCNN_REQUIRE(bytes && num_counters, PCNN_ENULL, "fc_skinny_workspace_bytes: NULL output");
  PCNN_REQUIRE(rows_capacity >= 1 && rows_capacity <= 32 && in_features >= 16 && in_features % 16 == 0 && out_features >= 1,
               PCNN_EINVAL, "fc_skinny_workspace_bytes: rows 1..32, in_features a multiple of 16 (got %d x %d -> %d)",
               rows_capacity, in_features, out_features);
  const int groups = (out_features + SK_COLS - 1) / SK_COLS;
  const int MB = rows_capacity > 16 ? 2 : 1;
  *bytes = sizeof(float) * (size_t)sk_splits(in_features, out_features) * 16 * MB * groups * SK_COLS;
  *num_counters = groups;
  return PCNN_OK;
}

extern "C" int pcnn_fc_skinny_fwd(const float* x, const float* wt, const float* bias, int rows_capacity, int in_features,
                                  int out_features, int activation, const int32_t* num_rows_dev, float* y, float* y_act,
                                  void* workspace, size_t workspace_bytes, int32_t* counters, int num_counters, void* stream_)
{
  PCNN_REQUIRE(rows_capacity >= 1 && rows_capacity <= 32, PCNN_EINVAL, "fc_skinny: 1..32 rows (got %d); more rows go to pcnn_fc_rows_fwd", rows_capacity);
  PCNN_REQUIRE(in_features >= 16 && in_features % 16 == 0, PCNN_EINVAL, "fc_skinny: in_features must be a multiple of 16 (got %d)", in_features);
  PCNN_REQUIRE(out_features >= 1, PCNN_EINVAL, "fc_skinny: out_features must be positive");
  PCNN_REQUIRE(activation >= 0 && activation <= 2, PCNN_EINVAL, "fc_skinny: activation 0 (none), 1 (ReLU) or 2 (tanh)");
  PCNN_REQUIRE(x && wt && bias && y && workspace && counters, PCNN_ENULL, "fc_skinny: NULL pointer");
  PCNN_REQUIRE(aligned16(x) && aligned16(wt), PCNN_EINVAL, "fc_skinny: x and wt must be 16-byte aligned");
  size_t need = 0;
  int groups = 0;
  pcnn_fc_skinny_workspace_bytes(rows_capacity, in_features, out_features, &need, &groups);
  PCNN_REQUIRE(workspace_bytes >= need && num_counters >= groups, PCNN_EWORKSPACE,
               "fc_skinny: workspace %zu bytes / %d counters, need %zu / %d", workspace_bytes, num_counters, need, groups);
  hipStream_t stream = (hipStream_t)stream_;
  const int S = sk_splits(in_features, out_features);
  const int KS = in_features / 16;
  const int per = (KS + S - 1) / S;
  float* part = static_cast<float*>(workspace);
  static const bool fenced = [] { const char* e = getenv("PCNN_FC_SKINNY_FENCED"); return e && e[0] == '1'; }();
#define SK_GO(MBV, FV) PCNN_LAUNCH((fc_skinny_kernel<MBV, 2, FV>), dim3(groups, S), dim3(64 * SK_WAVES), 0, stream, x, wt, bias, y, y_act, in_features, \
                out_features, rows_capacity, activation, num_rows_dev, part, counters, per)
  // (groups of 2 K steps: 66 / 80 VGPRs; groups of 4 need 114 / 140 and measured 5-10 % slower)
  if (fenced) { if (rows_capacity > 16) SK_GO(2, true); else SK_GO(1, true); }
  else { if (rows_capacity > 16) SK_GO(2, false); else SK_GO(1, false); }
#undef SK_GO
  const int rc = check_launch("fc_skinny_fwd");
  if (rc != PCNN_OK) (void)hipMemsetAsync(counters, 0, sizeof(int32_t) * (size_t)groups, stream);   // a refused launch must not leave a ticket half-drawn
  return rc;
}
