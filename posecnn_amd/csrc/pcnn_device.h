// pcnn_device.h — shared device-side helpers for the gfx950 kernels of libposecnn_hip.so.
//
// Numerics contract: every kernel in this library is compiled with
// -ffp-contract=off and uses IEEE-correct f32 divide / sqrt, so that a result is a pure function
// of the reference's expression tree (hough_voting_gpu_op.cu.cc etc.) with one rounding per
// operation.  Fused multiply-adds appear only where written explicitly (__builtin_fmaf) inside
// conservative *filters* whose outcome is re-derived exactly when it could matter.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/posecnn_hip.h"

#define PCNN_WAVE 64
#define PCNN_MAX_CLASSES 64  // per-class state lives in one wave's lanes / small LDS arrays

namespace pcnn {

// ---- error plumbing (host) -------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> PCNN_EHIP
int zero_async(void* p, size_t bytes, hipStream_t stream, const char* what);  // zero-fill kernel (graph-capture safe)

#define PCNN_REQUIRE(cond, status, ...)  \
  do {                                   \
    if (!(cond)) {                       \
      ::pcnn::set_error(__VA_ARGS__);    \
      return (status);                   \
    }                                    \
  } while (0)

// ---- launch + optional per-kernel HIP-event timing (pcnn_profile_*) ---------------------------
extern bool g_profile_on;
void profile_begin(const char* name, hipStream_t stream);
void profile_end(hipStream_t stream);

#define PCNN_LAUNCH(kernel, grid, block, shmem, stream, ...)                      \
  do {                                                                            \
    if (::pcnn::g_profile_on) ::pcnn::profile_begin(#kernel, (stream));           \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__);  \
    if (::pcnn::g_profile_on) ::pcnn::profile_end((stream));                      \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- data exchanged between workgroups INSIDE one launch (split-K partials summed by the last workgroup) -------------
// gfx950 has one L2 per XCD and they are only made coherent at kernel boundaries. Inside a kernel, a value another
// XCD's workgroup must see is stored with agent scope (sc1: written through to the level all XCDs share) and loaded
// with agent scope (sc1: never served from a line another XCD may have overwritten) — no cache-wide write-back /
// invalidate, which a release / acquire FENCE at agent scope would cost (measured in csrc/fc_skinny.hip).
__device__ __forceinline__ void store_coherent(float* p, float v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_coherent(const float* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- exact f32 primitives --------------------------------------------------------------------
// NOTE: HIP's __fsqrt_rn() is __ocml_native_sqrt_f32 (v_sqrt_f32, ~1 ulp) unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined — it is NOT correctly rounded. Plain `/` and
// __builtin_sqrtf are, under -fhip-fp32-correctly-rounded-divide-sqrt (set in the Makefile; the
// gpu tests check both against the host bit for bit).
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }

// Canonical expf: exp evaluated in IEEE double — range reduction by ln2 (hi/lo),
// degree-13 Taylor polynomial in Horner form, scale by 2^k — then ONE rounding to float.  Every
// operation is a basic IEEE double op, so the host restatement of the same sequence agrees bit
// for bit.  (The bits of CUDA's expf, which the reference calls at
// hough_voting_gpu_op.cu.cc:280, are not reproducible off NVIDIA hardware.)
__device__ __forceinline__ float exp_f32(float xf)
{
  if (xf != xf) return xf;
  double x = (double)xf;
  if (x > 130.0) x = 130.0;
  if (x < -150.0) x = -150.0;
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double kd = floor(x * LOG2E + 0.5);
  double r = (x - kd * LN2_HI) - kd * LN2_LO;
  double p = 1.6059043836821614599e-10;
  p = p * r + 2.0876756987868098979e-09;
  p = p * r + 2.5052108385441718775e-08;
  p = p * r + 2.7557319223985890653e-07;
  p = p * r + 2.7557319223985892511e-06;
  p = p * r + 2.4801587301587301566e-05;
  p = p * r + 1.9841269841269841253e-04;
  p = p * r + 1.3888888888888889419e-03;
  p = p * r + 8.3333333333333332177e-03;
  p = p * r + 4.1666666666666664354e-02;
  p = p * r + 1.6666666666666665741e-01;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  int k = (int)kd;
  double two_k = __longlong_as_double((long long)(k + 1023) << 52);
  return (float)(p * two_k);
}

// Canonical exp of the SOFTMAX layers (network.py:474-488: tf.exp(x - max)), all-f32: TF's / cuDNN's
// bits are unknowable, so any fixed IEEE sequence is as canonical as another — this one costs ~25 f32
// VALU operations instead of the ~30 f64 ones of exp_f32 (the label-head epilogue evaluates 22 per
// pixel and was ALU-bound on it). Every step is one IEEE f32 operation with one rounding (no FMA:
// the library is built -ffp-contract=off), so the C oracle and the numpy restatement reproduce it bit
// for bit:  k = rint(x log2e);  r = (x - k ln2_hi) - k ln2_lo  (ln2_hi has 15 significant bits: k ln2_hi
// is exact);  p = Horner degree 7 (1/n!);  result = p 2^k, scaled in two exact steps below 2^-126.
// Relative error <= 2e-7 (the truncation r^8/8! is 5e-9).
__device__ __forceinline__ float exp_softmax_f32(float x)
{
  if (x != x) return x;
  x = fminf(fmaxf(x, -104.f), 88.f);
  const float kf = __builtin_rintf(x * 1.44269502f);
  float r = x - kf * 0.693145752f;
  r = r - kf * 1.42860677e-06f;
  float p = 1.98412698e-04f;
  p = p * r + 1.38888889e-03f;
  p = p * r + 8.33333377e-03f;
  p = p * r + 4.16666679e-02f;
  p = p * r + 1.66666672e-01f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  const int k = (int)kf;
  if (k < -126) return (p * __int_as_float((k + 64 + 127) << 23)) * 5.42101086e-20f;   // 2^-64
  return p * __int_as_float((k + 127) << 23);
}

// float -> int as the reference's `int v = round(x)` behaves on the GPU: saturating, NaN -> 0.
__device__ __forceinline__ int round_to_int_sat(float x)
{
  float r = roundf(x);
  if (r != r) return 0;
  if (r >= 2147483648.f) return INT32_MAX;
  if (r <= -2147483648.f) return INT32_MIN;
  return (int)r;
}

// ---- wave helpers (wave = 64 lanes on gfx950) --------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long lanemask_lt()
{
  return (1ull << lane_id()) - 1ull;
}

}  // namespace pcnn
