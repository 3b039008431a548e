// tools/mfma_bare.hip — what the fp32 matrix pipe of a gfx950 SIMD sustains with NOTHING else in the loop: dependent chains
// of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on 1, 2 or 4 accumulators, one or two waves per SIMD, with and without an
// s_nop between the MFMAs (the assembler puts one between adjacent MFMAs of inline asm). The ceilings the trunk kernels
// (csrc/wino_mfma.hip) are held against. Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bare.hip -o tools/mfma_bare && tools/mfma_bare
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NACC, int NOP, int FILL>
__global__ __launch_bounds__(512) void bare32(float* out, int iters, float a0, float b0)
{
  v16f acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  int s = threadIdx.x;
  int su = __builtin_amdgcn_readfirstlane(iters);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 32; k++) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[k % NACC]) : "v"(a), "v"(b));
      if (NOP) asm volatile("s_nop 0");
#pragma unroll
      for (int f = 0; f < (FILL & 63); f++) {
        if (FILL & 64) asm volatile("v_add_u32 %0, %0, 1" : "+v"(s));   // VALU fillers
        else if (FILL & 128) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(su));   // SALU fillers
        else asm volatile("s_nop 0");
      }
    }
  }
  float r = s + su;
#pragma unroll
  for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][7];
  if (r == 1234.5f) out[threadIdx.x] = r;
}

template <int NACC>
__global__ __launch_bounds__(512) void bare16(float* out, int iters, float a0, float b0)
{
  v4f acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 32; k++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[k % NACC]) : "v"(a), "v"(b));
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][3];
  if (r == 1234.5f) out[threadIdx.x] = r;
}


// fillers per group of 8 MFMAs: NR ds_read_b128 (waited for at the end of the group, like an operand prefetch one group
// ahead) and ND global_load_lds_dwordx4 (waited for 16 MFMAs later), spread behind the MFMAs one per shadow
template <int NR, int ND>
__global__ __launch_bounds__(256) void bare32_mem(float* out, const float* src, int iters, float a0, float b0)
{
  __shared__ __attribute__((aligned(16))) float lds[32768];   // 128 KB
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  v4f rd[8];
#pragma unroll
  for (int i = 0; i < 8; i++) rd[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
  const unsigned laddr = lbase + (threadIdx.x & 255) * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ldst = lbase + 65536 + wave * 8192;
  const unsigned voff = (threadIdx.x & 63) * 16;
  const char* sp = reinterpret_cast<const char*>(src) + (size_t)(blockIdx.x & 255) * 65536;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (k < NR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[k]) : "v"(laddr), "n"(4096 * (k & 7)));
        else if (k - NR < ND && k >= NR) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sp), "s"(ldst + 1024 * (k & 7)) : "memory", "m0");
      }
      if (NR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ND) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND) : "memory");
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) r += rd[i][0];
  r += acc[0] + acc[7];
  if (r == 1234.5f) out[threadIdx.x] = r;
}


// the small MFMA the one-wave kernel folds with: v_mfma_f32_4x4x1_16b_f32 (64 lanes x 4 FMAs), NACC independent accumulators
template <int NACC>
__global__ __launch_bounds__(256) void bare4(float* out, int iters, float a0, float b0)
{
  v4f acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 32; k++) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k % NACC]) : "v"(a), "v"(b));
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][3];
  if (r == 1234.5f) out[threadIdx.x] = r;
}

template <typename F>
static double timeit(F launch)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; i++) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main()
{
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 4000, grid = 256 * 8;   // 8 rounds of one workgroup per CU
  for (int w = 0; w < 200; w++) hipLaunchKernelGGL((bare32<4, 0, 0>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);   // clocks up
  hipDeviceSynchronize();
#define R32(NACC, NOP, FILL, THREADS) { const double ms = timeit([&] { hipLaunchKernelGGL((bare32<NACC, NOP, FILL>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.f, 2.f); }); \
    const double fl = 2.0 * 32 * 32 * 2 * 32.0 * iters * (THREADS / 64) * grid; \
    printf("32x32x2  acc %d  nop %d  fill %2d  waves/SIMD %d : %8.3f ms  %6.1f TF\n", NACC, NOP, FILL, THREADS / 256, ms, fl / ms / 1e9); }
#define R16(NACC, THREADS) { const double ms = timeit([&] { hipLaunchKernelGGL((bare16<NACC>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.f, 2.f); }); \
    const double fl = 2.0 * 16 * 16 * 4 * 32.0 * iters * (THREADS / 64) * grid; \
    printf("16x16x4  acc %d                  waves/SIMD %d : %8.3f ms  %6.1f TF\n", NACC, THREADS / 256, ms, fl / ms / 1e9); }
  R32(1, 0, 0, 256) R32(2, 0, 0, 256) R32(4, 0, 0, 256)
  R32(1, 1, 0, 256) R32(4, 1, 0, 256)
  R32(1, 0, 4, 256) R32(1, 0, 8, 256) R32(1, 0, 12, 256) R32(1, 0, 14, 256) R32(1, 0, 16, 256) R32(1, 0, 20, 256)
  R32(1, 0, 64 + 4, 256) R32(1, 0, 64 + 8, 256) R32(1, 0, 64 + 12, 256) R32(1, 0, 64 + 16, 256)
  R32(1, 0, 128 + 4, 256) R32(1, 0, 128 + 8, 256) R32(1, 0, 128 + 12, 256) R32(1, 0, 128 + 16, 256)
  R32(1, 0, 0, 512) R32(1, 0, 64 + 8, 512) R32(1, 0, 64 + 16, 512)
  float* src;
  hipMalloc(&src, 256 * 65536);
  hipMemset(src, 0, 256 * 65536);
#define RM(NR, ND) { const double ms = timeit([&] { hipLaunchKernelGGL((bare32_mem<NR, ND>), dim3(grid), dim3(256), 0, 0, out, src, iters, 1.f, 2.f); }); \
    const double fl = 2.0 * 32 * 32 * 2 * 32.0 * iters * 4 * grid; \
    printf("32x32x2  per 8 MFMAs: %d ds_read_b128 + %d LDS-DMA, one wave/SIMD : %8.3f ms  %6.1f TF\n", NR, ND, ms, fl / ms / 1e9); }
  RM(0, 0) RM(2, 0) RM(4, 0) RM(8, 0) RM(0, 1) RM(0, 2) RM(0, 4) RM(4, 2) RM(4, 4)
#define R4(NACC) { const double ms = timeit([&] { hipLaunchKernelGGL((bare4<NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); }); \
    const double n = 32.0 * iters * 8;   /* instructions per SIMD: 8 rounds of one wave */ \
    printf("4x4x1    acc %d  one wave/SIMD : %8.3f ms  %5.1f cycles per instruction at 2.4 GHz\n", NACC, ms, ms * 1e-3 * 2.4e9 / n); }
  R4(1) R4(2) R4(4) R4(8)
  R16(1, 256) R16(2, 256) R16(4, 256) R16(1, 512) R16(2, 512) R16(4, 512)
  return 0;
}
