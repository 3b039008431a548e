// tools/valu_rate_probe.hip — what does one vector f32 instruction cost a gfx950 SIMD? Sixteen independent accumulators per
// lane, one instruction kind per kernel, 1 / 2 / 4 / 8 waves per SIMD (single-wave workgroups); cycles per wave-instruction of wave 0 (s_memtime) and per SIMD over
// the whole launch (events).
// Not part of the library. Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_rate_probe.hip -o tools/valu_rate_probe && tools/valu_rate_probe
#include <cstdio>
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* cyc, int iters, float b, float c)
{
  float a[16];
  v2f p[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = (v2f){a[i], a[i] + 0.5f}; }
  v2f bb = (v2f){b, b}, cc = (v2f){c, c};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 1) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bb));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 4) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bb));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 5) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(bb), "v"(cc));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 6) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 7) {   // one dependent chain
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 8) {   // sub (VOP2 with a VGPR second source, like ex = x1 - qx)
#define X(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    } else if (KIND == 9) {   // compare + select pair (the nearest-neighbour walk's chain)
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0]) : "v"(a[i]) : "vcc");
      REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* cyc)
{
  const int iters = 2000;
  printf("%-34s", name);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4, 8, 16, 32}) {   // waves per CU, as single-wave workgroups (the dispatcher spreads them over the four SIMDs)
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256 * wpc), dim3(64), 0, 0, out, cyc, iters, 1.0000001f, 1e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256 * wpc), dim3(64), 0, 0, out, cyc, iters, 1.0000001f, 1e-9f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // the whole launch: wave-instructions per second per SIMD -> cycles per wave-instruction per SIMD at 2.4 GHz
    const double winstr = 256.0 * wpc * iters * 64.0;
    printf("  %2d w/CU: wave0 %5.2f cyc/instr, launch %5.2f cyc/instr/SIMD", wpc, (double)c / (iters * 64.0),
           ms * 1e-3 * 2.4e9 / (winstr / 1024.0));
  }
  printf("\n");
}

int main()
{
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024 * 4); hipMalloc(&cyc, 8);
  run<0>("v_add_f32", out, cyc);
  run<8>("v_sub_f32", out, cyc);
  run<1>("v_mul_f32", out, cyc);
  run<2>("v_fma_f32", out, cyc);
  run<6>("v_min_f32", out, cyc);
  run<3>("v_pk_add_f32", out, cyc);
  run<4>("v_pk_mul_f32", out, cyc);
  run<5>("v_pk_fma_f32", out, cyc);
  run<7>("v_add_f32, one dependent chain", out, cyc);
  run<9>("v_cmp_lt + v_cndmask chain (x2)", out, cyc);
  return 0;
}
