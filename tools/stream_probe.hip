// stream_probe.hip — how fast can 2048 waves pull a [4096][25088] f32 matrix (fc6's weights, 411 MB) out of HBM,
// as a function of what one load instruction covers?   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe
//   pattern 0: 16 rows x 64 B per instruction (the MFMA 16x16x4 operand layout of fc_mfma / fc_skinny)
//   pattern 1:  8 rows x 128 B per instruction (full cache lines; the 4x4x1-MFMA layout)
//   pattern 2:  1 row x 1 KB per instruction (GEMV layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int PAT, int UNR>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ w, float* __restrict__ out, int K, int N, int rows_per_wave, int S)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * rows_per_wave;
  const int s = blockIdx.y;
  const int kper = K / S;            // floats per slice
  const int k0 = s * kper;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (PAT == 0) {          // lane (r = lane & 15, q = lane >> 4): row r, 16 B at k + 4 q; step 16 floats
    for (int nb = 0; nb < rows_per_wave; nb += 32) {
      const float* p0 = w + (size_t)(n0 + nb + (lane & 15)) * K + k0 + 4 * (lane >> 4);
      const float* p1 = p0 + (size_t)16 * K;
      for (int k = 0; k < kper; k += 16 * UNR) {
        v4f t[2 * UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { t[2 * u] = *(const v4f*)(p0 + k + 16 * u); t[2 * u + 1] = *(const v4f*)(p1 + k + 16 * u); }
#pragma unroll
        for (int u = 0; u < 2 * UNR; u++) acc += t[u];
      }
    }
  } else if (PAT == 1) {   // lane: row = lane >> 3 (8 rows), 16 B at k + 4 (lane & 7); step 32 floats
    for (int nb = 0; nb < rows_per_wave; nb += 32) {
      const float* p0 = w + (size_t)(n0 + nb + (lane >> 3)) * K + k0 + 4 * (lane & 7);
      for (int k = 0; k < kper; k += 32 * UNR) {
        v4f t[4 * UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
          for (int j = 0; j < 4; j++) t[4 * u + j] = *(const v4f*)(p0 + (size_t)(8 * j) * K + k + 32 * u);
#pragma unroll
        for (int u = 0; u < 4 * UNR; u++) acc += t[u];
      }
    }
  } else {                 // lane: 16 B at k + 4 lane of ONE row; step 256 floats; rows one after the other
    for (int nb = 0; nb < rows_per_wave; nb += 4 * UNR) {
      const float* p0 = w + (size_t)(n0 + nb) * K + k0 + 4 * lane;
      for (int k = 0; k < kper; k += 256) {
        v4f t[4 * UNR];
#pragma unroll
        for (int u = 0; u < 4 * UNR; u++) t[u] = *(const v4f*)(p0 + (size_t)u * K + k);
#pragma unroll
        for (int u = 0; u < 4 * UNR; u++) acc += t[u];
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

int main()
{
  const int N = 4096, K = 25088;
  float *w, *out;
  hipMalloc(&w, sizeof(float) * (size_t)N * K);
  hipMalloc(&out, 4096);
  hipMemset(w, 0, sizeof(float) * (size_t)N * K);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int rows_per_wave, int S) {
    dim3 grid(N / (4 * rows_per_wave), S);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, w, out, K, N, rows_per_wave, S);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, w, out, K, N, rows_per_wave, S);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s rows/wave %3d S %3d waves %5d : %7.1f us  %6.0f GB/s\n", name, rows_per_wave, S, grid.x * grid.y * 4, ms * 100, 4.0 * N * K / (ms / 10 * 1e-3) / 1e9);
  };
  // K = 25088 = 98 * 256: slices must be multiples of the pattern's step x unroll
  run("16 rows x 64 B, unroll 4", probe<0, 4>, 32, 14);    // kper 1792 = 28 x 64
  run("16 rows x 64 B, unroll 4", probe<0, 4>, 32, 7);
  run("16 rows x 64 B, unroll 4", probe<0, 4>, 64, 14);
  run("16 rows x 64 B, unroll 7", probe<0, 7>, 32, 14);
  run("8 rows x 128 B, unroll 2", probe<1, 2>, 32, 14);
  run("8 rows x 128 B, unroll 2", probe<1, 2>, 32, 7);
  run("8 rows x 128 B, unroll 4", probe<1, 4>, 32, 14);
  run("8 rows x 128 B, unroll 2", probe<1, 2>, 64, 14);
  run("1 row x 1 KB, 8 rows in flight", probe<2, 2>, 32, 14);
  run("1 row x 1 KB, 8 rows in flight", probe<2, 2>, 32, 7);
  run("1 row x 1 KB, 16 rows in flight", probe<2, 4>, 32, 14);
  run("1 row x 1 KB, 16 rows in flight", probe<2, 4>, 64, 7);
  run("1 row x 1 KB, 16 rows in flight", probe<2, 4>, 16, 14);
  return 0;
}
