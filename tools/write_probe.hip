// write_probe.hip — what does the LAYOUT of the Winograd-domain tensor V cost the kernels that write it?
//   hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o tools/write_probe
// conv3x3_c3_wino43_kernel / wino43_input_kernel write, per workgroup, 36 chunks (one per transform plane) of
// (tiles x channels) floats. Plane-major V ([36][T][C], what the library used through round 3) puts those chunks
// T*C*4 bytes apart (157 MB at conv1_2); a block-major V ([T/32][36][32][C]) keeps them 32*C*4 bytes apart (8 KB).
// This probe issues exactly the store instructions of the c3 kernel's phase 2 (192 threads, 12 dwordx4 stores each,
// 4 tiles x 64 channels per workgroup) with nothing else in the kernel, in both layouts, next to a linear stream of
// the same volume.
//   mode 0: plane-major   mode 1: block-major (32-tile blocks)   mode 2: linear (each workgroup one 36 KB run)
//   mode 3: plane-major, 8 tiles per workgroup (2 KB chunks)     mode 4: block-major, 8 tiles per workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* __restrict__ v, long long T, int C, float seed)
{
  constexpr int TPW = (MODE == 3 || MODE == 4) ? 8 : 4;     // tiles per workgroup
  const int tid = threadIdx.x, wave = tid >> 6;
  if (wave >= 3) return;
  const long long plane = T * C;
  const f4 val = {seed, seed + 1.f, seed + 2.f, seed + 3.f};
#pragma unroll
  for (int rep = 0; rep < TPW / 4; rep++) {
    const int t = (tid & 63) >> 4, q = tid & 15, pr = wave;
    const long long tile = (long long)blockIdx.x * TPW + rep * 4 + t;
    if (tile >= T) return;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int k = 6 * (2 * pr + i) + j;
        long long off;
        if (MODE == 0 || MODE == 3) off = k * plane + tile * C + q * 4;
        else if (MODE == 1 || MODE == 4) off = (((tile >> 5) * 36 + k) * 32 + (tile & 31)) * C + q * 4;
        else off = ((long long)blockIdx.x * 36 + k) * (4 * C) + (tid & 63) * 4;
        *reinterpret_cast<f4*>(v + off) = val;
      }
  }
}

int main(int argc, char** argv)
{
  const long long T = argc > 1 ? atoll(argv[1]) : 614400;   // conv1_2, two towers x 16 frames
  const int C = 64;
  const size_t bytes = (size_t)36 * T * C * 4;
  float* v;
  if (hipMalloc(&v, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(v, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[5] = {"plane-major 4 tiles/wg", "block-major 4 tiles/wg", "linear", "plane-major 8 tiles/wg", "block-major 8 tiles/wg"};
  for (int mode = 0; mode < 5; mode++) {
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int r = 0; r < reps + 2; r++) {
      const unsigned grid = (unsigned)((mode >= 3) ? (T + 7) / 8 : (T + 3) / 4);
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, v, T, C, (float)r); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, v, T, C, (float)r); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, v, T, C, (float)r); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, v, T, C, (float)r); break;
        default: hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, v, T, C, (float)r); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("{\"mode\": \"%s\", \"GB\": %.3f, \"avg_ms\": %.4f, \"best_ms\": %.4f, \"avg_TBps\": %.3f}\n", names[mode], bytes / 1e9, sum / reps,
           best, bytes / (sum / reps) / 1e9);
  }
  hipFree(v);
  return 0;
}
