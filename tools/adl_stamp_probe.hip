// tools/adl_stamp_probe.hip — where does one row's time go inside adl_terms_kernel and adl_sum_kernel? Thread 0 of workgroup
// (0,0) writes the 100 MHz wall clock and the shader clock at each stage (ADL_STAMP in average_distance.hip, empty in the
// library). One live row of a symmetric class, P = 2620, 3024-row capacity, like tools/probe_adl.py's "1 live" cases.
// Not part of the library. Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -Iposecnn_amd/csrc -Iinclude tools/adl_stamp_probe.hip posecnn_amd/csrc/common.hip -o tools/adl_stamp_probe
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <hip/hip_runtime.h>

__device__ unsigned long long g_wall[32], g_clk[32];
#define ADL_STAMP(k)                                                              \
  do {                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                 \
      g_wall[k] = wall_clock64();                                                 \
      g_clk[k] = clock64();                                                       \
    }                                                                             \
  } while (0)

#include "../posecnn_amd/csrc/average_distance.hip"

int main(int argc, char** argv)
{
  const int C = 22, P = 2620, CAP = 3024;
  const int sym = argc > 1 ? atoi(argv[1]) : 1;
  std::vector<float> w((size_t)CAP * 4 * C, 0.f), t(w), p(w), pts((size_t)C * P * 3), sy(C, 0.f);
  srand(1);
  for (auto& v : pts) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  const int cls = 16;
  sy[cls] = sym ? 1.f : 0.f;
  float q1[4] = {0.5f, 0.5f, 0.5f, 0.5f}, q2[4] = {0.1f, 0.7f, -0.3f, 0.64f};
  float n2 = std::sqrt(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3]);
  for (int i = 0; i < 4; i++) { w[4 * cls + i] = 1.f; t[4 * cls + i] = q1[i]; p[4 * cls + i] = q2[i] / n2; }
  float *dw, *dt, *dp, *dpts, *dsy, *loss, *bd; void* ws; int* cnt;
  size_t wsb = 0;
  pcnn_average_distance_workspace_bytes(CAP, C, P, &wsb);
  hipMalloc(&dw, w.size() * 4); hipMalloc(&dt, w.size() * 4); hipMalloc(&dp, w.size() * 4);
  hipMalloc(&dpts, pts.size() * 4); hipMalloc(&dsy, C * 4); hipMalloc(&loss, 4); hipMalloc(&bd, w.size() * 4);
  hipMalloc(&ws, wsb); hipMalloc(&cnt, 4);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, t.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dp, p.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dpts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsy, sy.data(), C * 4, hipMemcpyHostToDevice);
  const int one = 1;
  hipMemcpy(cnt, &one, 4, hipMemcpyHostToDevice);
  for (int it = 0; it < 4; it++) {
    int rc = pcnn_average_distance_fwd(dp, dt, dw, dpts, dsy, CAP, C, P, 0.01f, cnt, loss, bd, ws, wsb, nullptr);
    if (rc) { printf("rc=%d %s\n", rc, pcnn_last_error_string()); return 1; }
    hipDeviceSynchronize();
  }
  unsigned long long wl[32], ck[32];
  hipMemcpyFromSymbol(wl, HIP_SYMBOL(g_wall), sizeof(wl)); hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_clk), sizeof(ck));
  float l; hipMemcpy(&l, loss, 4, hipMemcpyDeviceToHost);
  printf("symmetric=%d loss=%.9g\n", sym, l);
  const char* nm[32] = {};
  nm[0] = "sum: start"; nm[1] = "sum: class found"; nm[2] = "sum: row staged in LDS"; nm[3] = "sum: chains done";
  nm[10] = "terms: start"; nm[11] = "terms: row header"; nm[12] = "terms: own point rotated"; nm[13] = "terms: first tile in LDS";
  nm[14] = "terms: first tile scanned"; nm[15] = "terms: all tiles scanned"; nm[16] = "terms: in-trip walk done"; nm[17] = "terms: stored";
  for (int base : {10, 0})
    for (int k = base; k < base + 8; k++) {
      if (!nm[k] || !wl[k]) continue;
      printf("%-28s wall %8.2f us   shader clock %9llu cycles\n", nm[k], (wl[k] - wl[base]) * 0.01, ck[k] - ck[base]);
    }
  return 0;
}
