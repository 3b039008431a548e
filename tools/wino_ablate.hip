// tools/wino_ablate.hip — "ablate before optimizing" (cdna_hip_programming.md §5.4): the fused Winograd
// MFMA kernel of csrc/wino_mfma.hip with one ingredient of its K loop removed per variant, timed with HIP
// events on synthetic buffers at a trunk layer's shape. Not part of the library. Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iposecnn_amd/csrc -Iinclude \
//         tools/wino_ablate.hip posecnn_amd/csrc/common.hip -o tools/wino_ablate && tools/wino_ablate 9600 512 512
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "../posecnn_amd/csrc/wino_mfma.hip"

template <int ABL, int WR>
static float run(const float* v, const float* ut, const float* bias, float* y, int H, int W, int Cin, int Cout, long long T, int iters)
{
  const int Ht = (H + 3) / 4, Wt = (W + 3) / 4;
  const long long nbt = (T + 32 * WR - 1) / (32 * WR);
  const int ncb = Cout / 64;
  const long long blocks = ((nbt + 7) / 8) * 8 * ncb;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; i++)
    hipLaunchKernelGGL((wino43_mfma_kernel<0, WR, ABL>), dim3((unsigned)blocks), dim3(256 * WR), 0, 0, v, ut, bias, y, (float*)nullptr, H, W, Cin, Cout, Ht, Wt, T, T, 1, (int)nbt, ncb, 1, 0, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; i++)
    hipLaunchKernelGGL((wino43_mfma_kernel<0, WR, ABL>), dim3((unsigned)blocks), dim3(256 * WR), 0, 0, v, ut, bias, y, (float*)nullptr, H, W, Cin, Cout, Ht, Wt, T, T, 1, (int)nbt, ncb, 1, 0, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main(int argc, char** argv)
{
  // image geometry: B images of H x W with T = B * (H/4) * (W/4) tiles
  const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 60, W = argc > 3 ? atoi(argv[3]) : 80;
  const int Cin = argc > 4 ? atoi(argv[4]) : 512, Cout = argc > 5 ? atoi(argv[5]) : 512;
  const int wr = argc > 6 ? atoi(argv[6]) : 2;
  const long long T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
  float *v, *ut, *bias, *y;
  hipMalloc(&v, sizeof(float) * 36 * ((T + 63) / 64 * 64) * Cin);
  hipMalloc(&ut, sizeof(float) * 36 * (size_t)Cout * Cin);
  hipMalloc(&bias, sizeof(float) * Cout);
  hipMalloc(&y, sizeof(float) * (size_t)B * H * W * Cout);
  std::vector<float> h(1 << 20);
  for (auto& f : h) f = (float)rand() / RAND_MAX - 0.5f;
  for (size_t o = 0; o < 36ull * T * Cin; o += h.size()) hipMemcpy(v + o, h.data(), sizeof(float) * std::min<size_t>(h.size(), (size_t)(36ull * T * Cin - o)), hipMemcpyHostToDevice);
  for (size_t o = 0; o < 36ull * Cout * Cin; o += h.size()) hipMemcpy(ut + o, h.data(), sizeof(float) * std::min<size_t>(h.size(), (size_t)(36ull * Cout * Cin - o)), hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), sizeof(float) * Cout, hipMemcpyHostToDevice);
  const double fl = 2.0 * 36 * T * Cin * Cout;
  const int it = 10;
#define R(ABL, WHAT) { float ms = wr == 2 ? run<ABL, 2>(v, ut, bias, y, H, W, Cin, Cout, T, it) : run<ABL, 1>(v, ut, bias, y, H, W, Cin, Cout, T, it); printf("%-44s %8.3f ms  %6.1f TFLOP/s-equivalent\n", WHAT, ms, fl / ms / 1e9); }
  printf("T=%lld Cin=%d Cout=%d WR=%d\n", T, Cin, Cout, wr);
  for (int w = 0; w < 250; w++) run<0, 1>(v, ut, bias, y, H, W, Cin, Cout, T, 5);   // clocks up (they ramp for seconds)
  R(0, "full kernel");
  R(64, "full kernel, every operand L2-resident");
  R(0, "full kernel (again)");
  R(32, "no epilogue");
  R(32 | 16, "no epilogue, no column fold");
  R(32 | 1, "no epilogue, no barrier");
  R(32 | 2, "no epilogue, no DMA");
  R(32 | 4, "no epilogue, no LDS reads");
  R(32 | 2 | 4, "no epilogue, no DMA, no LDS reads");
  R(32 | 1 | 2 | 4 | 16, "MFMAs + loop control only");
  R(32 | 8, "no epilogue, no MFMAs");
  R(32 | 8 | 4, "no epilogue, no MFMAs, no LDS reads (DMA only)");
  R(32 | 64, "no epilogue, every operand L2-resident");
  R(512, "full kernel, non-temporal output stores");
  R(0, "full kernel (again 3)");
  R(512, "full kernel, non-temporal output stores (again)");
  R(128, "epilogue without its global stores");
  R(256, "epilogue = bias + ReLU only (no staging, no stores)");
  R(0, "full kernel (again 2)");
  R(0, "full kernel (last)");
  return 0;
}
