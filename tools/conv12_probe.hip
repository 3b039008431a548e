// tools/conv12_probe.hip — where does a workgroup of the fused conv1_1 -> conv1_2 kernel (first version) spend its time?
// Wall-clock stamps (100 MHz) of one mid-grid workgroup's phases under a full-size launch. Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iposecnn_amd/csrc -Iinclude \
//         tools/conv12_probe.hip posecnn_amd/csrc/common.hip -o tools/conv12_probe && tools/conv12_probe
#define CONV12_PROBE 20000
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../posecnn_amd/csrc/conv_first.hip"

int main(int argc, char** argv)
{
  const int frag = argc > 1 ? atoi(argv[1]) : 0;     // 1: the B-operand loads of the fragment-major filter-bank layout
  const int B = 32, H = 480, W = 640;
  float *x, *w1, *b1, *ut2, *b2, *y;
  hipMalloc(&x, sizeof(float) * (size_t)B * H * W * 3);
  hipMalloc(&w1, sizeof(float) * 2 * 27 * 64);
  hipMalloc(&b1, sizeof(float) * 2 * 64);
  hipMalloc(&ut2, sizeof(float) * 2 * 36 * 64 * 64);
  hipMalloc(&b2, sizeof(float) * 2 * 64);
  hipMalloc(&y, sizeof(float) * (size_t)B * (H / 2) * (W / 2) * 64);
  hipMemset(x, 0, sizeof(float) * (size_t)B * H * W * 3);
  hipMemset(w1, 0, sizeof(float) * 2 * 27 * 64);
  hipMemset(b1, 0, sizeof(float) * 2 * 64);
  hipMemset(ut2, 0, sizeof(float) * 2 * 36 * 64 * 64);
  hipMemset(b2, 0, sizeof(float) * 2 * 64);
  const RawFrames none = {nullptr, nullptr, 0, {0.0, 0.0, 0.0}};
  const unsigned blocks = B * (H / 16) * (W / 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv12_wino43_fused_kernel<false>), dim3(blocks), dim3(512), 0, 0, x, w1, b1, ut2, b2, y, H, W, W / 16, H / 16, B / 2, 1, 1, none, frag);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ts[64];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_conv12_probe), sizeof(ts));
    auto us = [&](int a, int b) { return (double)(ts[b] - ts[a]) / 100.0; };
    printf("[ut2_layout %d] launch %.3f ms | block %d: window %.2f us, conv1_1 %.2f us, planes %.2f us, output transform %.2f us, store %.2f us, total %.2f us\n",
           frag, ms, CONV12_PROBE, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 5));
    printf("  steps (consumer wave 0 done / producer wave 4 done, us after phase-2 start):");
    for (int s = 0; s < 7; s++) printf(" [%.2f / %.2f]", us(2, 10 + 2 * s), us(2, 11 + 2 * s));
    printf("\n");
  }
  return 0;
}
