// tools/wino_w1_probe.hip — the one-wave-per-SIMD 32x32x2 kernel (wino43_mfma_w1_kernel) against the library's
// wino43_mfma_kernel at trunk shapes: bit equality of every output (all three pool modes, two filter groups, a ragged tile
// count) and timing, plus the K-loop ablations of the new kernel. Not part of the library. Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iposecnn_amd/csrc -Iinclude \
//         tools/wino_w1_probe.hip posecnn_amd/csrc/common.hip -o tools/wino_w1_probe && tools/wino_w1_probe
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../posecnn_amd/csrc/wino_mfma.hip"
#include "variants/wino43_mfma_w1_kernel.inc"   // (round 5: the kernel under test lives here, no longer in the library)

struct Shape { const char* name; int B, H, W, Cin, Cout, groups; };

template <int POOL>
static float run_old(const float* v, const float* ut, const float* bias, float* y, float* yp, const Shape& s, int iters)
{
  const int Ht = (s.H + 3) / 4, Wt = (s.W + 3) / 4;
  const long long T = (long long)s.B * Ht * Wt, tpg = T / s.groups;
  const long long nbt = (long long)s.groups * ((tpg + 31) / 32);
  const int ncb = s.Cout / 64;
  const long long blocks = ((nbt + 7) / 8) * 8 * ncb;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < iters + 1; i++) {
    if (i == 1) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wino43_mfma_kernel<POOL, 1, 0, 1>), dim3((unsigned)blocks, 1), dim3(256), 0, 0, v, ut, bias, y, yp, s.H, s.W, s.Cin,
                       s.Cout, Ht, Wt, T, tpg, 1, (int)nbt, ncb, 1, 0ll, 0);
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

template <int POOL, int ABL, int NB = 3, int SB = 0>
static float run_w1(const float* v, const float* ut, const float* bias, float* y, float* yp, const Shape& s, int iters)
{
  const int Ht = (s.H + 3) / 4, Wt = (s.W + 3) / 4;
  const long long T = (long long)s.B * Ht * Wt, tpg = T / s.groups;
  const long long nbt = (long long)s.groups * ((tpg + 63) / 64);
  const int ncb = s.Cout / 64;
  const long long blocks = ((nbt + 7) / 8) * 8 * ncb;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < iters + 1; i++) {
    if (i == 1) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wino43_mfma_w1_kernel<POOL, ABL, NB, SB>), dim3((unsigned)blocks), dim3(256), 0, 0, v, ut, bias, y, yp, s.H, s.W, s.Cin,
                       s.Cout, Ht, Wt, T, tpg, 1, (int)nbt, ncb, 0);
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

static size_t diff_count(const float* a, const float* b, size_t n, std::vector<float>& ha, std::vector<float>& hb)
{
  ha.resize(n); hb.resize(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  size_t d = 0;
  for (size_t i = 0; i < n; i++) d += memcmp(&ha[i], &hb[i], 4) != 0;
  return d;
}

int main(int argc, char** argv)
{
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const Shape shapes[] = {
    {"ragged 3x60x80 256->128 g1", 3, 60, 80, 256, 128, 1},
    {"ragged 6x36x44 128->64 g2", 6, 36, 44, 128, 64, 2},
    {"conv4_2 32x60x80 512->512 g2", 32, 60, 80, 512, 512, 2},
    {"conv5_1 32x30x40 512->512 g2", 32, 30, 40, 512, 512, 2},
    {"conv4_1 32x60x80 256->512 g2", 32, 60, 80, 256, 512, 2},
    {"conv3_2 32x120x160 256->256 g2", 32, 120, 160, 256, 256, 2},
    {"conv3_1 32x120x160 128->256 g2", 32, 120, 160, 128, 256, 2},
    {"conv2_2 32x240x320 128->128 g2", 32, 240, 320, 128, 128, 2},
    {"conv2_1 32x240x320 64->128 g2", 32, 240, 320, 64, 128, 2},
  };
  std::vector<float> h(1 << 22), ha, hb;
  srand(7);
  for (auto& f : h) { const float u = (float)rand() / RAND_MAX - 0.5f; f = (rand() & 3) ? 4.f * u * u * u : 0.f; }   // heavy-tailed, a quarter zeros
  int bad = 0;
  bool warmed = false;
  for (const Shape& s : shapes) {
    const int Ht = (s.H + 3) / 4, Wt = (s.W + 3) / 4;
    const long long T = (long long)s.B * Ht * Wt;
    const size_t nv = 36ull * (T + 64) * s.Cin, nu = 36ull * s.groups * s.Cout * s.Cin, ny = (size_t)s.B * s.H * s.W * s.Cout;
    float *v, *ut, *bias, *y0, *y1, *p0, *p1;
    hipMalloc(&v, nv * 4); hipMalloc(&ut, nu * 4); hipMalloc(&bias, sizeof(float) * s.groups * s.Cout);
    hipMalloc(&y0, ny * 4); hipMalloc(&y1, ny * 4); hipMalloc(&p0, ny); hipMalloc(&p1, ny);
    for (size_t o = 0; o < nv; o += h.size()) hipMemcpy(v + o, h.data(), 4 * std::min(h.size(), nv - o), hipMemcpyHostToDevice);
    for (size_t o = 0; o < nu; o += h.size()) hipMemcpy(ut + o, h.data() + 12345, 4 * std::min(h.size() - 12345, nu - o), hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data() + 77, sizeof(float) * s.groups * s.Cout, hipMemcpyHostToDevice);
    const double fl = 2.0 * 36 * T * s.Cin * s.Cout;
    const int it = 5;
    printf("== %s  T=%lld\n", s.name, T);
    // pool 0
    hipMemset(y0, 0xff, ny * 4); hipMemset(y1, 0xff, ny * 4);
    if (s.B == 32 && !warmed) {   // the clocks ramp for seconds: without this the first kernels measured look 15 % slower
      for (int w = 0; w < 300; w++) run_old<0>(v, ut, bias, y0, p0, s, 5);
      warmed = true;
    }
    float a = 1e9f, b = 1e9f;
    for (int rep = 0; rep < (s.B == 32 ? 4 : 1); rep++) {   // A / B alternating, best of 4
      a = std::min(a, run_old<0>(v, ut, bias, y0, p0, s, it));
      b = std::min(b, run_w1<0, 0>(v, ut, bias, y1, p1, s, it));
    }
    if (s.B == 32) {
      float c = 1e9f;
      for (int rep = 0; rep < 4; rep++) c = std::min(c, run_w1<0, 0, 4>(v, ut, bias, y1, p1, s, it));
      size_t d4 = diff_count(y0, y1, ny, ha, hb);
      printf("  ring of 4: w1 %7.3f ms %6.1f TF | %zu differ\n", c, fl / c / 1e9, d4);
      bad += d4 != 0;
      hipMemset(y1, 0xff, ny * 4);
      c = 1e9f;
      for (int rep = 0; rep < 4; rep++) c = std::min(c, run_w1<0, 0, 4, 1>(v, ut, bias, y1, p1, s, it));
      d4 = diff_count(y0, y1, ny, ha, hb);
      printf("  LDS counters instead of the barrier: w1 %7.3f ms %6.1f TF | %zu differ\n", c, fl / c / 1e9, d4);
      bad += d4 != 0;
      run_w1<0, 0>(v, ut, bias, y1, p1, s, 1);
    }
    size_t d = diff_count(y0, y1, ny, ha, hb);
    printf("  pool 0: library %7.3f ms %6.1f TF | w1 %7.3f ms %6.1f TF | %zu of %zu outputs differ\n", a, fl / a / 1e9, b, fl / b / 1e9, d, ny);
    bad += d != 0;
    if (!quick || s.B < 32) {
      // pool 1 (pooled only, written to y) and pool 2 (both)
      hipMemset(y0, 0xff, ny * 4); hipMemset(y1, 0xff, ny * 4);
      a = run_old<1>(v, ut, bias, y0, p0, s, 2); b = run_w1<1, 0>(v, ut, bias, y1, p1, s, 2);
      d = diff_count(y0, y1, ny / 4, ha, hb);
      printf("  pool 1: library %7.3f ms | w1 %7.3f ms | %zu of %zu differ\n", a, b, d, ny / 4);
      bad += d != 0;
      hipMemset(y0, 0xff, ny * 4); hipMemset(y1, 0xff, ny * 4); hipMemset(p0, 0xff, ny); hipMemset(p1, 0xff, ny);
      a = run_old<2>(v, ut, bias, y0, p0, s, 2); b = run_w1<2, 0>(v, ut, bias, y1, p1, s, 2);
      d = diff_count(y0, y1, ny, ha, hb);
      const size_t d2 = diff_count(p0, p1, ny / 4, ha, hb);
      printf("  pool 2: library %7.3f ms | w1 %7.3f ms | %zu of %zu, pooled %zu of %zu differ\n", a, b, d, ny, d2, ny / 4);
      bad += d != 0 || d2 != 0;
    }
    if (getenv("W1_ABLATE") && s.B == 32 && (s.Cin == 512 || s.Cin == 128) && s.H != 30 && s.Cout == s.Cin) {
#define R(ABL, WHAT) { const float ms = run_w1<0, ABL>(v, ut, bias, y1, p1, s, it); printf("    %-44s %8.3f ms  %6.1f TF-equivalent\n", WHAT, ms, fl / ms / 1e9); }
      R(32, "no epilogue");
      R(32 | 16, "no epilogue, no column fold");
      R(32 | 1, "no epilogue, no barrier");
      R(32 | 2, "no epilogue, no DMA");
      R(32 | 4, "no epilogue, no LDS reads");
      R(32 | 2 | 4, "no epilogue, no DMA, no LDS reads");
      R(32 | 1 | 2 | 4 | 16, "MFMAs + loop control only");
      R(32 | 8, "no epilogue, no MFMAs");
#undef R
    }
    hipFree(v); hipFree(ut); hipFree(bias); hipFree(y0); hipFree(y1); hipFree(p0); hipFree(p1);
  }
  printf(bad ? "MISMATCH in %d comparisons\n" : "all comparisons bit-identical (%d)\n", bad);
  return bad != 0;
}
