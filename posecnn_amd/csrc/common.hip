// common.hip — error plumbing and ABI bookkeeping for libposecnn_hip.so.
// Replaces the reference's fprintf(stderr)+exit(-1) on CUDA errors
// (hough_voting_gpu_op.cu.cc:679-684, roi_pooling_op_gpu.cu.cc:123-128) and OP_REQUIRES
// InvalidArgument (hough_voting_gpu_op.cc:328-332) with status codes + a thread-local message.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pcnn_device.h"

namespace pcnn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int check_launch(const char* what)
{
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PCNN_EHIP;
  }
  return PCNN_OK;
}

// ---- zero fill ------------------------------------------------------------------------------
// A kernel, not hipMemsetAsync: inside a stream capture (posecnn_amd/pipeline.py GraphedStep) the memset
// node of this ROCm corrupted neighbouring allocations on replay with changed inputs
// (tests/test_gpu_round2.py::test_hipgraph_replay_equals_the_eager_step found it).
__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned int* __restrict__ p, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

int zero_async(void* p, size_t bytes, hipStream_t stream, const char* what)
{
  if (bytes == 0) return PCNN_OK;
  const size_t n = bytes / 4;   // every buffer of this library is a multiple of 4 bytes
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream,
                     static_cast<unsigned int*>(p), n);
  return check_launch(what);
}

// ---- per-kernel timing ----------------------------------------------------------------------
bool g_profile_on = false;

namespace {
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
};
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
std::mutex g_prof_mu;

hipEvent_t take_event()
{
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void profile_begin(const char* name, hipStream_t stream)
{
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{name, take_event(), take_event()};
  (void)hipEventRecord(r.e0, stream);
  g_recs.push_back(r);
}

void profile_end(hipStream_t stream)
{
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, stream);
}

}  // namespace pcnn

extern "C" int pcnn_profile_reset(void)
{
  std::lock_guard<std::mutex> lk(pcnn::g_prof_mu);
  for (auto& r : pcnn::g_recs) {
    (void)hipEventSynchronize(r.e1);
    pcnn::g_pool.push_back(r.e0);
    pcnn::g_pool.push_back(r.e1);
  }
  pcnn::g_recs.clear();
  return PCNN_OK;
}

extern "C" int pcnn_profile_enable(int on)
{
  pcnn_profile_reset();
  pcnn::g_profile_on = on != 0;
  return PCNN_OK;
}

extern "C" long pcnn_profile_report(char* buf, long cap)
{
  std::lock_guard<std::mutex> lk(pcnn::g_prof_mu);
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& r : pcnn::g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      std::string n(r.name);
      size_t p = n.rfind("::");
      if (p != std::string::npos) n = n.substr(p + 2);
      size_t lt = n.find('<');
      if (lt != std::string::npos) n = n.substr(0, lt);
      auto& a = agg[n];
      a.first += 1;
      a.second += ms;
    }
  }
  std::string out = "{";
  bool first = true;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof line, "%s\"%s\": {\"calls\": %ld, \"total_ms\": %.6f, \"avg_us\": %.3f}",
             first ? "" : ", ", kv.first.c_str(), kv.second.first, kv.second.second,
             1000.0 * kv.second.second / (double)kv.second.first);
    out += line;
    first = false;
  }
  out += "}";
  if (buf && cap > 0) {
    long n = (long)out.size() < cap - 1 ? (long)out.size() : cap - 1;
    memcpy(buf, out.data(), (size_t)n);
    buf[n] = 0;
  }
  return (long)out.size() + 1;
}

extern "C" int pcnn_abi_version(void) { return PCNN_ABI_VERSION; }

extern "C" const char* pcnn_last_error_string(void) { return pcnn::g_err; }

extern "C" const char* pcnn_status_string(int status)
{
  switch (status) {
    case PCNN_OK: return "ok";
    case PCNN_EINVAL: return "invalid argument";
    case PCNN_EWORKSPACE: return "workspace missing, misaligned or too small";
    case PCNN_EHIP: return "HIP runtime error";
    case PCNN_ENULL: return "null pointer";
    default: return "unknown status";
  }
}

// ---- CRC32C (Castagnoli), host side ------------------------------------------------------------
// TensorFlow checkpoints (tensor bundles / LevelDB tables) protect every table block and every tensor
// payload with a masked CRC32C; posecnn_amd/tf_checkpoint.py verifies them through this entry so that
// a 0.5 GB checkpoint checks in a fraction of a second (SSE4.2 crc32 instruction; table fallback).
namespace {
uint32_t g_crc_table[256];
bool g_crc_table_ready = false;

void crc32c_table_init()
{
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
    g_crc_table[i] = c;
  }
  g_crc_table_ready = true;
}

uint32_t crc32c_sw(const unsigned char* p, size_t n, uint32_t crc)
{
  if (!g_crc_table_ready) crc32c_table_init();
  for (size_t i = 0; i < n; i++) crc = g_crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return crc;
}

#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const unsigned char* p, size_t n, uint32_t crc)
{
  uint64_t c = crc;
  while (n && ((uintptr_t)p & 7)) { c = __builtin_ia32_crc32qi((uint32_t)c, *p++); n--; }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c = __builtin_ia32_crc32di(c, v);
    p += 8;
    n -= 8;
  }
  while (n) { c = __builtin_ia32_crc32qi((uint32_t)c, *p++); n--; }
  return (uint32_t)c;
}
#endif
}  // namespace

extern "C" uint32_t pcnn_crc32c(const void* data, size_t n, uint32_t seed)
{
  uint32_t crc = ~seed;
  const unsigned char* p = (const unsigned char*)data;
  if (p && n) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("sse4.2")) crc = crc32c_hw(p, n, crc);
    else
#endif
      crc = crc32c_sw(p, n, crc);
  }
  return ~crc;
}
