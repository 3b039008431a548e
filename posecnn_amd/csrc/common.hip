// common.hip — error plumbing and ABI bookkeeping for libposecnn_hip.so.
// Replaces the reference's fprintf(stderr)+exit(-1) on CUDA errors
// (hough_voting_gpu_op.cu.cc:679-684, roi_pooling_op_gpu.cu.cc:123-128) and OP_REQUIRES
// InvalidArgument (hough_voting_gpu_op.cc:328-332) with status codes + a thread-local message.
#include <stdarg.h>
#include <stdio.h>

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

}  // namespace pcnn

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
