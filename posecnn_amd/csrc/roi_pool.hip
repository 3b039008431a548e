// roi_pool.hip — gfx950 ROI max pooling, NHWC (replaces TF1 ops "RoiPool"/"RoiPoolGrad",
// lib/roi_pooling_layer/roi_pooling_op.cc:306-347,384-461 + roi_pooling_op_gpu.cu.cc:20-254).
//
// Layout-driven design: features are NHWC, so the 512 channels of one spatial position are one
// contiguous 2 KB row. A workgroup owns one (roi, ph, pw) bin; its threads each own 4 consecutive
// channels (one dwordx4 per position) and walk the bin h-major, so every load instruction of a
// wave is a fully coalesced 1 KB row segment and the bin geometry (the reference recomputes it
// per output element, :45-75) is computed once per workgroup in scalar registers.
// The fused `add2` entry pools conv5_3 and conv4_3 and adds them (vgg16_convs.py:177-187) without
// writing either pooled tensor or any argmax (inference does not consume them).
#include <cfloat>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

struct Bin {
  int hstart, hend, wstart, wend, batch, cls;
  bool empty;
};

// roi_pooling_op_gpu.cu.cc:45-75
__device__ __forceinline__ Bin make_bin(const float* __restrict__ roi, float scale, int ph, int pw,
                                        int PH, int PW, int H, int W)
{
  Bin b;
  b.batch = (int)roi[0];
  b.cls = (int)roi[1];
  int roi_start_w = (int)roundf(roi[2] * scale);
  int roi_start_h = (int)roundf(roi[3] * scale);
  int roi_end_w = (int)roundf(roi[4] * scale);
  int roi_end_h = (int)roundf(roi[5] * scale);
  int roi_width = max(roi_end_w - roi_start_w + 1, 1);
  int roi_height = max(roi_end_h - roi_start_h + 1, 1);
  float bin_size_h = div_rn((float)roi_height, (float)PH);
  float bin_size_w = div_rn((float)roi_width, (float)PW);
  int hstart = (int)floorf((float)ph * bin_size_h);
  int wstart = (int)floorf((float)pw * bin_size_w);
  int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
  int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
  b.hstart = min(max(hstart + roi_start_h, 0), H);
  b.hend = min(max(hend + roi_start_h, 0), H);
  b.wstart = min(max(wstart + roi_start_w, 0), W);
  b.wend = min(max(wend + roi_start_w, 0), W);
  b.empty = (b.hend <= b.hstart) || (b.wend <= b.wstart);
  return b;
}

// one workgroup per (roi, ph, pw); thread t owns channels [4t, 4t+3] (+ 4*blockDim strides)
__global__ __launch_bounds__(128) void roi_pool_fwd_vec4(const float* __restrict__ data,
                                                         const float* __restrict__ rois,
                                                         float* __restrict__ top,
                                                         int* __restrict__ argmax, int B, int H,
                                                         int W, int C, int roi_cols, int PH, int PW,
                                                         float scale)
{
  const int bin = blockIdx.x;
  const int pw = bin % PW, ph = (bin / PW) % PH, n = bin / (PW * PH);
  const Bin b = make_bin(rois + (size_t)n * roi_cols, scale, ph, pw, PH, PW, H, W);
  if (b.batch < 0 || b.batch >= B) {  // CHECK_GE/LT in the CPU op (roi_pooling_op.cc:146-147); GPU op reads OOB
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
      float4 z = make_float4(0, 0, 0, 0);
      *reinterpret_cast<float4*>(top + (size_t)bin * C + c) = z;
      if (argmax) *reinterpret_cast<int4*>(argmax + (size_t)bin * C + c) = make_int4(-1, -1, -1, -1);
    }
    return;
  }
  const float* img = data + (size_t)b.batch * H * W * C;
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    const float init = b.empty ? 0.f : -FLT_MAX;
    float4 mv = make_float4(init, init, init, init);
    int4 mi = make_int4(-1, -1, -1, -1);
    for (int h = b.hstart; h < b.hend; ++h)
      for (int w = b.wstart; w < b.wend; ++w) {
        const int base = (h * W + w) * C + c;
        const float4 v = *reinterpret_cast<const float4*>(img + base);
        if (v.x > mv.x) { mv.x = v.x; mi.x = base; }
        if (v.y > mv.y) { mv.y = v.y; mi.y = base + 1; }
        if (v.z > mv.z) { mv.z = v.z; mi.z = base + 2; }
        if (v.w > mv.w) { mv.w = v.w; mi.w = base + 3; }
      }
    *reinterpret_cast<float4*>(top + (size_t)bin * C + c) = mv;
    if (argmax) *reinterpret_cast<int4*>(argmax + (size_t)bin * C + c) = mi;
  }
}

// generic path: any C, and pool_channel == 1 (pools only channel roi_cls, :87-88); one thread per output
__global__ __launch_bounds__(256) void roi_pool_fwd_scalar(const float* __restrict__ data,
                                                           const float* __restrict__ rois,
                                                           float* __restrict__ top,
                                                           int* __restrict__ argmax, long long total,
                                                           int B, int H, int W, int C, int roi_cols,
                                                           int PH, int PW, float scale,
                                                           int pool_channel)
{
  for (long long index = (long long)blockIdx.x * 256 + threadIdx.x; index < total;
       index += (long long)gridDim.x * 256) {
    long long t = index;
    int c = 1;
    if (!pool_channel) { c = (int)(t % C); t /= C; }
    int pw = (int)(t % PW); t /= PW;
    int ph = (int)(t % PH); t /= PH;
    int n = (int)t;
    const Bin b = make_bin(rois + (size_t)n * roi_cols, scale, ph, pw, PH, PW, H, W);
    float maxval = b.empty ? 0.f : -FLT_MAX;
    int maxidx = -1;
    const int cc = pool_channel ? b.cls : c;
    if (b.batch >= 0 && b.batch < B && cc >= 0 && cc < C) {
      const float* img = data + (size_t)b.batch * H * W * C;
      for (int h = b.hstart; h < b.hend; ++h)
        for (int w = b.wstart; w < b.wend; ++w) {
          int bottom_index = (h * W + w) * C + cc;
          float v = img[bottom_index];
          if (v > maxval) { maxval = v; maxidx = bottom_index; }
        }
    } else {
      maxval = 0.f;
    }
    top[index] = maxval;
    if (argmax) argmax[index] = maxidx;
  }
}

// pool_score = roi_pool(conv5_3, 1/16) + roi_pool(conv4_3, 1/8), vgg16_convs.py:177-187
__global__ __launch_bounds__(128) void roi_pool_add2_vec4(
    const float* __restrict__ data_a, int Ha, int Wa, float scale_a,
    const float* __restrict__ data_b, int Hb, int Wb, float scale_b,
    const float* __restrict__ rois, float* __restrict__ out, int B, int C, int roi_cols, int PH,
    int PW, const int* __restrict__ num_rows_dev)
{
  const int bin = blockIdx.x;
  const int pw = bin % PW, ph = (bin / PW) % PH, n = bin / (PW * PH);
  const float* roi = rois + (size_t)n * roi_cols;
  const Bin ba = make_bin(roi, scale_a, ph, pw, PH, PW, Ha, Wa);
  const Bin bb = make_bin(roi, scale_b, ph, pw, PH, PW, Hb, Wb);
  // rows past the device-side count are padding of a capacity-sized ROI buffer: they pool to 0
  const bool ok = ba.batch >= 0 && ba.batch < B && (num_rows_dev == nullptr || n < num_rows_dev[0]);
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    float4 ma = make_float4(0, 0, 0, 0), mb = ma;
    if (ok) {
      const float* ia = data_a + (size_t)ba.batch * Ha * Wa * C;
      const float* ib = data_b + (size_t)bb.batch * Hb * Wb * C;
      const float inita = ba.empty ? 0.f : -FLT_MAX, initb = bb.empty ? 0.f : -FLT_MAX;
      ma = make_float4(inita, inita, inita, inita);
      mb = make_float4(initb, initb, initb, initb);
      for (int h = ba.hstart; h < ba.hend; ++h)
        for (int w = ba.wstart; w < ba.wend; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(ia + (h * Wa + w) * C + c);
          ma.x = v.x > ma.x ? v.x : ma.x; ma.y = v.y > ma.y ? v.y : ma.y;
          ma.z = v.z > ma.z ? v.z : ma.z; ma.w = v.w > ma.w ? v.w : ma.w;
        }
      for (int h = bb.hstart; h < bb.hend; ++h)
        for (int w = bb.wstart; w < bb.wend; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(ib + (h * Wb + w) * C + c);
          mb.x = v.x > mb.x ? v.x : mb.x; mb.y = v.y > mb.y ? v.y : mb.y;
          mb.z = v.z > mb.z ? v.z : mb.z; mb.w = v.w > mb.w ? v.w : mb.w;
        }
    }
    // tf.add_n([pool5, pool4]) (network.py:362-369): pool5 + pool4
    float4 r = make_float4(ma.x + mb.x, ma.y + mb.y, ma.z + mb.z, ma.w + mb.w);
    *reinterpret_cast<float4*>(out + (size_t)bin * C + c) = r;
  }
}

// ROIPoolBackward, roi_pooling_op_gpu.cu.cc:135-229 (gather form; ROIs ascending => deterministic)
__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(
    const float* __restrict__ top_diff, const float* __restrict__ rois,
    const int* __restrict__ argmax, float* __restrict__ bottom_diff, long long total, int H, int W,
    int C, int R, int roi_cols, int PH, int PW, float scale, int pool_channel)
{
  for (long long index = (long long)blockIdx.x * 256 + threadIdx.x; index < total;
       index += (long long)gridDim.x * 256) {
    long long t = index;
    int c = (int)(t % C); t /= C;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H); t /= H;
    int n = (int)t;
    float gradient = 0;
    for (int roi_n = 0; roi_n < R; ++roi_n) {
      const float* roi = rois + (size_t)roi_n * roi_cols;
      int roi_batch_ind = (int)roi[0];
      int roi_cls = (int)roi[1];
      if (n != roi_batch_ind) continue;
      if (pool_channel && c != roi_cls) continue;
      int roi_start_w = (int)roundf(roi[2] * scale);
      int roi_start_h = (int)roundf(roi[3] * scale);
      int roi_end_w = (int)roundf(roi[4] * scale);
      int roi_end_h = (int)roundf(roi[5] * scale);
      if (!(w >= roi_start_w && w <= roi_end_w && h >= roi_start_h && h <= roi_end_h)) continue;
      size_t offset = pool_channel ? (size_t)roi_n * PH * PW : (size_t)roi_n * PH * PW * C;
      const float* otd = top_diff + offset;
      const int* oam = argmax + offset;
      int roi_width = max(roi_end_w - roi_start_w + 1, 1);
      int roi_height = max(roi_end_h - roi_start_h + 1, 1);
      float bin_size_h = div_rn((float)roi_height, (float)PH);
      float bin_size_w = div_rn((float)roi_width, (float)PW);
      int phstart = (int)floorf(div_rn((float)(h - roi_start_h), bin_size_h));
      int phend = (int)ceilf(div_rn((float)(h - roi_start_h + 1), bin_size_h));
      int pwstart = (int)floorf(div_rn((float)(w - roi_start_w), bin_size_w));
      int pwend = (int)ceilf(div_rn((float)(w - roi_start_w + 1), bin_size_w));
      phstart = min(max(phstart, 0), PH);
      phend = min(max(phend, 0), PH);
      pwstart = min(max(pwstart, 0), PW);
      pwend = min(max(pwend, 0), PW);
      for (int ph = phstart; ph < phend; ++ph)
        for (int pw = pwstart; pw < pwend; ++pw) {
          if (pool_channel) {
            if (oam[ph * PW + pw] == (h * W + w) * C + c) gradient += otd[ph * PW + pw];
          } else {
            if (oam[(ph * PW + pw) * C + c] == (h * W + w) * C + c)
              gradient += otd[(ph * PW + pw) * C + c];
          }
        }
    }
    bottom_diff[index] = gradient;
  }
}

int validate(int B, int H, int W, int C, int R, int roi_cols, int PH, int PW, float scale,
             int pool_channel)
{
  PCNN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, PCNN_EINVAL, "roi_pool: data must be 4-dimensional (got %dx%dx%dx%d)", B, H, W, C);
  PCNN_REQUIRE(R >= 0, PCNN_EINVAL, "roi_pool: rois must be 2-dimensional (num_rois %d)", R);
  PCNN_REQUIRE(roi_cols >= 6, PCNN_EINVAL, "roi_pool: rois need >= 6 columns (batch, cls, x1, y1, x2, y2), got %d", roi_cols);
  // attribute checks, roi_pooling_op.cc:60-69 (`>= 0` there; 0 makes the output empty and divides by zero)
  PCNN_REQUIRE(PH >= 1, PCNN_EINVAL, "roi_pool: Need pooled_height >= 1, got %d", PH);
  PCNN_REQUIRE(PW >= 1, PCNN_EINVAL, "roi_pool: Need pooled_width >= 1, got %d", PW);
  PCNN_REQUIRE(pool_channel == 0 || pool_channel == 1, PCNN_EINVAL, "roi_pool: pool_channel must be 0 or 1");
  PCNN_REQUIRE((long long)H * W * C < (1ll << 31), PCNN_EINVAL, "roi_pool: image too large for int32 argmax");
  (void)scale;
  return PCNN_OK;
}

}  // namespace

extern "C" int pcnn_roi_pool_fwd(const float* data, const float* rois, int B, int H, int W, int C,
                                 int R, int roi_cols, int PH, int PW, float scale,
                                 int pool_channel, float* top, int32_t* argmax, void* stream_)
{
  int st = validate(B, H, W, C, R, roi_cols, PH, PW, scale, pool_channel);
  if (st != PCNN_OK) return st;
  if (R == 0) return PCNN_OK;
  PCNN_REQUIRE(data && rois && top, PCNN_ENULL, "roi_pool: NULL pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const bool vec = !pool_channel && (C % 4 == 0) && aligned16(data) && aligned16(top) &&
                   (!argmax || aligned16(argmax));
  if (vec) {
    const int threads = C >= 512 ? 128 : 64;
    PCNN_LAUNCH(roi_pool_fwd_vec4, dim3(R * PH * PW), dim3(threads), 0, stream, data, rois,
                       top, argmax, B, H, W, C, roi_cols, PH, PW, scale);
  } else {
    long long total = (long long)R * PH * PW * (pool_channel ? 1 : C);
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    PCNN_LAUNCH(roi_pool_fwd_scalar, dim3(blocks), dim3(256), 0, stream, data, rois, top,
                       argmax, total, B, H, W, C, roi_cols, PH, PW, scale, pool_channel);
  }
  return check_launch("roi_pool_fwd");
}

extern "C" int pcnn_roi_pool_add2_fwd(const float* data_a, int Ha, int Wa, float scale_a,
                                      const float* data_b, int Hb, int Wb, float scale_b,
                                      const float* rois, int B, int C, int R, int roi_cols, int PH,
                                      int PW, const int32_t* num_rows_dev, float* out, void* stream_)
{
  int st = validate(B, Ha, Wa, C, R, roi_cols, PH, PW, scale_a, 0);
  if (st != PCNN_OK) return st;
  st = validate(B, Hb, Wb, C, R, roi_cols, PH, PW, scale_b, 0);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(C % 4 == 0, PCNN_EINVAL, "roi_pool_add2: channels must be a multiple of 4 (got %d)", C);
  if (R == 0) return PCNN_OK;
  PCNN_REQUIRE(data_a && data_b && rois && out, PCNN_ENULL, "roi_pool_add2: NULL pointer");
  PCNN_REQUIRE(aligned16(data_a) && aligned16(data_b) && aligned16(out), PCNN_EINVAL,
               "roi_pool_add2: tensors must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int threads = C >= 512 ? 128 : 64;
  PCNN_LAUNCH(roi_pool_add2_vec4, dim3(R * PH * PW), dim3(threads), 0, stream, data_a, Ha,
                     Wa, scale_a, data_b, Hb, Wb, scale_b, rois, out, B, C, roi_cols, PH, PW, num_rows_dev);
  return check_launch("roi_pool_add2_fwd");
}

extern "C" int pcnn_roi_pool_bwd(const float* top_diff, const float* rois, const int32_t* argmax,
                                 int B, int H, int W, int C, int R, int roi_cols, int PH, int PW,
                                 float scale, int pool_channel, float* bottom_diff, void* stream_)
{
  int st = validate(B, H, W, C, R, roi_cols, PH, PW, scale, pool_channel);
  if (st != PCNN_OK) return st;
  PCNN_REQUIRE(bottom_diff, PCNN_ENULL, "roi_pool_bwd: NULL output");
  PCNN_REQUIRE(R == 0 || (top_diff && rois && argmax), PCNN_ENULL, "roi_pool_bwd: NULL input");
  hipStream_t stream = (hipStream_t)stream_;
  long long total = (long long)B * H * W * C;
  int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  PCNN_LAUNCH(roi_pool_bwd_kernel, dim3(blocks), dim3(256), 0, stream, top_diff, rois,
                     argmax, bottom_diff, total, H, W, C, R, roi_cols, PH, PW, scale, pool_channel);
  return check_launch("roi_pool_bwd");
}
