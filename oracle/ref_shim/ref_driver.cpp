// ref_driver.cpp — runs the REFERENCE's own kernel bodies on the CPU.
//
// TEST INFRASTRUCTURE ONLY. The `__global__` / `__device__` functions of the reference's
// `*_op_gpu.cu.cc` files are #included below from line-range extracts that oracle/ref_shim/Makefile
// cuts out of /root/reference at build time (into a temporary directory; nothing of the reference
// is stored in this repo or in oracle/_ref, only the compiled libposecnn_ref.so). They compile
// unchanged against cuda_on_cpu.h; a launch is emulated as n sequential thread invocations.
//
// What is NOT the reference's code here: the host launchers. `HoughVotingLaucher` & co. are TF
// OpKernel code with <<<>>> launches, thrust and blocking cudaMemcpy; they cannot be compiled
// without TensorFlow/CUDA. The functions below restate just their orchestration (which kernel, which
// buffers, which host-side filter), citing the launcher lines they follow.
#include "cuda_on_cpu.h"

#include <cstdlib>
#include <vector>

#define VERTEX_CHANNELS 3  // hough_voting_gpu_op.cu.cc:13
#define MAX_ROI 128        // hough_voting_gpu_op.cu.cc:14
#define POSE_CHANNELS 4    // average_distance_loss_op_gpu.cu.cc:14

#include "hough_a.inc"        // hough_voting_gpu_op.cu.cc:23-187
#include "hough_b.inc"        // hough_voting_gpu_op.cu.cc:253-576
#include "roi_fwd.inc"        // roi_pooling_op_gpu.cu.cc:19-101
#include "roi_bwd.inc"        // roi_pooling_op_gpu.cu.cc:134-229
#include "hard_fwd.inc"       // hard_label_op_gpu.cu.cc:16-29
#include "adl_fwd.inc"        // average_distance_loss_op_gpu.cu.cc:34-206
#include "adl_sum.inc"        // average_distance_loss_op_gpu.cu.cc:209-252
#include "adl_bwd.inc"        // average_distance_loss_op_gpu.cu.cc:346-354
#include "bp_fwd.inc"         // backprojecting_op_gpu.cu.cc:16-126
#include "bp_bwd.inc"         // backprojecting_op_gpu.cu.cc:158-217

extern "C" {

// HoughvotinggpuOp<GpuDevice>::Compute (hough_voting_gpu_op.cc:349-428) + HoughVotingLaucher
// (hough_voting_gpu_op.cu.cc:615-797). Outputs: capacity MAX_ROI*9 rows, zero-filled; num_rois[0] =
// rows returned (dummy row when none), num_rois[1] = true count.
int ref_hough_voting(const int* label, const float* vertex, const float* extents, const float* meta,
                     const float* gt, int batch_size, int height, int width, int num_classes,
                     int num_meta_data, int num_gt, int is_train, float votingThreshold,
                     float perThreshold, int skip_pixels, float inlierThreshold, int labelThreshold,
                     float* top_box, float* top_pose, float* top_target, float* top_weight,
                     int* top_domain, int* num_rois_out, float* hs_debug)
{
  const int num = MAX_ROI * 9;  // reset_outputs :579-588
  memset(top_box, 0, num * 7 * sizeof(float));
  memset(top_pose, 0, num * 7 * sizeof(float));
  memset(top_target, 0, (size_t)num * 4 * num_classes * sizeof(float));
  memset(top_weight, 0, (size_t)num * 4 * num_classes * sizeof(float));
  memset(top_domain, 0, num * sizeof(int));
  int num_rois = 0;
  const int HW = height * width;
  std::vector<float> gt0(13, 0.f);
  if (!gt) gt = gt0.data();
  for (int n = 0; n < batch_size; n++) {  // :369-377
    const int* labelmap = label + (size_t)n * HW;
    const float* vertmap = vertex + (size_t)n * HW * VERTEX_CHANNELS * num_classes;
    const float* meta_data = meta + (size_t)n * num_meta_data;
    // step 1 :626-647
    std::vector<int> arrays((size_t)num_classes * HW, 0), array_sizes(num_classes, 0);
    PCNN_LAUNCH_1D(HW, compute_arrays_kernel(HW, labelmap, arrays.data(), array_sizes.data(), height, width));
    // class indexes :649-670
    std::vector<int> class_indexes;
    for (int c = 1; c < num_classes; c++)
      if (array_sizes[c] > labelThreshold) class_indexes.push_back(c);
    const int count = (int)class_indexes.size();
    if (count == 0) continue;
    // step 2 :686-714
    std::vector<float> hough_space((size_t)count * HW, 0.f), hough_data((size_t)count * HW * 3, 0.f);
    PCNN_LAUNCH_1D(count * HW, compute_hough_kernel(count * HW, hough_space.data(), hough_data.data(), labelmap, vertmap, extents,
                         meta_data, arrays.data(), array_sizes.data(), class_indexes.data(), height,
                         width, num_classes, count, inlierThreshold, skip_pixels));
    if (hs_debug)
      for (int i = 0; i < count; i++)
        memcpy(hs_debug + ((size_t)n * num_classes + class_indexes[i]) * HW, hough_space.data() + (size_t)i * HW, sizeof(float) * HW);
    // step 3 :723-762
    int num_max = 0;
    const int index_size = MAX_ROI / batch_size;
    std::vector<int> max_indexes(std::max(index_size, count) + 1, 0);
    if (votingThreshold > 0) {
      PCNN_LAUNCH_1D(count * HW, compute_max_indexes_kernel(count * HW, max_indexes.data(), index_size, &num_max, hough_space.data(),
                                 hough_data.data(), height, width, votingThreshold, perThreshold));
    } else {
      for (int i = 0; i < count; i++) {  // thrust::max_element: first maximum
        float* hmax = std::max_element(hough_space.data() + (size_t)i * HW, hough_space.data() + (size_t)(i + 1) * HW);
        max_indexes[i] = (int)(hmax - hough_space.data());
      }
      num_max = count;
    }
    // step 4 :770-785
    int num_max_host = num_max;
    if (num_max_host >= index_size) num_max_host = index_size;
    if (num_max_host > 0)
      PCNN_LAUNCH_1D(num_max_host, compute_rois_kernel(num_max_host, top_box, top_pose, top_target, top_weight, top_domain, extents,
                          meta_data, gt, hough_space.data(), hough_data.data(), max_indexes.data(),
                          class_indexes.data(), is_train, n, height, width, num_classes, num_gt, &num_rois));
  }
  num_rois_out[1] = num_rois;
  num_rois_out[0] = num_rois == 0 ? 1 : num_rois;  // hough_voting_gpu_op.cc:381-383
  return 0;
}

// ROIPoolForwardLaucher, roi_pooling_op_gpu.cu.cc:103-131
int ref_roi_pool(const float* data, const float* rois, int height, int width, int channels, int num_rois,
                 int channel_rois, int PH, int PW, float scale, int pool_channel, float* top, int* argmax)
{
  int output_size = pool_channel ? num_rois * PH * PW : num_rois * PH * PW * channels;
  PCNN_LAUNCH_1D(output_size, ROIPoolForward<float>(output_size, data, scale, pool_channel, height, width, channels, PH, PW, channel_rois,
                        rois, top, argmax));
  return 0;
}

// ROIPoolBackwardLaucher, roi_pooling_op_gpu.cu.cc:232-254
int ref_roi_pool_bwd(const float* top_diff, const float* rois, const int* argmax, int batch, int height,
                     int width, int channels, int num_rois, int channel_rois, int PH, int PW, float scale,
                     int pool_channel, float* bottom_diff)
{
  PCNN_LAUNCH_1D(batch * height * width * channels, ROIPoolBackward<float>(batch * height * width * channels, top_diff, argmax, num_rois, channel_rois, scale,
                         pool_channel, height, width, channels, PH, PW, bottom_diff, rois));
  return 0;
}

// HardlabelForwardLaucher, hard_label_op_gpu.cu.cc:32-51
int ref_hard_label(const float* prob, const int* gt, int n_pixels, int num_classes, float threshold, float* out)
{
  PCNN_LAUNCH_1D(n_pixels, HardlabelForward<float>(n_pixels, prob, gt, num_classes, threshold, out));
  return 0;
}

// AveragedistanceForwardLaucher, average_distance_loss_op_gpu.cu.cc:256-343
int ref_average_distance(const float* prediction, const float* target, const float* weight, const float* point,
                         const float* symmetry, int batch_size, int num_classes, int num_points, float margin,
                         float* loss, float* bottom_diff)
{
  std::vector<float> losses((size_t)batch_size * num_points, 0.f), loss_batch(batch_size, 0.f);
  std::vector<float> diffs((size_t)batch_size * num_points * POSE_CHANNELS * num_classes, 0.f);
  std::vector<float> rotations((size_t)batch_size * num_points * 6 * 9, 0.f);
  PCNN_LAUNCH_1D(batch_size * num_points, AveragedistanceForward<float>(batch_size * num_points, prediction, target, weight, point, symmetry, batch_size,
                                num_classes, num_points, margin, rotations.data(), losses.data(), diffs.data()));
  memset(bottom_diff, 0, sizeof(float) * batch_size * POSE_CHANNELS * num_classes);
  PCNN_LAUNCH_1D(batch_size * POSE_CHANNELS * num_classes, sum_losses_gradients<float>(batch_size * POSE_CHANNELS * num_classes, losses.data(), diffs.data(), batch_size,
                              num_classes, num_points, loss_batch.data(), bottom_diff));
  float total = 0.f;  // thrust::reduce :333-335, canonical ascending
  for (int n = 0; n < batch_size; n++) total += loss_batch[n];
  loss[0] = total;
  return 0;
}

int ref_average_distance_bwd(const float* grad, const float* bottom_diff, int n, float* out)
{
  PCNN_LAUNCH_1D(n, AveragedistanceBackward<float>(n, grad, bottom_diff, out));
  return 0;
}

// BackprojectForwardLaucher, backprojecting_op_gpu.cu.cc:129-155
int ref_backproject(const float* data, const float* label, const float* depth, const float* meta,
                    const float* label_3d, int batch, int height, int width, int channels, int num_classes,
                    int num_meta, int grid, int ksize, float threshold, float* top_data, float* top_label,
                    float* top_flag)
{
  PCNN_LAUNCH_1D(batch * grid * grid * grid * channels, BackprojectForward<float>(batch * grid * grid * grid * channels, data, label, depth, meta, label_3d, height, width,
                            channels, num_classes, num_meta, grid, ksize, threshold, top_data, top_label, top_flag));
  return 0;
}

// BackprojectBackwardLaucher, backprojecting_op_gpu.cu.cc:220-244
int ref_backproject_bwd(const float* top_diff, const float* depth, const float* meta, int batch, int height,
                        int width, int channels, int num_meta, int grid, float* bottom_diff)
{
  PCNN_LAUNCH_1D(batch * height * width * channels, BackprojectBackward<float>(batch * height * width * channels, top_diff, depth, meta, height, width, channels,
                             num_meta, grid, bottom_diff));
  return 0;
}

}  // extern "C"
