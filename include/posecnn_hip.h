/*
 * posecnn_hip.h — C-ABI of libposecnn_hip.so, the MI355X (gfx950) replacement for the
 * PoseCNN TF1 custom-op plugins on the single-frame inference hot path.
 *
 * Every entry point replaces one `REGISTER_OP` + GPU `OpKernel::Compute` + `*Laucher`
 * triple of the reference (citations are relative to the reference tree):
 *
 *   pcnn_hough_voting_*     lib/hough_voting_gpu_layer/hough_voting_gpu_op.cc:37-52,321-429
 *                           lib/hough_voting_gpu_layer/hough_voting_gpu_op.cu.cc:615-797
 *   pcnn_roi_pool_*         lib/roi_pooling_layer/roi_pooling_op.cc:29-50,306-347
 *                           lib/roi_pooling_layer/roi_pooling_op_gpu.cu.cc:103-131,232-254
 *   pcnn_hard_label_*       lib/hard_label_layer/hard_label_op.cc:30-44,143-188
 *                           lib/hard_label_layer/hard_label_op_gpu.cu.cc:32-51,66-85
 *   pcnn_average_distance_* lib/average_distance_loss/average_distance_loss_op.cc:38-54,253-314
 *                           lib/average_distance_loss/average_distance_loss_op_gpu.cu.cc:256-377
 *   pcnn_backproject_*      lib/backprojecting_layer/backprojecting_op.cc:30-53,295-383
 *                           lib/backprojecting_layer/backprojecting_op_gpu.cu.cc:129-155,220-244
 *
 * Conventions
 *   - Plain C: raw DEVICE pointers, ints, floats. No torch/TF types cross this boundary.
 *   - Tensors are dense, row-major, NHWC exactly as the reference ops see them; f32 / int32.
 *   - The library never allocates, frees, synchronises or exits. The caller owns inputs,
 *     outputs and workspace; `pcnn_*_workspace_bytes` sizes the workspace. Workspace and
 *     outputs need 16-byte alignment.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and is legal inside hipGraph capture (no host round trips, no device-wide syncs; the
 *     reference launchers block ≥6x per image, hough_voting_gpu_op.cu.cc:647-784).
 *   - Return value: PCNN_OK (0) or a negative pcnn_status. Argument errors mirror the
 *     reference's OP_REQUIRES / attribute checks; nothing is launched when an error is returned.
 *     `pcnn_last_error_string()` (thread-local) describes the most recent failure.
 *   - Results are bit-identical to oracle/ (the CPU restatement of the reference GPU kernels
 *     with canonical orderings) — custom kernels are built -ffp-contract=off.
 */
#ifndef POSECNN_HIP_H_
#define POSECNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCNN_ABI_VERSION 2

/* hough_voting_gpu_op.cc:31-32 */
#define PCNN_VERTEX_CHANNELS 3
#define PCNN_MAX_ROI 128
/* rows of the scratch outputs the reference allocates: MAX_ROI * 9 (hough_voting_gpu_op.cc:94) */
#define PCNN_HOUGH_ROWS_CAPACITY (PCNN_MAX_ROI * 9)
/* average_distance_loss_op.cc:33 */
#define PCNN_POSE_CHANNELS 4

typedef enum pcnn_status {
  PCNN_OK = 0,
  PCNN_EINVAL = -1,     /* bad shape / attribute (reference: errors::InvalidArgument) */
  PCNN_EWORKSPACE = -2, /* workspace NULL, misaligned or too small */
  PCNN_EHIP = -3,       /* a HIP launch failed (reference: fprintf + exit(-1)) */
  PCNN_ENULL = -4       /* a required pointer is NULL */
} pcnn_status;

int pcnn_abi_version(void);
const char* pcnn_last_error_string(void);
const char* pcnn_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Hough voting  (REGISTER_OP("Houghvotinggpu"), hough_voting_gpu_op.cc:37-52)
 *
 *   label    int32 [B,H,W]          bottom_label
 *   vertex   f32   [B,H,W,3*C]      bottom_vertex  (u, v, log depth per class)
 *   extents  f32   [C,3]            bottom_extents
 *   meta     f32   [B,num_meta]     bottom_meta_data ([B,1,1,48]); uses fx=0, px=2, fy=4, py=5
 *   gt       f32   [num_gt,13]      bottom_gt = (batch, cls, box4, quat wxyz, trans3); may be NULL iff num_gt==0
 *
 *   top_box    f32   [rows_capacity,7]   (batch, cls, x1, y1, x2, y2, votes)
 *   top_pose   f32   [rows_capacity,7]   (1,0,0,0, tx, ty, tz)
 *   top_target f32   [rows_capacity,4*C]
 *   top_weight f32   [rows_capacity,4*C]
 *   top_domain int32 [rows_capacity]
 *   num_rois   int32 [2]   [0] = rows the reference would return (>=1: a single all-zero dummy
 *                          row when nothing was detected, hough_voting_gpu_op.cc:381-383),
 *                          [1] = true detection row count (may be 0)
 *
 * All five outputs are zero-filled by the call (reset_outputs, .cu.cc:579-588); rows beyond
 * num_rois[0] stay zero. Row order is canonical: image ascending, then maxima ascending
 * (class slot, cell index) — the reference's order is atomicAdd-dependent.
 * attrs: is_train>=0, threshold_vote, threshold_percentage, skip_pixels>=1; inlier_threshold
 * and label_threshold are the constants 0.9 / 500 of hough_voting_gpu_op.cc:356-357.
 *
 * Capacity. The reference keeps at most index_size = MAX_ROI / batch maxima per image
 * (hough_voting_gpu_op.cu.cc:733,773-774) in scratch outputs of PCNN_HOUGH_ROWS_CAPACITY rows: its
 * test loop feeds one frame at a time (lib/fcn/test.py:1867), so every frame gets 128. A batched
 * caller would silently lose detections (batch 16 -> 8 per frame). `rois_per_image`:
 *   0   the reference rule (MAX_ROI / batch); rows_capacity = PCNN_HOUGH_ROWS_CAPACITY suffices;
 *   k>0 the first k maxima of EVERY image, whatever the batch: exactly the rows `batch`
 *       single-frame calls of the reference return when no frame has more than k maxima.
 * `rows_capacity` = rows allocated in each of the five outputs,
 * >= batch * capacity * (is_train ? 9 : 1) (and >= 1 for the dummy row).
 * ------------------------------------------------------------------------------------------ */
int pcnn_hough_voting_workspace_bytes(int batch, int height, int width, int num_classes,
                                      float threshold_vote, int skip_pixels, int rois_per_image,
                                      size_t* bytes);

int pcnn_hough_voting_fwd(const int32_t* label, const float* vertex, const float* extents,
                          const float* meta, const float* gt,
                          int batch, int height, int width, int num_classes,
                          int num_meta, int num_gt,
                          int is_train, float threshold_vote, float threshold_percentage,
                          int skip_pixels, float inlier_threshold, int label_threshold,
                          int rois_per_image, int rows_capacity,
                          float* top_box, float* top_pose, float* top_target, float* top_weight,
                          int32_t* top_domain, int32_t* num_rois,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Fused vertex head -> Hough voting (replaces the pair
 * `deconv(16,16,128,8,8,'upscore_vertex') -> conv(1,1,3C,'vertex_pred')` + Houghvotinggpu of
 * lib/networks/vgg16_convs.py:152-163 without materialising `vertex_pred`).
 *   z     f32 [B, H/stride, W/stride, 3*C]   the 1x1 `vertex_pred` conv applied at low resolution
 *                                            (it commutes with the fixed bilinear deconv), no bias
 *   bias  f32 [3*C]                          `vertex_pred/biases`
 * The op behaves exactly like pcnn_hough_voting_fwd on
 * vertex = pcnn_deconv_bilinear_fwd(z, kernel, stride, bias): only the own-class (u, v, log d) of
 * the sampled foreground pixels are interpolated (same arithmetic, same bits), so the
 * [B,H,W,3C] tensor (81 MB/frame at 640x480, C=22) never exists. Everything else (outputs,
 * workspace, canonical order, attrs) as above. */
int pcnn_hough_voting_lowres_fwd(const int32_t* label, const float* z, const float* bias,
                                 int kernel, int stride, const float* extents,
                                 const float* meta, const float* gt,
                                 int batch, int height, int width, int num_classes,
                                 int num_meta, int num_gt,
                                 int is_train, float threshold_vote, float threshold_percentage,
                                 int skip_pixels, float inlier_threshold, int label_threshold,
                                 int rois_per_image, int rows_capacity,
                                 float* top_box, float* top_pose, float* top_target,
                                 float* top_weight, int32_t* top_domain, int32_t* num_rois,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostics for tests: byte offsets of intermediate buffers inside the workspace.
 * offsets[8] = { hough space f32 [B][C-1][H*W] (SIZE_MAX unless threshold_vote > 0), pixel records
 * (48 B each), class totals i32 [B][C], slot classes i32 [B][C], slot counts i32 [B], record
 * offsets i32 [B][C], row maxima int2 [B][C-1][H] (votes, first cell), record capacity per image (a count) }. */
int pcnn_hough_voting_debug_layout(int batch, int height, int width, int num_classes,
                                   float threshold_vote, int skip_pixels, int rois_per_image,
                                   size_t* offsets);

/* HoughvotinggpuGrad (hough_voting_gpu_op.cc:440-484, set_gradients .cu.cc:608-612): zeros. */
int pcnn_hough_voting_bwd(float* grad_label, float* grad_vertex,
                          int batch, int height, int width, int num_classes, void* stream);

/* ------------------------------------------------------------------------------------------
 * ROI pooling  (REGISTER_OP("RoiPool"), roi_pooling_op.cc:29-38)
 *   data  f32 [B,H,W,C] NHWC;  rois f32 [R,roi_cols] with roi_cols>=6: (batch, cls, x1,y1,x2,y2,..)
 *   top   f32 [R,PH,PW,C]  (C -> 1 when pool_channel==1);  argmax int32 same shape, may be NULL
 *   argmax is image-relative: (h*W+w)*C + c, -1 for an empty bin (roi_pooling_op_gpu.cu.cc:75-99).
 * ------------------------------------------------------------------------------------------ */
int pcnn_roi_pool_fwd(const float* data, const float* rois,
                      int batch, int height, int width, int channels,
                      int num_rois, int roi_cols,
                      int pooled_height, int pooled_width, float spatial_scale, int pool_channel,
                      float* top, int32_t* argmax, void* stream);

/* RoiPoolGrad (roi_pooling_op.cc:384-461, roi_pooling_op_gpu.cu.cc:135-254): bottom_diff [B,H,W,C]. */
int pcnn_roi_pool_bwd(const float* top_diff, const float* rois, const int32_t* argmax,
                      int batch, int height, int width, int channels,
                      int num_rois, int roi_cols,
                      int pooled_height, int pooled_width, float spatial_scale, int pool_channel,
                      float* bottom_diff, void* stream);

/* Fused pool5 + pool4 -> add ('pool_score', vgg16_convs.py:177-187): two RoiPool calls with
 * pool_channel=0 and their element-wise sum, without materialising either pooled tensor.
 * out f32 [R,PH,PW,C]; data_a [B,Ha,Wa,C] with scale_a, data_b [B,Hb,Wb,C] with scale_b.
 * num_rows_dev (device int32[1], may be NULL): when `rois` is the capacity-sized buffer of the
 * sync-free Hough op, the true row count; rows at or past it pool to 0 without touching the maps. */
int pcnn_roi_pool_add2_fwd(const float* data_a, int height_a, int width_a, float scale_a,
                           const float* data_b, int height_b, int width_b, float scale_b,
                           const float* rois, int batch, int channels, int num_rois, int roi_cols,
                           int pooled_height, int pooled_width, const int32_t* num_rows_dev,
                           float* out, void* stream);
/* The same with the rows at or past *num_rows_dev (required) left UNTOUCHED instead of zero-filled: for a consumer
 * that masks those rows itself — pcnn_fc_rows_fwd / pcnn_fc_skinny_fwd take the same device-side count — three quarters
 * of a capacity-sized buffer (235 MB of zeros per 16-frame step) need not be written at all. */
int pcnn_roi_pool_add2_live_fwd(const float* data_a, int height_a, int width_a, float scale_a,
                                const float* data_b, int height_b, int width_b, float scale_b,
                                const float* rois, int batch, int channels, int num_rois, int roi_cols,
                                int pooled_height, int pooled_width, const int32_t* num_rows_dev,
                                float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Hard label  (REGISTER_OP("Hardlabel"), hard_label_op.cc:30-35; GPU semantics .cu.cc:17-29)
 *   prob f32 [N,C] (N = B*H*W), gt int32 [N] in {-1, 0..C-1}; out f32 [N,C]; threshold > 0.
 * ------------------------------------------------------------------------------------------ */
int pcnn_hard_label_fwd(const float* prob, const int32_t* gt, int64_t num_pixels, int num_classes,
                        float threshold, float* out, void* stream);

/* HardlabelGrad (hard_label_op_gpu.cu.cc:55-85): both gradients are zero. */
int pcnn_hard_label_bwd(float* grad_prob, float* grad_gt, int64_t num_pixels, int num_classes,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Average distance loss  (REGISTER_OP("Averagedistance"), average_distance_loss_op.cc:38-47)
 *   prediction/target/weight f32 [R,4*C]; point f32 [C,P,3]; symmetry f32 [C]; margin >= 0
 *   loss f32 [1]; bottom_diff f32 [R,4*C]
 *   num_rows_dev (device int32[1], may be NULL): the op's row count when the three inputs are
 *   capacity-sized buffers of the sync-free Hough op — the loss and gradient are normalised by
 *   min(R, *num_rows_dev) * P like the reference's exactly-sized call (:190,:203), rows past it
 *   contribute nothing and get a zero gradient. NULL: R rows.
 * ------------------------------------------------------------------------------------------ */
int pcnn_average_distance_workspace_bytes(int num_rois, int num_classes, int num_points,
                                          size_t* bytes);

int pcnn_average_distance_fwd(const float* prediction, const float* target, const float* weight,
                              const float* point, const float* symmetry,
                              int num_rois, int num_classes, int num_points, float margin,
                              const int32_t* num_rows_dev, float* loss, float* bottom_diff,
                              void* workspace, size_t workspace_bytes, void* stream);

/* AveragedistanceGrad (average_distance_loss_op_gpu.cu.cc:347-377): out = grad[0] * bottom_diff. */
int pcnn_average_distance_bwd(const float* grad, const float* bottom_diff, int num_rois,
                              int channels, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backprojecting  (REGISTER_OP("Backproject"), backprojecting_op.cc:30-43)
 *   data f32 [B,H,W,Cd]; label f32 [B,H,W,Cl]; depth f32 [B,H,W]; meta f32 [B,num_meta] (>=48);
 *   label_3d f32 [B,G,G,G,Cl];  top_data, top_flag f32 [B,G,G,G,Cd]; top_label f32 [B,G,G,G,Cl]
 * ------------------------------------------------------------------------------------------ */
int pcnn_backproject_fwd(const float* data, const float* label, const float* depth,
                         const float* meta, const float* label_3d,
                         int batch, int height, int width, int channels, int num_classes,
                         int num_meta, int grid_size, int kernel_size, float threshold,
                         float* top_data, float* top_label, float* top_flag, void* stream);

/* The same op with caller-owned scratch (added in round 5; no existing prototype changed). The reference's launcher has none (BackprojectForwardLaucher,
 * backprojecting_op_gpu.cu.cc:129-155: every (voxel, channel) thread rescans its depth window); here the workspace holds
 * the (min, max) depth of every (2k+1)^2 window that meets the image — [B][H+2k][W+2k] pairs, built by one small launch in
 * front of the voxel kernel — so that a voxel whose Z1 cannot match any pixel of its window skips the scan. Results are
 * bit-identical with and without it (workspace NULL / 0 bytes = pcnn_backproject_fwd). kernel_size > 3 needs 0 bytes. */
int pcnn_backproject_workspace_bytes(int batch, int height, int width, int kernel_size, size_t* bytes);
int pcnn_backproject_ws_fwd(const float* data, const float* label, const float* depth,
                            const float* meta, const float* label_3d,
                            int batch, int height, int width, int channels, int num_classes,
                            int num_meta, int grid_size, int kernel_size, float threshold,
                            float* top_data, float* top_label, float* top_flag,
                            void* workspace, size_t workspace_bytes, void* stream);

/* BackprojectGrad (backprojecting_op_gpu.cu.cc:159-244): bottom_diff [B,H,W,Cd] from top_diff [B,G,G,G,Cd]. */
int pcnn_backproject_bwd(const float* top_diff, const float* depth, const float* meta,
                         int batch, int height, int width, int channels, int num_meta,
                         int grid_size, float* bottom_diff, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused label head epilogue (vgg16_convs.py:140-149, network.py:432-434,474-488):
 *   score f32 [N,C] (already bias+ReLU'd) -> prob_normalized f32 [N,C] (softmax, max-subtracted),
 *   label_2d int32 [N] (first argmax of prob_normalized). prob may be NULL (label only).
 * ------------------------------------------------------------------------------------------ */
int pcnn_softmax_argmax_fwd(const float* score, int64_t num_pixels, int num_classes,
                            float* prob, int32_t* label, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fixed bilinear "deconv" (lib/networks/network.py:141-157 make_deconv_filter, :207-222 deconv):
 * tf.nn.conv2d_transpose 'SAME', stride s, kernel k, with the diagonal bilinear filter, NHWC.
 *   in f32 [B,H,W,C] -> out f32 [B,H*s,W*s,C];  out = deconv(in) [+ add1] [+ add2] [+ bias[C]] [ReLU]
 *   (add1/add2/bias may be NULL; they fuse `add_score` (vgg16_convs.py:135-136) and the bias of a
 *   1x1 conv hoisted in front of the deconv).
 * pcnn_upscore_softmax_argmax_fwd: label head epilogue without materialising the full-resolution
 *   score: score = [ReLU](deconv(z) + bias), prob = softmax(score) (network.py:474-488),
 *   label = first argmax(prob) (:432-434). z f32 [B,H,W,C]; score_out/prob f32 [B,H*s,W*s,C] or
 *   NULL; label int32 [B,H*s,W*s].
 * ------------------------------------------------------------------------------------------ */
int pcnn_deconv_bilinear_fwd(const float* in, int batch, int height, int width, int channels,
                             int kernel, int stride, const float* add1, const float* add2,
                             const float* bias, int relu, float* out, void* stream);

/* Gradient of pcnn_deconv_bilinear_fwd w.r.t. `in` (the transposed interpolation; TF's
 * conv2d_transpose gradient for the fixed filter): grad_out f32 [B,H*s,W*s,C] -> grad_in f32
 * [B,H,W,C]; sum order: output rows ascending, output columns ascending. */
int pcnn_deconv_bilinear_bwd(const float* grad_out, int batch, int height, int width, int channels,
                             int kernel, int stride, float* grad_in, void* stream);

/* y = [ReLU](x + bias[c]) over [num_pixels, channels] NHWC rows; y may alias x. The bias_add + relu
 * of Network.conv (network.py:181-187) in one pass. */
int pcnn_bias_act_fwd(const float* x, const float* bias, int64_t num_pixels, int channels, int relu,
                      float* y, void* stream);

/* First layer of a VGG tower (conv1_1 / conv1_1_p, vgg16_convs.py:36,53): 3x3, stride 1, SAME,
 * 3 input channels, with bias_add + ReLU (network.py:181-187) fused.
 *   x f32 [B,H,W,3];  weights f32 [3,3,3,Cout] = (ky, kx, ci, co), the layout of the TF variable
 *   `conv1_1/weights` (network.py:169-180);  bias f32 [Cout];  Cout % 64 == 0
 *   y f32 [B,H,W,Cout] = [ReLU](conv(x) + bias); acc = fma(w, x, acc) over (ky, kx, ci) ascending.
 * Bound by writing y once (HBM); the library GEMM path + separate bias pass is 3.5x slower here. */
int pcnn_conv3x3_c3_fwd(const float* x, const float* weights, const float* bias, int batch,
                        int height, int width, int out_channels, int relu, float* y, void* stream);

/* pcnn_conv3x3_c3_fwd fused with the F(4x4,3x3) input transform of the NEXT 3x3 layer (conv1_1 ->
 * conv1_2, vgg16_convs.py:36-37): v f32 [36][T][Cout] = pcnn_winograd43_input_fwd(pcnn_conv3x3_c3_fwd(x)),
 * bit for bit, without the [B,H,W,Cout] activation in between. groups >= 1: weights f32
 * [groups][3,3,3,Cout], bias [groups][Cout], image b uses set b / (batch / groups) (both towers of an
 * RGB-D network — conv1_1 and conv1_1_p, vgg16_convs.py:36,53 — in one launch). */
int pcnn_conv3x3_c3_winograd43_fwd(const float* x, const float* weights, const float* bias, int batch,
                                   int height, int width, int out_channels, int groups, int relu,
                                   float* v, void* stream);

/* The same, fed with the frames as the sensor delivers them instead of the f32 blobs of lib/fcn/test.py:56-74:
 *   color_bgr uint8 [num_color][H][W][3] (OpenCV order) and / or depth uint16 [num_depth][H][W]; either count may be 0.
 *   The blob values — (float)((double)bgr - pixel_means[c]) and (float)((double)(clip(depth / 2000, 0, 1) * 255) - pixel_means[c])
 *   tiled over the 3 channels, i.e. numpy's float32 -= float64 of `_get_image_blob` — are formed while the kernel stages its
 *   input window, so v is bit-identical to pcnn_conv3x3_c3_winograd43_fwd on the host-built blobs, while 0.9 MB (colour) /
 *   0.6 MB (depth) per 640x480 frame cross PCIe instead of 3.7 MB each and the blobs never exist in HBM.
 *   Colour frames come first in v's tile order and use filter set 0; depth frames follow and use the next set
 *   (weights [sets][3,3,3,Cout], bias [sets][Cout], sets = (num_color > 0) + (num_depth > 0)). pixel_means: 3 finite doubles
 *   behind a HOST pointer, read when the call is made and passed to the kernel by value (the caller may free or change them
 *   as soon as the call returns; non-finite values are refused with PCNN_EINVAL). */
int pcnn_conv3x3_c3_winograd43_raw_fwd(const uint8_t* color_bgr, int num_color, const uint16_t* depth, int num_depth,
                                       const double* pixel_means, const float* weights, const float* bias, int height,
                                       int width, int out_channels, int relu, float* v, void* stream);

/* conv1_1 -> conv1_2 -> pool1 of a VGG16 tower (vgg16_convs.py:36-38: conv(3,3,64) -> conv(3,3,64) -> max_pool(2,2)) in ONE kernel:
 *   y_pool [B, H/2, W/2, 64] = max_pool_2x2(act2(conv3x3(act1(conv3x3(x, w1) + b1), W2) + b2)),  H and W multiples of 16.
 *   x f32 [B,H,W,3] blobs; w1 [groups][3,3,3,64] (ky,kx,ci,co), b1 [groups][64]; ut2 = the F(4x4,3x3) transform of conv1_2's
 *   filter, U^T [groups][36][64 out][64 in] (what pcnn_winograd43_conv_fwd takes), b2 [groups][64]; image b uses set
 *   b / (B / groups). Bit-identical to pcnn_conv3x3_c3_winograd43_fwd followed by pcnn_winograd43_conv_fwd(pool = 1), without
 *   the 2.25 x 78.6 MB per frame of transformed input between them ever touching HBM. The _raw form takes the frames as
 *   pcnn_conv3x3_c3_winograd43_raw_fwd does (colour frames use set 0, depth frames the next).
 *   ut2_layout 1: the same numbers fragment-major, [groups][36][4][4][64][4] with element (k, w, g, lane, i) =
 *   U^T[k][16 w + (lane & 15)][16 g + 4 (lane >> 4) + i] — every B-operand load of a wave is then 1 KB of contiguous memory
 *   (the kernel re-reads the whole bank from L2 for every 16 tiles: DESIGN.md §3.2b). */
int pcnn_conv1_1_conv1_2_fused_fwd(const float* x, const float* w1, const float* b1, const float* ut2, int ut2_layout,
                                   const float* b2, int batch, int height, int width, int groups, int relu1, int relu2,
                                   float* y_pool, void* stream);
int pcnn_conv1_1_conv1_2_fused_raw_fwd(const uint8_t* color_bgr, int num_color, const uint16_t* depth, int num_depth,
                                       const double* pixel_means, const float* w1, const float* b1, const float* ut2,
                                       int ut2_layout, const float* b2, int height, int width, int relu1, int relu2,
                                       float* y_pool, void* stream);

/* Data transforms of a Winograd F(2x2,3x3) evaluation of `Network.conv` for the deep 3x3 / stride 1 /
 * SAME layers of the trunk (network.py:159-187; vgg16_convs.py:42-52). All f32.
 *   pcnn_winograd_input_fwd : x f32 [B,H,W,C] (H, W even, C % 4 == 0) -> v f32 [16][T][C],
 *       T = B*(H/2)*(W/2) tiles in (b, ty, tx) order, v[4i+j] = (B^T d B)[i][j] of the 4x4 input
 *       patch at (2ty-1, 2tx-1) (zero outside the image).
 *   (caller) m[k] = v[k] (T x Cin) * u[k] (Cin x Cout) for k = 0..15 with u[4i+j] = (G g G^T)[i][j].
 *   pcnn_winograd_output_fwd: m f32 [16][T][C] -> y = [ReLU](A^T m A + bias) as f32 [B,H,W,C], or with
 *       pool != 0 its 2x2 max-pool f32 [B,H/2,W/2,C] (an output tile is one pooling window).
 * Transform order: rows, then columns, sums left to right, bias last. */
int pcnn_winograd_input_fwd(const float* x, int batch, int height, int width, int channels, float* v,
                            void* stream);
int pcnn_winograd_output_fwd(const float* m, const float* bias, int batch, int height, int width,
                             int channels, int relu, int pool, float* y, void* stream);

/* The same with F(4x4,3x3) (36 multiplies per 16 outputs; interpolation points 0, +-1, +-2, inf):
 * T = B*ceil(H/4)*ceil(W/4) tiles, v / m are f32 [36][T][C] with index 6i+j; any H, W >= 1 (partial
 * tiles read zeros and store only inside the image); pool needs even H, W. */
int pcnn_winograd43_input_fwd(const float* x, int batch, int height, int width, int channels, float* v,
                              void* stream);
int pcnn_winograd43_output_fwd(const float* m, const float* bias, int batch, int height, int width,
                               int channels, int relu, int pool, float* y, void* stream);

/* F(4x4,3x3) contractions + output transform in ONE fp32-MFMA kernel (csrc/wino_mfma.hip), for every
 * 3x3 / stride 1 / SAME layer of the trunk with Cin % 64 == 0 and Cout % 64 == 0 (conv1_2 ... conv5_3
 * and the depth tower's twins, vgg16_convs.py:37-52,54-67):
 *   y = [ReLU](A^T (v[k] . u[k]) A + bias)  — the transform-domain product never reaches HBM.
 *   v    f32 [36][T][Cin]           from pcnn_winograd43_input_fwd / pcnn_conv3x3_c3_winograd43_fwd,
 *                                   T = batch * ceil(H/4) * ceil(W/4)
 *   ut   f32 [groups][36][Cout][Cin]  filter transforms TRANSPOSED: ut[g][6i+j][co][ci] = (G g_g G^T)[i][j]
 *   bias f32 [groups][Cout]
 *   groups >= 1: image b uses filter set b / (batch / groups) — the colour and depth towers of an RGB-D
 *                network as one launch (batch % groups == 0)
 *   pool 0: y f32 [batch,H,W,Cout];  1: y = max_pool_2x2 f32 [batch,H/2,W/2,Cout] only;
 *        2: both (y and y_pool), for a layer like conv4_3 whose un-pooled output is read as well.
 *   workspace (optional, pcnn_winograd43_conv_workspace_bytes; 0 bytes for launches that fill the chip): lets a
 *        small launch (batch-1 conv4_x / conv5_x: tens of workgroups of 288 stages each) split Cin over up to 8
 *        workgroups per output block; the partial outputs are summed in a fixed order before bias / ReLU / pooling.
 *        NULL or too small: no split (same result up to f32 summation order).
 *   Environment (experiments and the variant-equality test; every variant yields the same bits): PCNN_WINO_MODE, an integer
 *        read once per process — bit 0 the channel-block-major workgroup map, bit 1 the one-wave-per-SIMD kernel
 *        (wino43_mfma_w1_kernel, v_mfma_f32_32x32x2_f32, launches without a Cin split), bit 3 the round-3 accumulator zeroing.
 *        Unset: the library's choice per launch. */
int pcnn_winograd43_conv_workspace_bytes(int batch, int height, int width, int in_channels, int out_channels,
                                         int groups, size_t* bytes);
int pcnn_winograd43_conv_fwd(const float* v, const float* ut, const float* bias, int batch, int height,
                             int width, int in_channels, int out_channels, int groups, int relu,
                             int pool, float* y, float* y_pool, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Fully connected layer over a capacity-sized row buffer (`Network.fc`, network.py:392-422; fc6 / fc7 of
 * vgg16_convs.py:188-192 behind the sync-free Hough layer): y[m] = [ReLU](x[m] . W + bias) for the rows
 * m < min(rows_capacity, *num_rows_dev); rows at or past the count are written as zeros without touching
 * an operand (a library GEMM would need the count on the host, or compute every padded row).
 *   x    f32 [rows_capacity][in_features]      in_features % 64 == 0, >= 128
 *   wt   f32 [out_features][in_features]       the TF weight variable [in, out] TRANSPOSED; out_features % 64 == 0
 *   bias f32 [out_features];  num_rows_dev device int32[1] or NULL (= rows_capacity);  y f32 [rows_capacity][out_features]
 *   addend f32 [rows_capacity][out_features] or NULL: added before the ReLU — a 1x1 convolution over a channel
 *        concatenation (`concat` + `conv(1, 1, ...)` of the RGB-D heads, vgg16_convs.py:104-113) is the sum of
 *        two such products, one per tower, so the concatenated tensor is never built
 *   workspace (optional, pcnn_fc_rows_workspace_bytes): lets launches with few live rows split K over up to 8
 *        workgroups per output block — decided on the device from *num_rows_dev — and sum the partial products
 *        in a fixed order; without it (NULL / too small) every block runs its whole K (same results up to f32
 *        summation order, slower when the live rows are few: fc6 at 75 rows is bound by HBM latency then)
 * fp32 MFMA (exact f32), sum over k in ascending order within a lane-fixed interleave. */
int pcnn_fc_rows_workspace_bytes(int rows_capacity, int in_features, int out_features, size_t* bytes);
/* The same product for a layer whose width is no multiple of 64 (round 5: fc8, 4096 -> 4 C = 88, vgg16_convs.py:192-193, at
 * more rows than pcnn_fc_skinny_fwd takes): wt [out_padded][in_features] and bias [out_padded] are zero-padded to a multiple
 * of 64, y / y_tanh are [rows_capacity][out_features] (out_features % 4 == 0). activation 0 none, 1 ReLU, 2 tanh: y = the
 * linear output, y_tanh = tanh(y) (fc8 and poses_tanh in one launch). Rows at or past *num_rows_dev: zeros. */
int pcnn_fc_rows_cols_fwd(const float* x, const float* wt, const float* bias, int rows_capacity,
                          int in_features, int out_padded, int out_features, int activation,
                          const int32_t* num_rows_dev, float* y, float* y_tanh, void* stream);
/* Two layers that read the same rows as one product (round 5: score_conv4 + score_conv4_vertex on conv4_3, vgg16_convs.py:128-133,
 * 151-157): wt [out_a + out_b][in_features] (the two filters one after the other), bias likewise, y_a [rows][out_a] and y_b
 * [rows][out_b] with a ReLU flag each; out_a, out_b multiples of 64. Bit-identical to two pcnn_fc_rows_fwd calls. */
int pcnn_fc_rows_split_fwd(const float* x, const float* wt, const float* bias, int rows_capacity, int in_features,
                           int out_a, int out_b, int relu_a, int relu_b, const int32_t* num_rows_dev,
                           float* y_a, float* y_b, void* stream);
int pcnn_fc_rows_fwd(const float* x, const float* wt, const float* bias, int rows_capacity,
                     int in_features, int out_features, int relu, const int32_t* num_rows_dev,
                     const float* addend, float* y, void* workspace, size_t workspace_bytes, void* stream);

/* The same layer when only a handful of rows exist (the single-frame loop of lib/fcn/test.py:1867-1888: one image,
 * <= 21 detections): at <= 32 rows fc6 is a 411 MB weight stream, not a GEMM. Waves pull their weight rows straight
 * from HBM into registers (no LDS ring, no barriers), the rows ride along as the A operand of the fp32 MFMA, K is
 * split over grid.y and the LAST workgroup of a column group sums the partial products in ascending order
 * (deterministic), adds the bias and applies the activation — one launch.
 *   x f32 [rows_capacity <= 32][in_features % 16 == 0];  wt f32 [out_features][in_features] (any out_features >= 1:
 *   fc8's 4 * num_classes included);  activation 0 none, 1 ReLU (`relu_layer`), 2 tanh: y_act = tanh(y), y stays
 *   linear (fc8 -> poses_tanh, vgg16_convs.py:192-193; y_act may be NULL)
 *   num_rows_dev device int32[1] or NULL;  rows at or past it are written as zeros
 *   workspace / counters: pcnn_fc_skinny_workspace_bytes gives the bytes and the number of int32 tickets; the
 *   tickets must be ZERO on entry and are zero again on exit (zero them once; no other launch may share them
 *   concurrently). */
int pcnn_fc_skinny_workspace_bytes(int rows_capacity, int in_features, int out_features, size_t* bytes,
                                   int* num_counters);
int pcnn_fc_skinny_fwd(const float* x, const float* wt, const float* bias, int rows_capacity, int in_features,
                       int out_features, int activation, const int32_t* num_rows_dev, float* y, float* y_act,
                       void* workspace, size_t workspace_bytes, int32_t* counters, int num_counters, void* stream);

/* The 1/8-resolution part of a PoseCNN head in one launch (vgg16_convs.py:128-142 label head, :151-163 vertex head,
 * in the commuted order DESIGN.md §3.2 "fused_heads" explains):
 *   add_out = score4 + deconv_{kernel,stride}(score5) [+ planted]     (`add_score` / `dropout` at keep_prob 1)
 *   z       = add_out . W                                             (the 1x1 `score` / `vertex_pred` product, no bias)
 *   score4 f32 [B,h,w,units];  score5 f32 [B,h/stride,w/stride,units];  planted f32 [B,h,w,units] or NULL
 *   weights_t f32 [units][out_channels] (the TF variable [1,1,units,out] as it is);  z f32 [B,h,w,out_channels]
 * The deconv is the fixed bilinear filter of network.py:141-157 in the canonical tap order of csrc/bilinear.h
 * (same bits as pcnn_deconv_bilinear_fwd); the product accumulates k ascending with fused multiply-adds. */
int pcnn_head_lowres_fwd(const float* score4, const float* score5, const float* planted, const float* weights_t,
                         int batch, int height, int width, int units, int out_channels, int kernel, int stride,
                         float* add_out, float* z, void* stream);
/* The same step with the 1x1 product on the matrix cores, for many pixels per launch (round 5): `weights_nk` is the filter
 * N-MAJOR, [ceil(out_channels / 16) * 16][units] with K contiguous and zero rows past out_channels; units % 16 == 0,
 * out_channels <= 96. add_out has the bits of pcnn_head_lowres_fwd; z sums K in one fixed (matrix-core) order. */
int pcnn_head_lowres_mfma_fwd(const float* score4, const float* score5, const float* planted,
                              const float* weights_nk, int batch, int height, int width, int units, int out_channels,
                              int kernel, int stride, float* add_out, float* z, void* stream);

/* lib/fcn/test.py:197-211 on the device, minus the NMS: detection rows for the host / the all-gather.
 *   det_rows[i] = rois[i s][0:7] | poses_tanh[i s][4 c : 4 c + 4] | top_pose[i s][4:7],  c = int(rois[i s][1]) clamped
 *   to [0, num_classes), for i s < *num_rows_dev; zeros after.  s = row_stride: 9 in training mode (the un-jittered
 *   first row of each group, hough_voting_gpu_op.cu.cc:440-466), else 1.  det_count[0] = *num_rows_dev / s.
 *   rois f32 [rows][7], poses_tanh f32 [rows][4 num_classes], top_pose f32 [rows][7], det_rows f32 [ceil(rows/s)][14] */
int pcnn_det_assemble_fwd(const float* rois, const float* poses_tanh, const float* top_pose,
                          const int32_t* num_rows_dev, int rows, int row_stride, int num_classes, float* det_rows,
                          int32_t* det_count, void* stream);
/* The same rows as the block one rank hands to the detection all-gather: det_block [ceil(rows / row_stride) + 1][14],
 * column 0 of the live rows shifted by `frame_offset` (global frame index = rank * B + local), last row = (count, 0, ...). */
int pcnn_det_assemble_packed_fwd(const float* rois, const float* poses_tanh, const float* top_pose,
                                 const int32_t* num_rows_dev, int rows, int row_stride, int num_classes,
                                 float frame_offset, float* det_block, int32_t* det_count, void* stream);
/* poses_pred = l2_normalize(poses_tanh * poses_weight, dim 1) (vgg16_convs.py:195-197; network.py:573-577: x * rsqrt(max(sum
 * x^2, 1e-12))) on [rows][cols <= 256] buffers; rows at or past *num_rows_dev (NULL: none) are zeros. Fixed summation order. */
int pcnn_pose_l2_normalize_fwd(const float* poses_tanh, const float* poses_weight, const int32_t* num_rows_dev,
                               int rows, int cols, float* poses_pred, void* stream);

/* pcnn_winograd43_output_fwd writing BOTH the activation y f32 [B,H,W,C] and its 2x2 max-pool y_pool f32
 * [B,H/2,W/2,C] in one pass (conv4_3 -> pool4, whose un-pooled output score_conv4 and roi_pool read too). */
int pcnn_winograd43_output_both_fwd(const float* m, const float* bias, int batch, int height, int width,
                                    int channels, int relu, float* y, float* y_pool, void* stream);

/* y[b,oy,ox,c] = max over the 2x2 window of [ReLU](x + bias[c]): the `conv -> max_pool(2,2,2,2)`
 * pairs of the VGG trunk (vgg16_convs.py:36-49; network.py:181-187 + :189-196) from the raw
 * convolution output x f32 [B,H,W,C] (H, W even) to y f32 [B,H/2,W/2,C], same bits as
 * max_pool(bias_act(x)) without the intermediate tensor. */
int pcnn_bias_relu_pool2_fwd(const float* x, const float* bias, int batch, int height, int width,
                             int channels, int relu, float* y, void* stream);

int pcnn_upscore_softmax_argmax_fwd(const float* z, const float* bias, int batch, int height,
                                    int width, int num_classes, int kernel, int stride, int relu,
                                    float* score_out, float* prob, int32_t* label, void* stream);

/* pcnn_upscore_softmax_argmax_fwd that also evaluates the "Hardlabel" op (lib/hard_label_layer/hard_label_op.cc:143-188,
 * hard_label_op_gpu.cu.cc:17-29) on the probabilities it has just computed — the training graph feeds that op
 * prob_normalized and gt_label_2d (vgg16_convs.py:148-149): hard f32 [B,H*s,W*s,C] = pcnn_hard_label_fwd(prob, gt,
 * threshold), same bits, without a second pass over the probabilities. gt int32 [B,H*s,W*s]; threshold > 0 (the op's
 * attribute check, hard_label_op.cc:150-155); gt and hard must not be NULL; prob / score_out may be. */
int pcnn_upscore_softmax_argmax_hard_fwd(const float* z, const float* bias, int batch, int height,
                                         int width, int num_classes, int kernel, int stride, int relu,
                                         float* score_out, float* prob, int32_t* label,
                                         const int32_t* gt, float threshold, float* hard, void* stream);

/* ------------------------------------------------------------------------------------------
 * smooth_l1_loss_vertex (lib/fcn/train.py:564-573), the vertex regression loss of the training
 * graph, over n = B*H*W*3C elements of pred / target / weight (f32):
 *   diff = w*(pred - target);  in = |diff| < 1/sigma^2 ? diff^2*sigma^2/2 : |diff| - 0.5/sigma^2
 *   out[0] = loss = sum(in) / (sum(w) + 1e-10),  out[1] = sum(in),  out[2] = sum(w)   (f32 [3])
 * bwd: grad_pred = upstream[0] * w * (|diff| < 1/sigma^2 ? sigma^2*diff : sign(diff)) / (sum(w) + 1e-10)
 *   (`out` is the forward's; upstream f32 [1] on the device, NULL = 1). The reduction order is
 *   fixed so the loss is reproducible bit for bit.
 * ------------------------------------------------------------------------------------------ */
int pcnn_smooth_l1_vertex_workspace_bytes(size_t* bytes);
int pcnn_smooth_l1_vertex_fwd(const float* pred, const float* target, const float* weight, int64_t n,
                              float sigma, float* out, void* workspace, size_t workspace_bytes,
                              void* stream);
int pcnn_smooth_l1_vertex_bwd(const float* pred, const float* target, const float* weight,
                              const float* out, const float* upstream, int64_t n, float sigma,
                              float* grad_pred, void* stream);

/* ------------------------------------------------------------------------------------------
 * Depth-based pose refinement, first slice: the projective point-to-plane ICP that
 * Synthesizer::refinePose runs per object (lib/synthesize/synthesize.cpp:2020-2026 -> df::icp,
 * lib/kinect_fusion/src/optimization/icp.cpp:20-106 with the per-pixel kernel icp.cu:25-136), called from
 * lib/fcn/test.py:1925-1933 through Synthesizer::solveICP (the rest of solveICP: next block).
 *
 * pcnn_icp_backproject_fwd: synthesize.cpp:2139-2155 + df backproject (src/image/backprojection.cu:10-27, Poly3 camera
 *   with k = 0): vertex_map[y][x] = ((x - px)/fx d, (y - py)/fy d, d), d = depth[y][x] / factor_depth where
 *   label[y][x] == obj_id (label NULL: everywhere), else 0.   depth uint16 [H][W], label int32 [H][W] or NULL,
 *   vertex_map f32 [H][W][3].
 * pcnn_icp_refine_fwd: df::icp for num_objects independent problems in one call.
 *   live_vertices f32 [N][H][W][3] (the backprojected, masked depth); pred_vertices / pred_normals f32
 *   [N][H][W][pred_channels] (3 or 4 floats per pixel as the renderer's RGB(A) float textures; a predicted depth
 *   outside [z_near, z_far] — the render's background — marks a pixel without model surface)
 *   per iteration, per pixel (x, y) of the predicted maps: p = update * pred_vertex; (u, v) = round(project(p));
 *   skipped unless 2 < u < W-3, 2 < v < H-3, live depth in range, -ray.normal >= 0.1, |n.(live - p)| <= max_error;
 *   J = (1/live_z) [n^T | (p x n)^T], r = (1/live_z) n.(live - p); solve (sum J^T J) x = sum J^T r;
 *   update = exp(x) * update.                 update f64 [N][12] (row-major 3x4, starts at the identity)
 *   stats f32 [N][iterations][2] = (inliers, sum r^2) before each step, or NULL.
 *   Reductions in a fixed order (256-pixel halving trees in f32; block sums in f64, 8 contiguous segments ascending), solve / exp in f64:
 *   bit-identical to oracle_icp_refine. No host synchronisation between iterations.
 * ------------------------------------------------------------------------------------------ */
int pcnn_icp_backproject_fwd(const uint16_t* depth, const int32_t* label, int height, int width, int obj_id,
                             float factor_depth, float fx, float fy, float px, float py, float* vertex_map,
                             void* stream);
int pcnn_icp_refine_workspace_bytes(int num_objects, int height, int width, size_t* bytes);
int pcnn_icp_refine_fwd(const float* live_vertices, const float* pred_vertices, const float* pred_normals,
                        int num_objects, int height, int width, int pred_channels, float fx, float fy, float px,
                        float py, float z_near, float z_far, float max_error, int iterations, double* update,
                        float* stats, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Depth-based pose refinement, second slice: the rest of Synthesizer::solveICP (lib/synthesize/synthesize.cpp:2052-2380).
 *
 * pcnn_render_mesh_fwd replaces the two OpenGL passes of solveICP / refinePose (:2104-2136, :1972-1991; shaders
 *   lib/kinect_fusion/shaders/vertsAndNorms.{vert,frag}, canonicalVerts.{vert,frag}; projection :2088 = pixel centres at
 *   integer (u, v), u = fx X / Z + px): a triangle mesh rendered at num_poses poses in one call.
 *   vertices f32 [Nv][3] (object frame), normals f32 [Nv][3] or NULL, faces int32 [Nf][3], poses f32 [N][12] (row-major
 *   3x4, camera <- object; device memory like everything else).
 *   out_vertices f32 [N][H][W][4] = (camera-frame surface point, 1), out_normals f32 [N][H][W][4] = (per-vertex normals
 *   rotated and normalised per vertex, interpolated, 0), out_canonical f32 [N][H][W][3] = object-frame surface point with
 *   canon_x_offset (the model index, synthesize.cpp:270) added to x; each may be NULL; NaN where no surface is hit
 *   (the reference clears to NaN, :2113). Perspective-correct interpolation, z-buffer on (depth, face index), depth
 *   kept in [z_near, z_far]; triangles with a vertex in front of z_near are dropped. Deterministic and bit-identical to
 *   oracle_render_mesh. workspace: 8 bytes per pixel and pose (pcnn_render_mesh_workspace_bytes), 8-byte aligned.
 * pcnn_icp_center_fwd (:2157-2207): over the pixels with label == obj_id, live depth > 0 and a rendered canonical
 *   vertex m (x - round(x) strips the model index): mask = 1 (the (depth point, model point) pairs of the scoring step);
 *   those with |n.(d - v)| < max_error add (d - m) to the translation estimate.
 *   sums f64 [5] = (sum dx, sum dy, sum dz, count, valid pairs); mask uint8 [H][W]. Fixed-order reduction like pcnn_icp_refine_fwd.
 * pcnn_icp_score_fwd (:2302-2343, the SegICP score): for each hypothesis pose (f32 [M][12]) every model point of the mask,
 *   moved by the pose, marks its nearest depth point strictly inside `radius` (0.01 in the reference; ties to the lower
 *   pixel index); hits int32 [M] = distinct marked depth points (the reference's score times the number of model points).
 *   The reference searches a kd-tree built on the host; here the depth points are still an image, so the search is a window
 *   around the projection (exact: a point within r of (X, Y, Z) projects within fx r (1 + |X/Z|) / (Z - r) pixels).
 *   workspace: one bit per pixel and hypothesis (pcnn_icp_score_workspace_bytes).
 * pcnn_icp_polish_fwd (Synthesizer::poseWithOpt :2529-2570 with the objective optEnergy :2476-2526; refinePose case 0):
 *   Nelder-Mead over an update pose x = (quaternion wxyz, translation) in the box (1,0,0,0,0,0,0) +- (0.1 x4, 0.01, 0.01, 0.1),
 *   minimising the mean distance |x * pred_vertex - live_vertex| over the pixels with label == obj_id whose live and moved
 *   depths lie inside (z_near, z_far); max_evaluations = 50 in the reference (>= 8: the initial simplex). pred_vertices is
 *   the render at the pose the update multiplies from the left. update f64 [7] = the best vertex (quaternion NOT normalised,
 *   like nlopt hands it back; the caller normalises as Sophus::SE3f does), info f64 [2] = (its energy, evaluations used).
 *   nlopt itself is not available here: the algorithm is the published one of its nldrmd.c (Box's bound handling, alpha 1,
 *   beta 0.5, gamma 2, delta 0.5, initial step (ub - lb) / 4) restated, not its bits. The whole optimisation is one launch of
 *   one workgroup (simplex in LDS, an evaluation = a sweep of the label's bounding box); bit-identical to oracle_icp_polish.
 * ------------------------------------------------------------------------------------------ */
int pcnn_icp_polish_fwd(const int32_t* label, const float* live_vertices, const float* pred_vertices, int pred_channels,
                        int height, int width, int obj_id, float z_near, float z_far, int max_evaluations, double* update,
                        double* info, void* stream);
int pcnn_render_mesh_workspace_bytes(int num_poses, int height, int width, size_t* bytes);
int pcnn_render_mesh_fwd(const float* vertices, const float* normals, const int32_t* faces, int num_vertices,
                         int num_faces, const float* poses, int num_poses, int height, int width, float fx, float fy,
                         float px, float py, float z_near, float z_far, float canon_x_offset, float* out_vertices,
                         float* out_normals, float* out_canonical, void* workspace, size_t workspace_bytes,
                         void* stream);
int pcnn_icp_center_workspace_bytes(int height, int width, size_t* bytes);
int pcnn_icp_center_fwd(const int32_t* label, const float* live_vertices, const float* canonical,
                        const float* pred_vertices, const float* pred_normals, int pred_channels, int height, int width,
                        int obj_id, float max_error, double* sums, uint8_t* mask, void* workspace,
                        size_t workspace_bytes, void* stream);
int pcnn_icp_score_workspace_bytes(int num_hypotheses, int height, int width, size_t* bytes);
int pcnn_icp_score_fwd(const float* live_vertices, const float* canonical, const uint8_t* mask, int height, int width,
                       const float* hypotheses, int num_hypotheses, float fx, float fy, float px, float py, float radius,
                       int32_t* hits, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Kernel timing diagnostics (off by default; the reference's only instrumentation is the
 * wall-clock Timer of lib/utils/timer.py:10-32 around im_segment).
 * While enabled, every kernel launch of this library is bracketed by hipEventRecord on the launch
 * stream. pcnn_profile_report synchronises the recorded events and writes a JSON object
 * {"kernel": {"calls": n, "total_ms": t, "avg_us": a}, ...} into buf (NUL terminated, truncated to
 * cap); it returns the number of bytes needed. pcnn_profile_enable(0) disables and clears.
 * Not legal during hipGraph capture.
 * ------------------------------------------------------------------------------------------ */
int pcnn_profile_enable(int on);
int pcnn_profile_reset(void);
long pcnn_profile_report(char* buf, long cap);

/* ------------------------------------------------------------------------------------------
 * Host utility: CRC32C (Castagnoli, reflected polynomial 0x82F63B78) of a HOST buffer, continuing
 * from `seed` (0 for a fresh checksum) — the checksum TensorFlow's tensor-bundle checkpoints carry
 * per table block and per tensor (tensorflow/core/lib/hash/crc32c.h); used by the checkpoint reader
 * behind Network.load_file (lib/fcn/test.py:1809-1811 `saver.restore`). Never touches the GPU.
 * ------------------------------------------------------------------------------------------ */
uint32_t pcnn_crc32c(const void* data, size_t num_bytes, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* POSECNN_HIP_H_ */
