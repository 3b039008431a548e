/*
 * pcnn_oracle.c — CPU oracle for the PoseCNN custom-op hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under posecnn_amd/ may include, link, import or execute this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it (as the
 * checker / the timed CPU baseline, never as the product path).
 *
 * PINNING: the reference has no golden vectors, known-answer tests or unit tests for this path
 *  and its TF ops cannot be built here. This file restates, in plain C, the
 * reference's *GPU* kernels (`*_op_gpu.cu.cc`), which BASELINE.json names as the parity target, and
 * is pinned by (a) oracle/_ref: the reference's own __global__ kernel bodies compiled unchanged for
 * the CPU from the sources where they lie (oracle/ref_shim/) — tests/test_oracle_vs_reference.py
 * requires bit-identical outputs for all five ops, forward and backward; (b) an independent numpy
 * restatement (tests/np_ref.py); (c) hand-derived known answers (tests/test_oracle_kat.py).
 * The dense layers of the graph (TF/cuDNN conv, matmul) have no such pin: "parity unpinned" there.
 *
 * Canonical choices where the reference is order dependent (atomicAdd order, thrust): serial
 * execution order — ascending pixel index for the per-class pixel arrays, ascending cell index for
 * maxima, ascending (image, maximum) for output rows, sequential ascending sums.  FP contraction:
 * none (build with -ffp-contract=off); expf: the canonical pcnn_exp_f32 below (the bits of CUDA's
 * expf are unknowable here; both this file and the HIP kernels evaluate the same IEEE double
 * sequence, so they agree bit for bit).
 *
 * Each function cites the reference file:line it follows (paths relative to the reference tree).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VERTEX_CHANNELS 3
#define MAX_ROI 128
#define POSE_CHANNELS 4

/* ------------------------------------------------------------------------------------------ */
/* canonical expf: exp evaluated in IEEE double (range reduction + degree-13 Taylor/Horner),     */
/* rounded once to float.  |rel err| ~1e-16 before the final rounding.                           */
/* ------------------------------------------------------------------------------------------ */
float oracle_expf(float xf)
{
  if (xf != xf) return xf;
  double x = (double)xf;
  if (x > 130.0) x = 130.0;   /* -> +inf after the float conversion */
  if (x < -150.0) x = -150.0; /* -> 0 */
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double kd = floor(x * LOG2E + 0.5);
  double r = (x - kd * LN2_HI) - kd * LN2_LO;
  double p = 1.6059043836821614599e-10; /* 1/13! */
  p = p * r + 2.0876756987868098979e-09; /* 1/12! */
  p = p * r + 2.5052108385441718775e-08; /* 1/11! */
  p = p * r + 2.7557319223985890653e-07; /* 1/10! */
  p = p * r + 2.7557319223985892511e-06; /* 1/9! */
  p = p * r + 2.4801587301587301566e-05; /* 1/8! */
  p = p * r + 1.9841269841269841253e-04; /* 1/7! */
  p = p * r + 1.3888888888888889419e-03; /* 1/6! */
  p = p * r + 8.3333333333333332177e-03; /* 1/5! */
  p = p * r + 4.1666666666666664354e-02; /* 1/4! */
  p = p * r + 1.6666666666666665741e-01; /* 1/3! */
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  int k = (int)kd;
  uint64_t bits = (uint64_t)(k + 1023) << 52;
  double two_k;
  memcpy(&two_k, &bits, sizeof two_k);
  return (float)(p * two_k);
}

/* ------------------------------------------------------------------------------------------ */
/* Hough voting                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* angle_distance, hough_voting_gpu_op.cu.cc:32-42 */
static inline float angle_distance(int cx, int cy, int x, int y, float u, float v)
{
  float dx = (float)(cx - x);
  float dy = (float)(cy - y);
  float n1 = sqrtf(u * u + v * v);
  float n2 = sqrtf(dx * dx + dy * dy);
  float dot = u * dx + v * dy;
  return dot / (n1 * n2);
}

/* project_box, hough_voting_gpu_op.cu.cc:84-120 (factor = 0.6 at every call site :285,317) */
static float project_box(int cls, const float* extents, const float* meta, float distance)
{
  float xHalf = (float)((double)extents[cls * 3 + 0] * 0.5);
  float yHalf = (float)((double)extents[cls * 3 + 1] * 0.5);
  float zHalf = (float)((double)extents[cls * 3 + 2] * 0.5);
  float bb[24];
  bb[0] = xHalf;   bb[1] = yHalf;   bb[2] = zHalf + distance;
  bb[3] = -xHalf;  bb[4] = yHalf;   bb[5] = zHalf + distance;
  bb[6] = xHalf;   bb[7] = -yHalf;  bb[8] = zHalf + distance;
  bb[9] = -xHalf;  bb[10] = -yHalf; bb[11] = zHalf + distance;
  bb[12] = xHalf;  bb[13] = yHalf;  bb[14] = -zHalf + distance;
  bb[15] = -xHalf; bb[16] = yHalf;  bb[17] = -zHalf + distance;
  bb[18] = xHalf;  bb[19] = -yHalf; bb[20] = -zHalf + distance;
  bb[21] = -xHalf; bb[22] = -yHalf; bb[23] = -zHalf + distance;
  float fx = meta[0], fy = meta[4], px = meta[2], py = meta[5];
  float minX = 1e8f, maxX = -1e8f, minY = 1e8f, maxY = -1e8f;
  for (int i = 0; i < 8; i++) {
    float x = fx * (bb[i * 3] / bb[i * 3 + 2]) + px;
    float y = fy * (bb[i * 3 + 1] / bb[i * 3 + 2]) + py;
    minX = fminf(minX, x);
    minY = fminf(minY, y);
    maxX = fmaxf(maxX, x);
    maxY = fmaxf(maxY, y);
  }
  float width = maxX - minX + 1;
  float height = maxY - minY + 1;
  return fmaxf(width, height) * 0.6f;
}

/* IoU, hough_voting_gpu_op.cu.cc:73-82 */
static float box_iou(const float* a, const float* b)
{
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

/* compute_box_overlap, hough_voting_gpu_op.cu.cc:123-172. The rotation is Eigen's
 * Quaternionf(w,x,y,z).toRotationMatrix() (Eigen/src/Geometry/Quaternion.h, unpinned version):
 * tx=2x.. twx=tx*w.. R00=1-(tyy+tzz) ...; the 3x3 * 3x8 product is summed k = 0,1,2. */
static float compute_box_overlap(int cls, const float* extents, const float* meta,
                                 const float* pose, const float* box)
{
  float xHalf = (float)((double)extents[cls * 3 + 0] * 0.5);
  float yHalf = (float)((double)extents[cls * 3 + 1] * 0.5);
  float zHalf = (float)((double)extents[cls * 3 + 2] * 0.5);
  float bb[8][3] = {{xHalf, yHalf, zHalf},   {-xHalf, yHalf, zHalf},  {xHalf, -yHalf, zHalf},
                    {-xHalf, -yHalf, zHalf}, {xHalf, yHalf, -zHalf},  {-xHalf, yHalf, -zHalf},
                    {xHalf, -yHalf, -zHalf}, {-xHalf, -yHalf, -zHalf}};
  float qw = pose[6], qx = pose[7], qy = pose[8], qz = pose[9];
  float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  float txx = tx * qx, txy = ty * qx, txz = tz * qx;
  float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  float R[3][3];
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz;       R[0][2] = txz + twy;
  R[1][0] = txy + twz;       R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;       R[2][1] = tyz + twx;       R[2][2] = 1 - (txx + tyy);
  float fx = meta[0], fy = meta[4], px = meta[2], py = meta[5];
  float x1 = 1e8f, x2 = -1e8f, y1 = 1e8f, y2 = -1e8f;
  for (int i = 0; i < 8; i++) {
    float X = (R[0][0] * bb[i][0] + R[0][1] * bb[i][1] + R[0][2] * bb[i][2]) + pose[10];
    float Y = (R[1][0] * bb[i][0] + R[1][1] * bb[i][1] + R[1][2] * bb[i][2]) + pose[11];
    float Z = (R[2][0] * bb[i][0] + R[2][1] * bb[i][1] + R[2][2] * bb[i][2]) + pose[12];
    float x = fx * (X / Z) + px;
    float y = fy * (Y / Z) + py;
    x1 = fminf(x1, x);
    y1 = fminf(y1, y);
    x2 = fmaxf(x2, x);
    y2 = fmaxf(y2, y);
  }
  float box_gt[4] = {x1, y1, x2, y2};
  return box_iou(box, box_gt);
}

typedef struct {
  int x, y;
  float u, v, d, thr;
} hv_pixel;

/* One Hough cell: compute_hough_kernel, hough_voting_gpu_op.cu.cc:253-333, restricted to the
 * sampled pixels pix[0..m) (= arrays[cls][0::skip] in ascending pixel order). The box test and
 * the angle test are both pure, so evaluating the (cheap) box test first is result-identical. */
static float cell_votes(const hv_pixel* pix, int m, int cx, int cy, float inlier, float* sumd)
{
  float votes = 0.f, distance = 0.f;
  for (int i = 0; i < m; i++) {
    float dx = fabsf((float)(pix[i].x - cx));
    float dy = fabsf((float)(pix[i].y - cy));
    if (dx < pix[i].thr && dy < pix[i].thr) {
      if (angle_distance(cx, cy, pix[i].x, pix[i].y, pix[i].u, pix[i].v) > inlier) {
        votes++;
        distance += pix[i].d;
      }
    }
  }
  *sumd = distance;
  return votes;
}

/* second loop of compute_hough_kernel (:296-331): hough_data = (distance, 2*bb_height, 2*bb_width) */
static void cell_data(const hv_pixel* pix, int m, int cx, int cy, float inlier, int cls,
                      const float* extents, const float* meta, float votes, float sumd,
                      float* hd3)
{
  hd3[0] = hd3[1] = hd3[2] = 0.f; /* cudaMemset(hough_data, 0, ...) :706 */
  if (votes > 0) {
    float distance = sumd / votes;
    float bb_width = -1, bb_height = -1;
    float threshold = project_box(cls, extents, meta, distance);
    for (int i = 0; i < m; i++) {
      if (angle_distance(cx, cy, pix[i].x, pix[i].y, pix[i].u, pix[i].v) > inlier) {
        float dx = fabsf((float)(pix[i].x - cx));
        float dy = fabsf((float)(pix[i].y - cy));
        if (dx > bb_width && dx < threshold && dy < threshold) bb_width = dx;
        if (dy > bb_height && dx < threshold && dy < threshold) bb_height = dy;
      }
    }
    hd3[0] = distance;
    hd3[1] = 2 * bb_height;
    hd3[2] = 2 * bb_width;
  }
}

/* compute_rois_kernel, hough_voting_gpu_op.cu.cc:386-576, for one maximum; appends 1 or 9 rows. */
static void emit_rois(int x, int y, int cls, float votes, const float* hd3, const float* extents,
                      const float* meta, const float* gt, int num_gt, int is_train,
                      int batch_index, int C, float* top_box, float* top_pose, float* top_target,
                      float* top_weight, int* top_domain, int* num_rois)
{
  float scale = 0.05f;
  float fx = meta[0], fy = meta[4], px = meta[2], py = meta[5];
  float rx = ((float)x - px) / fx;
  float ry = ((float)y - py) / fy;
  float bb_distance = hd3[0], bb_height = hd3[1], bb_width = hd3[2];
  double k = 0.5 + (double)scale;
  int roi_index = *num_rois;
  *num_rois += is_train ? 9 : 1;
  float* b = top_box + (size_t)roi_index * 7;
  b[0] = (float)batch_index;
  b[1] = (float)cls;
  b[2] = (float)((double)x - (double)bb_width * k);
  b[3] = (float)((double)y - (double)bb_height * k);
  b[4] = (float)((double)x + (double)bb_width * k);
  b[5] = (float)((double)y + (double)bb_height * k);
  b[6] = votes;
  int nrows = is_train ? 9 : 1;
  for (int i = 0; i < nrows; i++) {
    float* p = top_pose + (size_t)(roi_index + i) * 7;
    p[0] = 1; p[1] = 0; p[2] = 0; p[3] = 0;
    p[4] = rx * bb_distance;
    p[5] = ry * bb_distance;
    p[6] = bb_distance;
    if (is_train) top_domain[roi_index + i] = (num_gt == 0) ? 1 : 0;
  }
  if (!is_train) return;

  /* pose target :440-466 */
  for (int i = 0; i < num_gt; i++) {
    int gt_batch = (int)gt[i * 13 + 0];
    int gt_id = (int)gt[i * 13 + 1];
    if (cls == gt_id && batch_index == gt_batch) {
      float overlap = compute_box_overlap(cls, extents, meta, gt + i * 13, b + 2);
      if ((double)overlap > 0.2) { /* `overlap > 0.2`: float promoted to double (:449) */
        for (int j = 0; j < 9; j++) {
          for (int q = 0; q < 4; q++) {
            top_target[(size_t)(roi_index + j) * 4 * C + 4 * cls + q] = gt[i * 13 + 6 + q];
            top_weight[(size_t)(roi_index + j) * 4 * C + 4 * cls + q] = 1;
          }
        }
        break;
      }
    }
  }

  /* jittered boxes :468-554; `0.05 * ww` is double arithmetic */
  float x1 = b[2], y1 = b[3], x2 = b[4], y2 = b[5];
  float ww = x2 - x1, hh = y2 - y1;
  static const int sx[8] = {-1, +1, -1, +1, 0, -1, 0, +1};
  static const int sy[8] = {-1, -1, +1, +1, -1, 0, +1, 0};
  for (int j = 0; j < 8; j++) {
    float* r = top_box + (size_t)(roi_index + 1 + j) * 7;
    r[0] = (float)batch_index;
    r[1] = (float)cls;
    if (sx[j] < 0) r[2] = (float)((double)x1 - 0.05 * (double)ww);
    else if (sx[j] > 0) r[2] = (float)((double)x1 + 0.05 * (double)ww);
    else r[2] = x1;
    if (sy[j] < 0) r[3] = (float)((double)y1 - 0.05 * (double)hh);
    else if (sy[j] > 0) r[3] = (float)((double)y1 + 0.05 * (double)hh);
    else r[3] = y1;
    r[4] = r[2] + ww;
    r[5] = r[3] + hh;
    r[6] = votes;
  }
}

/* Collect arrays[cls][0::skip] with per-pixel u, v, d=exp(.), thr=project_box(d): the loop
 * header of compute_hough_kernel (:269-285) hoisted per pixel (pure functions of the pixel). */
static int collect_pixels(const int* labelmap, const float* vertmap, const float* extents,
                          const float* meta, int H, int W, int C, int cls, int skip, hv_pixel* out)
{
  int rank = 0, m = 0;
  for (int i = 0; i < H * W; i++) {
    if (labelmap[i] != cls) continue;
    if (rank % skip == 0) {
      int x = i % W, y = i / W;
      size_t off = (size_t)VERTEX_CHANNELS * cls + (size_t)VERTEX_CHANNELS * C * ((size_t)y * W + x);
      hv_pixel p;
      p.x = x; p.y = y;
      p.u = vertmap[off];
      p.v = vertmap[off + 1];
      p.d = oracle_expf(vertmap[off + 2]);
      p.thr = project_box(cls, extents, meta, p.d);
      out[m++] = p;
    }
    rank++;
  }
  return m;
}

/*
 * HoughvotinggpuOp<GpuDevice>::Compute + HoughVotingLaucher
 * (hough_voting_gpu_op.cc:321-429, hough_voting_gpu_op.cu.cc:615-797).
 * Outputs have PCNN capacity MAX_ROI*9 rows and are zero-filled (reset_outputs :579-588).
 * num_rois[0] = rows the op returns (>=1, dummy row), num_rois[1] = true count.
 * hs_debug (optional, may be NULL): [B][C][H*W] votes of every slot class, for cross checks.
 */
int oracle_hough_voting_ex(const int* label, const float* vertex, const float* extents,
                           const float* meta, const float* gt, int B, int H, int W, int C,
                           int num_meta, int num_gt, int is_train, float vote_thr, float per_thr,
                           int skip, float inlier, int label_thr, int rois_per_image,
                           int rows_capacity, float* top_box, float* top_pose,
                           float* top_target, float* top_weight, int* top_domain, int* num_rois,
                           float* hs_debug);

int oracle_hough_voting(const int* label, const float* vertex, const float* extents,
                        const float* meta, const float* gt, int B, int H, int W, int C,
                        int num_meta, int num_gt, int is_train, float vote_thr, float per_thr,
                        int skip, float inlier, int label_thr, float* top_box, float* top_pose,
                        float* top_target, float* top_weight, int* top_domain, int* num_rois,
                        float* hs_debug)
{
  return oracle_hough_voting_ex(label, vertex, extents, meta, gt, B, H, W, C, num_meta, num_gt,
                                is_train, vote_thr, per_thr, skip, inlier, label_thr, 0, MAX_ROI * 9,
                                top_box, top_pose, top_target, top_weight, top_domain, num_rois,
                                hs_debug);
}

/* The same op with the per-image capacity lifted off the reference's `index_size = MAX_ROI /
 * batch_size` (:733): rois_per_image > 0 keeps the first rois_per_image maxima of EVERY image
 * whatever the batch size (what B single-frame calls of the reference yield, each with
 * index_size = 128); rois_per_image = 0 is the reference rule. Outputs hold rows_capacity rows
 * (>= B * capacity * (is_train ? 9 : 1)). */
int oracle_hough_voting_ex(const int* label, const float* vertex, const float* extents,
                           const float* meta, const float* gt, int B, int H, int W, int C,
                           int num_meta, int num_gt, int is_train, float vote_thr, float per_thr,
                           int skip, float inlier, int label_thr, int rois_per_image,
                           int rows_capacity, float* top_box, float* top_pose,
                           float* top_target, float* top_weight, int* top_domain, int* num_rois,
                           float* hs_debug)
{
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || skip <= 0 || rois_per_image < 0) return -1;
  const int cap_rows = rows_capacity;
  memset(top_box, 0, sizeof(float) * cap_rows * 7);
  memset(top_pose, 0, sizeof(float) * cap_rows * 7);
  memset(top_target, 0, sizeof(float) * (size_t)cap_rows * 4 * C);
  memset(top_weight, 0, sizeof(float) * (size_t)cap_rows * 4 * C);
  memset(top_domain, 0, sizeof(int) * cap_rows);
  int rows = 0;
  const int HW = H * W;
  const int index_size = rois_per_image > 0 ? rois_per_image : MAX_ROI / B; /* :733 */
  if ((long long)B * index_size * (is_train ? 9 : 1) > cap_rows) return -1;

  hv_pixel* pix = (hv_pixel*)malloc(sizeof(hv_pixel) * (size_t)(HW / skip + 2));
  float* hs = (float*)malloc(sizeof(float) * (size_t)HW);
  float* sd = (float*)malloc(sizeof(float) * (size_t)HW);
  int* sizes = (int*)malloc(sizeof(int) * C);

  for (int n = 0; n < B; n++) {
    const int* labelmap = label + (size_t)n * HW;
    const float* vertmap = vertex + (size_t)n * HW * VERTEX_CHANNELS * C;
    const float* md = meta + (size_t)n * num_meta;

    /* step 1 :626-663 */
    memset(sizes, 0, sizeof(int) * C);
    for (int i = 0; i < HW; i++) {
      int c = labelmap[i];
      if (c > 0 && c < C) sizes[c]++;
    }
    int count = 0;
    int slots[256];
    for (int c = 1; c < C; c++)
      if (sizes[c] > label_thr) slots[count++] = c;
    if (count == 0) continue;

    int num_max = 0; /* maxima accepted for this image, <= index_size */
    int total_max = 0;
    for (int s = 0; s < count; s++) {
      int cls = slots[s];
      int m = collect_pixels(labelmap, vertmap, extents, md, H, W, C, cls, skip, pix);
      /* step 2 :686-714 */
#pragma omp parallel for schedule(dynamic, 16)
      for (int cy = 0; cy < H; cy++)
        for (int cx = 0; cx < W; cx++)
          hs[cy * W + cx] = cell_votes(pix, m, cx, cy, inlier, &sd[cy * W + cx]);
      if (hs_debug) memcpy(hs_debug + ((size_t)n * C + cls) * HW, hs, sizeof(float) * HW);

      /* step 3 + 4 :723-785 */
      if (vote_thr > 0) {
        /* compute_max_indexes_kernel :335-383, cells in ascending index order */
        for (int cy = 0; cy < H; cy++) {
          for (int cx = 0; cx < W; cx++) {
            float v = hs[cy * W + cx];
            if (!(v > vote_thr)) continue;
            int flag = 0;
            for (int x = cx - 3; x <= cx + 3 && !flag; x++)
              for (int y = cy - 3; y <= cy + 3; y++)
                if (x >= 0 && x < W && y >= 0 && y < H && hs[y * W + x] > v) { flag = 1; break; }
            if (flag) continue;
            float hd3[3];
            cell_data(pix, m, cx, cy, inlier, cls, extents, md, v, sd[cy * W + cx], hd3);
            if (!(hd3[1] > 0 && hd3[2] > 0)) continue;
            if (v / (hd3[1] * hd3[2]) < per_thr) continue;
            total_max++;
            if (num_max < index_size) {
              num_max++;
              emit_rois(cx, cy, cls, v, hd3, extents, md, gt, num_gt, is_train, n, C, top_box,
                        top_pose, top_target, top_weight, top_domain, &rows);
            }
          }
        }
      } else {
        /* thrust::max_element :752-762: first maximum */
        int best = 0;
        for (int i = 1; i < HW; i++)
          if (hs[i] > hs[best]) best = i;
        total_max++;
        if (num_max < index_size) {
          num_max++;
          int cx = best % W, cy = best / W;
          float hd3[3];
          cell_data(pix, m, cx, cy, inlier, cls, extents, md, hs[best], sd[best], hd3);
          emit_rois(cx, cy, cls, hs[best], hd3, extents, md, gt, num_gt, is_train, n, C, top_box,
                    top_pose, top_target, top_weight, top_domain, &rows);
        }
      }
    }
    (void)total_max;
  }
  free(pix); free(hs); free(sd); free(sizes);
  num_rois[1] = rows;
  num_rois[0] = rows == 0 ? 1 : rows; /* hough_voting_gpu_op.cc:381-383 */
  return 0;
}

/* Full-fidelity Hough space of one (image, class): votes and hough_data for EVERY cell, exactly
 * as compute_hough_kernel writes them (no lazy evaluation). For cross checks at small sizes. */
int oracle_hough_space(const int* labelmap, const float* vertmap, const float* extents,
                       const float* meta, int H, int W, int C, int cls, int skip, float inlier,
                       float* hough_space, float* hough_data)
{
  hv_pixel* pix = (hv_pixel*)malloc(sizeof(hv_pixel) * (size_t)(H * W / skip + 2));
  int m = collect_pixels(labelmap, vertmap, extents, meta, H, W, C, cls, skip, pix);
#pragma omp parallel for schedule(dynamic, 4)
  for (int cy = 0; cy < H; cy++)
    for (int cx = 0; cx < W; cx++) {
      float sumd;
      float v = cell_votes(pix, m, cx, cy, inlier, &sumd);
      hough_space[cy * W + cx] = v;
      cell_data(pix, m, cx, cy, inlier, cls, extents, meta, v, sumd, hough_data + 3 * (cy * W + cx));
    }
  free(pix);
  return m;
}

float oracle_project_box(int cls, const float* extents, const float* meta, float distance)
{
  return project_box(cls, extents, meta, distance);
}

/* ------------------------------------------------------------------------------------------ */
/* ROI pooling: ROIPoolForward, roi_pooling_op_gpu.cu.cc:20-101                                */
/* ------------------------------------------------------------------------------------------ */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

int oracle_roi_pool(const float* data, const float* rois, int B, int H, int W, int C, int R,
                    int roi_cols, int PH, int PW, float scale, int pool_channel, float* top,
                    int* argmax)
{
  const int Cout = pool_channel ? 1 : C;
#pragma omp parallel for
  for (int n = 0; n < R; n++) {
    const float* roi = rois + (size_t)n * roi_cols;
    int roi_batch_ind = (int)roi[0];
    int roi_cls = (int)roi[1];
    int roi_start_w = (int)roundf(roi[2] * scale);
    int roi_start_h = (int)roundf(roi[3] * scale);
    int roi_end_w = (int)roundf(roi[4] * scale);
    int roi_end_h = (int)roundf(roi[5] * scale);
    int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
    int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
    float bin_size_h = (float)roi_height / (float)PH;
    float bin_size_w = (float)roi_width / (float)PW;
    /* canonical guard: the GPU op reads out of bounds for a bad batch index (the CPU op CHECKs,
       roi_pooling_op.cc:146-147); such ROIs pool to 0 / argmax -1 */
    const int bad = roi_batch_ind < 0 || roi_batch_ind >= B;
    const float* img = data + (size_t)(bad ? 0 : roi_batch_ind) * C * H * W;
    for (int ph = 0; ph < PH; ph++)
      for (int pw = 0; pw < PW; pw++) {
        int hstart = (int)floorf((float)ph * bin_size_h);
        int wstart = (int)floorf((float)pw * bin_size_w);
        int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
        int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
        hstart = imin(imax(hstart + roi_start_h, 0), H);
        hend = imin(imax(hend + roi_start_h, 0), H);
        wstart = imin(imax(wstart + roi_start_w, 0), W);
        wend = imin(imax(wend + roi_start_w, 0), W);
        int is_empty = (hend <= hstart) || (wend <= wstart);
        for (int c = 0; c < Cout; c++) {
          float maxval = is_empty ? 0 : -FLT_MAX;
          int maxidx = -1;
          int cc = pool_channel ? roi_cls : c;
          if (bad || cc < 0 || cc >= C) {
            size_t index0 = (((size_t)n * PH + ph) * PW + pw) * Cout + c;
            top[index0] = 0;
            if (argmax) argmax[index0] = -1;
            continue;
          }
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w) {
              int bottom_index = (h * W + w) * C + cc;
              if (img[bottom_index] > maxval) {
                maxval = img[bottom_index];
                maxidx = bottom_index;
              }
            }
          size_t index = (((size_t)n * PH + ph) * PW + pw) * Cout + c;
          top[index] = maxval;
          if (argmax) argmax[index] = maxidx;
        }
      }
  }
  return 0;
}

/* ROIPoolBackward, roi_pooling_op_gpu.cu.cc:135-229 (gather form, ROIs ascending) */
int oracle_roi_pool_bwd(const float* top_diff, const float* rois, const int* argmax, int B, int H,
                        int W, int C, int R, int roi_cols, int PH, int PW, float scale,
                        int pool_channel, float* bottom_diff)
{
#pragma omp parallel for
  for (long index = 0; index < (long)B * H * W * C; index++) {
    long t = index;
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
      int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
      int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
      float bin_size_h = (float)roi_height / (float)PH;
      float bin_size_w = (float)roi_width / (float)PW;
      int phstart = (int)floorf((float)(h - roi_start_h) / bin_size_h);
      int phend = (int)ceilf((float)(h - roi_start_h + 1) / bin_size_h);
      int pwstart = (int)floorf((float)(w - roi_start_w) / bin_size_w);
      int pwend = (int)ceilf((float)(w - roi_start_w + 1) / bin_size_w);
      phstart = imin(imax(phstart, 0), PH);
      phend = imin(imax(phend, 0), PH);
      pwstart = imin(imax(pwstart, 0), PW);
      pwend = imin(imax(pwend, 0), PW);
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
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Hard label: HardlabelForward, hard_label_op_gpu.cu.cc:17-29                                  */
/* ------------------------------------------------------------------------------------------ */
int oracle_hard_label(const float* prob, const int* gt, long N, int C, float threshold, float* out)
{
#pragma omp parallel for
  for (long index = 0; index < N; index++) {
    for (int c = 0; c < C; c++) out[index * C + c] = 0.0f;
    int gt_label = gt[index];
    /* labels outside [-1, C) index out of bounds in the reference; canonical: ignored */
    if (gt_label >= 0 && gt_label < C && (gt_label > 0 || prob[index * C + gt_label] < threshold))
      out[index * C + gt_label] = 1.0f;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Average distance loss: AveragedistanceForward + sum_losses_gradients + thrust::reduce,       */
/* average_distance_loss_op_gpu.cu.cc:35-206, 210-252, 323-335                                  */
/* ------------------------------------------------------------------------------------------ */
static void quat_rot(float s, float u, float v, float w, float* r)
{
  r[0] = s * s + u * u - v * v - w * w;
  r[1] = 2 * (u * v - s * w);
  r[2] = 2 * (u * w + s * v);
  r[3] = 2 * (u * v + s * w);
  r[4] = s * s - u * u + v * v - w * w;
  r[5] = 2 * (v * w - s * u);
  r[6] = 2 * (u * w - s * v);
  r[7] = 2 * (v * w + s * u);
  r[8] = s * s - u * u - v * v + w * w;
}

int oracle_average_distance(const float* prediction, const float* target, const float* weight,
                            const float* point, const float* symmetry, int R, int C, int P,
                            float margin, float* loss, float* bottom_diff)
{
  const int CH = POSE_CHANNELS * C;
  float* losses = (float*)calloc((size_t)R * P, sizeof(float));
  float* diffs = (float*)calloc((size_t)R * P * 4, sizeof(float)); /* only the class' 4 channels are ever non-zero */
  int* cls_of = (int*)malloc(sizeof(int) * (R > 0 ? R : 1));
  for (int n = 0; n < R; n++) {
    int index_cls = -1;
    float rot[54];
    float s = 0, u = 0, v = 0, w = 0;
    for (int i = 0; i < CH; i += POSE_CHANNELS) {
      int index = n * CH + i;
      if (weight[index] > 0) {
        index_cls = i / POSE_CHANNELS;
        quat_rot(target[index], target[index + 1], target[index + 2], target[index + 3], rot);
        s = prediction[index + 0]; u = prediction[index + 1];
        v = prediction[index + 2]; w = prediction[index + 3];
        quat_rot(s, u, v, w, rot + 9);
        break;
      }
    }
    cls_of[n] = index_cls;
    if (index_cls == -1) continue;
    float* d = rot + 18; /* derivatives of Ru w.r.t. (s,u,v,w) :96-139 */
    d[0] = 2 * s;  d[1] = -2 * w; d[2] = 2 * v;  d[3] = 2 * w;  d[4] = 2 * s;  d[5] = -2 * u; d[6] = -2 * v; d[7] = 2 * u;  d[8] = 2 * s;
    d += 9;
    d[0] = 2 * u;  d[1] = 2 * v;  d[2] = 2 * w;  d[3] = 2 * v;  d[4] = -2 * u; d[5] = -2 * s; d[6] = 2 * w;  d[7] = 2 * s;  d[8] = -2 * u;
    d += 9;
    d[0] = -2 * v; d[1] = 2 * u;  d[2] = 2 * s;  d[3] = 2 * u;  d[4] = 2 * v;  d[5] = 2 * w;  d[6] = -2 * s; d[7] = 2 * w;  d[8] = -2 * v;
    d += 9;
    d[0] = -2 * w; d[1] = -2 * s; d[2] = 2 * u;  d[3] = 2 * s;  d[4] = -2 * w; d[5] = 2 * v;  d[6] = 2 * u;  d[7] = 2 * v;  d[8] = 2 * w;

    const float* pts = point + (size_t)index_cls * P * 3;
    const int sym = symmetry[index_cls] > 0;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; p++) {
      const float* pt = pts + p * 3;
      float x1 = rot[9 + 0] * pt[0] + rot[9 + 1] * pt[1] + rot[9 + 2] * pt[2];
      float y1 = rot[9 + 3] * pt[0] + rot[9 + 4] * pt[1] + rot[9 + 5] * pt[2];
      float z1 = rot[9 + 6] * pt[0] + rot[9 + 7] * pt[1] + rot[9 + 8] * pt[2];
      int qmin = p;
      float x2, y2, z2;
      if (sym) {
        float dmin = FLT_MAX;
        for (int i = 0; i < P; i++) {
          const float* q = pts + i * 3;
          x2 = rot[0] * q[0] + rot[1] * q[1] + rot[2] * q[2];
          y2 = rot[3] * q[0] + rot[4] * q[1] + rot[5] * q[2];
          z2 = rot[6] * q[0] + rot[7] * q[1] + rot[8] * q[2];
          float distance = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2);
          if (distance < dmin) { dmin = distance; qmin = i; }
        }
      }
      const float* q = pts + qmin * 3;
      x2 = rot[0] * q[0] + rot[1] * q[1] + rot[2] * q[2];
      y2 = rot[3] * q[0] + rot[4] * q[1] + rot[5] * q[2];
      z2 = rot[6] * q[0] + rot[7] * q[1] + rot[8] * q[2];
      float distance = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2);
      if (distance < margin) continue;
      losses[(size_t)n * P + p] = (float)((double)(distance - margin) / (2.0 * R * P));
      float* df = diffs + ((size_t)n * P + p) * 4;
      for (int j = 0; j < 3; j++) {
        float diff = j == 0 ? x1 - x2 : (j == 1 ? y1 - y2 : z1 - z2);
        for (int k = 0; k < 3; k++) {
          float den = (float)(R * P);
          df[0] += diff * pt[k] * rot[18 + j * 3 + k] / den;
          df[1] += diff * pt[k] * rot[27 + j * 3 + k] / den;
          df[2] += diff * pt[k] * rot[36 + j * 3 + k] / den;
          df[3] += diff * pt[k] * rot[45 + j * 3 + k] / den;
        }
      }
    }
  }
  /* sum_losses_gradients :210-252 (p ascending), then thrust::reduce over n (ascending) */
  float total = 0.f;
  for (int n = 0; n < R; n++) {
    for (int c = 0; c < CH; c++) bottom_diff[(size_t)n * CH + c] = 0;
    int cls = cls_of[n];
    if (cls >= 0) {
      for (int k = 0; k < 4; k++) {
        float acc = 0;
        for (int p = 0; p < P; p++) acc += diffs[((size_t)n * P + p) * 4 + k];
        bottom_diff[(size_t)n * CH + 4 * cls + k] = acc;
      }
    }
    float lb = 0;
    for (int p = 0; p < P; p++) lb += losses[(size_t)n * P + p];
    total += lb;
  }
  loss[0] = total;
  free(losses); free(diffs); free(cls_of);
  return 0;
}

/* AveragedistanceBackward :347-354 */
int oracle_average_distance_bwd(const float* grad, const float* bottom_diff, int R, int channels,
                                float* out)
{
  for (long i = 0; i < (long)R * channels; i++) out[i] = grad[0] * bottom_diff[i];
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Backprojecting: BackprojectForward, backprojecting_op_gpu.cu.cc:17-126                       */
/* ------------------------------------------------------------------------------------------ */
/* one voxel of BackprojectForward; (td, tl, tf) point at this voxel's output rows */
static void backproject_voxel(const float* data, const float* label, const float* depth, const float* meta,
                              const float* label_3d, int H, int W, int Cd, int Cl, int num_meta, int G,
                              int ksize, float threshold, long vox, float* td, float* tl, float* tf)
{
  {
    long t = vox;
    int w = (int)(t % G); t /= G;
    int h = (int)(t % G); t /= G;
    int d = (int)(t % G); t /= G;
    int n = (int)t;
    const float* md = meta + (size_t)n * num_meta;
    float X = d * md[42] + md[45];
    float Y = h * md[43] + md[46];
    float Z = w * md[44] + md[47];
    float X1 = md[18] * X + md[19] * Y + md[20] * Z + md[21];
    float Y1 = md[22] * X + md[23] * Y + md[24] * Z + md[25];
    float Z1 = md[26] * X + md[27] * Y + md[28] * Z + md[29];
    float x1 = md[0] * X1 + md[1] * Y1 + md[2] * Z1;
    float x2 = md[3] * X1 + md[4] * Y1 + md[5] * Z1;
    float x3 = md[6] * X1 + md[7] * Y1 + md[8] * Z1;
    /* `int px = round(x1 / x3)`: out-of-int-range values (x3 == 0) are undefined in the
       reference; the canonical conversion saturates and maps NaN to 0 (GPU cvt semantics). */
    float fpx = roundf(x1 / x3), fpy = roundf(x2 / x3);
    int px = fpx != fpx ? 0 : (fpx >= 2147483648.f ? INT32_MAX : (fpx <= -2147483648.f ? INT32_MIN : (int)fpx));
    int py = fpy != fpy ? 0 : (fpy >= 2147483648.f ? INT32_MAX : (fpy <= -2147483648.f ? INT32_MIN : (int)fpy));
    for (int c = 0; c < Cd; c++) td[c] = 0;
    for (int c = 0; c < Cl; c++) tl[c] = 0;
    int count = 0;
    /* window bounds in 64-bit: px +- ksize may overflow int for saturated px */
    long xlo = (long)px - ksize, xhi = (long)px + ksize, ylo = (long)py - ksize, yhi = (long)py + ksize;
    if (xlo < 0) xlo = 0;
    if (ylo < 0) ylo = 0;
    if (xhi > W - 1) xhi = W - 1;
    if (yhi > H - 1) yhi = H - 1;
    for (long x = xlo; x <= xhi; x++)
      for (long y = ylo; y <= yhi; y++) {
        long index_pixel = (long)n * H * W + y * W + x;
        float dep = depth[index_pixel];
        if (fabsf(dep - Z1) < threshold) {
          count++;
          for (int c = 0; c < Cd; c++) td[c] += data[index_pixel * Cd + c];
          for (int c = 0; c < Cl; c++) tl[c] += label[index_pixel * Cl + c];
        }
      }
    if (count == 0) {
      for (int c = 0; c < Cd; c++) tf[c] = 0;
      for (int c = 0; c < Cl; c++) tl[c] = label_3d[vox * Cl + c];
    } else {
      for (int c = 0; c < Cd; c++) { td[c] /= count; tf[c] = 1; }
      for (int c = 0; c < Cl; c++) tl[c] /= count;
    }
  }
}

int oracle_backproject(const float* data, const float* label, const float* depth,
                       const float* meta, const float* label_3d, int B, int H, int W, int Cd,
                       int Cl, int num_meta, int G, int ksize, float threshold, float* top_data,
                       float* top_label, float* top_flag)
{
  const long nvox = (long)B * G * G * G;
#pragma omp parallel for schedule(static)
  for (long vox = 0; vox < nvox; vox++)
    backproject_voxel(data, label, depth, meta, label_3d, H, W, Cd, Cl, num_meta, G, ksize, threshold, vox,
                      top_data + vox * Cd, top_label + vox * Cl, top_flag + vox * Cd);
  return 0;
}

/* The same voxels, every `stride`-th one starting at `first`, written compactly (row i = voxel first + i stride):
 * lets a test hold the kernels to the reference default grid_size = 256 (lib/fcn/config.py:106,222 — 16.7 M
 * voxels, 4.3 GB per output tensor) without the checker producing the full tensors. */
int oracle_backproject_sample(const float* data, const float* label, const float* depth,
                              const float* meta, const float* label_3d, int B, int H, int W, int Cd,
                              int Cl, int num_meta, int G, int ksize, float threshold, long first, long stride,
                              float* top_data, float* top_label, float* top_flag)
{
  const long nvox = (long)B * G * G * G;
  if (first < 0 || stride < 1) return -1;
  const long rows = first < nvox ? (nvox - first + stride - 1) / stride : 0;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < rows; i++)
    backproject_voxel(data, label, depth, meta, label_3d, H, W, Cd, Cl, num_meta, G, ksize, threshold,
                      first + i * stride, top_data + i * Cd, top_label + i * Cl, top_flag + i * Cd);
  return 0;
}

/* BackprojectBackward, backprojecting_op_gpu.cu.cc:159-217 */
int oracle_backproject_bwd(const float* top_diff, const float* depth, const float* meta, int B,
                           int H, int W, int Cd, int num_meta, int G, float* bottom_diff)
{
#pragma omp parallel for schedule(static)
  for (long pix = 0; pix < (long)B * H * W; pix++) {
    long t = pix;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H); t /= H;
    int n = (int)t;
    const float* md = meta + (size_t)n * num_meta;
    float dep = depth[pix];
    /* backproject the pixel: RX = depth * Kinv * (w, h, 1) */
    float RX = md[9] * w + md[10] * h + md[11];
    float RY = md[12] * w + md[13] * h + md[14];
    float RZ = md[15] * w + md[16] * h + md[17];
    float X = dep * RX, Y = dep * RY, Z = dep * RZ;
    float X1 = md[30] * X + md[31] * Y + md[32] * Z + md[33];
    float Y1 = md[34] * X + md[35] * Y + md[36] * Z + md[37];
    float Z1 = md[38] * X + md[39] * Y + md[40] * Z + md[41];
    float fvd = roundf((X1 - md[45]) / md[42]);
    float fvh = roundf((Y1 - md[46]) / md[43]);
    float fvw = roundf((Z1 - md[47]) / md[44]);
    int vd = fvd != fvd ? 0 : (fvd >= 2147483648.f ? INT32_MAX : (fvd <= -2147483648.f ? INT32_MIN : (int)fvd));
    int vh = fvh != fvh ? 0 : (fvh >= 2147483648.f ? INT32_MAX : (fvh <= -2147483648.f ? INT32_MIN : (int)fvh));
    int vw = fvw != fvw ? 0 : (fvw >= 2147483648.f ? INT32_MAX : (fvw <= -2147483648.f ? INT32_MIN : (int)fvw));
    for (int c = 0; c < Cd; c++) {
      float g = 0;
      if (vd >= 0 && vd < G && vh >= 0 && vh < G && vw >= 0 && vw < G)
        g = top_diff[((((long)n * G + vd) * G + vh) * G + vw) * Cd + c];
      bottom_diff[pix * Cd + c] = g;
    }
  }
  return 0;
}

/* canonical exp of the softmax layers (network.py:474-488), all-f32, one IEEE operation per step, no
 * FMA (this file is built -ffp-contract=off) — the same sequence as exp_softmax_f32 in
 * posecnn_amd/csrc/pcnn_device.h and tests/np_ref.py; the bits of TF's exp are unknowable. */
float oracle_exp_softmax(float x)
{
  if (x != x) return x;
  x = fminf(fmaxf(x, -104.f), 88.f);
  const float kf = rintf(x * 1.44269502f);
  float r = x - kf * 0.693145752f;
  r = r - kf * 1.42860677e-06f;
  float p = 1.98412698e-04f;
  p = p * r + 1.38888889e-03f;
  p = p * r + 8.33333377e-03f;
  p = p * r + 4.16666679e-02f;
  p = p * r + 1.66666672e-01f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  const int k = (int)kf;
  union { unsigned u; float f; } sc;
  if (k < -126) { sc.u = (unsigned)(k + 64 + 127) << 23; return (p * sc.f) * 5.42101086e-20f; }
  sc.u = (unsigned)(k + 127) << 23;
  return p * sc.f;
}

/* ------------------------------------------------------------------------------------------ */
/* softmax_high_dimension + argmax_2d, lib/networks/network.py:474-488, 432-434                 */
/* (tf.reduce_max, tf.exp(x - m), tf.reduce_sum ascending, tf.div, tf.argmax = first maximum)    */
/* ------------------------------------------------------------------------------------------ */
int oracle_softmax_argmax(const float* score, long N, int C, float* prob, int* label)
{
#pragma omp parallel for
  for (long i = 0; i < N; i++) {
    const float* s = score + i * C;
    float m = s[0];
    for (int c = 1; c < C; c++) m = fmaxf(m, s[c]);
    float e[1024];
    float sum = 0;
    for (int c = 0; c < C; c++) { e[c] = oracle_exp_softmax(s[c] - m); sum += e[c]; }
    int best = 0;
    float bestp = e[0] / sum;
    for (int c = 0; c < C; c++) {
      float p = e[c] / sum;
      if (prob) prob[i * C + c] = p;
      if (p > bestp) { bestp = p; best = c; }
    }
    label[i] = best;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Fixed bilinear "deconv": tf.nn.conv2d_transpose 'SAME' with make_deconv_filter's diagonal     */
/* filter (lib/networks/network.py:141-157, 207-222): out[b, s*i + t - pad, ...] += f[ty]*f[tx]*in */
/* Canonical arithmetic (no bit-truth exists for cuDNN): gather form, input rows then cols       */
/* ascending, acc += (fy*fx)*in; then + add1, + add2, + bias, ReLU.                              */
/* ------------------------------------------------------------------------------------------ */
static float bilinear_tap(int t, int k)
{
  const int f = (k + 1) / 2;
  const double c = (double)(2 * f - 1 - f % 2) / (2.0 * (double)f);
  return (float)(1.0 - fabs((double)t / (double)f - c));
}

int oracle_deconv_bilinear(const float* in, int B, int H, int W, int C, int k, int s,
                           const float* add1, const float* add2, const float* bias, int relu,
                           float* out)
{
  const int pad = (k - s) / 2, Ho = H * s, Wo = W * s;
#pragma omp parallel for schedule(static)
  for (long row = 0; row < (long)B * Ho; row++) {
    const int b = (int)(row / Ho), oy = (int)(row % Ho);
    for (int ox = 0; ox < Wo; ox++) {
      float* o = out + ((size_t)row * Wo + ox) * C;
      for (int c = 0; c < C; c++) o[c] = 0.f;
      int iy0 = (oy + pad - k + 1) / s - 1, ix0 = (ox + pad - k + 1) / s - 1;
      if (iy0 < 0) iy0 = 0;
      if (ix0 < 0) ix0 = 0;
      for (int iy = iy0; iy < H && s * iy <= oy + pad; iy++) {
        const int ty = oy + pad - s * iy;
        if (ty < 0 || ty >= k) continue;
        for (int ix = ix0; ix < W && s * ix <= ox + pad; ix++) {
          const int tx = ox + pad - s * ix;
          if (tx < 0 || tx >= k) continue;
          const float w = bilinear_tap(ty, k) * bilinear_tap(tx, k);
          const float* p = in + (((size_t)b * H + iy) * W + ix) * C;
          for (int c = 0; c < C; c++) o[c] = o[c] + w * p[c];
        }
      }
      const size_t off = ((size_t)row * Wo + ox) * C;
      for (int c = 0; c < C; c++) {
        float v = o[c];
        if (add1) v = v + add1[off + c];
        if (add2) v = v + add2[off + c];
        if (bias) v = v + bias[c];
        if (relu) v = v > 0.f ? v : 0.f;
        o[c] = v;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Gradient of the fixed bilinear deconv (the transpose of oracle_deconv_bilinear; what TF's
 * Conv2DBackpropInput gradient computes for network.py:207-222 with the make_deconv_filter
 * weights): grad_in[b,i,j,c] = sum_{oy,ox} tap(oy+pad-s*i) * tap(ox+pad-s*j) * grad_out[b,oy,ox,c].
 * Canonical order: oy ascending, ox ascending, acc = acc + (wy*wx)*g.
 * ---------------------------------------------------------------------------------------------- */
int oracle_deconv_bilinear_bwd(const float* grad_out, int B, int H, int W, int C, int k, int s,
                               float* grad_in)
{
  const int pad = (k - s) / 2, Ho = H * s, Wo = W * s;
#pragma omp parallel for schedule(static)
  for (long row = 0; row < (long)B * H; row++) {
    const int b = (int)(row / H), i = (int)(row % H);
    for (int j = 0; j < W; j++) {
      float* gi = grad_in + ((size_t)row * W + j) * C;
      for (int c = 0; c < C; c++) gi[c] = 0.f;
      for (int ty = 0; ty < k; ty++) {
        const int oy = s * i + ty - pad;
        if (oy < 0 || oy >= Ho) continue;
        for (int tx = 0; tx < k; tx++) {
          const int ox = s * j + tx - pad;
          if (ox < 0 || ox >= Wo) continue;
          const float w = bilinear_tap(ty, k) * bilinear_tap(tx, k);
          const float* g = grad_out + (((size_t)b * Ho + oy) * Wo + ox) * C;
          for (int c = 0; c < C; c++) gi[c] = gi[c] + w * g[c];
        }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * smooth_l1_loss_vertex (lib/fcn/train.py:564-573):
 *   diff = w * (pred - target); in = |diff| < 1/sigma^2 ? diff^2 * sigma^2/2 : |diff| - 0.5/sigma^2
 *   loss = sum(in) / (sum(w) + 1e-10)
 * TF's reduce_sum order is unspecified; canonical order = the HIP kernel's: SL1_BLOCKS x 256
 * virtual threads, thread (blk, t) adds elements (blk*256 + t) + m*SL1_BLOCKS*256, m ascending,
 * into f32 accumulators; a 256-leaf then a SL1_BLOCKS-leaf halving tree (x[t] += x[t + stride],
 * stride = n/2 .. 1) combines them. out[0] = loss, out[1] = sum(in), out[2] = sum(w).
 * grad (d loss / d pred) = w * (|diff| < 1/sigma^2 ? sigma^2 * diff : sign(diff)) / (sum(w) + 1e-10).
 * ---------------------------------------------------------------------------------------------- */
#define SL1_BLOCKS 1024

static void sl1_elem(float p, float t, float w, float sigma2, float* in_loss, float* dpred)
{
  const float diff = w * (p - t);
  const float ad = fabsf(diff);
  const float inv = 1.0f / sigma2;
  if (ad < inv) {
    *in_loss = (diff * diff) * (sigma2 / 2.0f);
    *dpred = w * (sigma2 * diff);
  } else {
    *in_loss = ad - 0.5f / sigma2;
    *dpred = w * (diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f));
  }
}

int oracle_smooth_l1_vertex(const float* pred, const float* target, const float* weight, long n,
                            float sigma, float* out, float* grad)
{
  const float sigma2 = sigma * sigma;
  float* pl = (float*)malloc(sizeof(float) * SL1_BLOCKS * 2);
  float* pw = pl + SL1_BLOCKS;
#pragma omp parallel for schedule(static)
  for (int blk = 0; blk < SL1_BLOCKS; blk++) {
    float sl[256], sw[256];
    for (int t = 0; t < 256; t++) {
      float al = 0.f, aw = 0.f;
      for (long i = (long)blk * 256 + t; i < n; i += (long)SL1_BLOCKS * 256) {
        float il, dp;
        sl1_elem(pred[i], target[i], weight[i], sigma2, &il, &dp);
        al = al + il;
        aw = aw + weight[i];
      }
      sl[t] = al; sw[t] = aw;
    }
    for (int st = 128; st >= 1; st >>= 1)
      for (int t = 0; t < st; t++) { sl[t] = sl[t] + sl[t + st]; sw[t] = sw[t] + sw[t + st]; }
    pl[blk] = sl[0]; pw[blk] = sw[0];
  }
  for (int st = SL1_BLOCKS / 2; st >= 1; st >>= 1)
    for (int t = 0; t < st; t++) { pl[t] = pl[t] + pl[t + st]; pw[t] = pw[t] + pw[t + st]; }
  const float denom = pw[0] + 1e-10f;
  out[0] = pl[0] / denom; out[1] = pl[0]; out[2] = pw[0];
  free(pl);
  if (grad) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
      float il, dp;
      sl1_elem(pred[i], target[i], weight[i], sigma2, &il, &dp);
      grad[i] = dp / denom;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * H7 — the reference's CPU kernel of the same op (HoughvotinggpuOp<CPUDevice>,
 * hough_voting_gpu_op.cc:132-297; hough_voting :486-672; compute_width_height :679-758). It is a
 * DIFFERENT algorithm from the GPU kernels (every foreground pixel marches a ray along its
 * predicted direction and increments the cells it crosses; one maximum per class, >= 50 votes)
 * and therefore NOT a parity target — it is restated only because it is what `demo.sh` executes on a
 * machine without a GPU (BASELINE configs[0]); bench.py times it beside the GPU path.
 * The OpenCV types of the original (cv::Mat, projectPoints with zero rotation) are written out:
 * a corner (X,Y,Z) of the extents box at distance d projects to fx*X/(Z+d)+px, fy*Y/(Z+d)+py.
 * out rows: (batch, cls, x1, y1, x2, y2, votes, 1,0,0,0, tx, ty, tz); returns the row count.
 * ---------------------------------------------------------------------------------------------- */
#define PCNN_MAX_CLASSES_ORACLE 256

static int cmp_float(const void* a, const void* b)
{
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

int oracle_hough_cpu_kernel(const int* label, const float* vertex, const float* extents,
                            const float* meta, int B, int H, int W, int C, int num_meta, float* out,
                            int max_rows)
{
  const float inlier = 0.9f;
  const int voting_threshold = 50;
  int rows = 0;
  if (C < 1 || C > PCNN_MAX_CLASSES_ORACLE) return -1;
  int* hs = (int*)malloc(sizeof(int) * (size_t)H * W * C);
  float* dxs = (float*)malloc(sizeof(float) * (size_t)H * W);
  float* dys = (float*)malloc(sizeof(float) * (size_t)H * W);
  for (int n = 0; n < B; n++) {
    const int* lab = label + (size_t)n * H * W;
    const float* vm = vertex + (size_t)n * H * W * 3 * C;
    const float fx = meta[n * num_meta + 0], px = meta[n * num_meta + 2];
    const float fy = meta[n * num_meta + 4], py = meta[n * num_meta + 5];
    memset(hs, 0, sizeof(int) * (size_t)H * W * C);  /* (:501-502 clears a quarter; fresh pages are zero) */
    char flags[PCNN_MAX_CLASSES_ORACLE];
    memset(flags, 0, sizeof flags);
    for (int x = 0; x < W; x++)
      for (int y = 0; y < H; y++) {                                            /* :507-545 */
        const int c = lab[y * W + x];
        if (c <= 0 || c >= C) continue;
        flags[c] = 1;
        const float* vp = vm + (size_t)(y * W + x) * 3 * C + 3 * c;
        float u = vp[0], v = vp[1];
        const float nrm = sqrtf(u * u + v * v);
        u /= nrm; v /= nrm;
        const float delta = 1.0f / fabsf(u);
        if (!(delta < 1e30f)) continue;      /* u == 0 or NaN: the original never terminates / is undefined */
        float cx = (float)x, cy = (float)y;
        for (;;) {
          cx += delta * u; cy += delta * v;
          if (!(fabsf(cy) < 1e9f)) break;    /* int(cy) would overflow (undefined in the original) */
          const int ix = (int)cx, iy = (int)cy;   /* truncation: (-1, 0) counts as cell 0, as in the original */
          if (ix >= 0 && ix < W && iy >= 0 && iy < H) hs[c + C * (iy * W + ix)] += 1;
          else break;
        }
      }
    for (int c = 1; c < C; c++) {                                              /* :548-566 */
      if (!flags[c]) continue;
      int max_vote = 0, max_x = 0, max_y = 0;
      for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
          const int v = hs[c + C * (y * W + x)];
          if (v > max_vote) { max_vote = v; max_x = x; max_y = y; }
        }
      if (max_vote < voting_threshold) continue;
      /* compute_width_height :679-758 */
      float dsum = 0.f;
      int cnt = 0;
      for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
          if (lab[y * W + x] != c) continue;
          const float* vp = vm + (size_t)(y * W + x) * 3 * C + 3 * c;
          float u = vp[0], v = vp[1];
          const float dist = oracle_expf(vp[2]);
          const float nrm = sqrtf(u * u + v * v);
          u /= nrm; v /= nrm;
          const float ddx = (float)max_x - x, ddy = (float)max_y - y;
          const float ang = (u * ddx + v * ddy) / (sqrtf(u * u + v * v) * sqrtf(ddx * ddx + ddy * ddy));
          if (ang > inlier) { dxs[cnt] = fabsf(ddx); dys[cnt] = fabsf(ddy); dsum += dist; cnt++; }
        }
      if (cnt == 0) continue;                /* the original divides by zero and indexes an empty vector */
      const float bb_distance = dsum / cnt;
      int minX = 100000000, maxX = -100000000, minY = 100000000, maxY = -100000000;
      for (int i = 0; i < 8; i++) {
        const float X = (i & 1 ? 0.5f : -0.5f) * extents[3 * c + 0];
        const float Y = (i & 2 ? 0.5f : -0.5f) * extents[3 * c + 1];
        const float Z = (i & 4 ? 0.5f : -0.5f) * extents[3 * c + 2] + bb_distance;
        const float qx = fx * X / Z + px, qy = fy * Y / Z + py;
        minX = (int)fminf((float)minX, qx); maxX = (int)fmaxf((float)maxX, qx);
        minY = (int)fminf((float)minY, qy); maxY = (int)fmaxf((float)maxY, qy);
      }
      const float lim = (float)((maxX - minX + 1) > (maxY - minY + 1) ? (maxX - minX + 1) : (maxY - minY + 1));
      int nx = 0, ny = 0;
      for (int i = 0; i < cnt; i++) if (!(dxs[i] > lim)) dxs[nx++] = dxs[i];
      for (int i = 0; i < cnt; i++) if (!(dys[i] > lim)) dys[ny++] = dys[i];
      if (nx == 0 || ny == 0) continue;
      qsort(dxs, nx, sizeof(float), cmp_float);
      qsort(dys, ny, sizeof(float), cmp_float);
      const int bb_w = (int)(2 * dxs[(int)(nx * 0.95)]), bb_h = (int)(2 * dys[(int)(ny * 0.95)]);
      if (rows < max_rows) {                                                   /* :575-603 */
        float* r = out + 14 * rows;
        const float scale = 0.05f;
        r[0] = (float)n; r[1] = (float)c;
        r[2] = max_x - bb_w * (0.5f + scale); r[3] = max_y - bb_h * (0.5f + scale);
        r[4] = max_x + bb_w * (0.5f + scale); r[5] = max_y + bb_h * (0.5f + scale);
        r[6] = (float)max_vote;
        r[7] = 1; r[8] = 0; r[9] = 0; r[10] = 0;
        r[11] = (max_x - px) / fx * bb_distance; r[12] = (max_y - py) / fy * bb_distance; r[13] = bb_distance;
        rows++;
      }
    }
  }
  free(hs); free(dxs); free(dys);
  return rows;
}

/* ================================================================================================== */
/* Depth-based pose refinement, first slice: the projective point-to-plane ICP core.   */
/*                                                                                                      */
/* Follows lib/kinect_fusion/src/optimization/icp.cu:25-136 (`icpKernel`: one Jacobian row + residual    */
/* per pixel of the PREDICTED vertex / normal maps), lib/kinect_fusion/src/optimization/icp.cpp:20-106   */
/* (`df::icp`: numIterations x { reduce J^T J / J^T r, LDLT solve, update = exp(solution),              */
/* accumulated = update * accumulated }), include/df/optimization/linearSystems.h:181-203 (the sum),     */
/* include/df/camera/poly3.h (projection with k1 = k2 = k3 = 0, as lib/synthesize/synthesize.cpp:2060-  */
/* 2069 builds the model) and the masked depth -> vertex-map step of Synthesizer::solveICP               */
/* (lib/synthesize/synthesize.cpp:2139-2155 + src/image/backprojection.cu:10-27). Called from            */
/* Synthesizer::refinePose (synthesize.cpp:2020-2026), i.e. lib/fcn/test.py:1925-1933.                   */
/*                                                                                                      */
/* PARITY UNPINNED: df::icp needs Eigen, Sophus, thrust and CUDA (none present) and the reference holds  */
/* no test vectors for it. The restatement is anchored statement by statement on icpKernel and pinned    */
/* only by hand-derived known answers (tests/test_icp.py). Canonical choices where the reference is      */
/* unspecified or order dependent:                                                                      */
/*   - thrust::transform_reduce has no defined order -> pixels in raster order in blocks of 256, a        */
/*     halving tree (i += i + 128, 64, ... 1) in f32 inside a block; the block sums added in f64 in 8      */
/*     contiguous segments (ascending inside a segment, then the segments ascending);                    */
/*   - the 6x6 solve and the pose bookkeeping are float in the reference (Eigen LDLT with pivoting,       */
/*     Sophus::SE3f): here LDL^T without pivoting, exp and the accumulated update in f64, with sin/cos    */
/*     replaced by fixed 10-term Taylor polynomials in theta^2 (pure + and *: identical bits on every     */
/*     IEEE machine); the kernel sees the accumulated update rounded to f32 like an SE3f;                */
/*   - a pivot that is not clearly positive drops its variable (no inlier at all: the update is unchanged). */
/* ================================================================================================== */
#define ICP_BLOCK 256
#define ICP_NSUM 29 /* 21 upper-triangular J^T J (row-major), 6 J^T r, inlier count, sum r^2 */

/* solveICP :2139-2155 (depth of the object's pixels, 0 elsewhere) + backprojectKernel + Poly3 unproject, k = 0 */
int oracle_icp_backproject(const uint16_t* depth, const int* label, int H, int W, int obj_id, float factor,
                           float fx, float fy, float px, float py, float* vertex_map)
{
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const long i = (long)y * W + x;
      const float d = (label == NULL || label[i] == obj_id) ? (float)depth[i] / factor : 0.f;
      vertex_map[3 * i + 0] = ((float)x - px) / fx * d;
      vertex_map[3 * i + 1] = ((float)y - py) / fy * d;
      vertex_map[3 * i + 2] = d;
    }
  return 0;
}

/* float -> int the way the GPU converts (`const int u = projected + 0.5` of icp.cu:78-79 is undefined in C++ for NaN /
   out-of-range values; PTX cvt.rzi and gfx950 v_cvt_i32_f32: NaN -> 0, saturating) */
static int icp_cvt_rz(float f)
{
  if (f != f) return 0;
  if (f >= 2147483648.f) return INT32_MAX;
  if (f <= -2147483648.f) return INT32_MIN;
  return (int)f;
}

/* The content of the Sophus::SE3f that icpKernel receives (`updatedPose`): unit quaternion (w, x, y, z) + translation,
   from the accumulated 3x4 transform T (row-major; f64 in the solve, rounded to f32 here). Rotation matrix -> quaternion
   by Eigen's published algorithm (Geometry/Quaternion.h, quaternionbase_assign_impl<.,3,3>: the trace branch or the
   largest-diagonal branch), then Sophus's normalisation (coefficients / sqrt(squaredNorm), 4-term reduction in Eigen's
   unrolled order over (x, y, z, w)) — all in f32 with correctly rounded sqrt and division. */
static void icp_se3f_from_matrix(const float* T /* [12] */, float* q /* wxyz */, float* t)
{
  const float m[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
  float qq[4];   /* x, y, z, w */
  float tr = (m[0][0] + m[1][1]) + m[2][2];
  if (tr > 0.f) {
    tr = sqrtf(tr + 1.0f);
    qq[3] = 0.5f * tr;
    tr = 0.5f / tr;
    qq[0] = (m[2][1] - m[1][2]) * tr;
    qq[1] = (m[0][2] - m[2][0]) * tr;
    qq[2] = (m[1][0] - m[0][1]) * tr;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    tr = sqrtf(((m[i][i] - m[j][j]) - m[k][k]) + 1.0f);
    qq[i] = 0.5f * tr;
    tr = 0.5f / tr;
    qq[3] = (m[k][j] - m[j][k]) * tr;
    qq[j] = (m[j][i] + m[i][j]) * tr;
    qq[k] = (m[k][i] + m[i][k]) * tr;
  }
  const float n = sqrtf((qq[0] * qq[0] + qq[1] * qq[1]) + (qq[2] * qq[2] + qq[3] * qq[3]));
  q[0] = qq[3] / n; q[1] = qq[0] / n; q[2] = qq[1] / n; q[3] = qq[2] / n;
  t[0] = T[3]; t[1] = T[7]; t[2] = T[11];
}

/* Sophus::SE3f * point = so3 * p + translation, so3 * p = Eigen's Quaternion::_transformVector:
   uv = 2 (q.vec x p); (p + w uv) + q.vec x uv */
static void icp_se3f_apply(const float* q /* wxyz */, const float* t, float x, float y, float z, float* o)
{
  const float w = q[0], a = q[1], b = q[2], c = q[3];
  float ux = b * z - c * y, uy = c * x - a * z, uz = a * y - b * x;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = b * uz - c * uy, cy = c * ux - a * uz, cz = a * uy - b * ux;
  o[0] = ((x + w * ux) + cx) + t[0];
  o[1] = ((y + w * uy) + cy) + t[1];
  o[2] = ((z + w * uz) + cz) + t[2];
}

/* icpKernel (icp.cu:25-136) for one pixel, in the body's own order of tests and — where the result of a float
   expression depends on it — in the evaluation order Eigen publishes for fixed-size 3-vectors (3-term reductions as
   t0 + (t1 + t2); oracle/ref_shim/eigen_sophus_on_cpu.h has the list). tests/test_oracle_vs_reference.py holds this to
   the reference's kernel body, compiled unchanged, bit for bit (J, r and the exit reason of every pixel).
   Returns the exit reason — 0 contributes (J[6], r filled), 1 predicted depth out of range (:60), 2 projects onto
   the border (:81), 3 live depth out of range (:92), 4 ray / normal angle (:104), 5 |error| > maxError (:115). */
enum { ICP_OK = 0, ICP_PRED_DEPTH = 1, ICP_BORDER = 2, ICP_LIVE_DEPTH = 3, ICP_RAY_NORMAL = 4, ICP_ERROR = 5 };
static int icp_pixel(const float* live, const float* pv, const float* pn, int W, int H, const float* q /* wxyz */, const float* t,
                     float fx, float fy, float px, float py, float znear, float zfar, float max_error, float* J, float* r)
{
  const float border = 2.f, ray_norm_dot_threshold = 0.1f;
  const float pdepth = pv[2];
  if ((pdepth < znear) || pdepth > zfar) return ICP_PRED_DEPTH;      /* :60 (a NaN — the render's clear colour — passes, as in the body) */
  float u3[3];
  icp_se3f_apply(q, t, pv[0], pv[1], pv[2], u3);                     /* :67 */
  const float ux = u3[0], uy = u3[1], uz = u3[2];
  /* :69 Poly3CameraModel::project with k1 = k2 = k3 = 0 (poly3.h:72-88, cameraModel.h:72-82): the distortion factor is
     ((1 + 0 r2) + 0 r4) + 0 r6 — exactly 1 for finite r2, NaN (0 x inf) when the squared radius overflows */
  const float dhx = ux / uz, dhy = uy / uz;
  const float r2 = dhx * dhx + dhy * dhy, r4 = r2 * r2, r6 = r4 * r2;
  const float factor = ((1.f + 0.f * r2) + 0.f * r4) + 0.f * r6;
  const float projx = (factor * dhx) * fx + px, projy = (factor * dhy) * fy + py;
  const int u = icp_cvt_rz(projx + 0.5f), v = icp_cvt_rz(projy + 0.5f);          /* :78-79 */
  if (((float)u <= border) || ((float)u >= (float)(unsigned)(W - 1) - border) || ((float)v <= border) || ((float)v >= (float)(unsigned)(H - 1) - border))
    return ICP_BORDER;                                                /* :81 */
  const float* lv = live + 3 * ((long)v * W + u);
  const float ldepth = lv[2];
  if ((ldepth < znear) || (ldepth > zfar)) return ICP_LIVE_DEPTH;    /* :92 */
  const float sq = ux * ux + (uy * uy + uz * uz);                    /* normalized(): squaredNorm, then v / sqrt(z) when z > 0 */
  float rx = ux, ry = uy, rz = uz;
  if (sq > 0.f) { const float nrm = sqrtf(sq); rx = ux / nrm; ry = uy / nrm; rz = uz / nrm; }   /* :100 */
  const float dotrn = rx * pn[0] + (ry * pn[1] + rz * pn[2]);
  if (-dotrn < ray_norm_dot_threshold) return ICP_RAY_NORMAL;        /* :104 */
  const float ex = lv[0] - ux, ey = lv[1] - uy, ez = lv[2] - uz;
  const float error = pn[0] * ex + (pn[1] * ey + pn[2] * ez);        /* :111 */
  if (fabsf(error) > max_error) return ICP_ERROR;                    /* :115 */
  const float w = 1.f / ldepth;                                      /* :122 */
  const float wx = w * pn[0], wy = w * pn[1], wz = w * pn[2];        /* (weightSqrt n^T) first, then x [I | -[p]x] (:124-130), */
  J[0] = wx * 1.f + (wy * 0.f + wz * 0.f);                           /* coefficient by coefficient: c0 + (c1 + c2) */
  J[1] = wx * 0.f + (wy * 1.f + wz * 0.f);
  J[2] = wx * 0.f + (wy * 0.f + wz * 1.f);
  J[3] = wx * 0.f + (wy * (-uz) + wz * uy);
  J[4] = wx * uz + (wy * 0.f + wz * (-ux));
  J[5] = wx * (-uy) + (wy * ux + wz * 0.f);
  *r = w * error;
  return ICP_OK;
}

/* the per-pixel records of one icpKernel launch: J [P,6], r [P] (zero where the pixel does not contribute, as the body
   leaves them), reason [P] (the enum above). pose = the content of the SE3f: q wxyz, t. */
int oracle_icp_terms(const float* live, const float* pred_v, const float* pred_n, int H, int W, int pc, const float* q,
                     const float* t, float fx, float fy, float px, float py, float znear, float zfar, float max_error,
                     float* J_out, float* r_out, unsigned char* reason_out)
{
  const long P = (long)H * W;
  for (long p = 0; p < P; p++) {
    float J[6] = {0, 0, 0, 0, 0, 0}, r = 0.f;
    const int why = icp_pixel(live, pred_v + pc * p, pred_n + pc * p, W, H, q, t, fx, fy, px, py, znear, zfar, max_error, J, &r);
    for (int k = 0; k < 6; k++) J_out[6 * p + k] = why == ICP_OK ? J[k] : 0.f;
    r_out[p] = why == ICP_OK ? r : 0.f;
    reason_out[p] = (unsigned char)why;
  }
  return 0;
}

/* the SE3f content icp_refine hands to the per-pixel step for an accumulated transform T (f64 [12]) */
int oracle_icp_se3f(const double* T, float* q, float* t)
{
  float Tf[12];
  for (int i = 0; i < 12; i++) Tf[i] = (float)T[i];
  icp_se3f_from_matrix(Tf, q, t);
  return 0;
}

static void icp_exp_se3(const double* xi, double* U /* [12] */)
{
  /* Sophus::SE3::exp: xi = (upsilon, omega); R = I + A W + B W^2, t = (I + B W + C W^2) upsilon,
     A = sin(th)/th, B = (1 - cos th)/th^2, C = (th - sin th)/th^3 as 10-term Taylor series in th^2 (Horner) */
  const double wx = xi[3], wy = xi[4], wz = xi[5];
  const double t2 = (wx * wx + wy * wy) + wz * wz;
  static const double fa[10] = {1.0, 6.0, 120.0, 5040.0, 362880.0, 39916800.0, 6227020800.0, 1307674368000.0, 355687428096000.0, 121645100408832000.0};              /* (2k+1)! */
  static const double fb[10] = {2.0, 24.0, 720.0, 40320.0, 3628800.0, 479001600.0, 87178291200.0, 20922789888000.0, 6402373705728000.0, 2432902008176640000.0};        /* (2k+2)! */
  static const double fc[10] = {6.0, 120.0, 5040.0, 362880.0, 39916800.0, 6227020800.0, 1307674368000.0, 355687428096000.0, 121645100408832000.0, 51090942171709440000.0}; /* (2k+3)! */
  double A = 0, B = 0, C = 0;
  for (int k = 9; k >= 0; k--) {
    const double s = (k & 1) ? -1.0 : 1.0;
    A = A * t2 + s / fa[k];
    B = B * t2 + s / fb[k];
    C = C * t2 + s / fc[k];
  }
  const double Wm[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W2[3 * i + j] = (Wm[3 * i] * Wm[j] + Wm[3 * i + 1] * Wm[3 + j]) + Wm[3 * i + 2] * Wm[6 + j];
  for (int i = 0; i < 3; i++) {
    double t = 0;
    for (int j = 0; j < 3; j++) {
      const double id = i == j ? 1.0 : 0.0;
      U[4 * i + j] = (id + A * Wm[3 * i + j]) + B * W2[3 * i + j];
      const double V = (id + B * Wm[3 * i + j]) + C * W2[3 * i + j];
      t = t + V * xi[j];
    }
    U[4 * i + 3] = t;
  }
}

/* one Gauss-Newton solve + pose bookkeeping (icp.cpp:58-100) */
static int icp_solve_update(const double* S /* [ICP_NSUM] */, double* T /* [12], in/out */)
{
  double A[6][6], b[6], L[6][6], d[6], y[6], x[6];
  int q = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) { A[i][j] = A[j][i] = S[q]; q++; }
  for (int i = 0; i < 6; i++) b[i] = S[21 + i];
  /* LDL^T without pivoting. A pivot that is not clearly positive (below 1e-10 of the largest diagonal entry: a direction
     the visible surface does not constrain — e.g. sliding along the only two faces in view — or no inlier at all) is
     dropped: that variable stays 0 instead of the solve blowing up (Eigen's pivoted LDLT of the reference does the
     equivalent for a rank-deficient system). */
  double maxdiag = 0.0;
  int skip[6];
  for (int i = 0; i < 6; i++) if (A[i][i] > maxdiag) maxdiag = A[i][i];
  const double tol = 1e-10 * maxdiag;
  for (int j = 0; j < 6; j++) {
    double dj = A[j][j];
    for (int k = 0; k < j; k++) dj = dj - (L[j][k] * L[j][k]) * d[k];
    skip[j] = !(dj > tol) || !(dj < 1e300);
    d[j] = skip[j] ? 0.0 : dj;
    for (int i = j + 1; i < 6; i++) {
      double v = A[i][j];
      for (int k = 0; k < j; k++) v = v - (L[i][k] * L[j][k]) * d[k];
      L[i][j] = skip[j] ? 0.0 : v / dj;
    }
  }
  for (int i = 0; i < 6; i++) { double v = b[i]; for (int k = 0; k < i; k++) v = v - L[i][k] * y[k]; y[i] = v; }
  for (int i = 0; i < 6; i++) y[i] = skip[i] ? 0.0 : y[i] / d[i];
  for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v = v - L[k][i] * x[k]; x[i] = v; }
  double U[12], N[12];
  icp_exp_se3(x, U);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) N[4 * i + j] = (U[4 * i] * T[j] + U[4 * i + 1] * T[4 + j]) + U[4 * i + 2] * T[8 + j];
    N[4 * i + 3] = ((U[4 * i] * T[3] + U[4 * i + 1] * T[7]) + U[4 * i + 2] * T[11]) + U[4 * i + 3];
  }
  memcpy(T, N, sizeof(N));
  return 1;
}

/* the canonical f64 sum of per-block f32 partial rows [nblocks][nq]: ICP_NSEG contiguous segments of
   ceil(nblocks / ICP_NSEG) blocks, each added in ascending block order, then the segment sums in ascending order */
#define ICP_NSEG 8
static void icp_segmented_sums(const float* rows, long nblocks, int nq, double* S)
{
  const long L = (nblocks + ICP_NSEG - 1) / ICP_NSEG;
  for (int q = 0; q < nq; q++) {
    double seg[ICP_NSEG];
    for (int sg = 0; sg < ICP_NSEG; sg++) {
      const long b0 = sg * L, b1 = (b0 + L < nblocks) ? b0 + L : nblocks;
      double acc = 0.0;
      for (long b = b0; b < b1; b++) acc = acc + (double)rows[b * nq + q];
      seg[sg] = acc;
    }
    double s = seg[0];
    for (int sg = 1; sg < ICP_NSEG; sg++) s = s + seg[sg];
    S[q] = s;
  }
}

/* df::icp for N independent (live, predicted) map triples. pred_* have `pc` (3 or 4) floats per pixel.
   update_out [N][12] f64 (row-major 3x4 accumulated update), stats_out [N][iterations][2] f32 (inliers, sum r^2) or NULL */
int oracle_icp_refine(const float* live, const float* pred_v, const float* pred_n, int N, int H, int W, int pc,
                      float fx, float fy, float px, float py, float znear, float zfar, float max_error,
                      int iterations, double* update_out, float* stats_out)
{
  const long P = (long)H * W;
  const long nblocks = (P + ICP_BLOCK - 1) / ICP_BLOCK;
  float* buf = (float*)malloc(sizeof(float) * ICP_NSUM * ICP_BLOCK);
  float* rows = (float*)malloc(sizeof(float) * ICP_NSUM * (size_t)nblocks);
  if (!buf || !rows) { free(buf); free(rows); return -1; }
  for (int n = 0; n < N; n++) {
    double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const float* lv = live + 3 * P * n;
    const float* pv = pred_v + (long)pc * P * n;
    const float* pn = pred_n + (long)pc * P * n;
    for (int it = 0; it < iterations; it++) {
      float Tf[12], qf[4], tf[3];
      for (int i = 0; i < 12; i++) Tf[i] = (float)T[i];
      icp_se3f_from_matrix(Tf, qf, tf);
      double S[ICP_NSUM];
      memset(rows, 0, sizeof(float) * ICP_NSUM * (size_t)nblocks);
      for (long b = 0; b < nblocks; b++) {
        int any = 0;
        memset(buf, 0, sizeof(float) * ICP_NSUM * ICP_BLOCK);
        for (int t = 0; t < ICP_BLOCK; t++) {
          const long p = b * ICP_BLOCK + t;
          if (p >= P) break;
          float J[6], r;
          if (icp_pixel(lv, pv + pc * p, pn + pc * p, W, H, qf, tf, fx, fy, px, py, znear, zfar, max_error, J, &r) != ICP_OK) continue;
          any = 1;
          int q = 0;
          for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++) buf[(q++) * ICP_BLOCK + t] = J[i] * J[j];
          for (int i = 0; i < 6; i++) buf[(21 + i) * ICP_BLOCK + t] = J[i] * r;
          buf[27 * ICP_BLOCK + t] = 1.f;
          buf[28 * ICP_BLOCK + t] = r * r;
        }
        if (!any) continue;   /* all zeros: a zero row */
        for (int q = 0; q < ICP_NSUM; q++) {
          float* a = buf + q * ICP_BLOCK;
          for (int s = ICP_BLOCK / 2; s >= 1; s >>= 1)
            for (int t = 0; t < s; t++) a[t] = a[t] + a[t + s];
          rows[b * ICP_NSUM + q] = a[0];
        }
      }
      icp_segmented_sums(rows, nblocks, ICP_NSUM, S);
      if (stats_out) {
        stats_out[((long)n * iterations + it) * 2 + 0] = (float)S[27];
        stats_out[((long)n * iterations + it) * 2 + 1] = (float)S[28];
      }
      icp_solve_update(S, T);
    }
    memcpy(update_out + 12 * n, T, sizeof(T));
  }
  free(buf);
  free(rows);
  return 0;
}

/* ================================================================================================== */
/* solveICP around the iterations: predicted maps, translation estimate, hypothesis selection          */
/* ================================================================================================== */
/* Follows lib/synthesize/synthesize.cpp:2104-2136 + :1972-1991 (what the two GL renderers produce:     */
/* shaders lib/kinect_fusion/shaders/vertsAndNorms.{vert,frag}, canonicalVerts.{vert,frag}; projection   */
/* matrix :2088 = pixel centres at integer (u, v) of u = fx X / Z + px; canonical x + model index        */
/* :266-271; NaN clear colour :2113), :2157-2207 (translation estimate), :2302-2343 (SegICP score).      */
/*                                                                                                      */
/* PARITY UNPINNED: OpenGL rasterisation (fixed-point snapping, fill rule, vendor interpolation) is not  */
/* reproducible off the reference's GPU + driver, PCL/FLANN are absent and the reference holds no test    */
/* vectors. The renderer is pinned to an analytic ray-caster (tests/test_icp.py); canonical choices:      */
/*   - coverage: float edge functions evaluated from the lower-numbered vertex of an edge (both triangles */
/*     sharing it see the same number), inclusive on both sides; visibility = minimum of (depth, face     */
/*     index); triangles with a vertex in front of z_near are dropped, not clipped;                       */
/*   - attributes: perspective-correct, weight_i = (e_i / area) / z_i normalised by their sum;            */
/*   - sums of the translation estimate: blocks of 256 pixels in raster order, halving tree in f32,        */
/*     block sums in f64 in 8 ascending segments like the ICP sums (the reference: sequential float);    */
/*   - radius search: the nearest depth point strictly inside the radius, ties to the lower pixel index   */
/*     (FLANN's order among equal distances is unspecified); the score counts distinct marked points      */
/*     (the reference marks from an OpenMP loop: same count whatever the order).                         */

typedef struct {
  float u[3], v[3], z[3];
  int flip[3];
  int x0, x1, y0, y1;
} RdTri;

static void rd_transform(const float* T, const float* p, float* c)
{
  const float x = p[0], y = p[1], z = p[2];
  c[0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  c[1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  c[2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

static float rd_edge(float au, float av, float bu, float bv, float x, float y)
{
  return (bu - au) * (y - av) - (bv - av) * (x - au);
}

static int rd_setup(const float* T, const float* vtx, const int* face, int W, int H, float fx, float fy, float px,
                    float py, float znear, RdTri* t, float cam[3][3])
{
  const int idx[3] = {face[0], face[1], face[2]};
  for (int k = 0; k < 3; k++) {
    rd_transform(T, vtx + 3 * (size_t)idx[k], cam[k]);
    t->z[k] = cam[k][2];
    t->u[k] = cam[k][0] / cam[k][2] * fx + px;
    t->v[k] = cam[k][1] / cam[k][2] * fy + py;
  }
  if (idx[0] == idx[1] || idx[1] == idx[2] || idx[0] == idx[2]) return 0;
  for (int k = 0; k < 3; k++)
    if (!(t->z[k] >= znear) || !(fabsf(t->u[k]) < 1e7f) || !(fabsf(t->v[k]) < 1e7f)) return 0;
  t->flip[0] = idx[1] > idx[2];
  t->flip[1] = idx[2] > idx[0];
  t->flip[2] = idx[0] > idx[1];
  const float umin = fminf(fminf(t->u[0], t->u[1]), t->u[2]), umax = fmaxf(fmaxf(t->u[0], t->u[1]), t->u[2]);
  const float vmin = fminf(fminf(t->v[0], t->v[1]), t->v[2]), vmax = fmaxf(fmaxf(t->v[0], t->v[1]), t->v[2]);
  const int cx0 = (int)ceilf(umin), cx1 = (int)floorf(umax), cy0 = (int)ceilf(vmin), cy1 = (int)floorf(vmax);
  t->x0 = cx0 > 0 ? cx0 : 0;
  t->x1 = cx1 < W - 1 ? cx1 : W - 1;
  t->y0 = cy0 > 0 ? cy0 : 0;
  t->y1 = cy1 < H - 1 ? cy1 : H - 1;
  return t->x0 <= t->x1 && t->y0 <= t->y1;
}

static int rd_weights(const RdTri* t, float x, float y, float* w, float* s)
{
  float e[3];
  e[0] = t->flip[0] ? -rd_edge(t->u[2], t->v[2], t->u[1], t->v[1], x, y) : rd_edge(t->u[1], t->v[1], t->u[2], t->v[2], x, y);
  e[1] = t->flip[1] ? -rd_edge(t->u[0], t->v[0], t->u[2], t->v[2], x, y) : rd_edge(t->u[2], t->v[2], t->u[0], t->v[0], x, y);
  e[2] = t->flip[2] ? -rd_edge(t->u[1], t->v[1], t->u[0], t->v[0], x, y) : rd_edge(t->u[0], t->v[0], t->u[1], t->v[1], x, y);
  const int pos = e[0] >= 0.f && e[1] >= 0.f && e[2] >= 0.f;
  const int neg = e[0] <= 0.f && e[1] <= 0.f && e[2] <= 0.f;
  if (!(pos || neg)) return 0;
  const float area = (e[0] + e[1]) + e[2];
  if (area == 0.f) return 0;
  w[0] = e[0] / area / t->z[0];
  w[1] = e[1] / area / t->z[1];
  w[2] = e[2] / area / t->z[2];
  *s = (w[0] + w[1]) + w[2];
  return *s > 0.f;
}

static float rd_nan(void)
{
  union { uint32_t u; float f; } c;
  c.u = 0x7fc00000u;
  return c.f;
}

/* vertices [Nv,3], normals [Nv,3] (or NULL), faces [Nf,3], poses [N,12] f32 (row-major 3x4 camera <- object).
   out_v / out_n [N,H,W,4], out_c [N,H,W,3] (each optional); NaN where nothing is hit */
int oracle_render_mesh(const float* vtx, const float* nrm, const int* faces, int nv, int nf, const float* poses, int N,
                       int H, int W, float fx, float fy, float px, float py, float znear, float zfar, float canon_x_offset,
                       float* out_v, float* out_n, float* out_c)
{
  (void)nv;
  const long P = (long)H * W;
  uint64_t* zbuf = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)P);
  if (!zbuf) return -1;
  const float qnan = rd_nan();
  for (int n = 0; n < N; n++) {
    const float* T = poses + 12 * (size_t)n;
    for (long i = 0; i < P; i++) zbuf[i] = ~(uint64_t)0;
    for (int f = 0; f < nf; f++) {
      RdTri t;
      float cam[3][3];
      if (!rd_setup(T, vtx, faces + 3 * (size_t)f, W, H, fx, fy, px, py, znear, &t, cam)) continue;
      for (int y = t.y0; y <= t.y1; y++)
        for (int x = t.x0; x <= t.x1; x++) {
          float w[3], s;
          if (!rd_weights(&t, (float)x, (float)y, w, &s)) continue;
          const float z = ((w[0] * t.z[0] + w[1] * t.z[1]) + w[2] * t.z[2]) / s;
          if (!(z >= znear) || !(z <= zfar)) continue;
          union { float f; uint32_t u; } c;
          c.f = z;
          const uint64_t key = ((uint64_t)c.u << 32) | (uint32_t)f;
          if (key < zbuf[(long)y * W + x]) zbuf[(long)y * W + x] = key;
        }
    }
    for (long i = 0; i < P; i++) {
      const uint64_t key = zbuf[i];
      float ov[3] = {qnan, qnan, qnan}, on[3] = {qnan, qnan, qnan}, oc[3] = {qnan, qnan, qnan};
      const int hit = key != ~(uint64_t)0;
      if (hit) {
        const int* face = faces + 3 * (size_t)(uint32_t)(key & 0xffffffffu);
        RdTri t;
        float cam[3][3], w[3], s;
        rd_setup(T, vtx, face, W, H, fx, fy, px, py, znear, &t, cam);
        if (rd_weights(&t, (float)(i % W), (float)(i / W), w, &s)) {
          for (int k = 0; k < 3; k++) ov[k] = ((w[0] * cam[0][k] + w[1] * cam[1][k]) + w[2] * cam[2][k]) / s;
          if (out_n) {
            float vn[3][3];
            for (int j = 0; j < 3; j++) {
              const float* p = nrm + 3 * (size_t)face[j];
              const float a = p[0], b = p[1], c = p[2];
              const float rx = (T[0] * a + T[1] * b) + T[2] * c;
              const float ry = (T[4] * a + T[5] * b) + T[6] * c;
              const float rz = (T[8] * a + T[9] * b) + T[10] * c;
              const float len = sqrtf((rx * rx + ry * ry) + rz * rz);
              const int ok = len > 0.f;
              vn[j][0] = ok ? rx / len : rx;
              vn[j][1] = ok ? ry / len : ry;
              vn[j][2] = ok ? rz / len : rz;
            }
            for (int k = 0; k < 3; k++) on[k] = ((w[0] * vn[0][k] + w[1] * vn[1][k]) + w[2] * vn[2][k]) / s;
          }
          if (out_c)
            for (int k = 0; k < 3; k++) {
              const float off = k == 0 ? canon_x_offset : 0.f;
              const float a0 = vtx[3 * (size_t)face[0] + k] + off, a1 = vtx[3 * (size_t)face[1] + k] + off,
                          a2 = vtx[3 * (size_t)face[2] + k] + off;
              oc[k] = ((w[0] * a0 + w[1] * a1) + w[2] * a2) / s;
            }
        }
      }
      const size_t o = (size_t)n * P + i;
      if (out_v) { out_v[4 * o] = ov[0]; out_v[4 * o + 1] = ov[1]; out_v[4 * o + 2] = ov[2]; out_v[4 * o + 3] = hit ? 1.f : qnan; }
      if (out_n) { out_n[4 * o] = on[0]; out_n[4 * o + 1] = on[1]; out_n[4 * o + 2] = on[2]; out_n[4 * o + 3] = hit ? 0.f : qnan; }
      if (out_c) { out_c[3 * o] = oc[0]; out_c[3 * o + 1] = oc[1]; out_c[3 * o + 2] = oc[2]; }
    }
  }
  free(zbuf);
  return 0;
}

#define ICP_NCEN 5
/* synthesize.cpp:2157-2207. sums[5] = sum (d - m).xyz over the agreeing pixels, their count, number of valid pairs; mask [H*W] */
int oracle_icp_center(const int* label, const float* live, const float* canon, const float* pred_v, const float* pred_n, int pc,
                      int H, int W, int obj_id, float max_error, double* sums, uint8_t* mask)
{
  const long P = (long)H * W;
  const long nblocks = (P + ICP_BLOCK - 1) / ICP_BLOCK;
  float buf[ICP_NCEN][ICP_BLOCK];
  float* rows = (float*)malloc(sizeof(float) * ICP_NCEN * (size_t)nblocks);
  if (!rows) return -1;
  for (long b = 0; b < nblocks; b++) {
    memset(buf, 0, sizeof(buf));
    for (int t = 0; t < ICP_BLOCK; t++) {
      const long p = b * ICP_BLOCK + t;
      if (p >= P) break;
      int valid = 0;
      if (label[p] == obj_id) {
        const float dx = live[3 * p], dy = live[3 * p + 1], dz = live[3 * p + 2];
        if (dz > 0.f) {                                                   /* :2166 */
          const float cx = canon[3 * p], vy = canon[3 * p + 1], vz = canon[3 * p + 2];
          const float vx = cx - roundf(cx);                                /* :2168 */
          if (vx == vx && vy == vy && vz == vz) {                          /* :2172 */
            valid = 1;
            const float* pv = pred_v + p * pc;
            const float* pn = pred_n + p * pc;
            const float error = (pn[0] * (dx - pv[0]) + pn[1] * (dy - pv[1])) + pn[2] * (dz - pv[2]);   /* :2177 */
            if (fabsf(error) < max_error) {                                /* :2178 */
              buf[0][t] = dx - vx; buf[1][t] = dy - vy; buf[2][t] = dz - vz; buf[3][t] = 1.f;
            }
            buf[4][t] = 1.f;
          }
        }
      }
      mask[p] = (uint8_t)valid;
    }
    for (int q = 0; q < ICP_NCEN; q++) {
      float* a = buf[q];
      for (int s = ICP_BLOCK / 2; s >= 1; s >>= 1)
        for (int t = 0; t < s; t++) a[t] = a[t] + a[t + s];
      rows[b * ICP_NCEN + q] = a[0];
    }
  }
  icp_segmented_sums(rows, nblocks, ICP_NCEN, sums);
  free(rows);
  return 0;
}

/* synthesize.cpp:2302-2343 by exhaustive search. hyps [M,12] f32; hits [M] = number of distinct depth points that are
   the nearest one (strictly inside `radius`) of some model point moved by the hypothesis */
int oracle_icp_score(const float* live, const float* canon, const uint8_t* mask, int H, int W, const float* hyps, int M,
                     float radius, int* hits)
{
  const long P = (long)H * W;
  uint8_t* flags = (uint8_t*)malloc((size_t)P);
  if (!flags) return -1;
  const float r2 = radius * radius;
  for (int m = 0; m < M; m++) {
    const float* T = hyps + 12 * (size_t)m;
    memset(flags, 0, (size_t)P);
    int score = 0;
    for (long p = 0; p < P; p++) {
      if (!mask[p]) continue;
      const float cx = canon[3 * p];
      const float mx = cx - roundf(cx), my = canon[3 * p + 1], mz = canon[3 * p + 2];
      const float qx = ((T[0] * mx + T[1] * my) + T[2] * mz) + T[3];
      const float qy = ((T[4] * mx + T[5] * my) + T[6] * mz) + T[7];
      const float qz = ((T[8] * mx + T[9] * my) + T[10] * mz) + T[11];
      if (!(qx == qx) || !(qy == qy) || !(qz == qz)) continue;
      float best = r2;
      long bi = -1;
      for (long i = 0; i < P; i++) {
        if (!mask[i]) continue;
        const float ex = live[3 * i] - qx, ey = live[3 * i + 1] - qy, ez = live[3 * i + 2] - qz;
        const float d2 = (ex * ex + ey * ey) + ez * ez;
        if (d2 < best) { best = d2; bi = i; }
      }
      if (bi >= 0 && !flags[bi]) { flags[bi] = 1; score++; }
    }
    hits[m] = score;
  }
  free(flags);
  return 0;
}

/* ================================================================================================== */
/* solveICP's polish between the translation estimate and the hypotheses: Synthesizer::poseWithOpt      */
/* (lib/synthesize/synthesize.cpp:2529-2570: nlopt LN_NELDERMEAD, 7 variables = quaternion wxyz +        */
/* translation of an UPDATE applied on the left of T_co, box +-0.1 / +-0.01 (x, y) / +-0.1 (z) around    */
/* (1,0,0,0,0,0,0), maxeval = 50) minimising optEnergy (:2476-2526): the mean distance between the       */
/* moved predicted vertex and the depth point of the same pixel, over the object's label pixels.          */
/*                                                                                                      */
/* PARITY UNPINNED: nlopt is absent; what follows restates the published algorithm of nlopt's            */
/* nldrmd.c (Nelder-Mead with Box's bound handling: a trial point outside the box is moved onto it;        */
/* alpha = 1, beta = 0.5, gamma = 2, delta = 0.5; initial simplex x0 and x0 + step_i e_i with nlopt's      */
/* default initial step (ub - lb) / 4) — NOT its code, and not its bits. Canonical choices:               */
/*   - the energy: pixels of the label's bounding box in raster order dealt round-robin to 1024 partial    */
/*     sums (f32 distance, int count), a halving tree over them, energy = sum / count in f32 (0 when       */
/*     no pixel qualifies); the per-pixel term is the reference body's own arithmetic — the SE3f that     */
/*     optEnergy builds (quaternion normalised in f32 by Sophus's constructor) applied through Eigen's    */
/*     quaternion transform — and is held to that body bit for bit (tests/test_oracle_vs_reference.py);   */
/*   - ties in the simplex ordering: the lower vertex index counts as better.                             */
#define NM_N 7
#define NM_LANES 1024

/* The SE3f optEnergy builds from the optimiser's point (:2481-2484): Eigen::Quaternionf(pose[0..3]) (double -> float),
   Sophus::SE3f(quaternion, translation) — whose constructor NORMALISES the quaternion: coefficients / sqrt(squaredNorm),
   4-term reduction in Eigen's unrolled order over (x, y, z, w) — in f32. q out as (w, x, y, z). */
static void nm_se3f(const double* x, float* q, float* t)
{
  const float w = (float)x[0], a = (float)x[1], b = (float)x[2], c = (float)x[3];
  const float n = sqrtf((a * a + b * b) + (c * c + w * w));
  q[0] = w / n; q[1] = a / n; q[2] = b / n; q[3] = c / n;
  t[0] = (float)x[4]; t[1] = (float)x[5]; t[2] = (float)x[6];
}

/* one pixel of optEnergy's loop (:2496-2519): 1 and the distance when the pixel counts */
static int nm_pixel(const float* lv /* [3] */, const float* pv, const float* q, const float* t, float znear, float zfar, float* dist)
{
  float m[3];
  icp_se3f_apply(q, t, pv[0], pv[1], pv[2], m);                      /* T_co * point, :2505 */
  const float vx = lv[0], vy = lv[1], vz = lv[2];
  if (m[0] == m[0] && m[1] == m[1] && m[2] == m[2] && vz > znear && vz < zfar && m[2] > znear && m[2] < zfar) {   /* :2515 */
    *dist = sqrtf(((m[0] - vx) * (m[0] - vx) + (m[1] - vy) * (m[1] - vy)) + (m[2] - vz) * (m[2] - vz));             /* :2517 */
    return 1;
  }
  return 0;
}

/* optEnergy :2476-2526; the SUM is the canonical parallel one (header above), the reference adds sequentially in f32 */
static float nm_energy(const int* label, const float* live, const float* pred_v, int pc, int W, const int* box, int obj,
                       float znear, float zfar, const double* x, float* part, int* cnt)
{
  float q[4], t[3];
  nm_se3f(x, q, t);
  const int bw = box[1] - box[0] + 1;
  const long nb = (long)bw * (box[3] - box[2] + 1);
  for (int l = 0; l < NM_LANES; l++) {
    float acc = 0.f;
    int c = 0;
    for (long j = l; j < nb; j += NM_LANES) {
      const long p = (long)(box[2] + j / bw) * W + (box[0] + j % bw);
      if (label[p] != obj) continue;
      float d;
      if (nm_pixel(live + 3 * p, pred_v + p * pc, q, t, znear, zfar, &d)) {
        acc = acc + d;
        c++;
      }
    }
    part[l] = acc;
    cnt[l] = c;
  }
  for (int s = NM_LANES / 2; s >= 1; s >>= 1)
    for (int l = 0; l < s; l++) { part[l] = part[l] + part[l + s]; cnt[l] = cnt[l] + cnt[l + s]; }
  return cnt[0] ? part[0] / (float)cnt[0] : 0.f;
}

/* probes for tests/test_oracle_vs_reference.py: the per-pixel term of the energy for EVERY pixel of the frame
   (dist_out [P], valid_out [P]), and the energy itself over the pixels labelled `obj` */
int oracle_icp_energy_terms(const float* live, const float* pred_v, int pc, int H, int W, float znear, float zfar,
                            const double* x, float* dist_out, unsigned char* valid_out)
{
  float q[4], t[3];
  nm_se3f(x, q, t);
  for (long p = 0; p < (long)H * W; p++) {
    float d = 0.f;
    valid_out[p] = (unsigned char)nm_pixel(live + 3 * p, pred_v + p * pc, q, t, znear, zfar, &d);
    dist_out[p] = valid_out[p] ? d : 0.f;
  }
  return 0;
}

double oracle_icp_energy(const int* label, const float* live, const float* pred_v, int pc, int H, int W, int obj, float znear,
                         float zfar, const double* x)
{
  int box[4] = {W, -1, H, -1};
  for (int y = 0; y < H; y++)
    for (int xx = 0; xx < W; xx++)
      if (label[(long)y * W + xx] == obj) {
        if (xx < box[0]) box[0] = xx;
        if (xx > box[1]) box[1] = xx;
        if (y < box[2]) box[2] = y;
        if (y > box[3]) box[3] = y;
      }
  if (box[1] < box[0]) return 0.0;
  float* part = (float*)malloc(sizeof(float) * NM_LANES);
  int* cnt = (int*)malloc(sizeof(int) * NM_LANES);
  if (!part || !cnt) { free(part); free(cnt); return -1.0; }
  const double e = (double)nm_energy(label, live, pred_v, pc, W, box, obj, znear, zfar, x, part, cnt);
  free(part);
  free(cnt);
  return e;
}

/* label int32 [H,W], live f32 [H,W,3], pred_v f32 [H,W,pc] (rendered at the pose the update will multiply).
   x_out f64 [7] = best update found (quaternion wxyz NOT normalised, translation), info f64 [2] = (its energy, evaluations) */
int oracle_icp_polish(const int* label, const float* live, const float* pred_v, int pc, int H, int W, int obj, float znear,
                      float zfar, int maxeval, double* x_out, double* info)
{
  int box[4] = {W, -1, H, -1};
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      if (label[(long)y * W + x] == obj) {
        if (x < box[0]) box[0] = x;
        if (x > box[1]) box[1] = x;
        if (y < box[2]) box[2] = y;
        if (y > box[3]) box[3] = y;
      }
  const double x0[NM_N] = {1, 0, 0, 0, 0, 0, 0};
  const double range[NM_N] = {0.1, 0.1, 0.1, 0.1, 0.01, 0.01, 0.1};      /* :2535-2557 */
  double lb[NM_N], ub[NM_N], P[NM_N + 1][NM_N], f[NM_N + 1];
  for (int i = 0; i < NM_N; i++) { lb[i] = x0[i] - range[i]; ub[i] = x0[i] + range[i]; }
  float* part = (float*)malloc(sizeof(float) * NM_LANES);
  int* cnt = (int*)malloc(sizeof(int) * NM_LANES);
  if (!part || !cnt) { free(part); free(cnt); return -1; }
  int evals = 0;
  if (box[1] < box[0]) {            /* no pixel of the object: nothing to minimise */
    memcpy(x_out, x0, sizeof(x0));
    info[0] = 0.0; info[1] = 0.0;
    free(part); free(cnt);
    return 0;
  }
#define NM_EVAL(X) ((double)nm_energy(label, live, pred_v, pc, W, box, obj, znear, zfar, (X), part, cnt))
  /* initial simplex: x0, x0 + step_i e_i, step = (ub - lb) / 4 (nlopt's default initial step for a centred start) */
  for (int k = 0; k <= NM_N; k++) {
    for (int i = 0; i < NM_N; i++) P[k][i] = x0[i];
    if (k > 0) P[k][k - 1] = x0[k - 1] + (ub[k - 1] - lb[k - 1]) * 0.25;
    f[k] = NM_EVAL(P[k]);
    evals++;
  }
  while (evals < maxeval) {
    /* best (lowest f, then lowest index), worst (highest f, then highest index), second worst */
    int lo = 0, hi = 0, nh = -1;
    for (int k = 1; k <= NM_N; k++) {
      if (f[k] < f[lo]) lo = k;
      if (f[k] >= f[hi]) hi = k;
    }
    for (int k = 0; k <= NM_N; k++)
      if (k != hi && (nh < 0 || f[k] >= f[nh])) nh = k;
    double c[NM_N], xr[NM_N], xe[NM_N];
    for (int i = 0; i < NM_N; i++) {
      double sacc = 0.0;
      for (int k = 0; k <= NM_N; k++)
        if (k != hi) sacc = sacc + P[k][i];
      c[i] = sacc / (double)NM_N;
    }
#define NM_POINT(DST, COEF)                                          \
    for (int i = 0; i < NM_N; i++) {                                 \
      double v_ = c[i] + (COEF) * (c[i] - P[hi][i]);                 \
      if (v_ < lb[i]) v_ = lb[i];                                    \
      if (v_ > ub[i]) v_ = ub[i];                                    \
      (DST)[i] = v_;                                                 \
    }
    NM_POINT(xr, 1.0);
    const double fr = NM_EVAL(xr);
    evals++;
    if (fr < f[lo]) {                                   /* new best: try to expand */
      if (evals < maxeval) {
        NM_POINT(xe, 2.0);
        const double fe = NM_EVAL(xe);
        evals++;
        if (fe < fr) { memcpy(P[hi], xe, sizeof(xe)); f[hi] = fe; }
        else { memcpy(P[hi], xr, sizeof(xr)); f[hi] = fr; }
      } else { memcpy(P[hi], xr, sizeof(xr)); f[hi] = fr; }
    } else if (fr < f[nh]) {                            /* better than the second worst: accept */
      memcpy(P[hi], xr, sizeof(xr)); f[hi] = fr;
    } else {                                            /* contract: outside if the reflection beat the worst, else inside */
      if (evals >= maxeval) { if (fr < f[hi]) { memcpy(P[hi], xr, sizeof(xr)); f[hi] = fr; } break; }
      const double coef = fr < f[hi] ? 0.5 : -0.5;
      NM_POINT(xe, coef);
      const double fc = NM_EVAL(xe);
      evals++;
      const double fref = fr < f[hi] ? fr : f[hi];
      if (fc < fref) { memcpy(P[hi], xe, sizeof(xe)); f[hi] = fc; }
      else {                                            /* shrink towards the best vertex */
        for (int k = 0; k <= NM_N && evals < maxeval; k++) {
          if (k == lo) continue;
          for (int i = 0; i < NM_N; i++) P[k][i] = P[lo][i] + 0.5 * (P[k][i] - P[lo][i]);
          f[k] = NM_EVAL(P[k]);
          evals++;
        }
      }
    }
  }
  int lo = 0;
  for (int k = 1; k <= NM_N; k++)
    if (f[k] < f[lo]) lo = k;
  memcpy(x_out, P[lo], sizeof(double) * NM_N);
  info[0] = f[lo];
  info[1] = (double)evals;
  free(part); free(cnt);
  return 0;
#undef NM_EVAL
#undef NM_POINT
}
