/* TEST INFRASTRUCTURE: the CPU checker (pcnn_oracle.c: pose refinement, then the five custom layers and the head kernels) compiled into one translation unit with
 * AddressSanitizer + UndefinedBehaviorSanitizer and driven over small, odd-sized inputs (objects cut by the image border,
 * empty masks, zero faces, budgets that stop the simplex in every branch). `make -C oracle asan` builds
 * oracle/_asan/asan_driver; tests/test_icp_render.py runs it. Exit code 0 and "asan_driver ok" = no report. */
#include "pcnn_oracle.c"

#include <stdio.h>

static float frand(unsigned* s)
{
  *s = *s * 1664525u + 1013904223u;
  return (float)((*s >> 8) & 0xffff) / 65536.f;
}

int main(void)
{
  const int H = 37, W = 53;
  const long P = (long)H * W;
  const float fx = 60.f, fy = 61.f, px = 25.3f, py = 18.1f;
  unsigned seed = 7;
  /* an octahedron around the origin, scaled; one degenerate and one repeated-index face */
  const float vtx[6 * 3] = {0.1f, 0, 0, -0.1f, 0, 0, 0, 0.08f, 0, 0, -0.08f, 0, 0, 0, 0.12f, 0, 0, -0.12f};
  float nrm[6 * 3];
  for (int i = 0; i < 18; i++) nrm[i] = vtx[i];
  const int faces[10 * 3] = {0, 2, 4, 2, 1, 4, 1, 3, 4, 3, 0, 4, 2, 0, 5, 1, 2, 5, 3, 1, 5, 0, 3, 5, 0, 0, 1, 2, 2, 2};
  /* poses: centred, cut by the right border, straddling z_near, beyond z_far */
  const float poses[4 * 12] = {1, 0, 0, 0.0f, 0, 1, 0, 0.0f, 0, 0, 1, 0.5f,
                               0.8f, 0, 0.6f, 0.2f, 0, 1, 0, 0.05f, -0.6f, 0, 0.8f, 0.5f,
                               1, 0, 0, 0.0f, 0, 1, 0, 0.0f, 0, 0, 1, 0.3f,
                               1, 0, 0, 0.0f, 0, 1, 0, 0.0f, 0, 0, 1, 7.0f};
  float* ov = (float*)malloc(sizeof(float) * 4 * P * 4);
  float* on = (float*)malloc(sizeof(float) * 4 * P * 4);
  float* oc = (float*)malloc(sizeof(float) * 4 * P * 3);
  if (oracle_render_mesh(vtx, nrm, faces, 6, 10, poses, 4, H, W, fx, fy, px, py, 0.25f, 6.f, 3.f, ov, on, oc)) return 2;
  if (oracle_render_mesh(vtx, NULL, faces, 6, 10, poses, 1, H, W, fx, fy, px, py, 0.25f, 6.f, 0.f, ov + 0, NULL, NULL)) return 3;
  if (oracle_render_mesh(vtx, nrm, faces, 6, 0, poses, 1, H, W, fx, fy, px, py, 0.25f, 6.f, 0.f, NULL, on, NULL)) return 4;   /* no faces */
  if (oracle_render_mesh(vtx, nrm, faces, 6, 10, poses, 4, H, W, fx, fy, px, py, 0.25f, 6.f, 3.f, ov, on, oc)) return 5;
  long hit = 0;
  for (long i = 0; i < P; i++) hit += ov[4 * i + 2] == ov[4 * i + 2];
  if (hit < 100) { printf("render covered %ld pixels\n", hit); return 6; }

  /* depth + label from the first render, with holes and a foreign label */
  uint16_t* depth = (uint16_t*)malloc(sizeof(uint16_t) * P);
  int* label = (int*)malloc(sizeof(int) * P);
  for (long i = 0; i < P; i++) {
    const float z = ov[4 * i + 2];
    const int on_obj = z == z;
    depth[i] = (uint16_t)(on_obj && (i % 9) ? z * 10000.f + 30.f * (frand(&seed) - 0.5f) : 0);
    label[i] = on_obj ? ((i % 31) ? 4 : 9) : 0;
  }
  float* live = (float*)malloc(sizeof(float) * 3 * P * 2);
  oracle_icp_backproject(depth, label, H, W, 4, 10000.f, fx, fy, px, py, live);
  oracle_icp_backproject(depth, NULL, H, W, 0, 10000.f, fx, fy, px, py, live + 3 * P);
  double sums[5];
  uint8_t* mask = (uint8_t*)malloc((size_t)P);
  if (oracle_icp_center(label, live, oc, ov, on, 4, H, W, 4, 0.01f, sums, mask)) return 7;
  if (!(sums[4] > 50)) { printf("center found %g pairs\n", sums[4]); return 8;
  }
  /* ICP: two problems (the second has nothing to align), 3 iterations, statistics on */
  double upd[2 * 12];
  float stats[2 * 3 * 2];
  memcpy(live + 3 * P, live, sizeof(float) * 3 * P);
  if (oracle_icp_refine(live, ov, on, 2, H, W, 4, fx, fy, px, py, 0.25f, 6.f, 0.01f, 3, upd, stats)) return 9;
  /* score: the rendered pose, one shifted by 3 mm, one far away, one at the camera (full-scan branch) */
  float hyps[4 * 12];
  for (int m = 0; m < 4; m++) memcpy(hyps + 12 * m, poses, sizeof(float) * 12);
  hyps[12 + 3] += 0.003f;
  hyps[24 + 11] += 0.5f;
  hyps[36 + 11] = 0.001f;
  int hits[4];
  if (oracle_icp_score(live, oc, mask, H, W, hyps, 4, 0.01f, hits)) return 10;
  if (!(hits[0] > 0 && hits[2] == 0)) { printf("hits %d %d %d %d\n", hits[0], hits[1], hits[2], hits[3]); return 11; }
  const int h0 = hits[0], h1 = hits[1], h2 = hits[2], h3 = hits[3];
  memset(mask, 0, (size_t)P);
  if (oracle_icp_score(live, oc, mask, H, W, hyps, 4, 0.01f, hits) || hits[0]) return 12;
  /* polish: every budget from the initial simplex up, an absent object */
  double x[7], info[2];
  for (int budget = 8; budget <= 40; budget++)
    if (oracle_icp_polish(label, live, ov, 4, H, W, 4, 0.25f, 6.f, budget, x, info) || (int)info[1] != budget) { printf("polish budget %d -> %g\n", budget, info[1]); return 13; }
  if (oracle_icp_polish(label, live, ov, 4, H, W, 17, 0.25f, 6.f, 50, x, info) || info[1] != 0.0) return 14;
  /* ---- the five custom layers + head kernels on small random inputs (odd sizes, malformed ROIs, ignore labels) ---- */
  {
    const int B = 2, h = 9, w = 11, C = 5, R = 6;
    float* data = (float*)malloc(sizeof(float) * B * h * w * C);
    for (int i = 0; i < B * h * w * C; i++) data[i] = frand(&seed) - 0.5f;
    float rois[6 * 7] = {0, 1, 10, 5, 100, 60, 1,   1, 2, -40, -30, 20, 15, 1,   0, 3, 150, 120, 400, 300, 1,
                         1, 4, 60, 50, 20, 10, 1,   5, 1, 0, 0, 10, 10, 1,     -1, 2, 0, 0, 10, 10, 1};   /* off-image, inverted, bad batch */
    float* top = (float*)malloc(sizeof(float) * R * 7 * 7 * C);
    int* arg = (int*)malloc(sizeof(int) * R * 7 * 7 * C);
    if (oracle_roi_pool(data, rois, B, h, w, C, 4, 7, 7, 7, 1.f / 16, 0, top, arg)) return 20;
    if (oracle_roi_pool(data, rois, B, h, w, C, 4, 7, 3, 2, 1.f / 8, 1, top, arg)) return 21;
    float* bd = (float*)calloc((size_t)B * h * w * C, sizeof(float));
    if (oracle_roi_pool(data, rois, B, h, w, C, 4, 7, 7, 7, 1.f / 16, 0, top, arg) || oracle_roi_pool_bwd(top, rois, arg, B, h, w, C, 4, 7, 7, 7, 1.f / 16, 0, bd)) return 22;
    const long N = (long)B * h * w;
    int* gt = (int*)malloc(sizeof(int) * N);
    for (long i = 0; i < N; i++) gt[i] = (int)(frand(&seed) * 8) - 2;     /* -2 .. 5: includes -1 (ignore) and out-of-range labels */
    float* hl = (float*)malloc(sizeof(float) * N * C);
    float* prob = (float*)malloc(sizeof(float) * N * C);
    int* lab2 = (int*)malloc(sizeof(int) * N);
    if (oracle_softmax_argmax(data, N, C, prob, lab2) || oracle_hard_label(prob, gt, N, C, 0.4f, hl)) return 23;
    float* up = (float*)malloc(sizeof(float) * B * h * 8 * w * 8 * C);
    float bias[5] = {0.1f, -0.2f, 0.3f, 0, 0.5f};
    if (oracle_deconv_bilinear(data, B, h, w, C, 4, 2, NULL, NULL, NULL, 0, up)) return 24;
    if (oracle_deconv_bilinear(data, B, h, w, C, 16, 8, NULL, NULL, bias, 1, up)) return 25;
    /* average distance: 4 rows (one without a class), 3 classes, 37 points, class 2 symmetric */
    const int AR = 4, AC = 3, AP = 37;
    float pred[4 * 12], targ[4 * 12], wgt[4 * 12], pts[3 * 37 * 3], sym[3] = {0, 0, 1}, loss, diff[4 * 12];
    for (int i = 0; i < AR * 4 * AC; i++) { pred[i] = frand(&seed) - 0.5f; targ[i] = frand(&seed) - 0.5f; wgt[i] = 0.f; }
    for (int i = 0; i < AC * AP * 3; i++) pts[i] = 0.1f * (frand(&seed) - 0.5f);
    for (int q = 0; q < 4; q++) { wgt[0 * 12 + 4 * 1 + q] = 1; wgt[1 * 12 + 4 * 2 + q] = 1; wgt[3 * 12 + 4 * 2 + q] = 1; }
    if (oracle_average_distance(pred, targ, wgt, pts, sym, AR, AC, AP, 0.01f, &loss, diff) || !(loss == loss)) return 26;
    /* smooth L1 over a length that is not a multiple of anything */
    const long n1 = 10007;
    float* a = (float*)malloc(sizeof(float) * n1 * 3);
    for (long i = 0; i < 3 * n1; i++) a[i] = frand(&seed) - 0.5f;
    float l1[3], *g1 = (float*)malloc(sizeof(float) * n1);     /* (loss, sum of terms, sum of weights) */
    if (oracle_smooth_l1_vertex(a, a + n1, a + 2 * n1, n1, 1.f, l1, g1)) return 27;
    /* Hough voting: 2 images of 48 x 64, 4 classes, a blob per class, both threshold branches, train mode */
    const int HB = 2, HH = 48, HW = 64, HC = 4;
    int* hlab = (int*)calloc((size_t)HB * HH * HW, sizeof(int));
    float* hver = (float*)malloc(sizeof(float) * HB * HH * HW * 3 * HC);
    for (long i = 0; i < (long)HB * HH * HW * 3 * HC; i++) hver[i] = 0.2f * (frand(&seed) - 0.5f);
    for (int b = 0; b < HB; b++)
      for (int c = 1; c < HC; c++) {
        const int cx = 12 + 16 * c, cy = 14 + 6 * c;
        for (int y = cy - 7; y <= cy + 7; y++)
          for (int x = cx - 7; x <= cx + 7; x++) {
            const long i = ((long)b * HH + y) * HW + x;
            hlab[i] = c;
            const float dx = (float)(cx - x), dy = (float)(cy - y), nn = sqrtf(dx * dx + dy * dy) + 1e-6f;
            hver[i * 3 * HC + 3 * c] = dx / nn; hver[i * 3 * HC + 3 * c + 1] = dy / nn; hver[i * 3 * HC + 3 * c + 2] = logf(0.8f);
          }
      }
    float ext[4 * 3] = {0, 0, 0, 0.1f, 0.1f, 0.1f, 0.12f, 0.08f, 0.1f, 0.1f, 0.09f, 0.11f};
    float meta[2 * 48];
    memset(meta, 0, sizeof(meta));
    for (int b = 0; b < HB; b++) { meta[48 * b] = 106.f; meta[48 * b + 2] = 31.f; meta[48 * b + 4] = 106.f; meta[48 * b + 5] = 24.f; meta[48 * b + 8] = 1.f; }
    float hgt[2 * 13] = {0, 1, 20, 10, 40, 30, 1, 0, 0, 0, 0, 0, 0.8f,   1, 2, 30, 20, 60, 40, 1, 0, 0, 0, 0, 0, 0.8f};
    const int cap = 128 * 9;
    float* tb = (float*)malloc(sizeof(float) * cap * 7);
    float* tp = (float*)malloc(sizeof(float) * cap * 7);
    float* tt = (float*)malloc(sizeof(float) * cap * 4 * HC);
    float* tw = (float*)malloc(sizeof(float) * cap * 4 * HC);
    int* td = (int*)malloc(sizeof(int) * cap);
    int nr[2];
    for (int train = 0; train <= 1; train++)
      for (int thr = 0; thr <= 1; thr++)
        if (oracle_hough_voting(hlab, hver, ext, meta, hgt, HB, HH, HW, HC, 48, 2, train, thr ? 5.f : -1.f, 0.02f, 3, 0.9f, 100, tb, tp, tt, tw, td, nr, NULL)) return 28;
    if (nr[0] < 1) return 29;
    free(data); free(top); free(arg); free(bd); free(gt); free(hl); free(prob); free(lab2); free(up); free(a); free(g1);
    free(hlab); free(hver); free(tb); free(tp); free(tt); free(tw); free(td);
  }
  free(ov); free(on); free(oc); free(depth); free(label); free(live); free(mask);
  printf("asan_driver ok: %ld rendered pixels, %g pairs, hits %d %d %d %d\n", hit, sums[4], h0, h1, h2, h3);
  return 0;
}
