/* TEST INFRASTRUCTURE: the pose-refinement part of the CPU checker (pcnn_oracle.c) compiled into one translation unit with
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
  free(ov); free(on); free(oc); free(depth); free(label); free(live); free(mask);
  printf("asan_driver ok: %ld rendered pixels, %g pairs, hits %d %d %d %d\n", hit, sums[4], h0, h1, h2, h3);
  return 0;
}
