/* A plain-C consumer of include/posecnn_hip.h: compiled by gcc against the
 * header and linked against libposecnn_hip.so, so a drift between the header's prototypes and the
 * library's definitions (argument count / order / type) breaks the BUILD or this program's checks,
 * not just a ctypes table. Host-side calls only: runs on a box without a GPU.
 * Built by __graft_entry__.build(); run by tests/test_capi_load.py. */
#include <stdio.h>
#include <string.h>

#include "posecnn_hip.h"

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "capi_consumer: %s failed (line %d): %s\n", #cond, __LINE__, pcnn_last_error_string()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(void)
{
  size_t small = 0, big = 0, wide = 0, adl = 0, offs[8];
  CHECK(pcnn_abi_version() == PCNN_ABI_VERSION);
  CHECK(strcmp(pcnn_status_string(PCNN_OK), "ok") == 0);
  /* Houghvotinggpu: workspace of the demo configuration, with and without the Hough space */
  CHECK(pcnn_hough_voting_workspace_bytes(16, 480, 640, 22, -1.0f, 10, 0, &small) == PCNN_OK);
  CHECK(pcnn_hough_voting_workspace_bytes(16, 480, 640, 22, 50.0f, 10, 0, &big) == PCNN_OK);
  CHECK(pcnn_hough_voting_workspace_bytes(16, 480, 640, 22, -1.0f, 10, 21, &wide) == PCNN_OK);
  CHECK(small > 0 && big > small && wide >= small);
  CHECK(pcnn_hough_voting_debug_layout(16, 480, 640, 22, 50.0f, 10, 0, offs) == PCNN_OK);
  CHECK(offs[0] < big && offs[1] < big);
  /* attribute checks of the reference (OP_REQUIRES) come back as PCNN_EINVAL, nothing is launched */
  CHECK(pcnn_hough_voting_workspace_bytes(1, 480, 640, 22, -1.0f, 0, 0, &small) == PCNN_EINVAL);
  CHECK(strstr(pcnn_last_error_string(), "skip_pixels") != NULL);
  CHECK(pcnn_hough_voting_fwd(NULL, NULL, NULL, NULL, NULL, 1, 480, 640, 22, 48, 0, 0, -1.0f, 0.02f, 10, 0.9f, 500,
                              0, PCNN_HOUGH_ROWS_CAPACITY, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) == PCNN_ENULL);
  CHECK(pcnn_hard_label_fwd(NULL, NULL, 10, 22, 0.0f, NULL, NULL) == PCNN_EINVAL);
  CHECK(pcnn_average_distance_workspace_bytes(128, 22, 2620, &adl) == PCNN_OK && adl >= 128u * 5u * 2620u * 4u);
  CHECK(pcnn_average_distance_fwd(NULL, NULL, NULL, NULL, NULL, 1, 22, 10, -1.0f, NULL, NULL, NULL, NULL, 0, NULL) == PCNN_EINVAL);
  CHECK(pcnn_roi_pool_fwd(NULL, NULL, 1, 30, 40, 512, 3, 5, 7, 7, 0.0625f, 0, NULL, NULL, NULL) == PCNN_EINVAL);
  CHECK(pcnn_backproject_fwd(NULL, NULL, NULL, NULL, NULL, 1, 8, 8, 4, 3, 48, 4, -1, 0.1f, NULL, NULL, NULL, NULL) == PCNN_EINVAL);
  /* Network.fc over a row buffer: split-K workspace only for the wide shape (fc6), none for the tall 1x1-conv shape */
  {
    size_t fc6 = 0, head = 1;
    CHECK(pcnn_fc_rows_workspace_bytes(336, 25088, 4096, &fc6) == PCNN_OK && fc6 >= 8u * 336u * 4096u * 4u);
    CHECK(pcnn_fc_rows_workspace_bytes(76800, 512, 64, &head) == PCNN_OK && head == 0);
    CHECK(pcnn_fc_rows_fwd(NULL, NULL, NULL, 16, 100, 64, 1, NULL, NULL, NULL, NULL, 0, NULL) == PCNN_EINVAL);
  }
  /* pose refinement (Synthesizer::solveICP): workspace rules and host-side argument checks of the render / centre / score entries */
  {
    size_t zb = 0, cw = 0, sw = 0;
    CHECK(pcnn_render_mesh_workspace_bytes(8, 480, 640, &zb) == PCNN_OK && zb == 8u * 480u * 640u * 8u);
    CHECK(pcnn_icp_center_workspace_bytes(480, 640, &cw) == PCNN_OK && cw >= 1200u * 5u * 4u);
    CHECK(pcnn_icp_score_workspace_bytes(8, 480, 640, &sw) == PCNN_OK && sw == 8u * 9600u * 4u);
    CHECK(pcnn_render_mesh_fwd(NULL, NULL, NULL, 3, 1, NULL, 1, 480, 640, 1066.f, 1067.f, 313.f, 241.f, 0.f, 6.f, 0.f, NULL, NULL, NULL, NULL, 0, NULL) == PCNN_EINVAL);
    CHECK(pcnn_render_mesh_fwd(NULL, NULL, NULL, 3, 1, NULL, 1, 480, 640, 1066.f, 1067.f, 313.f, 241.f, 0.25f, 6.f, 0.f, NULL, NULL, NULL, NULL, 0, NULL) == PCNN_ENULL);
    CHECK(pcnn_icp_center_fwd(NULL, NULL, NULL, NULL, NULL, 5, 480, 640, 1, 0.01f, NULL, NULL, NULL, 0, NULL) == PCNN_EINVAL);
    CHECK(pcnn_icp_score_fwd(NULL, NULL, NULL, 480, 640, NULL, 8, 1066.f, 1067.f, 313.f, 241.f, -1.f, NULL, NULL, 0, NULL) == PCNN_EINVAL);
    CHECK(pcnn_icp_polish_fwd(NULL, NULL, NULL, 4, 480, 640, 1, 0.25f, 6.f, 7, NULL, NULL, NULL) == PCNN_EINVAL);
  }
  printf("capi_consumer ok: abi %d, hough workspace %zu / %zu / %zu bytes\n", pcnn_abi_version(), small, big, wide);
  return 0;
}
