// render.hip — the predicted maps of the pose-refinement step: what Synthesizer::refinePose / solveICP obtain from
// OpenGL before every ICP call (lib/synthesize/synthesize.cpp:1972-1991, :2104-2136):
//   * renderer_vn_ (df::GLRenderer<VertAndNormalRenderType>, shaders lib/kinect_fusion/shaders/vertsAndNorms.{vert,frag}):
//       texture 0 = camera-frame position of the surface point behind each pixel, texture 1 = the per-vertex normal
//       (rotated into the camera frame and normalised PER VERTEX, then interpolated — not re-normalised per pixel);
//   * renderer_ (CanonicalVertRenderType, shaders canonicalVerts.{vert,frag}): the object-frame ("canonical") vertex,
//       x shifted by the model index (synthesize.cpp:266-271), background = the NaN clear colour (:2113).
// Camera convention: pangolin::ProjectionMatrixRDF_TopLeft(w, h, fx, -fy, px + 0.5, h - (py + 0.5), ...) (:2088) puts
// pixel CENTRES at integer (u, v) of u = fx X / Z + px — the convention of df's Poly3 camera model that the ICP kernel
// projects with — so the sample point of pixel (x, y) is (x, y) itself.
//
// There is no GL on this path. A frame's refinement renders one mesh at 8 hypothesis poses, a few objects per frame
// (synthesize.cpp:2272-2300): small triangles (YCB meshes: 10^4-10^5 faces over ~10^4 pixels), many poses. So:
//   render_raster_kernel   grid (faces / 256, poses): one thread per triangle sets it up (3 vertex transforms,
//                          projection, canonical edge functions) and walks its bounding box when that is small
//                          (<= 64 pixels, the common case); triangles with a larger box are queued in LDS and
//                          walked by the whole workgroup afterwards. Visibility = atomicMin on a 64-bit key
//                          (depth bits << 32 | face index): order-independent, ties broken by the face index,
//                          hence deterministic.
//   render_resolve_kernel  one thread per pixel and pose: re-derives the winning triangle's weights and writes the
//                          perspective-correct attributes (NaN where nothing was hit).
// Watertightness: the edge function of an edge is always evaluated from its lower-numbered vertex to the higher one
// (and negated for the triangle that runs it the other way), so two triangles sharing an edge see the SAME float on
// it; coverage is inclusive on both sides — a pixel exactly on a shared edge is claimed twice and the key decides.
// Triangles with a vertex in front of z_near are dropped, not clipped (objects under refinement lie wholly inside the
// depth range, synthesize.cpp passes z_near = 0.25 m).
//
// GL's own rasterisation (fixed-point snapping, top-left rule, vendor interpolation) is not reproducible bit for bit;
// the CPU checker restates THIS file's arithmetic (same expression trees, all IEEE f32, no contraction) and the tests
// pin both to an analytic ray-caster.
#include <algorithm>

#include "pcnn_device.h"

namespace {

using namespace pcnn;

constexpr unsigned long long RD_EMPTY = ~0ull;
constexpr int RD_SMALL = 64;     // bounding boxes up to this many pixels are walked by the triangle's own thread

struct RdTri {
  float u[3], v[3], z[3];   // projected vertices and their camera depths
  int flip[3];              // edge i (opposite vertex i) runs from the higher-numbered vertex to the lower one
  int x0, x1, y0, y1;       // clipped bounding box (inclusive); empty when x0 > x1
};

__device__ __forceinline__ void rd_transform(const float* __restrict__ T, const float* __restrict__ p, float* c)
{
  const float x = p[0], y = p[1], z = p[2];
  c[0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  c[1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  c[2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

// edge function from a (lower vertex number) to b at (x, y)
__device__ __forceinline__ float rd_edge(float au, float av, float bu, float bv, float x, float y)
{
  return (bu - au) * (y - av) - (bv - av) * (x - au);
}

__device__ __forceinline__ bool rd_setup(const float* __restrict__ T, const float* __restrict__ vtx, const int* __restrict__ face,
                                         int W, int H, float fx, float fy, float px, float py, float znear, RdTri& t, float cam[3][3])
{
  const int i0 = face[0], i1 = face[1], i2 = face[2];
  const int idx[3] = {i0, i1, i2};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    rd_transform(T, vtx + 3 * (size_t)idx[k], cam[k]);
    t.z[k] = cam[k][2];
    t.u[k] = cam[k][0] / cam[k][2] * fx + px;
    t.v[k] = cam[k][1] / cam[k][2] * fy + py;
  }
  t.x0 = 1; t.x1 = 0; t.y0 = 1; t.y1 = 0;
  if (i0 == i1 || i1 == i2 || i0 == i2) return false;
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (!(t.z[k] >= znear) || !(fabsf(t.u[k]) < 1e7f) || !(fabsf(t.v[k]) < 1e7f)) return false;
  t.flip[0] = i1 > i2;
  t.flip[1] = i2 > i0;
  t.flip[2] = i0 > i1;
  const float umin = fminf(fminf(t.u[0], t.u[1]), t.u[2]), umax = fmaxf(fmaxf(t.u[0], t.u[1]), t.u[2]);
  const float vmin = fminf(fminf(t.v[0], t.v[1]), t.v[2]), vmax = fmaxf(fmaxf(t.v[0], t.v[1]), t.v[2]);
  t.x0 = max(0, (int)ceilf(umin));
  t.x1 = min(W - 1, (int)floorf(umax));
  t.y0 = max(0, (int)ceilf(vmin));
  t.y1 = min(H - 1, (int)floorf(vmax));
  return t.x0 <= t.x1 && t.y0 <= t.y1;
}

// screen-space weights of pixel (x, y): e[i] = edge function opposite vertex i; false when the pixel is outside
__device__ __forceinline__ bool rd_weights(const RdTri& t, float x, float y, float* w, float& s)
{
  float e[3];
  // edge 0: vertices 1 -> 2, edge 1: 2 -> 0, edge 2: 0 -> 1
  e[0] = t.flip[0] ? -rd_edge(t.u[2], t.v[2], t.u[1], t.v[1], x, y) : rd_edge(t.u[1], t.v[1], t.u[2], t.v[2], x, y);
  e[1] = t.flip[1] ? -rd_edge(t.u[0], t.v[0], t.u[2], t.v[2], x, y) : rd_edge(t.u[2], t.v[2], t.u[0], t.v[0], x, y);
  e[2] = t.flip[2] ? -rd_edge(t.u[1], t.v[1], t.u[0], t.v[0], x, y) : rd_edge(t.u[0], t.v[0], t.u[1], t.v[1], x, y);
  const bool pos = e[0] >= 0.f && e[1] >= 0.f && e[2] >= 0.f;
  const bool neg = e[0] <= 0.f && e[1] <= 0.f && e[2] <= 0.f;
  if (!(pos || neg)) return false;
  const float area = (e[0] + e[1]) + e[2];
  if (area == 0.f) return false;
  // perspective-correct: weight_i = (e_i / area) / z_i, normalised by their sum
  w[0] = e[0] / area / t.z[0];
  w[1] = e[1] / area / t.z[1];
  w[2] = e[2] / area / t.z[2];
  s = (w[0] + w[1]) + w[2];
  return s > 0.f;
}

__device__ __forceinline__ void rd_pixel(const RdTri& t, int x, int y, int W, float znear, float zfar, unsigned tri,
                                         unsigned long long* __restrict__ zbuf)
{
  float w[3], s;
  if (!rd_weights(t, (float)x, (float)y, w, s)) return;
  const float z = ((w[0] * t.z[0] + w[1] * t.z[1]) + w[2] * t.z[2]) / s;
  if (!(z >= znear) || !(z <= zfar)) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | tri;
  atomicMin(&zbuf[(size_t)y * W + x], key);
}

__global__ __launch_bounds__(256) void render_clear_kernel(unsigned long long* __restrict__ zbuf, long long n)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) zbuf[i] = RD_EMPTY;
}

__global__ __launch_bounds__(256) void render_raster_kernel(
    const float* __restrict__ vtx, const int* __restrict__ faces, int nfaces, const float* __restrict__ poses, int H, int W,
    float fx, float fy, float px, float py, float znear, float zfar, unsigned long long* __restrict__ zbuf)
{
  __shared__ RdTri s_big[256];
  __shared__ unsigned s_bigid[256];
  __shared__ int s_nbig;
  const int tid = threadIdx.x, n = blockIdx.y;
  const int f = blockIdx.x * 256 + tid;
  if (tid == 0) s_nbig = 0;
  __syncthreads();
  const float* T = poses + 12 * (size_t)n;
  unsigned long long* zb = zbuf + (size_t)n * H * W;
  if (f < nfaces) {
    RdTri t;
    float cam[3][3];
    if (rd_setup(T, vtx, faces + 3 * (size_t)f, W, H, fx, fy, px, py, znear, t, cam)) {
      const long long box = (long long)(t.x1 - t.x0 + 1) * (t.y1 - t.y0 + 1);
      if (box <= RD_SMALL) {
        for (int y = t.y0; y <= t.y1; y++)
          for (int x = t.x0; x <= t.x1; x++) rd_pixel(t, x, y, W, znear, zfar, (unsigned)f, zb);
      } else {
        const int slot = atomicAdd(&s_nbig, 1);
        s_big[slot] = t;
        s_bigid[slot] = (unsigned)f;
      }
    }
  }
  __syncthreads();
  const int nbig = s_nbig;
  for (int q = 0; q < nbig; q++) {
    const RdTri& t = s_big[q];
    const int bw = t.x1 - t.x0 + 1;
    const long long box = (long long)bw * (t.y1 - t.y0 + 1);
    for (long long i = tid; i < box; i += 256) {
      const int y = t.y0 + (int)(i / bw), x = t.x0 + (int)(i % bw);
      rd_pixel(t, x, y, W, znear, zfar, s_bigid[q], zb);
    }
  }
}

__global__ __launch_bounds__(256) void render_resolve_kernel(
    const float* __restrict__ vtx, const float* __restrict__ nrm, const int* __restrict__ faces, const float* __restrict__ poses,
    int H, int W, float fx, float fy, float px, float py, float znear, float canon_x_offset,
    const unsigned long long* __restrict__ zbuf, float* __restrict__ out_v, float* __restrict__ out_n, float* __restrict__ out_c)
{
  const long long P = (long long)H * W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (i >= P) return;
  const unsigned long long key = zbuf[(size_t)n * P + i];
  const float qnan = __uint_as_float(0x7fc00000u);
  float ov[3] = {qnan, qnan, qnan}, on[3] = {qnan, qnan, qnan}, oc[3] = {qnan, qnan, qnan};
  if (key != RD_EMPTY) {
    const unsigned f = (unsigned)(key & 0xffffffffu);
    const int* face = faces + 3 * (size_t)f;
    const float* T = poses + 12 * (size_t)n;
    RdTri t;
    float cam[3][3], w[3], s;
    rd_setup(T, vtx, face, W, H, fx, fy, px, py, znear, t, cam);
    const int x = (int)(i % W), y = (int)(i / W);
    if (rd_weights(t, (float)x, (float)y, w, s)) {
#pragma unroll
      for (int k = 0; k < 3; k++) ov[k] = ((w[0] * cam[0][k] + w[1] * cam[1][k]) + w[2] * cam[2][k]) / s;
      if (out_n) {
        float vn[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const float* p = nrm + 3 * (size_t)face[j];
          const float a = p[0], b = p[1], c = p[2];
          const float rx = (T[0] * a + T[1] * b) + T[2] * c;
          const float ry = (T[4] * a + T[5] * b) + T[6] * c;
          const float rz = (T[8] * a + T[9] * b) + T[10] * c;
          const float len = sqrt_rn((rx * rx + ry * ry) + rz * rz);
          const bool ok = len > 0.f;      // vertsAndNorms.vert: normalised only when the length is positive
          vn[j][0] = ok ? rx / len : rx;
          vn[j][1] = ok ? ry / len : ry;
          vn[j][2] = ok ? rz / len : rz;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) on[k] = ((w[0] * vn[0][k] + w[1] * vn[1][k]) + w[2] * vn[2][k]) / s;
      }
      if (out_c) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float a0 = vtx[3 * (size_t)face[0] + k] + (k == 0 ? canon_x_offset : 0.f);
          const float a1 = vtx[3 * (size_t)face[1] + k] + (k == 0 ? canon_x_offset : 0.f);
          const float a2 = vtx[3 * (size_t)face[2] + k] + (k == 0 ? canon_x_offset : 0.f);
          oc[k] = ((w[0] * a0 + w[1] * a1) + w[2] * a2) / s;
        }
      }
    }
  }
  const size_t o = (size_t)n * P + i;
  if (out_v) { out_v[4 * o] = ov[0]; out_v[4 * o + 1] = ov[1]; out_v[4 * o + 2] = ov[2]; out_v[4 * o + 3] = key != RD_EMPTY ? 1.f : qnan; }
  if (out_n) { out_n[4 * o] = on[0]; out_n[4 * o + 1] = on[1]; out_n[4 * o + 2] = on[2]; out_n[4 * o + 3] = key != RD_EMPTY ? 0.f : qnan; }
  if (out_c) { out_c[3 * o] = oc[0]; out_c[3 * o + 1] = oc[1]; out_c[3 * o + 2] = oc[2]; }
}

}  // namespace

extern "C" int pcnn_render_mesh_workspace_bytes(int num_poses, int height, int width, size_t* bytes)
{
  PCNN_REQUIRE(bytes, PCNN_ENULL, "render_mesh_workspace_bytes: NULL output");
  PCNN_REQUIRE(num_poses >= 0 && height >= 1 && width >= 1, PCNN_EINVAL, "render_mesh_workspace_bytes: bad shape");
  *bytes = sizeof(unsigned long long) * (size_t)num_poses * height * width;
  return PCNN_OK;
}

extern "C" int pcnn_render_mesh_fwd(const float* vertices, const float* normals, const int32_t* faces, int num_vertices,
                                    int num_faces, const float* poses, int num_poses, int height, int width, float fx,
                                    float fy, float px, float py, float z_near, float z_far, float canon_x_offset,
                                    float* out_vertices, float* out_normals, float* out_canonical, void* workspace,
                                    size_t workspace_bytes, void* stream_)
{
  PCNN_REQUIRE(num_poses >= 0 && height >= 1 && width >= 1 && num_vertices >= 0 && num_faces >= 0, PCNN_EINVAL,
               "render_mesh: bad shape (%d poses, %dx%d, %d vertices, %d faces)", num_poses, height, width, num_vertices, num_faces);
  PCNN_REQUIRE(num_poses <= 65535, PCNN_EINVAL, "render_mesh: at most 65535 poses per call");
  PCNN_REQUIRE(z_near > 0 && z_far >= z_near && fx != 0 && fy != 0, PCNN_EINVAL, "render_mesh: need 0 < z_near <= z_far and non-zero focal lengths");
  if (num_poses == 0) return PCNN_OK;
  PCNN_REQUIRE(poses && workspace && (num_faces == 0 || (vertices && faces)), PCNN_ENULL, "render_mesh: NULL pointer");
  PCNN_REQUIRE(!out_normals || normals || num_faces == 0, PCNN_ENULL, "render_mesh: a normal map needs per-vertex normals");
  const size_t need = sizeof(unsigned long long) * (size_t)num_poses * height * width;
  PCNN_REQUIRE(workspace_bytes >= need, PCNN_EWORKSPACE, "render_mesh: workspace too small (%zu < %zu)", workspace_bytes, need);
  PCNN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, PCNN_EINVAL, "render_mesh: workspace must be 8-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  unsigned long long* zbuf = static_cast<unsigned long long*>(workspace);
  const long long P = (long long)height * width;
  const long long nz = P * num_poses;
  PCNN_LAUNCH(render_clear_kernel, dim3((unsigned)std::min<long long>((nz + 255) / 256, 8192)), dim3(256), 0, stream, zbuf, nz);
  if (num_faces > 0)
    PCNN_LAUNCH(render_raster_kernel, dim3((num_faces + 255) / 256, num_poses), dim3(256), 0, stream, vertices, faces, num_faces,
                poses, height, width, fx, fy, px, py, z_near, z_far, zbuf);
  PCNN_LAUNCH(render_resolve_kernel, dim3((unsigned)((P + 255) / 256), num_poses), dim3(256), 0, stream, vertices, normals, faces,
              poses, height, width, fx, fy, px, py, z_near, canon_x_offset, zbuf, out_vertices, out_normals, out_canonical);
  return check_launch("render_mesh_fwd");
}
