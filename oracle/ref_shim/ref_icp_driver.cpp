// ref_icp_driver.cpp — C entry points around the reference's pose-refinement bodies, compiled UNCHANGED for the CPU:
//   icp_kernel.inc       = lib/kinect_fusion/src/optimization/icp.cu:20-137        (df::icpKernel)
//   poly3_project.inc    = lib/kinect_fusion/include/df/camera/poly3.h:36-88       (accessors + Poly3CameraModel::project)
//   camera_helpers.inc   = lib/kinect_fusion/include/df/camera/cameraModel.h:72-82 (dehomogenize, applyFocalLengthAndPrincipalPoint)
//   opt_energy.inc       = lib/synthesize/synthesize.cpp:2476-2526                 (optEnergy, the objective of poseWithOpt)
// cut by oracle/ref_shim/Makefile out of the sources where they lie (never stored in this repository) and compiled
// against eigen_sophus_on_cpu.h. What is NOT the reference's code here: the class shells the cut member functions sit in,
// the DataForOpt shell (the fields optEnergy reads, synthesize.hpp:84-103), and these wrappers, which play the role of
// the launch (icp.cu:150-160: one thread per pixel) and hand the per-pixel records / the energy back.
// TEST INFRASTRUCTURE ONLY.
#include "eigen_sophus_on_cpu.h"

namespace df {

template <typename Scalar>
class Poly3CameraModel {
 public:
  Scalar params_[7];
  const Scalar* params() const { return params_; }

 protected:
#include "camera_helpers.inc"

 public:
#include "poly3_project.inc"
};

#include "icp_kernel.inc"

}  // namespace df

typedef Eigen::Matrix<float, 3, 1, Eigen::DontAlign> Vec3;   // synthesize.hpp:65

struct DataForOpt {   // synthesize.hpp:84-103, the members optEnergy reads
  int width, height, objID;
  const int* labelmap;
  std::vector<int>* label_indexes;
  Eigen::Vector2f depthRange;
  df::ManagedHostTensor2<Vec3>* vertex_map;
  df::ManagedHostTensor2<Eigen::UnalignedVec4<float> >* predicted_verts;
};

#include "opt_energy.inc"

template <int DPred>
static void run_icp_kernel(const float* live, const float* pv, const float* pn, int H, int W, const Sophus::SE3f& pose,
                           const df::Poly3CameraModel<float>& cam, float znear, float zfar, float max_error, float* J_out,
                           float* r_out, unsigned char* color_out)
{
  typedef Eigen::UnalignedVec<float, DPred> VecD;
  typedef Eigen::UnalignedVec4<uchar> Color;
  static_assert(sizeof(VecD) == DPred * sizeof(float) && sizeof(Eigen::UnalignedVec3<float>) == 12 && sizeof(Color) == 4, "layout");
  std::vector<df::internal::JacobianAndResidual<float, 1, 6> > jr((size_t)H * W);
  df::DeviceTensor2<Eigen::UnalignedVec3<float> > tl(W, H, (Eigen::UnalignedVec3<float>*)live);
  df::DeviceTensor2<VecD> tv(W, H, (VecD*)pv), tn(W, H, (VecD*)pn);
  df::DeviceTensor2<Color> dbg(W, H, (Color*)color_out);
  Eigen::Matrix<float, 6, 1> initial = Eigen::Matrix<float, 6, 1>::Zero();   // (unused by the kernel body)
  Eigen::Matrix<float, 2, 1> range(znear, zfar);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      blockIdx.x = x;
      blockIdx.y = y;
      df::icpKernel<float, df::Poly3CameraModel<float>, DPred>(jr.data(), tl, tv, tn, cam, pose, initial, range, max_error, dbg);
    }
  blockIdx.x = blockIdx.y = 0;
  for (size_t i = 0; i < jr.size(); i++) {
    for (int k = 0; k < 6; k++) J_out[6 * i + k] = jr[i].J(0, k);
    r_out[i] = jr[i].r;
  }
}

// live [H,W,3], pv / pn [H,W,pc], pose = the CONTENT of an SE3f (unit quaternion wxyz, translation) taken as it is,
// J_out [H*W,6], r_out [H*W], color_out [H*W,4] (the PixelDebugger colour = the body's own exit reason; 0 = untouched)
extern "C" int ref_icp_kernel(const float* live, const float* pv, const float* pn, int H, int W, int pc, const float* q_wxyz,
                              const float* t, float fx, float fy, float px, float py, float znear, float zfar, float max_error,
                              float* J_out, float* r_out, unsigned char* color_out)
{
  df::Poly3CameraModel<float> cam;
  const float prm[7] = {fx, fy, px, py, 0.f, 0.f, 0.f};
  for (int i = 0; i < 7; i++) cam.params_[i] = prm[i];
  const Sophus::SE3f pose = Sophus::SE3f::raw(Eigen::Quaternionf(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]), Sophus::SE3f::Point(t[0], t[1], t[2]));
  memset(color_out, 0, (size_t)H * W * 4);
  if (pc == 3) run_icp_kernel<3>(live, pv, pn, H, W, pose, cam, znear, zfar, max_error, J_out, r_out, color_out);
  else if (pc == 4) run_icp_kernel<4>(live, pv, pn, H, W, pose, cam, znear, zfar, max_error, J_out, r_out, color_out);
  else return -1;
  return 0;
}

// optEnergy on the pixel list `indexes` (the label_indexes of poseWithOpt's caller); live [H,W,3], pv [H,W,4]
extern "C" double ref_opt_energy(const double* pose7, const int* indexes, int n, const float* live, const float* pv, int H, int W,
                                 float znear, float zfar)
{
  std::vector<double> pose(pose7, pose7 + 7), grad;
  std::vector<int> idx(indexes, indexes + n);
  df::ManagedHostTensor2<Vec3> vm(W, H, (Vec3*)live);
  df::ManagedHostTensor2<Eigen::UnalignedVec4<float> > pvt(W, H, (Eigen::UnalignedVec4<float>*)pv);
  DataForOpt d;
  d.width = W; d.height = H; d.objID = 0; d.labelmap = nullptr;
  d.label_indexes = &idx;
  d.depthRange = Eigen::Vector2f(znear, zfar);
  d.vertex_map = &vm;
  d.predicted_verts = &pvt;
  return optEnergy(pose, grad, &d);
}
