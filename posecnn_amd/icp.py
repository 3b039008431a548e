"""Depth-based pose refinement, first slice: the projective point-to-plane ICP core of
`Synthesizer::solveICP` / `refinePose` (lib/synthesize/synthesize.cpp:2052-2380, :1969-2029 -> `df::icp`,
lib/kinect_fusion/src/optimization/icp.cpp:20-106), which lib/fcn/test.py:1925-1933 calls after the network when
`cfg.TEST.POSE_REFINE` is set:

    synthesizer.icp_python(labels_icp, im_depth, parameters, height, width, num_roi, channel_roi,
                           rois_icp, poses, poses_new, poses_icp, error_threshold)

What is here: the masked depth -> vertex-map step and the ICP iterations (gfx950 kernels behind
`pcnn_icp_backproject_fwd` / `pcnn_icp_refine_fwd`, all objects of a frame in one call, no host round trip between
iterations), and `refine_poses`, the per-frame driver with the reference's argument meaning. What is NOT here: the
OpenGL renderer that turns (mesh, pose) into the predicted vertex / normal maps — `refine_poses` takes it as a
callable —, the PCL kd-tree scoring of the 11 depth hypotheses and the nlopt stage (synthesize.cpp:2237-2343).
"""
import ctypes

import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .pose_error import quat2mat

Z_NEAR, Z_FAR = 0.25, 6.0          # lib/fcn/test.py:1905-1906
ERROR_THRESHOLD = 0.01             # lib/fcn/test.py:1909


def backproject(depth, label, obj_id, K, factor_depth):
    """synthesize.cpp:2139-2155: the depth of the object's pixels (0 elsewhere), back-projected with the intrinsics
    (df/image/backprojection). depth uint16 [H,W] (torch, GPU), label int32 [H,W] or None -> f32 [H,W,3]."""
    depth = ops._dev(depth, "depth", torch.uint16)
    lab = ops._dev(label, "label", torch.int32) if label is not None else None
    H, W = depth.shape
    out = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device)
    check("pcnn_icp_backproject_fwd",
          lib().pcnn_icp_backproject_fwd(ops._ptr(depth), ops._ptr(lab), H, W, int(obj_id), float(factor_depth), float(K[0, 0]), float(K[1, 1]),
                                         float(K[0, 2]), float(K[1, 2]), ops._ptr(out), ops._stream(depth)))
    return out


def icp(live_vertices, pred_vertices, pred_normals, K, depth_range=(Z_NEAR, Z_FAR), max_error=ERROR_THRESHOLD, iterations=8,
        want_stats=False):
    """`df::icp` for N independent problems: live_vertices f32 [N,H,W,3], pred_vertices / pred_normals f32 [N,H,W,3|4].
    Returns the accumulated update as f64 [N,3,4] (T_new = update * T_old, icp.cpp:96 / synthesize.cpp:2025) and, with
    want_stats, f32 [N,iterations,2] = (inliers, sum r^2) seen by each iteration."""
    live = ops._dev(live_vertices, "live_vertices", torch.float32)
    pv = ops._dev(pred_vertices, "pred_vertices", torch.float32)
    pn = ops._dev(pred_normals, "pred_normals", torch.float32)
    if live.dim() != 4 or live.shape[3] != 3 or pv.dim() != 4 or pv.shape[:3] != live.shape[:3] or pv.shape != pn.shape or pv.shape[3] not in (3, 4):
        raise ValueError("live_vertices [N,H,W,3], pred_vertices / pred_normals [N,H,W,3|4]")
    N, H, W, _ = live.shape
    update = torch.empty((N, 3, 4), dtype=torch.float64, device=live.device)
    stats = torch.zeros((N, max(iterations, 1), 2), dtype=torch.float32, device=live.device) if want_stats else None
    nbytes = ctypes.c_size_t()
    check("pcnn_icp_refine_workspace_bytes", lib().pcnn_icp_refine_workspace_bytes(N, H, W, ctypes.byref(nbytes)))
    ws = ops._ws(live.device, "icp").get(nbytes.value, live.device)
    check("pcnn_icp_refine_fwd",
          lib().pcnn_icp_refine_fwd(ops._ptr(live), ops._ptr(pv), ops._ptr(pn), N, H, W, int(pv.shape[3]), float(K[0, 0]), float(K[1, 1]),
                                    float(K[0, 2]), float(K[1, 2]), float(depth_range[0]), float(depth_range[1]), float(max_error),
                                    int(iterations), ops._ptr(update), ops._ptr(stats), ops._ptr(ws), nbytes.value, ops._stream(live)))
    return (update, stats) if want_stats else update


def mat2quat(R):
    """Rotation matrix -> unit quaternion (w, x, y, z), w >= 0 (what Sophus::SE3f::unit_quaternion() hands to the
    output arrays of solveICP, synthesize.cpp:2366-2375)."""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[1 + i] = 0.25 * s
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q /= np.linalg.norm(q)
    return q if q[0] >= 0 else -q


def refine_poses(labels, depth, K, factor_depth, rois, poses, render_fn, iterations=8, max_error=ERROR_THRESHOLD,
                 depth_range=(Z_NEAR, Z_FAR), min_pixels=400, device="cuda"):
    """The ICP leg of `icp_python(labels, depth, parameters, H, W, num_roi, channel_roi, rois, poses, outputs, outputs_icp,
    maxError)`: for every ROI with a class id > 0 and at least `min_pixels` label pixels (synthesize.cpp:2093, :2152),
    T_co <- icp(masked depth, maps rendered at T_co) * T_co. `render_fn(class_id, T_co[3,4] float64) -> (vertex_map,
    normal_map)` (numpy or torch f32 [H,W,3|4], camera coordinates, background depth outside depth_range) stands in for
    the reference's GL renderer (synthesize.cpp:1972-1991). labels int32 [H,W], depth uint16 [H,W] (numpy),
    rois [R,>=2], poses [R,7] = (quaternion wxyz, translation). Returns poses_icp f32 [R,7] (rows of skipped ROIs: 0)."""
    dev = torch.device(device)
    labels_t = torch.as_tensor(np.ascontiguousarray(labels, dtype=np.int32)).to(dev)
    depth_t = depth if isinstance(depth, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(depth, dtype=np.uint16)).to(dev)
    K = np.asarray(K, dtype=np.float64)
    rois, poses = np.asarray(rois), np.asarray(poses, dtype=np.float64)
    out = np.zeros((rois.shape[0], 7), dtype=np.float32)
    todo, lives, pvs, pns, T0 = [], [], [], [], []
    for i in range(rois.shape[0]):
        cls = int(rois[i, 1])
        if cls <= 0 or int((labels == cls).sum()) < min_pixels:
            continue
        T = np.zeros((3, 4))
        T[:, :3] = quat2mat(poses[i, :4])
        T[:, 3] = poses[i, 4:7]
        pv, pn = render_fn(cls, T)
        todo.append(i)
        T0.append(T)
        lives.append(backproject(depth_t, labels_t, cls, K, factor_depth))
        pvs.append(torch.as_tensor(pv, dtype=torch.float32).to(dev))
        pns.append(torch.as_tensor(pn, dtype=torch.float32).to(dev))
    if not todo:
        return out
    upd = icp(torch.stack(lives), torch.stack(pvs), torch.stack(pns), K, depth_range, max_error, iterations).cpu().numpy()
    for i, T, U in zip(todo, T0, upd):
        R = U[:, :3] @ T[:, :3]
        t = U[:, :3] @ T[:, 3] + U[:, 3]
        out[i, :4] = mat2quat(R)
        out[i, 4:] = t
    return out
