"""Depth-based pose refinement: `Synthesizer::solveICP` (lib/synthesize/synthesize.cpp:2052-2380), which
lib/fcn/test.py:1925-1933 calls after the network when `cfg.TEST.POSE_REFINE` is set:

    synthesizer.icp_python(labels_icp, im_depth, parameters, height, width, num_roi, channel_roi,
                           rois_icp, poses, poses_new, poses_icp, error_threshold)

`Synthesizer` below keeps that call (same argument meaning, results written into `poses_new` / `poses_icp`). Per ROI with a
class id > 0 and >= 400 label pixels, all on the GPU (gfx950 kernels of libposecnn_hip.so):

    render the mesh at the network's pose            pcnn_render_mesh_fwd     (the reference: two OpenGL passes, :2104-2136)
    masked depth -> live vertex map                  pcnn_icp_backproject_fwd (:2139-2155)
    translation from depth vs. rendered surface      pcnn_icp_center_fwd      (:2157-2225)
    re-render + Nelder-Mead polish of the pose       pcnn_icp_polish_fwd      (:2226-2235, poseWithOpt :2529-2570)  -> poses_new
    8 depth hypotheses, each: render + 8 ICP steps   pcnn_render_mesh_fwd + pcnn_icp_refine_fwd, the 8 in ONE call each (:2272-2300)
    SegICP score of the 8 refined hypotheses         pcnn_icp_score_fwd       (:2302-2343)  -> poses_icp = best

(nlopt is absent: the polish is its published Nelder-Mead restated, the whole optimisation in one launch.) The objects of a
frame run phase by phase on their own streams.

No assimp here: meshes are Wavefront OBJ read by `Mesh.load_obj` (positions, faces, optional normals; smooth normals are
generated like aiProcess_GenSmoothNormals when the file has none).
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


def _pose34(q_t):
    T = np.zeros((3, 4))
    T[:, :3] = quat2mat(np.asarray(q_t[:4], dtype=np.float64))
    T[:, 3] = q_t[4:7]
    return T


def _compose(U, T):
    """U * T for 3x4 rigid transforms (Sophus: update * T_co, synthesize.cpp:2025)"""
    out = np.zeros((3, 4))
    out[:, :3] = U[:, :3] @ T[:, :3]
    out[:, 3] = U[:, :3] @ T[:, 3] + U[:, 3]
    return out


class Mesh:
    """One object model as the reference keeps it after `loadTexturedMesh` + `initializeBuffers`
    (synthesize.cpp:197-320): positions, per-vertex normals, triangle indices — on the GPU."""

    def __init__(self, vertices, faces, normals=None, device="cuda"):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        if f.size and (f.min() < 0 or f.max() >= len(v)):
            raise ValueError("face index out of range (0..%d)" % (len(v) - 1))
        n = self.smooth_normals(v, f) if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        if n.shape != v.shape:
            raise ValueError("one normal per vertex")
        self.vertices_np, self.faces_np, self.normals_np = v, f, n
        dev = torch.device(device)
        self.vertices = torch.from_numpy(v).to(dev)
        self.faces = torch.from_numpy(f).to(dev)
        self.normals = torch.from_numpy(n).to(dev)

    @staticmethod
    def smooth_normals(v, f):
        """aiProcess_GenSmoothNormals (synthesize.cpp:199): face normals (v1-v0) x (v2-v0), un-normalised — i.e. area
        weighted —, summed over the faces around a vertex and normalised."""
        v = v.astype(np.float64)
        fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        n = np.zeros_like(v)
        for k in range(3):
            np.add.at(n, f[:, k], fn)
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        return (n / np.maximum(ln, 1e-30)).astype(np.float32)

    @classmethod
    def load_obj(cls, path, device="cuda"):
        """Wavefront OBJ: `v`, `vn`, `f` records (polygons are fanned into triangles; `a/b/c` index triples: when every
        corner's normal index equals its position index the file's normals are used, else smooth normals are generated).
        Vertices are NOT merged (assimp's JoinIdenticalVertices only matters for the generated normals of duplicated
        positions, which OBJ exports of the YCB / LINEMOD models do not have)."""
        vs, vns, faces, nidx_ok = [], [], [], True
        with open(path) as fh:
            for line in fh:
                t = line.split()
                if not t:
                    continue
                if t[0] == "v":
                    vs.append([float(x) for x in t[1:4]])
                elif t[0] == "vn":
                    vns.append([float(x) for x in t[1:4]])
                elif t[0] == "f":
                    idx = []
                    for c in t[1:]:
                        parts = c.split("/")
                        vi = int(parts[0])
                        vi = vi - 1 if vi > 0 else len(vs) + vi
                        if len(parts) >= 3 and parts[2]:
                            ni = int(parts[2])
                            ni = ni - 1 if ni > 0 else len(vns) + ni
                            nidx_ok = nidx_ok and ni == vi
                        else:
                            nidx_ok = False
                        idx.append(vi)
                    for k in range(1, len(idx) - 1):
                        faces.append([idx[0], idx[k], idx[k + 1]])
        normals = np.asarray(vns, dtype=np.float32) if (nidx_ok and len(vns) == len(vs) and vns) else None
        return cls(np.asarray(vs, dtype=np.float32), np.asarray(faces, dtype=np.int32), normals, device)


def render(mesh, poses, K, height, width, depth_range=(Z_NEAR, Z_FAR), model_index=0, want=("vertices", "normals")):
    """The predicted maps of `refinePose` / `solveICP` (synthesize.cpp:1972-1991, :2104-2136) for N poses in one call.
    poses [N,3,4] (camera <- object; numpy or torch). Returns a dict with the requested maps: "vertices" / "normals"
    f32 [N,H,W,4], "canonical" f32 [N,H,W,3] (object-frame point, x + model_index); NaN where no surface is hit."""
    dev = mesh.vertices.device
    P = torch.as_tensor(np.asarray(poses, dtype=np.float32) if not isinstance(poses, torch.Tensor) else poses, dtype=torch.float32).to(dev)
    P = P.reshape(-1, 12).contiguous()
    N = P.shape[0]
    out = {}
    for key, ch in (("vertices", 4), ("normals", 4), ("canonical", 3)):
        out[key] = torch.empty((N, height, width, ch), dtype=torch.float32, device=dev) if key in want else None
    nbytes = ctypes.c_size_t()
    check("pcnn_render_mesh_workspace_bytes", lib().pcnn_render_mesh_workspace_bytes(N, height, width, ctypes.byref(nbytes)))
    ws = ops._ws(dev, "render").get(nbytes.value, dev)
    check("pcnn_render_mesh_fwd",
          lib().pcnn_render_mesh_fwd(ops._ptr(mesh.vertices), ops._ptr(mesh.normals), ops._ptr(mesh.faces), mesh.vertices.shape[0],
                                     mesh.faces.shape[0], ops._ptr(P), N, height, width, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
                                     float(K[1, 2]), float(depth_range[0]), float(depth_range[1]), float(model_index),
                                     ops._ptr(out["vertices"]), ops._ptr(out["normals"]), ops._ptr(out["canonical"]), ops._ptr(ws),
                                     nbytes.value, ops._stream(mesh.vertices)))
    return {k: v for k, v in out.items() if v is not None}


def center(label, live_vertices, canonical, pred_vertices, pred_normals, obj_id, max_error=ERROR_THRESHOLD):
    """synthesize.cpp:2157-2207: returns (sums f64 [5] = sum (d - m).xyz over the pixels whose depth point agrees with the
    render, their count, number of valid (depth point, model point) pairs; mask uint8 [H,W] of those pairs)."""
    lab = ops._dev(label, "label", torch.int32)
    live = ops._dev(live_vertices, "live_vertices", torch.float32)
    can = ops._dev(canonical, "canonical", torch.float32)
    pv = ops._dev(pred_vertices, "pred_vertices", torch.float32)
    pn = ops._dev(pred_normals, "pred_normals", torch.float32)
    H, W = lab.shape
    if live.shape != (H, W, 3) or can.shape != (H, W, 3) or pv.shape[:2] != (H, W) or pv.shape != pn.shape or pv.shape[2] not in (3, 4):
        raise ValueError("label [H,W], live / canonical [H,W,3], pred_* [H,W,3|4]")
    sums = torch.empty((5,), dtype=torch.float64, device=lab.device)
    mask = torch.empty((H, W), dtype=torch.uint8, device=lab.device)
    nbytes = ctypes.c_size_t()
    check("pcnn_icp_center_workspace_bytes", lib().pcnn_icp_center_workspace_bytes(H, W, ctypes.byref(nbytes)))
    ws = ops._ws(lab.device, "icp_center").get(nbytes.value, lab.device)
    check("pcnn_icp_center_fwd",
          lib().pcnn_icp_center_fwd(ops._ptr(lab), ops._ptr(live), ops._ptr(can), ops._ptr(pv), ops._ptr(pn), int(pv.shape[2]), H, W,
                                    int(obj_id), float(max_error), ops._ptr(sums), ops._ptr(mask), ops._ptr(ws), nbytes.value,
                                    ops._stream(lab)))
    return sums, mask


def score(live_vertices, canonical, mask, hypotheses, K, radius=0.01):
    """synthesize.cpp:2302-2343: int32 [M] = distinct depth points marked by the model points moved by each hypothesis
    (hypotheses [M,3,4]); the reference's score is this divided by the number of model points."""
    live = ops._dev(live_vertices, "live_vertices", torch.float32)
    can = ops._dev(canonical, "canonical", torch.float32)
    msk = ops._dev(mask, "mask", torch.uint8)
    H, W = msk.shape
    hyp = torch.as_tensor(np.asarray(hypotheses, dtype=np.float32) if not isinstance(hypotheses, torch.Tensor) else hypotheses,
                          dtype=torch.float32).to(live.device).reshape(-1, 12).contiguous()
    M = hyp.shape[0]
    hits = torch.empty((M,), dtype=torch.int32, device=live.device)
    nbytes = ctypes.c_size_t()
    check("pcnn_icp_score_workspace_bytes", lib().pcnn_icp_score_workspace_bytes(M, H, W, ctypes.byref(nbytes)))
    ws = ops._ws(live.device, "icp_score").get(max(nbytes.value, 4), live.device)
    check("pcnn_icp_score_fwd",
          lib().pcnn_icp_score_fwd(ops._ptr(live), ops._ptr(can), ops._ptr(msk), H, W, ops._ptr(hyp), M, float(K[0, 0]), float(K[1, 1]),
                                   float(K[0, 2]), float(K[1, 2]), float(radius), ops._ptr(hits), ops._ptr(ws), nbytes.value,
                                   ops._stream(live)))
    return hits


def polish_async(label, live_vertices, pred_vertices, obj_id, depth_range=(Z_NEAR, Z_FAR), max_evaluations=50):
    """`Synthesizer::poseWithOpt` (synthesize.cpp:2529-2570): Nelder-Mead over an update pose, minimising `optEnergy`
    (:2476-2526), enqueued on the current stream. pred_vertices [H,W,3|4] = the render at the pose the update multiplies.
    Returns device tensors (x f64 [7] = quaternion wxyz un-normalised + translation, info f64 [2] = energy, evaluations)."""
    lab = ops._dev(label, "label", torch.int32)
    live = ops._dev(live_vertices, "live_vertices", torch.float32)
    pv = ops._dev(pred_vertices, "pred_vertices", torch.float32)
    H, W = lab.shape
    if live.shape != (H, W, 3) or pv.shape[:2] != (H, W) or pv.shape[2] not in (3, 4):
        raise ValueError("label [H,W], live [H,W,3], pred_vertices [H,W,3|4]")
    x = torch.empty((7,), dtype=torch.float64, device=lab.device)
    info = torch.empty((2,), dtype=torch.float64, device=lab.device)
    check("pcnn_icp_polish_fwd",
          lib().pcnn_icp_polish_fwd(ops._ptr(lab), ops._ptr(live), ops._ptr(pv), int(pv.shape[2]), H, W, int(obj_id), float(depth_range[0]),
                                    float(depth_range[1]), int(max_evaluations), ops._ptr(x), ops._ptr(info), ops._stream(lab)))
    return x, info


def polish(label, live_vertices, pred_vertices, obj_id, depth_range=(Z_NEAR, Z_FAR), max_evaluations=50):
    """polish_async + read-back: (update 3x4 float64 with the quaternion normalised as Sophus::SE3f does at :2014-2016,
    energy, evaluations, raw x)."""
    x, info = polish_async(label, live_vertices, pred_vertices, obj_id, depth_range, max_evaluations)
    x, info = x.cpu().numpy(), info.cpu().numpy()
    return _pose34(x), float(info[0]), int(info[1]), x


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


HYPOTHESIS_DZ = (0.0, -0.02, -0.01, 0.01, 0.02, 0.03, 0.04, 0.05)    # synthesize.cpp:2252-2270


class Synthesizer:
    """The slice of `libsynthesizer.Synthesizer` the test loop uses (lib/fcn/test.py:1863-1865, :1925-1933):
    `Synthesizer(model_file, pose_file)`, `setup(width, height)`, `icp_python(...)`. `model_file` lists one OBJ path per
    line (synthesize.cpp:147-160; class id c uses line c - 1); `meshes` passes `Mesh` objects directly instead."""

    def __init__(self, model_file=None, pose_file=None, meshes=None, device="cuda", polish_evaluations=50):
        self.model_file, self.pose_file, self.device = model_file, pose_file, torch.device(device)
        self.meshes = list(meshes) if meshes is not None else None
        self.polish_evaluations = polish_evaluations      # `iterations = 50` of synthesize.cpp:2226; 0 skips the polish
        self.width = self.height = None
        self.last = []                # per processed ROI: hits / pairs of each hypothesis and the chosen one
        self._streams = []            # one per object of a frame (created on first use)

    def setup(self, width, height):
        self.width, self.height = int(width), int(height)
        if self.meshes is None:
            if not self.model_file:
                raise ValueError("Synthesizer needs model_file or meshes")
            with open(self.model_file) as fh:
                self.meshes = [Mesh.load_obj(line.strip(), self.device) for line in fh if line.strip()]

    def icp_python(self, labelmap, depth, parameters, height, width, num_roi, channel_roi, rois, poses, outputs, outputs_icp,
                   maxError, iterations=8, min_pixels=400, radius=0.01):
        """`Synthesizer::icp_python` -> `solveICP`. parameters = (fx, fy, px, py, znear, zfar, factor_depth);
        labelmap int32 [H,W]; depth uint16 [H,W]; rois [num_roi, channel_roi] (class id in column 1); poses [num_roi,7]
        (quaternion wxyz, translation); outputs / outputs_icp f32 [num_roi,7] are filled in place (rows of skipped ROIs
        untouched): outputs = pose with the depth-based translation, outputs_icp = best refined hypothesis."""
        if self.meshes is None:
            self.setup(width, height)
        fx, fy, px, py, znear, zfar, factor = [float(x) for x in np.asarray(parameters).reshape(-1)[:7]]
        K = np.array([[fx, 0, px], [0, fy, py], [0, 0, 1]], dtype=np.float64)
        dev = self.device
        labels_np = np.ascontiguousarray(labelmap, dtype=np.int32).reshape(height, width)
        labels_t = torch.from_numpy(labels_np).to(dev)
        depth_t = depth.to(dev) if isinstance(depth, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(depth, dtype=np.uint16).reshape(height, width)).to(dev)
        rois = np.asarray(rois).reshape(num_roi, channel_roi)
        poses = np.asarray(poses, dtype=np.float64).reshape(num_roi, 7)
        counts = np.bincount(labels_np.reshape(-1).clip(min=0), minlength=len(self.meshes) + 2)
        self.last = []
        jobs = []
        for i in range(num_roi):
            obj = int(rois[i, 1])
            if obj <= 0:                                   # :2096
                continue
            if obj > len(self.meshes):
                raise ValueError("ROI %d: class id %d has no model (%d loaded)" % (i, obj, len(self.meshes)))
            if counts[obj] < min_pixels:                   # :2152 (the reference leaves the row as it was)
                continue
            jobs.append({"i": i, "obj": obj, "mesh": self.meshes[obj - 1], "T": _pose34(poses[i])})
        if not jobs:
            return
        # The objects of a frame are independent and most of the kernels are short or narrow (the polish is ONE workgroup for
        # 0.7 ms): each object's chain runs on its own stream, phase by phase, so they overlap on the chip; the host reads a
        # phase's few numbers back object by object while the others are still running.
        cur = torch.cuda.current_stream(dev)
        while len(self._streams) < len(jobs):
            self._streams.append(torch.cuda.Stream(device=dev))
        for job, st in zip(jobs, self._streams):
            job["stream"] = st
            st.wait_stream(cur)
        dr = (znear, zfar)
        for job in jobs:                                   # render at the network's pose, live vertex map, translation votes
            with torch.cuda.stream(job["stream"]):
                job["maps"] = render(job["mesh"], job["T"][None], K, height, width, dr, model_index=job["obj"] - 1,
                                     want=("vertices", "normals", "canonical"))
                job["live"] = backproject(depth_t, labels_t, job["obj"], K, factor)
                job["sums_t"], job["mask"] = center(labels_t, job["live"], job["maps"]["canonical"][0], job["maps"]["vertices"][0],
                                                    job["maps"]["normals"][0], job["obj"], maxError)
        for job in jobs:                                   # :2209-2232: new translation, then the polish (re-render + Nelder-Mead)
            with torch.cuda.stream(job["stream"]):
                sums = job["sums_t"].cpu().numpy()
                i, T_co = job["i"], job["T"]
                job["agree"], job["pairs"], job["Tz"] = int(sums[3]), int(sums[4]), T_co[2, 3]
                job["polish"] = None
                if job["agree"] > 0:
                    Tz = float(np.float32(sums[2]) / np.float32(job["agree"]))
                    rx = poses[i, 4] / poses[i, 6] if poses[i, 6] else 0.0
                    ry = poses[i, 5] / poses[i, 6] if poses[i, 6] else 0.0
                    T_co[:, 3] = (rx * Tz, ry * Tz, Tz)
                    job["Tz"] = Tz
                    if self.polish_evaluations:
                        pv = render(job["mesh"], T_co[None], K, height, width, dr, want=("vertices",))["vertices"][0]
                        job["polish"] = polish_async(labels_t, job["live"], pv, job["obj"], dr, self.polish_evaluations)
        for job in jobs:                                   # :2238-2300: outputs, the 8 depth hypotheses, 8 ICP iterations each
            with torch.cuda.stream(job["stream"]):
                i, T_co = job["i"], job["T"]
                if job["polish"] is not None:
                    x = job["polish"][0].cpu().numpy()
                    T_co = _compose(_pose34(x), T_co)
                    job["T"], job["Tz"] = T_co, T_co[2, 3]
                outputs[i, :4] = mat2quat(T_co[:, :3])
                outputs[i, 4:7] = T_co[:, 3]
                hyps = np.repeat(T_co[None], len(HYPOTHESIS_DZ), 0)
                hyps[:, 2, 3] = job["Tz"] + np.asarray(HYPOTHESIS_DZ)
                pm = render(job["mesh"], hyps, K, height, width, dr, want=("vertices", "normals"))
                live_n = job["live"].unsqueeze(0).expand(len(hyps), -1, -1, -1).contiguous()
                job["hyps"] = hyps
                job["upd_t"] = icp(live_n, pm["vertices"], pm["normals"], K, dr, maxError, iterations)
        for job in jobs:                                   # :2302-2343: SegICP score of the refined hypotheses
            with torch.cuda.stream(job["stream"]):
                upd = job["upd_t"].cpu().numpy()
                job["hyps"] = np.stack([_compose(U, T) for U, T in zip(upd, job["hyps"])])
                job["hits_t"] = score(job["live"], job["maps"]["canonical"][0], job["mask"], job["hyps"], K, radius) if job["pairs"] > 0 else None
        for job in jobs:
            with torch.cuda.stream(job["stream"]):
                choose, hits = 0, None
                if job["hits_t"] is not None:
                    hits = job["hits_t"].cpu().numpy()
                    choose = int(np.argmax(hits))          # first maximum (`score > max_score`, :2336)
                i = job["i"]
                self.last.append({"roi": i, "obj": job["obj"], "pairs": job["pairs"], "agree": job["agree"], "hits": hits, "choose": choose})
                outputs_icp[i, :4] = mat2quat(job["hyps"][choose][:, :3])
                outputs_icp[i, 4:7] = job["hyps"][choose][:, 3]
            cur.wait_stream(job["stream"])
