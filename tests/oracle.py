"""numpy/ctypes front-end of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see the header of
oracle/pcnn_oracle.c). Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_long, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None

MAX_ROI = 128
CAP = MAX_ROI * 9


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_expf.restype = c_float
        _lib.oracle_expf.argtypes = [c_float]
        _lib.oracle_project_box.restype = c_float
        _lib.oracle_project_box.argtypes = [c_int, c_void_p, c_void_p, c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else c_void_p(0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def exp_softmax(x):
    """oracle_exp_softmax (the all-f32 canonical exp of the softmax layers), element-wise"""
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    L.oracle_exp_softmax.restype = c_float
    L.oracle_exp_softmax.argtypes = [c_float]
    flat_in, flat_out = x.ravel(), out.ravel()
    for i in range(flat_in.size):
        flat_out[i] = L.oracle_exp_softmax(c_float(float(flat_in[i])))
    return out


def expf(x):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.ravel(), out.ravel()
    for i in range(flat_in.size):
        flat_out[i] = L.oracle_expf(c_float(float(flat_in[i])))
    return out


def project_box(cls, extents, meta, distance):
    extents, meta = _f32(extents), _f32(meta)
    return float(lib().oracle_project_box(int(cls), _p(extents), _p(meta), c_float(float(distance))))


def hough_voting(label, vertex, extents, meta, gt, is_train, vote_thr, per_thr, skip,
                 inlier=0.9, label_thr=500, want_hs=False, padded=False, rois_per_image=0):
    """rois_per_image = 0: the reference's capacity rule index_size = MAX_ROI / B (:733);
    > 0: that many maxima per image whatever B (outputs sized B * rois_per_image * (9 | 1) rows)."""
    label, vertex, extents, meta = _i32(label), _f32(vertex), _f32(extents), _f32(meta)
    B, H, W = label.shape
    C = vertex.shape[3] // 3
    num_meta = meta.shape[-1]
    meta = meta.reshape(B, num_meta)
    if gt is None or len(gt) == 0:
        gt_a, num_gt = None, 0
    else:
        gt_a = _f32(gt)
        num_gt = gt_a.shape[0]
    cap = CAP if not rois_per_image else max(1, B * int(rois_per_image) * (9 if is_train else 1))
    top_box = np.empty((cap, 7), np.float32)
    top_pose = np.empty((cap, 7), np.float32)
    top_target = np.empty((cap, 4 * C), np.float32)
    top_weight = np.empty((cap, 4 * C), np.float32)
    top_domain = np.empty((cap,), np.int32)
    num_rois = np.zeros(2, np.int32)
    hs = np.zeros((B, C, H * W), np.float32) if want_hs else None
    st = lib().oracle_hough_voting_ex(_p(label), _p(vertex), _p(extents), _p(meta), _p(gt_a),
                                      B, H, W, C, num_meta, num_gt, int(is_train), c_float(vote_thr),
                                      c_float(per_thr), int(skip), c_float(inlier), int(label_thr),
                                      int(rois_per_image), cap,
                                      _p(top_box), _p(top_pose), _p(top_target), _p(top_weight),
                                      _p(top_domain), _p(num_rois), _p(hs))
    assert st == 0
    if padded:
        res = (top_box, top_pose, top_target, top_weight, top_domain, num_rois)
    else:
        r = int(num_rois[0])
        res = (top_box[:r], top_pose[:r], top_target[:r], top_weight[:r], top_domain[:r])
    return res + (hs,) if want_hs else res


def hough_space(labelmap, vertmap, extents, meta, cls, skip, inlier=0.9):
    labelmap, vertmap, extents, meta = _i32(labelmap), _f32(vertmap), _f32(extents), _f32(meta)
    H, W = labelmap.shape
    C = vertmap.shape[2] // 3
    hs = np.empty((H, W), np.float32)
    hd = np.empty((H, W, 3), np.float32)
    m = lib().oracle_hough_space(_p(labelmap), _p(vertmap), _p(extents), _p(meta), H, W, C, int(cls),
                                 int(skip), c_float(inlier), _p(hs), _p(hd))
    return hs, hd, m


def roi_pool(data, rois, PH, PW, scale, pool_channel):
    data, rois = _f32(data), _f32(rois)
    B, H, W, C = data.shape
    R, cols = rois.shape
    Cout = 1 if pool_channel else C
    top = np.empty((R, PH, PW, Cout), np.float32)
    argmax = np.empty((R, PH, PW, Cout), np.int32)
    lib().oracle_roi_pool(_p(data), _p(rois), B, H, W, C, R, cols, PH, PW, c_float(scale),
                          int(pool_channel), _p(top), _p(argmax))
    return top, argmax


def roi_pool_bwd(top_diff, rois, argmax, B, H, W, C, PH, PW, scale, pool_channel):
    top_diff, rois, argmax = _f32(top_diff), _f32(rois), _i32(argmax)
    R, cols = rois.shape
    out = np.empty((B, H, W, C), np.float32)
    lib().oracle_roi_pool_bwd(_p(top_diff), _p(rois), _p(argmax), B, H, W, C, R, cols, PH, PW,
                              c_float(scale), int(pool_channel), _p(out))
    return out


def hard_label(prob, gt, threshold):
    prob, gt = _f32(prob), _i32(gt)
    C = prob.shape[-1]
    N = prob.size // C
    out = np.empty_like(prob)
    lib().oracle_hard_label(_p(prob), _p(gt), c_long(N), C, c_float(threshold), _p(out))
    return out


def average_distance(pred, target, weight, point, symmetry, margin):
    pred, target, weight, point, symmetry = map(_f32, (pred, target, weight, point, symmetry))
    R = pred.shape[0]
    C, P = point.shape[0], point.shape[1]
    loss = np.zeros(1, np.float32)
    diff = np.zeros((R, 4 * C), np.float32)
    lib().oracle_average_distance(_p(pred), _p(target), _p(weight), _p(point), _p(symmetry), R, C, P,
                                  c_float(margin), _p(loss), _p(diff))
    return loss, diff


def average_distance_bwd(grad, bottom_diff):
    grad, bottom_diff = _f32(grad), _f32(bottom_diff)
    out = np.empty_like(bottom_diff)
    lib().oracle_average_distance_bwd(_p(grad), _p(bottom_diff), bottom_diff.shape[0], bottom_diff.shape[1], _p(out))
    return out


def backproject(data, label, depth, meta, label_3d, G, ksize, threshold):
    data, label, depth, meta, label_3d = map(_f32, (data, label, depth, meta, label_3d))
    B, H, W, Cd = data.shape
    Cl = label.shape[3]
    num_meta = meta.shape[-1]
    top_data = np.empty((B, G, G, G, Cd), np.float32)
    top_flag = np.empty((B, G, G, G, Cd), np.float32)
    top_label = np.empty((B, G, G, G, Cl), np.float32)
    lib().oracle_backproject(_p(data), _p(label), _p(depth), _p(meta), _p(label_3d), B, H, W, Cd, Cl,
                             num_meta, G, int(ksize), c_float(threshold), _p(top_data), _p(top_label), _p(top_flag))
    return top_data, top_label, top_flag


def backproject_sample(data, label, depth, meta, label_3d, G, ksize, threshold, first, stride):
    """Rows first, first + stride, ... of backproject's three outputs ([rows, Cd], [rows, Cl], [rows, Cd])."""
    data, label, depth, meta, label_3d = map(_f32, (data, label, depth, meta, label_3d))
    B, H, W, Cd = data.shape
    Cl = label.shape[3]
    num_meta = meta.shape[-1]
    nvox = B * G * G * G
    rows = (nvox - first + stride - 1) // stride
    top_data = np.empty((rows, Cd), np.float32)
    top_flag = np.empty((rows, Cd), np.float32)
    top_label = np.empty((rows, Cl), np.float32)
    rc = lib().oracle_backproject_sample(_p(data), _p(label), _p(depth), _p(meta), _p(label_3d), B, H, W, Cd, Cl, num_meta, G,
                                         int(ksize), c_float(threshold), c_long(first), c_long(stride), _p(top_data), _p(top_label),
                                         _p(top_flag))
    assert rc == 0
    return top_data, top_label, top_flag


def backproject_bwd(top_diff, depth, meta, B, H, W, Cd, G):
    top_diff, depth, meta = map(_f32, (top_diff, depth, meta))
    num_meta = meta.shape[-1]
    out = np.empty((B, H, W, Cd), np.float32)
    lib().oracle_backproject_bwd(_p(top_diff), _p(depth), _p(meta), B, H, W, Cd, num_meta, G, _p(out))
    return out


def softmax_argmax(score):
    score = _f32(score)
    C = score.shape[-1]
    N = score.size // C
    prob = np.empty_like(score)
    label = np.empty(score.shape[:-1], np.int32)
    lib().oracle_softmax_argmax(_p(score), c_long(N), C, _p(prob), _p(label))
    return prob, label


def deconv_bilinear(x, k, s, add1=None, add2=None, bias=None, relu=False):
    x = _f32(x)
    B, H, W, C = x.shape
    out = np.empty((B, H * s, W * s, C), np.float32)
    a1 = _f32(add1) if add1 is not None else None
    a2 = _f32(add2) if add2 is not None else None
    bs = _f32(bias) if bias is not None else None
    lib().oracle_deconv_bilinear(_p(x), B, H, W, C, int(k), int(s), _p(a1), _p(a2), _p(bs), int(bool(relu)), _p(out))
    return out


def hough_cpu_kernel(label, vertex, extents, meta, max_rows=1024):
    """H7: the reference's CPU kernel semantics (ray marching) — a different algorithm, timing only."""
    label = np.ascontiguousarray(label, dtype=np.int32)
    vertex, extents, meta = _f32(vertex), _f32(extents), _f32(meta)
    B, H, W = label.shape
    C = vertex.shape[3] // 3
    out = np.zeros((max_rows, 14), np.float32)
    n = lib().oracle_hough_cpu_kernel(_p(label), _p(vertex), _p(extents), _p(meta), B, H, W, C, meta.shape[-1], _p(out), max_rows)
    return out[:max(n, 0)]


def deconv_bilinear_bwd(grad_out, k, s):
    g = _f32(grad_out)
    B, Ho, Wo, C = g.shape
    gin = np.empty((B, Ho // s, Wo // s, C), np.float32)
    lib().oracle_deconv_bilinear_bwd(_p(g), B, Ho // s, Wo // s, C, int(k), int(s), _p(gin))
    return gin


def smooth_l1_vertex(pred, target, weight, sigma=1.0, want_grad=True):
    """-> (out[3] = loss, sum(in_loss), sum(weight); grad d loss / d pred or None)"""
    import ctypes
    pred, target, weight = _f32(pred), _f32(target), _f32(weight)
    out = np.empty(3, np.float32)
    grad = np.empty_like(pred) if want_grad else None
    f = lib().oracle_smooth_l1_vertex
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    f(_p(pred), _p(target), _p(weight), pred.size, float(sigma), _p(out), _p(grad))
    return out, grad


def upscore_softmax_argmax(z, bias, k, s, relu=True):
    score = deconv_bilinear(z, k, s, bias=bias, relu=relu)
    prob, label = softmax_argmax(score)
    return score, prob, label


# ---- depth-based pose refinement ---------------------------------------------------------------
def icp_backproject(depth, label, obj_id, K, factor):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    H, W = depth.shape
    lab = None if label is None else _i32(label)
    out = np.empty((H, W, 3), np.float32)
    L = lib()
    L.oracle_icp_backproject.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p]
    L.oracle_icp_backproject(_p(depth), _p(lab), H, W, int(obj_id), float(factor), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), _p(out))
    return out


def icp_refine(live, pred_v, pred_n, K, depth_range=(0.25, 6.0), max_error=0.01, iterations=8):
    """oracle_icp_refine: live [N,H,W,3], pred_* [N,H,W,3|4] -> (update f64 [N,3,4], stats f32 [N,iterations,2])"""
    live, pred_v, pred_n = _f32(live), _f32(pred_v), _f32(pred_n)
    N, H, W, _ = live.shape
    pc = pred_v.shape[3]
    upd = np.empty((N, 3, 4), np.float64)
    stats = np.zeros((N, max(iterations, 1), 2), np.float32)
    L = lib()
    L.oracle_icp_refine.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                    c_float, c_float, c_float, c_int, c_void_p, c_void_p]
    rc = L.oracle_icp_refine(_p(live), _p(pred_v), _p(pred_n), N, H, W, pc, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                             float(depth_range[0]), float(depth_range[1]), float(max_error), int(iterations), _p(upd), _p(stats))
    assert rc == 0
    return upd, stats


def icp_se3f(T):
    """The content of the SE3f (unit quaternion wxyz f32 [4], translation f32 [3]) icp_refine hands to the per-pixel step
    for an accumulated 3x4 transform T (f64)."""
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(12))
    q, t = np.empty(4, np.float32), np.empty(3, np.float32)
    lib().oracle_icp_se3f(_p(T), _p(q), _p(t))
    return q, t


def icp_terms(live, pred_v, pred_n, q, t, K, depth_range=(0.25, 6.0), max_error=0.01):
    """oracle_icp_terms: the per-pixel records of ONE icpKernel launch at the SE3f (q wxyz, t) ->
    (J f32 [H,W,6], r f32 [H,W], reason uint8 [H,W]: 0 contributes, 1 pred depth, 2 border, 3 live depth, 4 ray/normal, 5 error)"""
    live, pred_v, pred_n, q, t = _f32(live), _f32(pred_v), _f32(pred_n), _f32(q), _f32(t)
    H, W, pc = pred_v.shape
    J, r, why = np.empty((H, W, 6), np.float32), np.empty((H, W), np.float32), np.empty((H, W), np.uint8)
    rc = lib().oracle_icp_terms(_p(live), _p(pred_v), _p(pred_n), H, W, pc, _p(q), _p(t), c_float(K[0, 0]), c_float(K[1, 1]),
                                c_float(K[0, 2]), c_float(K[1, 2]), c_float(depth_range[0]), c_float(depth_range[1]),
                                c_float(max_error), _p(J), _p(r), _p(why))
    assert rc == 0
    return J, r, why


def icp_energy_terms(live, pred_v, x, depth_range=(0.25, 6.0)):
    """per-pixel term of optEnergy at the optimiser's point x f64 [7] -> (dist f32 [H,W], valid uint8 [H,W])"""
    live, pred_v = _f32(live), _f32(pred_v)
    H, W, pc = pred_v.shape
    x = np.ascontiguousarray(x, dtype=np.float64)
    d, v = np.empty((H, W), np.float32), np.empty((H, W), np.uint8)
    lib().oracle_icp_energy_terms(_p(live), _p(pred_v), pc, H, W, c_float(depth_range[0]), c_float(depth_range[1]), _p(x), _p(d), _p(v))
    return d, v


def icp_energy(label, live, pred_v, obj_id, x, depth_range=(0.25, 6.0)):
    """the polish objective (canonical parallel sum) over the pixels labelled obj_id"""
    label, live, pred_v = _i32(label), _f32(live), _f32(pred_v)
    H, W, pc = pred_v.shape
    x = np.ascontiguousarray(x, dtype=np.float64)
    L = lib()
    L.oracle_icp_energy.restype = ctypes.c_double
    return float(L.oracle_icp_energy(_p(label), _p(live), _p(pred_v), pc, H, W, int(obj_id), c_float(depth_range[0]),
                                     c_float(depth_range[1]), _p(x)))


def render_mesh(vertices, normals, faces, poses, K, H, W, depth_range=(0.25, 6.0), model_index=0, want=("vertices", "normals", "canonical")):
    """oracle_render_mesh: poses [N,3,4] -> dict of "vertices" / "normals" f32 [N,H,W,4], "canonical" f32 [N,H,W,3] (NaN = no surface)"""
    v, f = _f32(vertices).reshape(-1, 3), _i32(faces).reshape(-1, 3)
    n = None if normals is None else _f32(normals).reshape(-1, 3)
    P = _f32(poses).reshape(-1, 12)
    N = P.shape[0]
    out = {"vertices": np.empty((N, H, W, 4), np.float32) if "vertices" in want else None,
           "normals": np.empty((N, H, W, 4), np.float32) if "normals" in want else None,
           "canonical": np.empty((N, H, W, 3), np.float32) if "canonical" in want else None}
    L = lib()
    L.oracle_render_mesh.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                     c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]
    rc = L.oracle_render_mesh(_p(v), _p(n), _p(f), len(v), len(f), _p(P), N, H, W, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                              float(depth_range[0]), float(depth_range[1]), float(model_index), _p(out["vertices"]), _p(out["normals"]), _p(out["canonical"]))
    assert rc == 0
    return {k: a for k, a in out.items() if a is not None}


def icp_center(label, live, canonical, pred_v, pred_n, obj_id, max_error=0.01):
    """oracle_icp_center -> (sums f64 [5], mask uint8 [H,W])"""
    label, live, canonical, pred_v, pred_n = _i32(label), _f32(live), _f32(canonical), _f32(pred_v), _f32(pred_n)
    H, W = label.shape
    sums = np.empty((5,), np.float64)
    mask = np.empty((H, W), np.uint8)
    L = lib()
    L.oracle_icp_center.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]
    rc = L.oracle_icp_center(_p(label), _p(live), _p(canonical), _p(pred_v), _p(pred_n), pred_v.shape[2], H, W, int(obj_id), float(max_error), _p(sums), _p(mask))
    assert rc == 0
    return sums, mask


def icp_score(live, canonical, mask, hypotheses, radius=0.01):
    """oracle_icp_score (exhaustive nearest-neighbour search) -> int32 [M]"""
    live, canonical = _f32(live), _f32(canonical)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = mask.shape
    hyp = _f32(hypotheses).reshape(-1, 12)
    hits = np.empty((hyp.shape[0],), np.int32)
    L = lib()
    L.oracle_icp_score.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_void_p]
    rc = L.oracle_icp_score(_p(live), _p(canonical), _p(mask), H, W, _p(hyp), hyp.shape[0], float(radius), _p(hits))
    assert rc == 0
    return hits


def icp_polish(label, live, pred_v, obj_id, depth_range=(0.25, 6.0), maxeval=50):
    """oracle_icp_polish -> (x f64 [7] = update (quaternion wxyz un-normalised, translation), energy, evaluations)"""
    label, live, pred_v = _i32(label), _f32(live), _f32(pred_v)
    H, W = label.shape
    x = np.empty((7,), np.float64)
    info = np.empty((2,), np.float64)
    L = lib()
    L.oracle_icp_polish.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p]
    rc = L.oracle_icp_polish(_p(label), _p(live), _p(pred_v), pred_v.shape[2], H, W, int(obj_id), float(depth_range[0]), float(depth_range[1]), int(maxeval), _p(x), _p(info))
    assert rc == 0
    return x, float(info[0]), int(info[1])
