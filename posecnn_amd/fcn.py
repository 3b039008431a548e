"""Single-frame inference driver: the part of lib/fcn/test.py that sits directly on the hot path
(`_get_image_blob` :37-110, `im_segment_single_frame` :113-239) plus the tiny host helpers it
calls (`lib/utils/nms.py:3-32`, `lib/utils/blob.py:48-71`), re-expressed for a batched,
device-resident pipeline:

  * `im_segment_single_frame` keeps the reference's signature/returns for one frame;
  * `im_segment_batch` runs B frames in one pass and keeps everything on the GPU — the Hough layer
    is called in its sync-free padded form, ROI pooling / fc6-8 run on the padded row capacity
    (rows past the count are all-zero ROIs and are masked out afterwards), NMS and the pose
    combine are done once per batch on the few hundred bytes of detections.
"""
import numpy as np
import torch

from . import ops
from .config import PIXEL_MEANS, make_meta_data

__all__ = ["pad_im", "unpad_im", "nms", "_get_image_blob", "im_segment_single_frame",
           "im_segment_batch", "combine_poses", "Detections"]


def pad_im(im, factor, value=0):
    """lib/utils/blob.py:48-59"""
    height, width = im.shape[0], im.shape[1]
    pad_height = int(np.ceil(height / float(factor)) * factor - height)
    pad_width = int(np.ceil(width / float(factor)) * factor - width)
    if im.ndim == 3:
        return np.pad(im, ((0, pad_height), (0, pad_width), (0, 0)), "constant", constant_values=value)
    return np.pad(im, ((0, pad_height), (0, pad_width)), "constant", constant_values=value)


def unpad_im(im, factor, orig_shape=None):
    """lib/utils/blob.py:62-71. The reference recomputes the pad from the *padded* size, which is
    always 0; callers there pass the original frame size implicitly. `orig_shape` makes it explicit."""
    if orig_shape is not None:
        return im[:orig_shape[0], :orig_shape[1]]
    height, width = im.shape[0], im.shape[1]
    pad_height = int(np.ceil(height / float(factor)) * factor - height)
    pad_width = int(np.ceil(width / float(factor)) * factor - width)
    return im[0:height - pad_height, 0:width - pad_width]


def nms(dets, thresh):
    """lib/utils/nms.py:3-32 — class-aware greedy NMS on 7-column ROIs
    (batch, cls, x1, y1, x2, y2, score): a box is suppressed only by a higher-scoring box of the
    same class with IoU > thresh (areas with +1). Returns kept row indices, best first.

    The reference walks the score order and recomputes the overlaps of the survivor against everything left — ~20 numpy
    calls per kept box, 126 us for the 5 detections of a frame, more than any kernel of the single-frame path but the
    trunk's. Here (round 5) the same float32 expressions are evaluated ONCE for all pairs — (a_i + a_j) - inter is the
    reference's `areas[i] + areas[order[1:]] - inter` pair by pair, float addition commutes — and the greedy walk runs on
    the resulting boolean matrix: same survivors, same order (`scores.argsort()[::-1]`, ties included)."""
    dets = np.asarray(dets)
    n = dets.shape[0]
    if n == 0:
        return []
    scores = dets[:, 6]
    order = scores.argsort()[::-1]
    if n == 1:
        return [int(order[0])]
    cls = dets[:, 1]
    x1, y1, x2, y2 = dets[:, 2], dets[:, 3], dets[:, 4], dets[:, 5]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    xx1 = np.maximum(x1[:, None], x1[None, :])
    yy1 = np.maximum(y1[:, None], y1[None, :])
    xx2 = np.minimum(x2[:, None], x2[None, :])
    yy2 = np.minimum(y2[:, None], y2[None, :])
    w = np.maximum(0.0, xx2 - xx1 + 1)
    h = np.maximum(0.0, yy2 - yy1 + 1)
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        ovr = inter / (areas[:, None] + areas[None, :] - inter)
    sup = (ovr > thresh) & (cls[:, None] == cls[None, :])
    if n <= 16:
        # a frame's handful of boxes: plain Python on nested lists beats n numpy calls of ~1 us each
        sup_l = sup.tolist()
        keep, dead = [], [False] * n
        for i in order.tolist():
            if dead[i]:
                continue
            keep.append(i)
            row = sup_l[i]
            for j in range(n):
                if row[j]:
                    dead[j] = True
        return keep
    # hundreds of rows (train mode, a gathered multi-rank set): one vector OR per KEPT box, nothing per pair in Python
    # (the nested-list walk was O(n^2) Python objects + O(n * kept) interpreter iterations)
    keep, dead = [], np.zeros(n, dtype=bool)
    for i in order.tolist():
        if dead[i]:
            continue
        keep.append(i)
        dead |= sup[i]
    return keep


def _get_image_blob(im, im_depth, scale=1.0):
    """lib/fcn/test.py:37-110 for INPUT in {COLOR, RGBD} and SCALES_BASE = (1.0,):
    BGR float32 minus PIXEL_MEANS; depth tower input = clip(depth/2000, 0, 1)*255 tiled to 3
    channels minus the same means. Returns (blob[1,H,W,3], blob_depth[1,H,W,3], im_scale)."""
    assert scale == 1.0, "only SCALES_BASE = (1.0,) is on the demo/test path (lov_color_2d.yml:39)"
    im_orig = im.astype(np.float32, copy=True)
    im_orig -= PIXEL_MEANS
    blob = im_orig[np.newaxis]
    blob_depth = None
    if im_depth is not None:
        d = im_depth.astype(np.float32, copy=True)
        d = np.clip(d / 2000.0, 0, 1) * 255
        d = np.tile(d[:, :, np.newaxis], (1, 1, 3))
        d -= PIXEL_MEANS
        blob_depth = d[np.newaxis]
    return blob, blob_depth, scale


def combine_poses(rois, poses_init, poses_pred):
    """lib/fcn/test.py:197-211: NMS(0.5), then poses[i,:4] = poses_tanh[i, 4c:4c+4] (raw tanh)."""
    keep = nms(rois, 0.5)
    rois = rois[keep, :]
    poses = poses_init[keep, :].copy()
    poses_pred = poses_pred[keep, :]
    for i in range(rois.shape[0]):
        class_id = int(rois[i, 1])
        if class_id >= 0:
            poses[i, :4] = poses_pred[i, 4 * class_id:4 * class_id + 4]
    return rois, poses, keep


def _feed(net, data, data_p, K, extents, points, symmetry, num_classes, device):
    B, H, W, _ = data.shape
    meta = np.stack([make_meta_data(K)] * B).reshape(B, 1, 1, 48)
    def t(a, dt=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(device=device, dtype=dt)
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device)

    feed = {
        "data": data if isinstance(data, torch.Tensor) else t(data),
        "gt_label_2d": torch.ones((B, H, W), dtype=torch.int32, device=device),  # fake label blob of ones (:154)
        "keep_prob": 1.0,
        "poses": torch.zeros((1, 13), dtype=torch.float32, device=device),       # pose_blob (:156)
        "extents": t(extents), "meta_data": t(meta), "points": t(points), "symmetry": t(symmetry),
    }
    if data_p is not None:
        feed["data_p"] = data_p if isinstance(data_p, torch.Tensor) else t(data_p)
    return feed


def im_segment_single_frame(net, im, im_depth, meta_data, extents, points, symmetry, num_classes, device="cuda"):
    """lib/fcn/test.py:113-239 (TEST.VERTEX_REG_2D and TEST.POSE_REG set, as in lov_color_2d.yml).
    `im` is BGR uint8 (OpenCV order), already padded to a multiple of 16 by the caller (:1870).
    Returns (labels_2d[H,W] int32, probs[H,W,C], vertex_pred[H,W,3C], rois[R,7], poses[R,7])."""
    blob, blob_depth, im_scale = _get_image_blob(im, im_depth)
    K = np.array(meta_data["intrinsic_matrix"], dtype=np.float64) * im_scale
    data_p = blob_depth if net.input_format == "RGBD" else None
    feed = _feed(net, blob, data_p, K, extents, points, symmetry, num_classes, torch.device(device))
    net.run(feed)
    g = lambda n: net.get_output(n).detach().cpu().numpy()
    labels_2d, probs, vertex_pred = g("label_2d"), g("prob_normalized"), g("vertex_pred")
    rois, poses_init, poses_pred = g("rois"), g("poses_init"), g("poses_tanh")
    rois, poses, _ = combine_poses(rois, poses_init, poses_pred)
    return labels_2d[0].astype(np.int32), probs[0], vertex_pred[0], rois, poses


class Detections(object):
    """Fixed-capacity per-batch detections kept on the device (what ranks exchange, §8e):
    rows[cap,14] = box7 | pose7 (quaternion already taken from poses_tanh), count[1]."""

    def __init__(self, rows, count, label_2d=None, packed=None):
        self.rows = rows
        self.count = count
        self.label_2d = label_2d
        self.packed = packed      # [cap + 1, 14]: the rows in global frame numbering + (count, 0, ...): what dist.all_gather_packed sends

    def to_host(self):
        n = int(self.count.item())
        r = self.rows[:n].cpu().numpy()
        return r[:, :7], r[:, 7:]


def im_segment_batch(net, data, K, extents, points, symmetry, data_p=None, planted=None, feed_cache=None,
                     with_losses=False, gt_poses=None, strict_reference=False, frame_offset=None):
    """B frames, one pass, no host synchronisation. `data` is the mean-subtracted BGR blob
    [B,H,W,3] already on the device. Returns `Detections` (device tensors; rows past count are 0).

    Capacity: the reference op keeps MAX_ROI / B maxima per image when handed a batch
    (hough_voting_gpu_op.cu.cc:733) — but its test loop feeds one frame at a time
    (lib/fcn/test.py:1867), so every frame gets all its classes. By default this batched driver
    therefore asks the Hough layer for C-1 maxima per image (threshold_vote <= 0: one per class;
    > 0: MAX_ROI) whatever B, i.e. the rows B single-frame calls return;
    `strict_reference=True` keeps the op's own rule (8 per frame at B = 16).

    `net.is_train` selects the Hough layer's training mode (9 jittered rows per maximum + pose
    targets for the `gt_poses` [N,13] rows, vgg16_convs.py:167-168); with `with_losses` the two loss
    layers of the graph (hard_label, average_distance_loss) are evaluated on those rows."""
    dev = data.device
    feed = feed_cache if feed_cache is not None else _feed(net, data, data_p, K, extents, points, symmetry, net.num_classes, dev)
    feed["data"] = data
    if data_p is not None:
        feed["data_p"] = data_p
    # run the graph up to vertex_pred, then the padded Hough + pooled head without a host sync
    saved = (net.vertex_reg_2d, net.fuse_hard_label)
    net.vertex_reg_2d = False
    net.fuse_hard_label = bool(with_losses)   # gt_label_weight rides in the label head's launch (Network._fused_hard_gt)
    try:
        net.run(feed, planted=planted)
    finally:
        net.vertex_reg_2d, net.fuse_hard_label = saved
    label_2d = net.get_output("label_2d")
    B = label_2d.shape[0]
    is_train = int(net.is_train)
    per_image = 0 if strict_reference else (net.num_classes - 1 if net.vote_threshold <= 0 else ops.MAX_ROI)
    if "vertex_pred_lowres" in net.layers:
        # fused heads: the Hough kernel interpolates the 1/8-resolution field itself; `vertex_pred`
        # stays a lazy layer that nobody fetches on this path
        top_box, top_pose, top_target, top_weight, top_domain, num_rois = ops.hough_voting_gpu_lowres_padded(
            label_2d, net.get_output("vertex_pred_lowres"), net.get_output("vertex_pred_bias"),
            int(16 * net.scale), int(8 * net.scale), feed["extents"], feed["meta_data"], gt_poses, is_train,
            net.vote_threshold, net.vote_percentage, net.skip_pixels, rois_per_image=per_image)
    else:
        top_box, top_pose, top_target, top_weight, top_domain, num_rois = ops.hough_voting_gpu_padded(
            label_2d, net.get_output("vertex_pred"), feed["extents"], feed["meta_data"], gt_poses, is_train,
            net.vote_threshold, net.vote_percentage, net.skip_pixels, rois_per_image=per_image)
    cap = top_box.shape[0]
    rois = top_box
    count = num_rois[1:2]
    # rows at or past the count are not written when fc6 runs on the library's row kernels, which mask them by the same
    # count (Network.fc); the framework fallback (a trainable graph under autograd) reads the whole buffer: zeros there
    masked_fc = net.fc_masks_dead_rows("fc6", 7 * 7 * 512, 4096)   # (the predicate Network.fc itself applies — ADVICE r4)
    pool = ops.roi_pool_add2(net.get_output("conv5_3"), 1.0 / 16.0, net.get_output("conv4_3"), 1.0 / 8.0, rois,
                             num_rows=count, dead_rows="keep" if masked_fc else "zero")
    net.layers["pool_score"] = pool
    net.rows_count = count   # fc6 / fc7 skip the rows past the device-side count
    try:
        (net.feed("pool_score")
            .fc(4096, height=7, width=7, channel=512, name="fc6")
            .fc(4096, num_in=4096, name="fc7")
            .fc_tanh(4 * net.num_classes, name="fc8", tanh_name="poses_tanh"))
    finally:
        net.rows_count = None
    poses_tanh = net.get_output("poses_tanh")
    # poses[i,:4] = poses_tanh[i, 4c:4c+4] (lib/fcn/test.py:206-211) and the detection rows box7 | quat4 | trans3, on
    # the device; training mode emits 9 rows per maximum (the box + 8 jitters, .cu.cc:440-466) — the detection
    # product is the un-jittered first row of each group
    packed = None
    if frame_offset is None:
        det_rows, det_count = ops.det_assemble(rois, poses_tanh, top_pose, count, row_stride=9 if is_train else 1)
    else:   # (a rank's first global frame index: the all-gather block comes out of the same launch)
        det_rows, det_count, packed = ops.det_assemble(rois, poses_tanh, top_pose, count, row_stride=9 if is_train else 1,
                                                       frame_offset=frame_offset)
    net.layers.update({"rois": rois, "poses_init": top_pose, "poses_tanh": poses_tanh,
                       "poses_target": top_target, "poses_weight": top_weight})
    if with_losses:
        # the two training-loss layers of the graph (vgg16_convs.py:148-149,195-200). TF prunes them
        # at test time; BASELINE config 2 lists them. Both run on the capacity-sized buffers with
        # the device-side row count, so the loss is normalised by the true number of rows.
        if "gt_label_weight" not in net.layers:   # (a with_losses graph has evaluated it in setup())
            net.layers["gt_label_weight"] = ops.hard_label(net.get_output("prob_normalized"), feed["gt_label_2d"],
                                                           net.threshold_label)
        if poses_tanh.is_cuda and poses_tanh.shape[1] <= 256 and not (torch.is_grad_enabled() and poses_tanh.requires_grad):
            pred = ops.pose_l2_normalize(poses_tanh, top_weight, num_rows=count)   # (one launch instead of six framework ops)
        else:
            mul = poses_tanh * top_weight
            pred = mul * torch.rsqrt(torch.clamp((mul * mul).sum(dim=1, keepdim=True), min=1e-12))
        net.layers["poses_pred"] = pred
        net.layers["loss_pose"] = ops.average_distance_loss(pred, top_target, top_weight, feed["points"],
                                                            feed["symmetry"], 0.01, num_rows=count)[0]
    return Detections(det_rows, det_count, label_2d, packed)


def finalize_batch(det_rows, count):
    """Host epilogue for a gathered batch: class-aware NMS per image (nms() compares classes, and
    boxes of different images never overlap a shared image index, so NMS is applied per image)."""
    rows = det_rows[:count]
    if rows.shape[0] == 0:
        return np.zeros((0, 7), np.float32), np.zeros((0, 7), np.float32)
    img = rows[:, 0]
    if img.min() == img.max():      # one frame (the single-frame loop, lib/fcn/test.py:1867-1888): no grouping pass
        keep = nms(rows[:, :7], 0.5)
        return rows[keep, :7], rows[keep, 7:]
    out_rois, out_poses = [], []
    for b in np.unique(img):
        r = rows[img == b]
        keep = nms(r[:, :7], 0.5)
        out_rois.append(r[keep, :7])
        out_poses.append(r[keep, 7:])
    return np.concatenate(out_rois), np.concatenate(out_poses)
