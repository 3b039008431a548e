"""The per-frame front-end of `ros/listener.py` (`ImageListener`, :13-90) without ROS: SURVEY.md §8f-4 lists the ROS node
next to the ICP refinement; rospy / cv_bridge / message_filters do not exist here (and a transport is not part of the hot
path), so this keeps what the node DOES with a synchronised (colour, depth) pair and leaves the transport to two callables:

    ImageListener(net, imdb, meta_data, publish=..., publish_label=...).callback(rgb, depth, depth_encoding="16UC1")

`callback` = listener.py:41-90: depth message -> uint16 millimetre-style depth (`32FC1` metres * 1000, `16UC1` as is,
anything else is reported and dropped, :42-51), `im_segment_single_frame` on the frame, the `PoseCNNMsg` fields
(`ros/src/synthesizer/msg/PoseCNNMsg.msg`) as a dict, the colour-coded label image of `imdb.labels_to_image`
(lib/datasets/lov.py:348-364). With a `synthesizer` (posecnn_amd.icp.Synthesizer) the message also carries the refined
poses (`poses_refined`, `poses_icp`), which the reference's node leaves to the subscriber of `posecnn_result`.
"""
import numpy as np

from . import fcn

# lib/datasets/lov.py:32-35
LOV_CLASS_COLORS = [(255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255), (0, 255, 255),
                    (128, 0, 0), (0, 128, 0), (0, 0, 128), (128, 128, 0), (128, 0, 128), (0, 128, 128),
                    (64, 0, 0), (0, 64, 0), (0, 0, 64), (64, 64, 0), (64, 0, 64), (0, 64, 64),
                    (192, 0, 0), (0, 192, 0), (0, 0, 192)]


def labels_to_image(labels, class_colors=LOV_CLASS_COLORS):
    """`imdb.labels_to_image` (lov.py:348-364): uint8 [H,W,3], colour i where labels == i, black elsewhere."""
    lut = np.zeros((max(int(labels.max()) + 1 if labels.size else 1, len(class_colors)), 3), dtype=np.uint8)
    lut[:len(class_colors)] = np.asarray(class_colors, dtype=np.uint8)
    return lut[np.clip(labels, 0, None)] * (labels >= 0)[..., None].astype(np.uint8)


def depth_from_message(depth, encoding):
    """listener.py:42-51: `32FC1` (metres) -> uint16(depth * 1000); `16UC1` -> as is; None for anything else."""
    if encoding == "32FC1":
        return np.array(np.asarray(depth, dtype=np.float32) * 1000, dtype=np.uint16)
    if encoding == "16UC1":
        return np.ascontiguousarray(depth, dtype=np.uint16)
    return None


class ImageListener:
    """`imdb` needs `.extents`, `.points` (-> (points, points_all)), `.num_classes` (posecnn_amd.datasets.YCBVideo has
    them); `symmetry`: the imdb's symmetry vector (config.LOV_SYMMETRY). `publish(msg)` / `publish_label(image)` stand in
    for the two rospy publishers (`posecnn_result`, `posecnn_label`, listener.py:27-28)."""

    def __init__(self, net, imdb, meta_data, symmetry, publish=None, publish_label=None, synthesizer=None, device="cuda",
                 log=None):
        self.net, self.imdb, self.meta_data, self.symmetry = net, imdb, meta_data, symmetry
        self.publish, self.publish_label, self.synthesizer, self.device = publish, publish_label, synthesizer, device
        self.log = log if log is not None else (lambda s: None)
        self.count = 0

    def callback(self, rgb, depth, depth_encoding="16UC1"):
        """rgb: BGR uint8 [H,W,3] (`imgmsg_to_cv2(rgb, 'bgr8')`, :54); depth: the depth image as the message holds it.
        Returns the message dict (None when the depth encoding is not supported, like the node's early return)."""
        depth_cv = depth_from_message(depth, depth_encoding)
        if depth_cv is None:
            self.log("Unsupported depth type. Expected 16UC1 or 32FC1, got %s" % depth_encoding)
            return None
        im = np.ascontiguousarray(rgb, dtype=np.uint8)
        self.count += 1
        h, w = im.shape[:2]
        _, points_all = self.imdb.points
        labels, probs, vertex_pred, rois, poses = fcn.im_segment_single_frame(
            self.net, fcn.pad_im(im, 16), fcn.pad_im(depth_cv, 16), self.meta_data, self.imdb.extents, points_all, self.symmetry,
            self.imdb.num_classes, device=self.device)
        labels = fcn.unpad_im(labels, 16, orig_shape=(h, w))
        K = np.asarray(self.meta_data["intrinsic_matrix"], dtype=np.float64)
        factor = float(np.asarray(self.meta_data["factor_depth"]).reshape(-1)[0])
        msg = {"height": int(h), "width": int(w), "roi_num": int(rois.shape[0]), "roi_channel": int(rois.shape[1]) if rois.ndim == 2 else 0,
               "fx": float(K[0, 0]), "fy": float(K[1, 1]), "px": float(K[0, 2]), "py": float(K[1, 2]), "factor": factor,
               "znear": 0.25, "zfar": 6.0, "label": labels.astype(np.uint8), "depth": depth_cv,
               "rois": rois.astype(np.float32).flatten().tolist(), "poses": poses.astype(np.float32).flatten().tolist()}
        if self.synthesizer is not None and rois.shape[0]:
            par = np.array([msg["fx"], msg["fy"], msg["px"], msg["py"], msg["znear"], msg["zfar"], factor], dtype=np.float32)
            pn = np.zeros((rois.shape[0], 7), dtype=np.float32)
            pi = np.zeros((rois.shape[0], 7), dtype=np.float32)
            lab = np.ascontiguousarray(labels, dtype=np.int32)
            self.synthesizer.icp_python(lab, depth_cv, par, h, w, rois.shape[0], rois.shape[1], rois, poses, pn, pi, 0.01)
            msg["poses_refined"], msg["poses_icp"] = pn.flatten().tolist(), pi.flatten().tolist()
        if self.publish is not None:
            self.publish(msg)
        im_label = labels_to_image(labels)
        if self.publish_label is not None:
            self.publish_label(im_label)
        msg["label_image"] = im_label
        return msg
