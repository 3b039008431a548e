"""YCB-Video ("LOV") dataset access and evaluation for the inference path.

What the reference does in `lib/datasets/lov.py`, re-expressed for this framework:

  * `load_object_extents` / `load_object_points`   the per-class 3-D extents and model point clouds the
    Hough layer and average_distance_loss consume (lov.py:141-170; data/LOV/extents.txt,
    data/LOV/models/<class>/points.xyz)
  * `YCBVideo`   the on-disk layout `<root>/data/<seq>/<frame>-{color,depth,label}.png` + `-meta.mat`
    indexed by an image-set file (`keyframe.txt`, `val.txt`, ...; lov.py:57-135), read with PIL / scipy
    (the reference uses OpenCV): frames come back as the BGR uint8 / uint16 arrays `fcn._get_image_blob`
    expects, metadata as the dict `im_segment_single_frame` takes.
  * `Evaluator`   `evaluate_result` + `evaluate_segmentations` (lov.py:397-680): confusion histogram ->
    per-class IoU / accuracies, and per ground-truth object the ADD (ADD-S for the three symmetric
    classes) of every detection of its class against the 0.1 x |extents| threshold, accumulated into
    per-class pose accuracy. Results are returned as data (the reference prints them); optional
    per-frame .mat dumps keep the reference's file format ('labels', 'rois', 'poses').

Host-side numpy; nothing here touches the GPU.
"""
import os

import numpy as np

from . import pose_error
from .config import LOV_CLASSES

# lov.py:465-468 / :605-608: the classes evaluated with ADD-S (adi)
ADI_CLASSES = ("024_bowl", "036_wood_block", "061_foam_brick")


def load_object_extents(extent_file, num_classes):
    """lov.py:161-170: rows 1.. of a [num_classes, 3] f32 array come from `extents.txt`, row 0 (background) is 0."""
    ext = np.zeros((num_classes, 3), dtype=np.float32)
    rows = np.loadtxt(extent_file, dtype=np.float64).reshape(-1, 3)
    if rows.shape[0] < num_classes - 1:
        raise ValueError("%s has %d rows, need %d" % (extent_file, rows.shape[0], num_classes - 1))
    ext[1:] = rows[:num_classes - 1]
    return ext


def load_object_points(models_dir, classes):
    """lov.py:141-158: `points.xyz` of every class; returns (list of [n_i, 3] arrays, points_all f32
    [num_classes, min_i n_i, 3] cut to the shortest model — what the `points` placeholder is fed)."""
    points = [np.zeros((0, 3))]
    for name in classes[1:]:
        path = os.path.join(models_dir, name, "points.xyz")
        if not os.path.exists(path):
            raise FileNotFoundError("Path does not exist: %s" % path)
        points.append(np.loadtxt(path, dtype=np.float64).reshape(-1, 3))
    num = min(p.shape[0] for p in points[1:])
    points_all = np.zeros((len(classes), num, 3), dtype=np.float32)
    for i in range(1, len(classes)):
        points_all[i] = points[i][:num]
    return points, points_all


def read_color_bgr(path):
    """An image file as OpenCV's imread returns it: uint8 [H, W, 3] in B, G, R order (alpha dropped:
    lib/fcn/test.py:1872-1876 zeroes RGB where alpha == 0, done here too)."""
    from PIL import Image
    im = np.asarray(Image.open(path))
    if im.ndim == 2:
        im = np.stack([im] * 3, axis=-1)
    if im.shape[2] == 4:
        rgb = im[:, :, :3].copy()
        rgb[im[:, :, 3] == 0] = 0
        im = rgb
    return np.ascontiguousarray(im[:, :, ::-1])


def read_depth(path):
    """uint16 depth image (`cv2.IMREAD_UNCHANGED` of a 16-bit PNG); metres = value / factor_depth."""
    from PIL import Image
    d = np.asarray(Image.open(path))
    return d.astype(np.uint16)


def read_label(path):
    """`-label.png`: the per-pixel class index image (lov.py:545-546)."""
    from PIL import Image
    return np.asarray(Image.open(path)).astype(np.int32)


class YCBVideo(object):
    """Frames of a YCB-Video tree by image-set index (lov.py:19-135)."""

    def __init__(self, root, image_set="keyframe", classes=LOV_CLASSES):
        self.root = root
        self.classes = tuple(classes)
        self.num_classes = len(self.classes)
        self.data_path = os.path.join(root, "data")
        set_file = os.path.join(root, image_set + ".txt")
        if not os.path.exists(set_file):
            raise FileNotFoundError("Path does not exist: %s" % set_file)
        with open(set_file) as f:
            self.image_index = [line.rstrip("\n") for line in f if line.strip()]
        self.extents = load_object_extents(os.path.join(root, "extents.txt"), self.num_classes)
        self._points = None

    def __len__(self):
        return len(self.image_index)

    @property
    def points(self):
        if self._points is None:
            self._points = load_object_points(os.path.join(self.root, "models"), self.classes)
        return self._points

    def path(self, i, what):
        ext = {"color": "-color.png", "depth": "-depth.png", "label": "-label.png", "meta": "-meta.mat"}[what]
        p = os.path.join(self.data_path, self.image_index[i] + ext)
        if not os.path.exists(p):
            raise FileNotFoundError("Path does not exist: %s" % p)
        return p

    def frame(self, i, with_label=True):
        """dict(color BGR uint8, depth uint16, label int32 | None, meta dict) of frame i. `meta` carries
        'intrinsic_matrix', 'factor_depth', 'poses' [3,4,n], 'cls_indexes' [n] as the -meta.mat does."""
        import scipy.io
        meta = scipy.io.loadmat(self.path(i, "meta"))
        meta = {k: v for k, v in meta.items() if not k.startswith("__")}
        if "cls_indexes" in meta:
            meta["cls_indexes"] = np.asarray(meta["cls_indexes"]).reshape(-1)
        return {"index": self.image_index[i], "color": read_color_bgr(self.path(i, "color")),
                "depth": read_depth(self.path(i, "depth")),
                "label": read_label(self.path(i, "label")) if with_label else None, "meta": meta}


def fast_hist(gt, pred, n):
    """Confusion histogram of two flat label arrays (lov.py `fast_hist`): rows = ground truth."""
    gt = np.asarray(gt).astype(np.int64).ravel()
    pred = np.asarray(pred).astype(np.int64).ravel()
    k = (gt >= 0) & (gt < n)
    return np.bincount(n * gt[k] + pred[k], minlength=n * n).reshape(n, n).astype(np.float64)


class Evaluator(object):
    """Accumulates what lov.py's evaluate_result / evaluate_segmentations compute over a frame stream."""

    def __init__(self, classes, extents, points, adi_classes=ADI_CLASSES):
        self.classes = tuple(classes)
        self.n = len(self.classes)
        self.extents = np.asarray(extents, dtype=np.float64)
        self.points = points                      # list / array indexed by class: [n_i, 3]
        self.adi = set(adi_classes)
        self.hist = np.zeros((self.n, self.n))
        self.count_all = np.zeros(self.n)
        self.count_correct = np.zeros(self.n)
        self.threshold = 0.1 * np.linalg.norm(self.extents, axis=1)   # lov.py:540-541
        self.frames = 0

    def pose_error(self, cls_index, quat_trans, RT_gt):
        """ADD (ADD-S for the symmetric classes) + rotation / translation error of one detection."""
        RT = np.zeros((3, 4))
        RT[:3, :3] = pose_error.quat2mat(quat_trans[:4])
        RT[:, 3] = quat_trans[4:7]
        pts = np.asarray(self.points[cls_index], dtype=np.float64)
        fn = pose_error.adi if self.classes[cls_index] in self.adi else pose_error.add
        return {"error": fn(RT[:3, :3], RT[:, 3], RT_gt[:3, :3], RT_gt[:, 3], pts),
                "rotation_error_deg": pose_error.re(RT[:3, :3], RT_gt[:3, :3]),
                "translation_error": pose_error.te(RT[:, 3], RT_gt[:, 3])}

    def evaluate_result(self, labels, rois, poses, gt_labels, meta_data, mat_path=None, poses_new=None, poses_icp=None):
        """One frame (lov.py:397-515): adds its confusion histogram and pose matches to the totals and
        returns {'iou': {class: v}, 'poses': [{class, error, threshold, correct, ...}]}. With the refined poses of
        cfg.TEST.POSE_REFINE (`poses_refined` / `poses_icp` of the segmentation record, lov.py:381-382, :463-511) every
        pose entry also carries error_new / error_icp (+ rotation / translation errors); the totals keep counting the
        network's pose, as the reference does (:517-520)."""
        h = fast_hist(gt_labels, labels, self.n)
        self.hist += h
        inter = np.diag(h)
        union = h.sum(1) + h.sum(0) - inter
        out = {"iou": {self.classes[i]: float(inter[i] / union[i]) for i in np.where(union > 0)[0]}, "poses": []}
        for tag, pp in (("poses_refined", poses_new), ("poses_icp", poses_icp)):   # before anything is written
            if pp is not None and np.asarray(pp).shape[0] != np.asarray(poses).shape[0]:
                raise ValueError("%s has %d rows for %d detections" % (tag, np.asarray(pp).shape[0], np.asarray(poses).shape[0]))
        if mat_path is not None:
            import scipy.io
            # the reference's record always carries both keys — empty lists when POSE_REFINE is off
            # (lib/fcn/test.py:1940, lov.py:389) — and downstream evaluation scripts read them
            empty = np.zeros((0, 7), np.float32)
            rec = {"labels": labels, "rois": rois, "poses": poses,
                   "poses_refined": empty if poses_new is None else poses_new,
                   "poses_icp": empty if poses_icp is None else poses_icp}
            scipy.io.savemat(mat_path, rec, do_compression=True)
        poses_gt = np.asarray(meta_data["poses"])
        if poses_gt.ndim == 2:
            poses_gt = poses_gt.reshape(3, 4, 1)
        cls_indexes = np.asarray(meta_data["cls_indexes"]).reshape(-1)
        for j in range(poses_gt.shape[2]):
            cj = int(cls_indexes[j])
            if cj <= 0:
                continue
            self.count_all[cj] += 1
            for k in range(rois.shape[0]):
                if int(rois[k, 1]) != cj:
                    continue
                e = self.pose_error(cj, poses[k], poses_gt[:, :, j])
                e.update({"class": self.classes[cj], "threshold": float(self.threshold[cj]),
                          "correct": bool(e["error"] < self.threshold[cj])})
                for tag, pp in (("new", poses_new), ("icp", poses_icp)):
                    if pp is not None:
                        e2 = self.pose_error(cj, pp[k], poses_gt[:, :, j])
                        e.update({k2 + "_" + tag: v for k2, v in e2.items()})
                if e["correct"]:
                    self.count_correct[cj] += 1
                out["poses"].append(e)
        self.frames += 1
        return out

    def summary(self):
        """evaluate_segmentations' closing numbers (lov.py:640-676)."""
        h = self.hist
        with np.errstate(divide="ignore", invalid="ignore"):
            iu = np.diag(h) / (h.sum(1) + h.sum(0) - np.diag(h))
            acc_cls = np.diag(h) / h.sum(1)
            freq = h.sum(1) / h.sum()
            pose_acc = self.count_correct / self.count_all
        return {"frames": self.frames,
                "overall_accuracy": float(np.diag(h).sum() / h.sum()) if h.sum() else float("nan"),
                "mean_accuracy": float(np.nanmean(acc_cls)) if h.sum() else float("nan"),
                "per_class_iu": {self.classes[i]: float(iu[i]) for i in range(self.n)},
                "mean_iu": float(np.nanmean(iu)) if h.sum() else float("nan"),
                "fwavacc": float((freq[freq > 0] * iu[freq > 0]).sum()) if h.sum() else float("nan"),
                "pose_accuracy": {self.classes[i]: (float(pose_acc[i]) if self.count_all[i] else None) for i in range(1, self.n)},
                "poses_correct": self.count_correct[1:].tolist(), "poses_all": self.count_all[1:].tolist(),
                "confusion_matrix": h}

    def write_reports(self, output_dir):
        """segmentation.txt + confusion_matrix.txt in the reference's formats (lov.py:654-664)."""
        os.makedirs(output_dir, exist_ok=True)
        s = self.summary()
        with open(os.path.join(output_dir, "segmentation.txt"), "wt") as f:
            for c in self.classes:
                f.write("{:f}\n".format(s["per_class_iu"][c]))
        with open(os.path.join(output_dir, "confusion_matrix.txt"), "wt") as f:
            for i in range(self.n):
                f.write(" ".join("{:f}".format(v) for v in self.hist[i]) + " \n")
        return s


def run_evaluation(net, dataset, points_all, symmetry, device="cuda", max_frames=None, evaluator=None, mat_dir=None,
                   synthesizer=None):
    """The evaluation loop of lib/fcn/test.py:1867-1945 (`test_net_single_frame`) without the
    visualisation branch: every frame of `dataset` -> pad to a multiple of 16 -> PoseCNN single
    frame inference -> un-padded labels, ROIs, poses -> [cfg.TEST.POSE_REFINE, :1896-1933: `synthesizer.icp_python` on the
    un-padded labels and the depth image -> poses_refined, poses_icp] -> `Evaluator.evaluate_result`. `synthesizer`: a
    `posecnn_amd.icp.Synthesizer` (None = POSE_REFINE off). Returns the evaluator."""
    from . import fcn
    if evaluator is None:
        evaluator = Evaluator(dataset.classes, dataset.extents, dataset.points[0])
    n = len(dataset) if max_frames is None else min(len(dataset), max_frames)
    for i in range(n):
        fr = dataset.frame(i)
        im = fcn.pad_im(fr["color"], 16)
        depth = fcn.pad_im(fr["depth"], 16)
        labels, probs, vertex_pred, rois, poses = fcn.im_segment_single_frame(
            net, im, depth, fr["meta"], dataset.extents, points_all, symmetry, dataset.num_classes, device=device)
        labels = fcn.unpad_im(labels, 16, orig_shape=fr["color"].shape[:2])
        mat = None if mat_dir is None else os.path.join(mat_dir, "%06d.mat" % i)
        poses_new = poses_icp = None
        if synthesizer is not None:                          # lib/fcn/test.py:1900-1933
            Km = np.asarray(fr["meta"]["intrinsic_matrix"], dtype=np.float64)
            parameters = np.array([Km[0, 0], Km[1, 1], Km[0, 2], Km[1, 2], 0.25, 6.0, float(np.asarray(fr["meta"]["factor_depth"]).reshape(-1)[0])],
                                  dtype=np.float32)
            poses_new = np.zeros((poses.shape[0], 7), dtype=np.float32)
            poses_icp = np.zeros((poses.shape[0], 7), dtype=np.float32)
            if rois.shape[0]:
                lab = np.ascontiguousarray(labels, dtype=np.int32)
                synthesizer.icp_python(lab, np.ascontiguousarray(fr["depth"], dtype=np.uint16), parameters, lab.shape[0], lab.shape[1],
                                       rois.shape[0], rois.shape[1], rois, poses, poses_new, poses_icp, 0.01)
        evaluator.evaluate_result(labels, rois, poses, fr["label"], fr["meta"], mat_path=mat, poses_new=poses_new, poses_icp=poses_icp)
    return evaluator
