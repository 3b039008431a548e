"""CPU restatement of the whole hot path: the `vgg16_convs` graph with PyTorch-CPU fp32 for the
dense layers and oracle/liboracle.so for the custom layers. TEST INFRASTRUCTURE ONLY — used by
the end-to-end parity tests and as bench.py's `cpu_baseline` ("port": the reference's TF1 graph
cannot run here)."""
import numpy as np
import torch

import oracle
from posecnn_amd.networks import layer, vgg16_convs


class vgg16_convs_cpu(vgg16_convs):
    """Same graph, same variables, custom layers routed to the oracle (numpy)."""

    def __init__(self, *args, **kw):
        kw["device"] = "cpu"
        vgg16_convs.__init__(self, *args, **kw)
        self.trainable = False  # a checker: its custom layers are numpy, nothing to differentiate

    def share_weights(self, other):
        self.vars = {k: v.detach().cpu() for k, v in other.vars.items()}

    def _bias_act(self, y, bias, relu):
        y = y + bias
        return torch.relu(y) if relu else y

    def _conv_first(self, x, w, bias, relu):
        y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1).permute(0, 2, 3, 1).contiguous()
        return self._bias_act(y, bias, relu)

    def _bias_relu_pool2(self, y, bias, relu):
        a = self._bias_act(y, bias, relu)
        return torch.nn.functional.max_pool2d(a.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()

    def _deconv_bilinear(self, x, k, s, add1=None, add2=None, bias=None, relu=False):
        n = lambda t: None if t is None else t.numpy()
        return torch.from_numpy(oracle.deconv_bilinear(x.numpy(), k, s, n(add1), n(add2), n(bias), relu))

    def _upscore_softmax_argmax(self, z, bias, k, s, relu=True, want_score=False, want_prob=True, hard_gt=None):
        assert hard_gt is None   # (the fused Hardlabel is a GPU-path choice: Network._fused_hard_gt)
        score, prob, label = oracle.upscore_softmax_argmax(z.numpy(), bias.numpy(), k, s, relu)
        return (torch.from_numpy(score) if want_score else None, torch.from_numpy(prob) if want_prob else None,
                torch.from_numpy(label))

    @layer
    def softmax_high_dimension(self, input, num_classes, name):
        p, l = oracle.softmax_argmax(input.numpy())
        self._argmax_cache = (None, torch.from_numpy(l))
        out = torch.from_numpy(p)
        self._argmax_cache = (out, torch.from_numpy(l))
        return out

    @layer
    def hough_voting_gpu(self, input, is_train, threshold, per_threshold, skip_pixels, name):
        gt = None if input[4] is None else input[4].numpy()
        out = oracle.hough_voting(input[0].numpy(), input[1].numpy(), input[2].numpy(), input[3].numpy(),
                                  gt, is_train, threshold, per_threshold, skip_pixels)
        return tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in out)

    @layer
    def hough_voting_gpu_lowres(self, input, kernel, stride, is_train, threshold, per_threshold, skip_pixels, name):
        # the checker takes the long way round: materialise vertex_pred, then the plain op
        vertex = oracle.deconv_bilinear(input[1].numpy(), kernel, stride, None, None, input[2].numpy(), False)
        gt = None if input[5] is None else input[5].numpy()
        out = oracle.hough_voting(input[0].numpy(), vertex, input[3].numpy(), input[4].numpy(),
                                  gt, is_train, threshold, per_threshold, skip_pixels)
        return tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in out)

    @layer
    def roi_pool(self, input, pooled_height, pooled_width, spatial_scale, pool_channel, name):
        t, a = oracle.roi_pool(input[0].numpy(), input[1].numpy(), pooled_height, pooled_width, spatial_scale, pool_channel)
        return torch.from_numpy(t), torch.from_numpy(a)

    @layer
    def hard_label(self, input, threshold, name):
        return torch.from_numpy(oracle.hard_label(input[0].numpy(), input[1].numpy(), threshold))

    @layer
    def average_distance_loss(self, input, margin, name):
        l, d = oracle.average_distance(*[i.numpy() for i in input], margin)
        return torch.from_numpy(l), torch.from_numpy(d)


def run_cpu_pipeline(net_cpu, data, K, extents, points, symmetry, planted=None, data_p=None, gt_poses=None):
    """B frames through the CPU graph; returns dict of numpy outputs (the fetch list of
    lib/fcn/test.py:193-195) + NMS'd rois/poses per lib/fcn/test.py:197-211. `data_p` feeds the
    depth tower of an RGBD network, `gt_poses` [N,13] the Hough layer's training mode."""
    from posecnn_amd import fcn
    from posecnn_amd.config import make_meta_data
    B, H, W, _ = data.shape
    meta = np.stack([make_meta_data(K)] * B).reshape(B, 1, 1, 48)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt)
    feed = {"data": t(data), "gt_label_2d": torch.ones((B, H, W), dtype=torch.int32), "keep_prob": 1.0,
            "poses": torch.zeros((1, 13)) if gt_poses is None else t(gt_poses), "extents": t(extents),
            "meta_data": t(meta), "points": t(points), "symmetry": t(symmetry)}
    if data_p is not None:
        feed["data_p"] = t(data_p)
    pl = None if planted is None else {k: t(v) for k, v in planted.items()}
    with torch.no_grad():
        net_cpu.run(feed, planted=pl)
    g = lambda n: net_cpu.get_output(n)
    out = {"label_2d": g("label_2d").numpy(), "vertex_pred": g("vertex_pred").numpy(),
           "rois": g("rois").numpy(), "poses_init": g("poses_init").numpy(), "poses_tanh": g("poses_tanh").numpy()}
    for name in ("poses_target", "poses_weight", "poses_pred"):
        if name in net_cpu.layers:
            out[name] = g(name).numpy()
    if "loss_pose" in net_cpu.layers:
        out["loss_pose"] = g("loss_pose")[0].numpy()
    rois_all, init_all, tanh_all = out["rois"], out["poses_init"], out["poses_tanh"]
    if net_cpu.is_train and rois_all.shape[0] % 9 == 0 and out["rois"][:, 6].any():
        # training mode: 9 rows per maximum (box + 8 jitters); the detection is the first of each group
        rois_all, init_all, tanh_all = rois_all[0::9], init_all[0::9], tanh_all[0::9]
    rois, poses = [], []
    for b in np.unique(rois_all[:, 0]):
        m = rois_all[:, 0] == b
        r, p, _ = fcn.combine_poses(rois_all[m], init_all[m], tanh_all[m])
        rois.append(r); poses.append(p)
    out["final_rois"] = np.concatenate(rois) if rois else np.zeros((0, 7), np.float32)
    out["final_poses"] = np.concatenate(poses) if poses else np.zeros((0, 7), np.float32)
    return out
