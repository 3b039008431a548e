"""Host-side operator surface of the PoseCNN custom layers on MI355X.

Names, argument order and attribute names mirror the reference's layer wrappers in
lib/networks/network.py (`hough_voting_gpu` :256-259, `roi_pool` :321-332, `hard_label` :338-340,
`average_distance_loss` :236-238, `backproject` :224-226) and the output tuples follow the
`REGISTER_OP` output order of each `*_op.cc`. Tensors are NHWC, float32 / int32, resident on the
GPU; every call enqueues hand-written gfx950 kernels from libposecnn_hip.so on torch's current
stream. There is no eager/PyTorch fallback: a missing library or a CPU tensor raises.
"""
import ctypes
from ctypes import c_size_t

import numpy as np

import torch

from . import _lib
from ._lib import HOUGH_ROWS_CAPACITY, MAX_ROI, POSE_CHANNELS, VERTEX_CHANNELS, check, lib

__all__ = [
    "hough_voting_gpu", "hough_voting_gpu_padded", "roi_pool", "roi_pool_add2", "hard_label", "conv1_1_conv1_2_fused",
    "conv1_1_conv1_2_fused_raw",
    "average_distance_loss", "backproject", "softmax_argmax", "deconv_bilinear", "bias_act_",
    "hough_voting_grad", "hard_label_grad", "hough_rows_capacity",
    "upscore_softmax_argmax", "Workspace",
]

INLIER_THRESHOLD = 0.9  # hough_voting_gpu_op.cc:356
LABEL_THRESHOLD = 500   # hough_voting_gpu_op.cc:357


def _dev(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s); posecnn_amd has no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _aligned16(t):
    """The float4 loads of the MFMA / reduction kernels need 16-byte aligned bases; a view that starts
    mid-allocation (e.g. one row of a packed bias table) is copied — the C-ABI itself rejects it (EINVAL)."""
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class Workspace:
    """Caller-owned scratch (the reference uses context->allocate_temp inside every launcher,
    e.g. hough_voting_gpu_op.cu.cc:633-640). Grows monotonically; one per stream."""

    def __init__(self):
        self._buf = None

    def get(self, nbytes, device):
        if self._buf is None or self._buf.numel() < nbytes or self._buf.device != device:
            self._buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self._buf


_default_ws = {}


def _ws(device, key):
    k = (device.index, key, torch.cuda.current_stream(device).cuda_stream)
    if k not in _default_ws:
        _default_ws[k] = Workspace()
    return _default_ws[k]


# ------------------------------------------------------------------------------------------------
def hough_rows_capacity(batch, is_train, rois_per_image=0):
    """Rows each Hough output needs: the reference's scratch size MAX_ROI * 9
    (hough_voting_gpu_op.cc:94) under its own capacity rule, batch * k * (9 | 1) for k maxima per image."""
    if not rois_per_image:
        return HOUGH_ROWS_CAPACITY
    return max(1, int(batch) * int(rois_per_image) * (9 if is_train else 1))


def _hough_common(label_2d, field, extents, meta_data, poses, threshold, skip_pixels, workspace, out, lowres,
                  is_train=0, rois_per_image=0):
    label_2d = _dev(label_2d, "label_2d", torch.int32)
    field = _dev(field, "vertex_pred", torch.float32)
    extents = _dev(extents, "extents", torch.float32)
    meta_data = _dev(meta_data, "meta_data", torch.float32)
    if label_2d.dim() != 3:
        raise ValueError("label must be 3-dimensional")  # hough_voting_gpu_op.cc:328-329
    if field.dim() != 4:
        raise ValueError("vertex must be 4-dimensional")  # :331-332
    B, H, W = label_2d.shape
    if lowres is None:
        want = (B, H, W)
    else:
        stride = int(lowres)
        if stride < 1 or H % stride or W % stride:
            raise ValueError("label map %dx%d is not a multiple of the stride %d" % (H, W, stride))
        want = (B, H // stride, W // stride)
    if tuple(field.shape[:3]) != want or field.shape[3] % VERTEX_CHANNELS:
        raise ValueError("vertex must be [B,%d,%d,3*num_classes] matching label" % want[1:])
    C = field.shape[3] // VERTEX_CHANNELS
    if extents.numel() != C * 3:
        raise ValueError("extents must be [num_classes,3]")
    num_meta = meta_data.shape[-1]
    if meta_data.numel() != B * num_meta:
        raise ValueError("meta_data must be [B,1,1,num_meta]")
    if poses is None or poses.numel() == 0:
        num_gt, gt = 0, None
    else:
        gt = _dev(poses, "poses", torch.float32)
        if gt.dim() != 2 or gt.shape[1] != 13:
            raise ValueError("poses (gt) must be [N,13]")
        num_gt = gt.shape[0]
    dev = label_2d.device
    nbytes = c_size_t(0)
    check("pcnn_hough_voting_workspace_bytes",
          lib().pcnn_hough_voting_workspace_bytes(B, H, W, C, float(threshold), int(skip_pixels),
                                                  int(rois_per_image), ctypes.byref(nbytes)))
    ws = (workspace or _ws(dev, "hough")).get(nbytes.value, dev)
    if out is None:
        cap = hough_rows_capacity(B, is_train, rois_per_image)
        out = (torch.empty((cap, 7), dtype=torch.float32, device=dev),
               torch.empty((cap, 7), dtype=torch.float32, device=dev),
               torch.empty((cap, POSE_CHANNELS * C), dtype=torch.float32, device=dev),
               torch.empty((cap, POSE_CHANNELS * C), dtype=torch.float32, device=dev),
               torch.empty((cap,), dtype=torch.int32, device=dev),
               torch.empty((2,), dtype=torch.int32, device=dev))
    return label_2d, field, extents, meta_data, gt, num_gt, (B, H, W, C, num_meta), ws, out


def hough_voting_gpu_padded(label_2d, vertex_pred, extents, meta_data, poses, is_train, threshold,
                            per_threshold, skip_pixels, workspace=None, out=None,
                            inlier_threshold=INLIER_THRESHOLD, label_threshold=LABEL_THRESHOLD,
                            rois_per_image=0):
    """Sync-free form: returns capacity-sized buffers and the device-side row counts.

    Returns (top_box[cap,7], top_pose[cap,7], top_target[cap,4C], top_weight[cap,4C],
    top_domain[cap] int32, num_rois[2] int32) where num_rois[0] is the number of rows the
    reference op returns (>= 1) and num_rois[1] the true detection row count; cap =
    `hough_rows_capacity(B, is_train, rois_per_image)` (1152 under the reference's capacity rule).
    rois_per_image: 0 = the reference's index_size = MAX_ROI / B per image; k > 0 = k maxima per
    image whatever B (include/posecnn_hip.h, "Capacity").
    """
    label_2d, vertex_pred, extents, meta_data, gt, num_gt, (B, H, W, C, num_meta), ws, out = _hough_common(
        label_2d, vertex_pred, extents, meta_data, poses, threshold, skip_pixels, workspace, out, None,
        is_train, rois_per_image)
    top_box, top_pose, top_target, top_weight, top_domain, num_rois = out
    check("pcnn_hough_voting_fwd",
          lib().pcnn_hough_voting_fwd(_ptr(label_2d), _ptr(vertex_pred), _ptr(extents), _ptr(meta_data), _ptr(gt),
                                      B, H, W, C, num_meta, num_gt,
                                      int(is_train), float(threshold), float(per_threshold), int(skip_pixels),
                                      float(inlier_threshold), int(label_threshold),
                                      int(rois_per_image), int(top_box.shape[0]),
                                      _ptr(top_box), _ptr(top_pose), _ptr(top_target), _ptr(top_weight),
                                      _ptr(top_domain), _ptr(num_rois),
                                      _ptr(ws), ws.numel(), _stream(label_2d)))
    return out


def hough_voting_gpu_lowres_padded(label_2d, z, bias, kernel, stride, extents, meta_data, poses, is_train,
                                   threshold, per_threshold, skip_pixels, workspace=None, out=None,
                                   inlier_threshold=INLIER_THRESHOLD, label_threshold=LABEL_THRESHOLD,
                                   rois_per_image=0):
    """Fused vertex head -> Hough voting: identical results to
    `hough_voting_gpu_padded(label_2d, deconv_bilinear(z, kernel, stride, bias=bias), ...)` without
    ever building the [B,H,W,3C] `vertex_pred` (vgg16_convs.py:152-163).
    z is the 1x1 `vertex_pred` conv evaluated at 1/stride resolution, [B,H/stride,W/stride,3C]."""
    label_2d, z, extents, meta_data, gt, num_gt, (B, H, W, C, num_meta), ws, out = _hough_common(
        label_2d, z, extents, meta_data, poses, threshold, skip_pixels, workspace, out, stride,
        is_train, rois_per_image)
    bias = _dev(bias, "bias", torch.float32)
    if bias.numel() != VERTEX_CHANNELS * C:
        raise ValueError("bias must be [3*num_classes]")
    top_box, top_pose, top_target, top_weight, top_domain, num_rois = out
    check("pcnn_hough_voting_lowres_fwd",
          lib().pcnn_hough_voting_lowres_fwd(_ptr(label_2d), _ptr(z), _ptr(bias), int(kernel), int(stride),
                                             _ptr(extents), _ptr(meta_data), _ptr(gt),
                                             B, H, W, C, num_meta, num_gt,
                                             int(is_train), float(threshold), float(per_threshold), int(skip_pixels),
                                             float(inlier_threshold), int(label_threshold),
                                             int(rois_per_image), int(top_box.shape[0]),
                                             _ptr(top_box), _ptr(top_pose), _ptr(top_target), _ptr(top_weight),
                                             _ptr(top_domain), _ptr(num_rois),
                                             _ptr(ws), ws.numel(), _stream(label_2d)))
    return out


def hough_voting_grad(label_2d, vertex_pred):
    """HoughvotinggpuGrad (hough_voting_gpu_op.cc:440-484 -> set_gradients, .cu.cc:608-612): the op
    is not differentiable; its registered gradient hands zeros to both inputs. Returns
    (grad_label f32 [B,H,W], grad_vertex f32 [B,H,W,3C])."""
    B, H, W = label_2d.shape
    C = vertex_pred.shape[3] // VERTEX_CHANNELS
    dev = vertex_pred.device
    gl = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    gv = torch.empty((B, H, W, VERTEX_CHANNELS * C), dtype=torch.float32, device=dev)
    check("pcnn_hough_voting_bwd", lib().pcnn_hough_voting_bwd(_ptr(gl), _ptr(gv), B, H, W, C, _stream(gv)))
    return gl, gv


class _HoughVotingFn(torch.autograd.Function):
    """The op inside an autograd graph (training graph, vgg16_convs.py:167-168): outputs carry no
    gradient information; backward = the registered zero gradient of the reference."""

    @staticmethod
    def forward(ctx, label_2d, vertex_pred, extents, meta_data, poses, cfg):
        is_train, threshold, per_threshold, skip_pixels, workspace, consts = cfg
        out = hough_voting_gpu_padded(label_2d, vertex_pred, extents, meta_data, poses, is_train, threshold,
                                      per_threshold, skip_pixels, workspace=workspace, **consts)
        ctx.save_for_backward(label_2d, vertex_pred)
        ctx.mark_non_differentiable(out[4], out[5])
        return out

    @staticmethod
    def backward(ctx, *_grads):
        label_2d, vertex_pred = ctx.saved_tensors
        _, gv = hough_voting_grad(label_2d, vertex_pred)
        return None, gv, None, None, None, None   # label is int32: no gradient slot


def hough_voting_gpu(label_2d, vertex_pred, extents, meta_data, poses, is_train, threshold,
                     per_threshold, skip_pixels, name=None, workspace=None, **consts):
    """Drop-in for `Network.hough_voting_gpu` (network.py:256-259): returns exactly-sized
    (top_box[R,7], top_pose[R,7], top_target[R,4C], top_weight[R,4C], top_domain[R]).
    Like the reference (hough_voting_gpu_op.cc:379-383) this reads the row count back to the host.
    """
    if torch.is_grad_enabled() and isinstance(vertex_pred, torch.Tensor) and vertex_pred.requires_grad:
        top_box, top_pose, top_target, top_weight, top_domain, num_rois = _HoughVotingFn.apply(
            label_2d, vertex_pred, extents, meta_data, poses,
            (is_train, threshold, per_threshold, skip_pixels, workspace, consts))
    else:
        top_box, top_pose, top_target, top_weight, top_domain, num_rois = hough_voting_gpu_padded(
            label_2d, vertex_pred, extents, meta_data, poses, is_train, threshold, per_threshold,
            skip_pixels, workspace=workspace, **consts)
    r = int(num_rois[0].item())
    return top_box[:r], top_pose[:r], top_target[:r], top_weight[:r], top_domain[:r]


# ------------------------------------------------------------------------------------------------
class _RoiPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, rois, pooled_height, pooled_width, spatial_scale, pool_channel):
        B, H, W, C = data.shape
        R, cols = rois.shape
        Cout = 1 if pool_channel else C
        top = torch.empty((R, pooled_height, pooled_width, Cout), dtype=torch.float32, device=data.device)
        argmax = torch.empty((R, pooled_height, pooled_width, Cout), dtype=torch.int32, device=data.device)
        check("pcnn_roi_pool_fwd",
              lib().pcnn_roi_pool_fwd(_ptr(data), _ptr(rois), B, H, W, C, R, cols, int(pooled_height),
                                      int(pooled_width), float(spatial_scale), int(pool_channel),
                                      _ptr(top), _ptr(argmax), _stream(data)))
        ctx.save_for_backward(rois, argmax)
        ctx.cfg = (B, H, W, C, R, cols, int(pooled_height), int(pooled_width), float(spatial_scale), int(pool_channel))
        ctx.mark_non_differentiable(argmax)
        return top, argmax

    @staticmethod
    def backward(ctx, grad_top, _grad_argmax):
        rois, argmax = ctx.saved_tensors
        B, H, W, C, R, cols, PH, PW, scale, pc = ctx.cfg
        grad_top = grad_top.contiguous()
        bottom = torch.empty((B, H, W, C), dtype=torch.float32, device=grad_top.device)
        check("pcnn_roi_pool_bwd",
              lib().pcnn_roi_pool_bwd(_ptr(grad_top), _ptr(rois), _ptr(argmax), B, H, W, C, R, cols, PH, PW,
                                      scale, pc, _ptr(bottom), _stream(grad_top)))
        return bottom, None, None, None, None, None


def roi_pool(data, rois, pooled_height, pooled_width, spatial_scale, pool_channel, name=None):
    """Drop-in for `Network.roi_pool` (network.py:321-332): NHWC max pooling of 7-column ROIs.
    Returns (top_data[R,PH,PW,C or 1], argmax int32)."""
    data = _dev(data, "data", torch.float32)
    rois = _dev(rois, "rois", torch.float32)
    if data.dim() != 4:
        raise ValueError("data must be 4-dimensional")  # roi_pooling_op.cc:313-314
    if rois.dim() != 2:
        raise ValueError("rois must be 2-dimensional")  # :317-318
    return _RoiPoolFn.apply(data, rois, pooled_height, pooled_width, spatial_scale, pool_channel)


def roi_pool_add2(data_a, scale_a, data_b, scale_b, rois, pooled_height=7, pooled_width=7, num_rows=None, dead_rows="zero", out=None):
    """Fused `pool_score` = roi_pool(conv5_3, 1/16) + roi_pool(conv4_3, 1/8)
    (vgg16_convs.py:177-187), inference only (no argmax). `num_rows` (device int32[1]): true row
    count of a capacity-sized `rois` buffer; rows past it pool to zero — or, with dead_rows="keep", are not
    written at all (for consumers that mask them by the same count: ops.fc_rows / ops.fc_skinny)."""
    data_a = _dev(data_a, "data_a", torch.float32)
    data_b = _dev(data_b, "data_b", torch.float32)
    rois = _dev(rois, "rois", torch.float32)
    B, Ha, Wa, C = data_a.shape
    Bb, Hb, Wb, Cb = data_b.shape
    if (B, C) != (Bb, Cb):
        raise ValueError("feature maps must share batch and channel dimensions")
    if dead_rows not in ("zero", "keep"):
        raise ValueError("dead_rows must be 'zero' or 'keep'")
    if dead_rows == "keep" and num_rows is None:
        raise ValueError("dead_rows='keep' needs num_rows")
    R, cols = rois.shape
    if out is None:
        out = torch.empty((R, pooled_height, pooled_width, C), dtype=torch.float32, device=data_a.device)
    else:
        out = _dev(out, "out", torch.float32)
        if tuple(out.shape) != (R, pooled_height, pooled_width, C):
            raise ValueError("out must be [R, PH, PW, C]")
    name = "pcnn_roi_pool_add2_live_fwd" if dead_rows == "keep" else "pcnn_roi_pool_add2_fwd"
    check(name,
          getattr(lib(), name)(_ptr(data_a), Ha, Wa, float(scale_a), _ptr(data_b), Hb, Wb, float(scale_b),
                               _ptr(rois), B, C, R, cols, int(pooled_height), int(pooled_width),
                               _ptr(_dev(num_rows, "num_rows", torch.int32) if num_rows is not None else None),
                               _ptr(out), _stream(data_a)))
    return out


# ------------------------------------------------------------------------------------------------
def _hard_label_raw(prob, gt_label, threshold):
    C = prob.shape[-1]
    N = prob.numel() // C
    out = torch.empty_like(prob)
    check("pcnn_hard_label_fwd",
          lib().pcnn_hard_label_fwd(_ptr(prob), _ptr(gt_label), N, C, float(threshold), _ptr(out), _stream(prob)))
    return out


def hard_label_grad(prob, gt_label):
    """HardlabelGrad (hard_label_op_gpu.cu.cc:55-85): zeros for both inputs.
    Returns (grad_prob f32 like prob, grad_gt f32 [N])."""
    C = prob.shape[-1]
    N = prob.numel() // C
    gp = torch.empty_like(prob)
    gg = torch.empty(gt_label.shape, dtype=torch.float32, device=prob.device)
    check("pcnn_hard_label_bwd", lib().pcnn_hard_label_bwd(_ptr(gp), _ptr(gg), N, C, _stream(prob)))
    return gp, gg


class _HardLabelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, gt_label, threshold):
        ctx.save_for_backward(prob, gt_label)
        return _hard_label_raw(prob, gt_label, threshold)

    @staticmethod
    def backward(ctx, _grad):
        prob, gt_label = ctx.saved_tensors
        return hard_label_grad(prob, gt_label)[0], None, None


def hard_label(prob, gt_label, threshold, name=None):
    """Drop-in for `Network.hard_label` (network.py:338-340); GPU-kernel semantics
    (hard_label_op_gpu.cu.cc:17-29). prob [B,H,W,C] f32, gt_label [B,H,W] int32 -> [B,H,W,C].
    Inside an autograd graph the registered zero gradient (HardlabelGrad) is what flows back."""
    prob = _dev(prob, "prob", torch.float32)
    gt_label = _dev(gt_label, "gt_label", torch.int32)
    C = prob.shape[-1]
    N = prob.numel() // C
    if gt_label.numel() != N:
        raise ValueError("gt_label must have one entry per pixel of prob")
    if torch.is_grad_enabled() and prob.requires_grad:
        return _HardLabelFn.apply(prob, gt_label, float(threshold))
    return _hard_label_raw(prob, gt_label, threshold)


def softmax_argmax(score, want_prob=True):
    """softmax_high_dimension + argmax_2d (network.py:474-488, 432-434) in one pass.
    score [..., C] f32 -> (prob_normalized [..., C] or None, label_2d [...] int32)."""
    score = _dev(score, "score", torch.float32)
    C = score.shape[-1]
    N = score.numel() // C
    prob = torch.empty_like(score) if want_prob else None
    label = torch.empty(score.shape[:-1], dtype=torch.int32, device=score.device)
    check("pcnn_softmax_argmax_fwd",
          lib().pcnn_softmax_argmax_fwd(_ptr(score), N, C, _ptr(prob), _ptr(label), _stream(score)))
    return prob, label


def bias_act_(x, bias, relu=True):
    """In place: x[..., c] = [ReLU](x[..., c] + bias[c]) for an NHWC-contiguous tensor."""
    x = _dev(x, "x", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    C = x.shape[-1]
    if bias.numel() != C:
        raise ValueError("bias must have one entry per channel")
    check("pcnn_bias_act_fwd",
          lib().pcnn_bias_act_fwd(_ptr(x), _ptr(bias), x.numel() // C, C, 1 if relu else 0, _ptr(x), _stream(x)))
    return x


def conv3x3_c3(x, weights, bias, relu=True):
    """[ReLU](conv3x3_SAME(x) + bias) for a 3-channel NHWC input: x [B,H,W,3], weights [3,3,3,Cout]
    laid out (ky, kx, ci, co) like the TF variable, Cout a multiple of 64. Returns [B,H,W,Cout]."""
    x = _dev(x, "x", torch.float32)
    weights = _dev(weights, "weights", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    if x.dim() != 4 or x.shape[3] != 3:
        raise ValueError("x must be [B,H,W,3]")
    B, H, W, _ = x.shape
    Cout = weights.shape[-1]
    if tuple(weights.shape) != (3, 3, 3, Cout) or bias.numel() != Cout:
        raise ValueError("weights must be [3,3,3,Cout] (ky,kx,ci,co) and bias [Cout]")
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    check("pcnn_conv3x3_c3_fwd",
          lib().pcnn_conv3x3_c3_fwd(_ptr(x), _ptr(weights), _ptr(bias), B, H, W, Cout, 1 if relu else 0, _ptr(y), _stream(x)))
    return y


def conv3x3_c3_winograd43(x, weights, bias, relu=True, groups=1):
    """winograd_input(conv3x3_c3(x, weights, bias, relu), tile=4) in one kernel: V [36, T, Cout].
    groups > 1: weights [groups,3,3,3,Cout], bias [groups,Cout]; image b uses set b // (B // groups)."""
    x = _dev(x, "x", torch.float32)
    weights = _dev(weights, "weights", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    if x.dim() != 4 or x.shape[3] != 3:
        raise ValueError("x must be [B,H,W,3]")
    B, H, W, _ = x.shape
    Cout = weights.shape[-1]
    if weights.numel() != groups * 27 * Cout or tuple(weights.shape[-4:]) != (3, 3, 3, Cout) or bias.numel() != groups * Cout:
        raise ValueError("weights must be [groups,3,3,3,Cout] (ky,kx,ci,co) and bias [groups,Cout]")
    v = torch.empty((36, B * ((H + 3) // 4) * ((W + 3) // 4), Cout), dtype=torch.float32, device=x.device)
    check("pcnn_conv3x3_c3_winograd43_fwd",
          lib().pcnn_conv3x3_c3_winograd43_fwd(_ptr(x), _ptr(weights), _ptr(bias), B, H, W, Cout, int(groups),
                                               1 if relu else 0, _ptr(v), _stream(x)))
    return v


def conv3x3_c3_winograd43_raw(color_bgr, depth, weights, bias, relu=True, pixel_means=None):
    """conv3x3_c3_winograd43 on the frames as the sensor delivers them: color_bgr uint8 [B,H,W,3] (or None), depth uint16
    [B,H,W] (or None); the blobs of lib/fcn/test.py:56-74 are formed inside the kernel, bit for bit. Colour frames first
    (filter set 0), depth frames after (next set): weights [sets,3,3,3,Cout], bias [sets,Cout]. Returns V [36, T, Cout]."""
    from .config import PIXEL_MEANS
    c = _dev(color_bgr, "color_bgr", torch.uint8) if color_bgr is not None else None
    d = _dev(depth, "depth", torch.uint16) if depth is not None else None
    if c is None and d is None:
        raise ValueError("need colour and / or depth frames")
    if c is not None and (c.dim() != 4 or c.shape[3] != 3):
        raise ValueError("color_bgr must be uint8 [B,H,W,3]")
    if d is not None and d.dim() == 4 and d.shape[3] == 1:
        d = d.reshape(d.shape[:3])
    if d is not None and d.dim() != 3:
        raise ValueError("depth must be uint16 [B,H,W]")
    H, W = (c.shape[1], c.shape[2]) if c is not None else (d.shape[1], d.shape[2])
    if c is not None and d is not None and tuple(d.shape[1:]) != (H, W):
        raise ValueError("colour and depth frames must have the same size")
    nc, nd = (0 if c is None else c.shape[0]), (0 if d is None else d.shape[0])
    sets = (nc > 0) + (nd > 0)
    dev = (c if c is not None else d).device
    weights = _dev(weights, "weights", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    Cout = weights.shape[-1]
    if weights.numel() != sets * 27 * Cout or tuple(weights.shape[-4:]) != (3, 3, 3, Cout) or bias.numel() != sets * Cout:
        raise ValueError("weights must be [sets,3,3,3,Cout] (ky,kx,ci,co) and bias [sets,Cout]")
    means = (ctypes.c_double * 3)(*[float(x) for x in np.asarray(PIXEL_MEANS if pixel_means is None else pixel_means, dtype=np.float64).reshape(-1)[:3]])
    v = torch.empty((36, (nc + nd) * ((H + 3) // 4) * ((W + 3) // 4), Cout), dtype=torch.float32, device=dev)
    check("pcnn_conv3x3_c3_winograd43_raw_fwd",
          lib().pcnn_conv3x3_c3_winograd43_raw_fwd(_ptr(c), nc, _ptr(d), nd, means, _ptr(weights), _ptr(bias), H, W, Cout,
                                                   1 if relu else 0, _ptr(v), _stream(v)))
    return v


def conv12_fragment_major(ut2):
    """U^T [..., 36, 64, 64] (plane, out channel, in channel) -> the fragment-major layout of pcnn_conv1_1_conv1_2_fused_fwd's
    ut2_layout = 1: [..., 36, 4, 4, 64, 4] with (k, w, g, lane, i) = U^T[k][16 w + (lane & 15)][16 g + 4 (lane >> 4) + i]."""
    lead = ut2.shape[:-3]
    t = ut2.reshape(*lead, 36, 4, 16, 4, 4, 4)          # k, w, lr, g, lk, i
    t = t.permute(*range(len(lead)), len(lead), len(lead) + 1, len(lead) + 3, len(lead) + 4, len(lead) + 2, len(lead) + 5)   # k, w, g, lk, lr, i
    return t.reshape(*lead, 36, 4, 4, 64, 4).contiguous()


def conv1_1_conv1_2_fused(x, w1, b1, ut2, b2, relu1=True, relu2=True, groups=1, ut2_layout=0):
    """max_pool_2x2(relu(conv1_2(relu(conv1_1(x))))) in one kernel (csrc/conv_first.hip): x [B,H,W,3] f32 blobs, H and W multiples
    of 16; w1 [groups,3,3,3,64], b1 [groups,64]; ut2 [groups,36,64,64] = winograd_filter(w2, 4).transpose(1, 2) per set;
    b2 [groups,64]. Bit-identical to conv3x3_c3_winograd43 + winograd43_conv(pool=1). Returns [B,H/2,W/2,64]."""
    x = _dev(x, "x", torch.float32)
    w1, b1, ut2, b2 = (_dev(t, n, torch.float32) for t, n in ((w1, "w1"), (b1, "b1"), (ut2, "ut2"), (b2, "b2")))
    if x.dim() != 4 or x.shape[3] != 3:
        raise ValueError("x must be [B,H,W,3]")
    B, H, W, _ = x.shape
    if w1.numel() != groups * 27 * 64 or b1.numel() != groups * 64 or ut2.numel() != groups * 36 * 64 * 64 or b2.numel() != groups * 64:
        raise ValueError("w1 [groups,3,3,3,64], b1 [groups,64], ut2 [groups,36,64,64], b2 [groups,64]")
    y = torch.empty((B, H // 2, W // 2, 64), dtype=torch.float32, device=x.device)
    check("pcnn_conv1_1_conv1_2_fused_fwd",
          lib().pcnn_conv1_1_conv1_2_fused_fwd(_ptr(x), _ptr(w1), _ptr(b1), _ptr(ut2), int(ut2_layout), _ptr(b2), B, H, W, int(groups),
                                               1 if relu1 else 0, 1 if relu2 else 0, _ptr(y), _stream(x)))
    return y


def conv1_1_conv1_2_fused_raw(color_bgr, depth, w1, b1, ut2, b2, relu1=True, relu2=True, pixel_means=None, ut2_layout=0):
    """conv1_1_conv1_2_fused on the frames as the sensor delivers them (see conv3x3_c3_winograd43_raw): colour frames first
    (filter set 0), depth frames after (next set)."""
    from .config import PIXEL_MEANS
    c = _dev(color_bgr, "color_bgr", torch.uint8) if color_bgr is not None else None
    d = _dev(depth, "depth", torch.uint16) if depth is not None else None
    if c is None and d is None:
        raise ValueError("need colour and / or depth frames")
    if c is not None and (c.dim() != 4 or c.shape[3] != 3):
        raise ValueError("color_bgr must be uint8 [B,H,W,3]")
    if d is not None and d.dim() == 4 and d.shape[3] == 1:
        d = d.reshape(d.shape[:3])
    if d is not None and d.dim() != 3:
        raise ValueError("depth must be uint16 [B,H,W]")
    H, W = (c.shape[1], c.shape[2]) if c is not None else (d.shape[1], d.shape[2])
    if c is not None and d is not None and tuple(d.shape[1:]) != (H, W):
        raise ValueError("colour and depth frames must have the same size")
    nc, nd = (0 if c is None else c.shape[0]), (0 if d is None else d.shape[0])
    sets = (nc > 0) + (nd > 0)
    dev = (c if c is not None else d).device
    w1, b1, ut2, b2 = (_dev(t, n, torch.float32) for t, n in ((w1, "w1"), (b1, "b1"), (ut2, "ut2"), (b2, "b2")))
    if w1.numel() != sets * 27 * 64 or b1.numel() != sets * 64 or ut2.numel() != sets * 36 * 64 * 64 or b2.numel() != sets * 64:
        raise ValueError("w1 [sets,3,3,3,64], b1 [sets,64], ut2 [sets,36,64,64], b2 [sets,64]")
    means = (ctypes.c_double * 3)(*[float(v) for v in np.asarray(PIXEL_MEANS if pixel_means is None else pixel_means, dtype=np.float64).reshape(-1)[:3]])
    y = torch.empty((nc + nd, H // 2, W // 2, 64), dtype=torch.float32, device=dev)
    check("pcnn_conv1_1_conv1_2_fused_raw_fwd",
          lib().pcnn_conv1_1_conv1_2_fused_raw_fwd(_ptr(c), nc, _ptr(d), nd, means, _ptr(w1), _ptr(b1), _ptr(ut2), int(ut2_layout),
                                                   _ptr(b2), H, W, 1 if relu1 else 0, 1 if relu2 else 0, _ptr(y), _stream(y)))
    return y


_WINO_G = {
    2: [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
    4: [[1 / 4, 0.0, 0.0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
        [1 / 24, -1 / 12, 1 / 6], [0.0, 0.0, 1.0]],
}


def _wino_tiles(B, H, W, tile):
    if tile == 2:
        return B * (H // 2) * (W // 2)
    return B * ((H + 3) // 4) * ((W + 3) // 4)


def winograd_filter(weights, tile=2):
    """torch filter [Cout, Cin, 3, 3] -> U f32 [(tile+2)^2, Cin, Cout], U[n*i+j] = (G g G^T)[i][j]
    (float64 arithmetic, one rounding). Host-side plumbing, done once per filter."""
    g = weights.detach().to(torch.float64)
    G = torch.tensor(_WINO_G[tile], dtype=torch.float64, device=g.device)
    u = torch.einsum("ir,ocrs,js->ijco", G, g, G)   # [n,n,Cin,Cout]
    n = tile + 2
    return u.reshape(n * n, g.shape[1], g.shape[0]).to(torch.float32).contiguous()


def winograd_input(x, tile=2):
    """x [B,H,W,C] -> V [(tile+2)^2, T, C]: the input transform B^T d B of every input patch
    (tile = 2: F(2x2,3x3), T = B*H/2*W/2; tile = 4: F(4x4,3x3), T = B*ceil(H/4)*ceil(W/4))."""
    x = _dev(x, "x", torch.float32)
    B, H, W, C = x.shape
    n = tile + 2
    v = torch.empty((n * n, _wino_tiles(B, H, W, tile), C), dtype=torch.float32, device=x.device)
    fn = "pcnn_winograd_input_fwd" if tile == 2 else "pcnn_winograd43_input_fwd"
    check(fn, getattr(lib(), fn)(_ptr(x), B, H, W, C, _ptr(v), _stream(x)))
    return v


def winograd_output(m, bias, B, H, W, relu=True, pool=False, tile=2):
    """M [(tile+2)^2, T, C] -> [ReLU](A^T M A + bias) as [B,H,W,C], or its 2x2 max-pool [B,H/2,W/2,C]."""
    m = _dev(m, "m", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    C = m.shape[2]
    n = tile + 2
    if m.shape[0] != n * n or m.shape[1] != _wino_tiles(B, H, W, tile) or bias.numel() != C:
        raise ValueError("m must be [%d, tiles, C] and bias [C]" % (n * n))
    shape = (B, H // 2, W // 2, C) if pool else (B, H, W, C)
    y = torch.empty(shape, dtype=torch.float32, device=m.device)
    fn = "pcnn_winograd_output_fwd" if tile == 2 else "pcnn_winograd43_output_fwd"
    check(fn, getattr(lib(), fn)(_ptr(m), _ptr(bias), B, H, W, C, 1 if relu else 0, 1 if pool else 0, _ptr(y), _stream(m)))
    return y


def winograd43_output_both(m, bias, B, H, W, relu=True):
    """F(4x4,3x3) output transform returning (y [B,H,W,C], max_pool_2x2(y) [B,H/2,W/2,C]) from one pass."""
    m = _dev(m, "m", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    C = m.shape[2]
    if m.shape[0] != 36 or m.shape[1] != _wino_tiles(B, H, W, 4) or bias.numel() != C:
        raise ValueError("m must be [36, tiles, C] and bias [C]")
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=m.device)
    yp = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=m.device)
    check("pcnn_winograd43_output_both_fwd",
          lib().pcnn_winograd43_output_both_fwd(_ptr(m), _ptr(bias), B, H, W, C, 1 if relu else 0, _ptr(y), _ptr(yp), _stream(m)))
    return y, yp


def winograd43_conv(v, ut, bias, B, H, W, relu=True, pool=0, groups=1):
    """The 36 Winograd-domain contractions + output transform of F(4x4,3x3) in one fp32-MFMA kernel:
    v [36,T,Cin], ut [groups,36,Cout,Cin] (= winograd_filter(w, 4).transpose(1, 2) per group), bias
    [groups,Cout] -> y [B,H,W,Cout] (pool=0), its 2x2 max-pool [B,H/2,W/2,Cout] (pool=1) or both (pool=2,
    returns (y, y_pool)). `groups`: image b uses filter set b // (B // groups)."""
    v = _dev(v, "v", torch.float32)
    ut = _dev(ut, "ut", torch.float32)
    bias = _aligned16(_dev(bias, "bias", torch.float32))
    Cin = v.shape[2]
    if ut.dim() == 3:
        ut = ut.unsqueeze(0)
    Cout = ut.shape[2]
    if (v.shape[0] != 36 or v.shape[1] != _wino_tiles(B, H, W, 4) or tuple(ut.shape) != (groups, 36, Cout, Cin)
            or bias.numel() != groups * Cout):
        raise ValueError("v must be [36, tiles, Cin], ut [groups, 36, Cout, Cin], bias [groups, Cout]")
    pool = int(pool)
    y = torch.empty((B, H // 2, W // 2, Cout) if pool == 1 else (B, H, W, Cout), dtype=torch.float32, device=v.device)
    yp = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.float32, device=v.device) if pool == 2 else None
    nbytes = ctypes.c_size_t()
    check("pcnn_winograd43_conv_workspace_bytes",
          lib().pcnn_winograd43_conv_workspace_bytes(B, H, W, Cin, Cout, int(groups), ctypes.byref(nbytes)))
    ws = _ws(v.device, "wino43_conv").get(nbytes.value, v.device) if nbytes.value else None   # Cin-split partials (small launches only)
    check("pcnn_winograd43_conv_fwd",
          lib().pcnn_winograd43_conv_fwd(_ptr(v), _ptr(ut), _ptr(bias), B, H, W, Cin, Cout, int(groups), 1 if relu else 0,
                                         pool, _ptr(y), _ptr(yp), _ptr(ws), nbytes.value, _stream(v)))
    return (y, yp) if pool == 2 else y


def fc_rows(x, wt, bias, relu=True, num_rows=None, addend=None):
    """`Network.fc` on a capacity-sized row buffer: y[m] = [ReLU](x[m] @ wt.T + bias [+ addend[m]]) for
    m < *num_rows (device int32[1]; None = every row), zeros past it. x [M, K], wt [N, K] (the TF weight
    [K, N] transposed), K % 64 == 0, N % 64 == 0. One fp32-MFMA kernel, no host synchronisation."""
    x = _dev(x, "x", torch.float32)
    wt = _dev(wt, "wt", torch.float32)
    bias = _aligned16(_dev(bias, "bias", torch.float32))
    if x.dim() != 2 or wt.dim() != 2 or wt.shape[1] != x.shape[1] or bias.numel() != wt.shape[0]:
        raise ValueError("x must be [M, K], wt [N, K], bias [N]")
    M, K = x.shape
    N = wt.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    nr = _dev(num_rows, "num_rows", torch.int32) if num_rows is not None else None
    ad = _dev(addend, "addend", torch.float32) if addend is not None else None
    if ad is not None and tuple(ad.shape) != (M, N):
        raise ValueError("addend must be [M, N]")
    nbytes = ctypes.c_size_t()
    check("pcnn_fc_rows_workspace_bytes", lib().pcnn_fc_rows_workspace_bytes(M, K, N, ctypes.byref(nbytes)))
    ws = _ws(x.device, "fc_rows").get(nbytes.value, x.device) if nbytes.value else None   # split-K partials (few live rows)
    check("pcnn_fc_rows_fwd", lib().pcnn_fc_rows_fwd(_ptr(x), _ptr(wt), _ptr(bias), M, K, N, 1 if relu else 0, _ptr(nr), _ptr(ad), _ptr(y),
                                                    _ptr(ws), nbytes.value, _stream(x)))
    return y


def fc_rows_split(x, wt, bias, out_a, relu_a=True, relu_b=False, num_rows=None):
    """Two layers on the same rows as one product: wt [out_a + out_b, K] and bias [out_a + out_b] hold the two filters one
    after the other; returns (y_a [M, out_a], y_b [M, out_b]) with a ReLU flag each. Bit-identical to two `fc_rows` calls."""
    x = _dev(x, "x", torch.float32)
    wt = _dev(wt, "wt", torch.float32)
    bias = _aligned16(_dev(bias, "bias", torch.float32))
    if x.dim() != 2 or wt.dim() != 2 or wt.shape[1] != x.shape[1] or bias.numel() != wt.shape[0] or not 0 < out_a < wt.shape[0]:
        raise ValueError("x must be [M, K], wt [out_a + out_b, K], bias [out_a + out_b]")
    M, K = x.shape
    out_b = wt.shape[0] - int(out_a)
    ya = torch.empty((M, int(out_a)), dtype=torch.float32, device=x.device)
    yb = torch.empty((M, out_b), dtype=torch.float32, device=x.device)
    nr = _dev(num_rows, "num_rows", torch.int32) if num_rows is not None else None
    check("pcnn_fc_rows_split_fwd", lib().pcnn_fc_rows_split_fwd(_ptr(x), _ptr(wt), _ptr(bias), M, K, int(out_a), out_b, 1 if relu_a else 0,
                                                                1 if relu_b else 0, _ptr(nr), _ptr(ya), _ptr(yb), _stream(x)))
    return ya, yb


def fc_rows_cols(x, wt_padded, bias_padded, out_features, activation="none", num_rows=None):
    """`fc_rows` for a width that is no multiple of 64 (fc8: 88): wt_padded [Npad, K] / bias_padded [Npad] zero-padded to a
    multiple of 64, y [M, out_features]. activation "none" | "relu" | "tanh"; "tanh" returns (linear, tanh(linear))."""
    x = _dev(x, "x", torch.float32)
    wt_padded = _dev(wt_padded, "wt_padded", torch.float32)
    bias_padded = _aligned16(_dev(bias_padded, "bias_padded", torch.float32))
    if x.dim() != 2 or wt_padded.dim() != 2 or wt_padded.shape[1] != x.shape[1] or bias_padded.numel() != wt_padded.shape[0]:
        raise ValueError("x must be [M, K], wt_padded [Npad, K], bias_padded [Npad]")
    M, K = x.shape
    act = {"none": 0, "relu": 1, "tanh": 2}[activation]
    y = torch.empty((M, int(out_features)), dtype=torch.float32, device=x.device)
    y2 = torch.empty_like(y) if act == 2 else None
    nr = _dev(num_rows, "num_rows", torch.int32) if num_rows is not None else None
    check("pcnn_fc_rows_cols_fwd", lib().pcnn_fc_rows_cols_fwd(_ptr(x), _ptr(wt_padded), _ptr(bias_padded), M, K, wt_padded.shape[0],
                                                              int(out_features), act, _ptr(nr), _ptr(y), _ptr(y2), _stream(x)))
    return (y, y2) if act == 2 else y


_tickets = {}


def _ticket_buffer(device, n):
    """int32 ticket counters of the split-K kernels that finish in their last workgroup: zero on entry, zero on
    exit (the kernel resets them), so one zero-filled buffer per stream serves every launch on that stream."""
    k = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _tickets.get(k)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(int(n), 256), dtype=torch.int32, device=device)
        _tickets[k] = buf
    return buf


# The crossover between the two few-row paths of fc6 / fc7 / fc8: at or below it csrc/fc_skinny.hip streams the
# weights with split-K over the whole chip; above it `fc_rows` / `fc_rows_cols` take the layer on 64-row blocks. fc8 through
# `fc_rows_cols` has no split-K: with tens of live rows a couple of workgroups stream its padded 128 x 4096 filter (2 MB) —
# ~10 us, launch-bound either way, and only cheaper per row from there.
SKINNY_MAX_ROWS = 32


def fc_skinny(x, wt, bias, activation="none", num_rows=None):
    """`Network.fc` for <= 32 rows (the single-frame loop): y = act(x @ wt.T + bias) as one weight-streaming
    launch (csrc/fc_skinny.hip). x [M <= 32, K % 16 == 0], wt [N, K] (the TF weight transposed), bias [N].
    activation "none" | "relu" | "tanh"; "tanh" returns (linear, tanh(linear)) — fc8 and poses_tanh."""
    x = _dev(x, "x", torch.float32)
    wt = _dev(wt, "wt", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    if x.dim() != 2 or wt.dim() != 2 or wt.shape[1] != x.shape[1] or bias.numel() != wt.shape[0]:
        raise ValueError("x must be [M, K], wt [N, K], bias [N]")
    act = {"none": 0, "relu": 1, "tanh": 2}[activation]
    M, K = x.shape
    N = wt.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    y2 = torch.empty((M, N), dtype=torch.float32, device=x.device) if act == 2 else None
    nr = _dev(num_rows, "num_rows", torch.int32) if num_rows is not None else None
    nbytes, ncnt = ctypes.c_size_t(), ctypes.c_int()
    check("pcnn_fc_skinny_workspace_bytes", lib().pcnn_fc_skinny_workspace_bytes(M, K, N, ctypes.byref(nbytes), ctypes.byref(ncnt)))
    ws = _ws(x.device, "fc_skinny").get(nbytes.value, x.device)
    cnt = _ticket_buffer(x.device, ncnt.value)
    check("pcnn_fc_skinny_fwd", lib().pcnn_fc_skinny_fwd(_ptr(x), _ptr(wt), _ptr(bias), M, K, N, act, _ptr(nr), _ptr(y), _ptr(y2),
                                                        _ptr(ws), nbytes.value, _ptr(cnt), cnt.numel(), _stream(x)))
    return (y, y2) if act == 2 else y


def head_lowres(score4, score5, weights_t, planted=None, kernel=4, stride=2):
    """add = score4 + deconv_{kernel,stride}(score5) [+ planted]; z = add . weights_t (1x1 conv, no bias) in one
    launch. score4 [B,h,w,U], score5 [B,h/stride,w/stride,U], weights_t [U, Cout] -> (add [B,h,w,U], z [B,h,w,Cout])."""
    score4 = _dev(score4, "score4", torch.float32)
    score5 = _dev(score5, "score5", torch.float32)
    weights_t = _dev(weights_t, "weights_t", torch.float32)
    pl = _dev(planted, "planted", torch.float32) if planted is not None else None
    B, h, w, U = score4.shape
    if (tuple(score5.shape) != (B, h // stride, w // stride, U) or weights_t.dim() != 2 or weights_t.shape[0] != U
            or (pl is not None and tuple(pl.shape) != tuple(score4.shape))):
        raise ValueError("score4 [B,h,w,U], score5 [B,h/s,w/s,U], weights_t [U,Cout], planted like score4")
    Cout = weights_t.shape[1]
    add = torch.empty_like(score4)
    z = torch.empty((B, h, w, Cout), dtype=torch.float32, device=score4.device)
    check("pcnn_head_lowres_fwd", lib().pcnn_head_lowres_fwd(_ptr(score4), _ptr(score5), _ptr(pl), _ptr(weights_t), B, h, w, U, Cout,
                                                            int(kernel), int(stride), _ptr(add), _ptr(z), _stream(score4)))
    return add, z


def head_lowres_mfma_filter(weights_t):
    """The filter of `head_lowres_mfma`: weights_t [U, Cout] (the TF variable [1,1,U,Cout] as it is) -> [ceil(Cout/16)*16, U],
    N-major with K contiguous, zero rows past Cout (cache it per weight version)."""
    U, Cout = weights_t.shape
    npad = (Cout + 15) // 16 * 16
    w = torch.zeros((npad, U), dtype=torch.float32, device=weights_t.device)
    w[:Cout] = weights_t.t()
    return w


def head_lowres_mfma(score4, score5, weights_nk, out_channels, planted=None, kernel=4, stride=2):
    """`head_lowres` for many pixels per launch: the same add (bit-identical `add`), the 1x1 product on the matrix cores.
    weights_nk from `head_lowres_mfma_filter`. U % 16 == 0, out_channels <= 96."""
    score4 = _dev(score4, "score4", torch.float32)
    score5 = _dev(score5, "score5", torch.float32)
    weights_nk = _dev(weights_nk, "weights_nk", torch.float32)
    pl = _dev(planted, "planted", torch.float32) if planted is not None else None
    B, h, w, U = score4.shape
    Cout = int(out_channels)
    if (tuple(score5.shape) != (B, h // stride, w // stride, U) or weights_nk.dim() != 2 or weights_nk.shape[1] != U
            or weights_nk.shape[0] != (Cout + 15) // 16 * 16 or (pl is not None and tuple(pl.shape) != tuple(score4.shape))):
        raise ValueError("score4 [B,h,w,U], score5 [B,h/s,w/s,U], weights_nk [ceil(Cout/16)*16, U], planted like score4")
    add = torch.empty_like(score4)
    z = torch.empty((B, h, w, Cout), dtype=torch.float32, device=score4.device)
    check("pcnn_head_lowres_mfma_fwd", lib().pcnn_head_lowres_mfma_fwd(_ptr(score4), _ptr(score5), _ptr(pl), _ptr(weights_nk), B, h, w, U, Cout,
                                                                      int(kernel), int(stride), _ptr(add), _ptr(z), _stream(score4)))
    return add, z


def pose_l2_normalize(poses_tanh, poses_weight, num_rows=None):
    """poses_pred = l2_normalize(poses_tanh * poses_weight, dim=1) (vgg16_convs.py:195-197) in one launch; rows at or past
    the device-side count are zeros."""
    x = _dev(poses_tanh, "poses_tanh", torch.float32)
    w = _dev(poses_weight, "poses_weight", torch.float32)
    if x.dim() != 2 or tuple(w.shape) != tuple(x.shape):
        raise ValueError("poses_tanh and poses_weight must be [R, 4C]")
    nr = _dev(num_rows, "num_rows", torch.int32) if num_rows is not None else None
    out = torch.empty_like(x)
    check("pcnn_pose_l2_normalize_fwd", lib().pcnn_pose_l2_normalize_fwd(_ptr(x), _ptr(w), _ptr(nr), x.shape[0], x.shape[1], _ptr(out), _stream(x)))
    return out


def det_assemble(rois, poses_tanh, top_pose, num_rows, row_stride=1, frame_offset=None):
    """lib/fcn/test.py:206-211 on the device: rows [ceil(R / stride), 14] = box7 | quaternion of the row's class |
    translation, zeros past the device-side count; count [1] int32 = *num_rows // stride.
    With `frame_offset` (a rank's first global frame index) the same launch writes the block that rank hands to the
    detection all-gather — rows with column 0 shifted to global frame numbering + a last row (count, 0, ...) — and
    returns (rows, count, block) with `rows` a view of the block's first rows."""
    rois = _dev(rois, "rois", torch.float32)
    poses_tanh = _dev(poses_tanh, "poses_tanh", torch.float32)
    top_pose = _dev(top_pose, "top_pose", torch.float32)
    nr = _dev(num_rows, "num_rows", torch.int32)
    R = rois.shape[0]
    if rois.dim() != 2 or rois.shape[1] != 7 or tuple(top_pose.shape) != (R, 7) or poses_tanh.shape[0] != R or poses_tanh.shape[1] % 4:
        raise ValueError("rois [R,7], top_pose [R,7], poses_tanh [R,4C]")
    n_out = (R + row_stride - 1) // row_stride
    count = torch.empty((1,), dtype=torch.int32, device=rois.device)
    if frame_offset is not None:
        block = torch.empty((n_out + 1, 14), dtype=torch.float32, device=rois.device)
        check("pcnn_det_assemble_packed_fwd",
              lib().pcnn_det_assemble_packed_fwd(_ptr(rois), _ptr(poses_tanh), _ptr(top_pose), _ptr(nr), R, int(row_stride),
                                                 poses_tanh.shape[1] // 4, float(frame_offset), _ptr(block), _ptr(count), _stream(rois)))
        return block[:n_out], count, block
    rows = torch.empty((n_out, 14), dtype=torch.float32, device=rois.device)
    check("pcnn_det_assemble_fwd", lib().pcnn_det_assemble_fwd(_ptr(rois), _ptr(poses_tanh), _ptr(top_pose), _ptr(nr), R, int(row_stride),
                                                              poses_tanh.shape[1] // 4, _ptr(rows), _ptr(count), _stream(rois)))
    return rows, count


def conv3x3_winograd(x, u, bias, relu=True, pool=False, tile=2):
    """3x3 / stride 1 / SAME convolution + bias [+ ReLU] [+ 2x2 max-pool] as Winograd F(tile x tile, 3x3):
    input transform (gfx950 kernel) -> (tile+2)^2 fp32 GEMMs (library, MFMA) -> output transform
    (gfx950 kernel). `u` comes from `winograd_filter(weights, tile)`."""
    B, H, W, _ = x.shape
    v = winograd_input(x, tile)
    m = torch.bmm(v, u)
    return winograd_output(m, bias, B, H, W, relu, pool, tile)


def bias_relu_pool2(x, bias, relu=True):
    """max_pool_2x2(ReLU(x + bias)) from the raw convolution output [B,H,W,C] (H, W even)."""
    x = _dev(x, "x", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    B, H, W, C = x.shape
    if bias.numel() != C:
        raise ValueError("bias must have one entry per channel")
    y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    check("pcnn_bias_relu_pool2_fwd",
          lib().pcnn_bias_relu_pool2_fwd(_ptr(x), _ptr(bias), B, H, W, C, 1 if relu else 0, _ptr(y), _stream(x)))
    return y


def _deconv_bilinear_raw(input, kernel, stride, add1=None, add2=None, bias=None, relu=False):
    input = _dev(input, "input", torch.float32)
    if input.dim() != 4:
        raise ValueError("deconv input must be 4-dimensional")
    B, H, W, C = input.shape
    out = torch.empty((B, H * stride, W * stride, C), dtype=torch.float32, device=input.device)
    a1 = _dev(add1, "add1", torch.float32) if add1 is not None else None
    a2 = _dev(add2, "add2", torch.float32) if add2 is not None else None
    bs = _dev(bias, "bias", torch.float32) if bias is not None else None
    for t in (a1, a2):
        if t is not None and tuple(t.shape) != tuple(out.shape):
            raise ValueError("addend must have the output shape %s" % (tuple(out.shape),))
    check("pcnn_deconv_bilinear_fwd",
          lib().pcnn_deconv_bilinear_fwd(_ptr(input), B, H, W, C, int(kernel), int(stride), _ptr(a1), _ptr(a2),
                                         _ptr(bs), 1 if relu else 0, _ptr(out), _stream(input)))
    return out


def deconv_bilinear_grad(grad_out, kernel, stride):
    """Gradient of `deconv_bilinear` w.r.t. its input: [B,H*s,W*s,C] -> [B,H,W,C]."""
    grad_out = _dev(grad_out, "grad_out", torch.float32)
    B, Ho, Wo, C = grad_out.shape
    if Ho % stride or Wo % stride:
        raise ValueError("grad_out is not a multiple of the stride")
    gin = torch.empty((B, Ho // stride, Wo // stride, C), dtype=torch.float32, device=grad_out.device)
    check("pcnn_deconv_bilinear_bwd",
          lib().pcnn_deconv_bilinear_bwd(_ptr(grad_out), B, Ho // stride, Wo // stride, C, int(kernel), int(stride),
                                         _ptr(gin), _stream(grad_out)))
    return gin


class _DeconvBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, kernel, stride):
        ctx.ks = (kernel, stride)
        return _deconv_bilinear_raw(input, kernel, stride)

    @staticmethod
    def backward(ctx, grad_out):
        return deconv_bilinear_grad(grad_out.contiguous(), *ctx.ks), None, None


def deconv_bilinear(input, kernel, stride, add1=None, add2=None, bias=None, relu=False):
    """The fixed bilinear `deconv` layer (network.py:207-222 with make_deconv_filter :141-157) as a
    per-channel interpolation: [B,H,W,C] -> [B,H*s,W*s,C], optionally fused with up to two addends
    (same shape as the output), a per-channel bias and ReLU. Differentiable w.r.t. `input` (the
    filter is not trainable in the reference: deconv(..., trainable=False)); when a gradient is
    needed the extras are applied as separate framework ops."""
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (input, add1, add2, bias))
    if not needs_grad:
        return _deconv_bilinear_raw(input, kernel, stride, add1, add2, bias, relu)
    out = _DeconvBilinearFn.apply(_dev(input, "input", torch.float32), int(kernel), int(stride))
    for t in (add1, add2, bias):
        if t is not None:
            out = out + t
    return torch.relu(out) if relu else out


class _SmoothL1VertexFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, sigma):
        n = pred.numel()
        out = torch.empty(3, dtype=torch.float32, device=pred.device)
        nbytes = c_size_t(0)
        check("pcnn_smooth_l1_vertex_workspace_bytes", lib().pcnn_smooth_l1_vertex_workspace_bytes(ctypes.byref(nbytes)))
        ws = _ws(pred.device, "smooth_l1").get(nbytes.value, pred.device)
        check("pcnn_smooth_l1_vertex_fwd",
              lib().pcnn_smooth_l1_vertex_fwd(_ptr(pred), _ptr(target), _ptr(weight), n, float(sigma), _ptr(out),
                                              _ptr(ws), ws.numel(), _stream(pred)))
        ctx.save_for_backward(pred, target, weight, out)
        ctx.sigma = float(sigma)
        return out[0], out[1:].clone()

    @staticmethod
    def backward(ctx, grad_loss, _grad_sums):
        pred, target, weight, out = ctx.saved_tensors
        grad = torch.empty_like(pred)
        up = grad_loss.reshape(1).to(torch.float32).contiguous()
        check("pcnn_smooth_l1_vertex_bwd",
              lib().pcnn_smooth_l1_vertex_bwd(_ptr(pred), _ptr(target), _ptr(weight), _ptr(out), _ptr(up), pred.numel(),
                                              ctx.sigma, _ptr(grad), _stream(pred)))
        return grad, None, None, None


def smooth_l1_loss_vertex(vertex_pred, vertex_targets, vertex_weights, sigma=1.0):
    """lib/fcn/train.py:564-573 as two streaming kernels (forward, backward w.r.t. vertex_pred).
    Returns the scalar loss = sum(in_loss) / (sum(weights) + 1e-10)."""
    pred = _dev(vertex_pred, "vertex_pred", torch.float32)
    target = _dev(vertex_targets, "vertex_targets", torch.float32)
    weight = _dev(vertex_weights, "vertex_weights", torch.float32)
    if pred.shape != target.shape or pred.shape != weight.shape:
        raise ValueError("vertex_pred, vertex_targets and vertex_weights must have the same shape")
    return _SmoothL1VertexFn.apply(pred, target, weight, float(sigma))[0]


def upscore_softmax_argmax(z, bias, kernel, stride, relu=True, want_score=False, want_prob=True, hard_gt=None,
                           hard_threshold=None):
    """Label-head epilogue in one pass: score = [ReLU](deconv(z) + bias) -> softmax -> first argmax.
    z [B,H,W,C] f32. Returns (score or None, prob or None, label int32 [B,H*s,W*s]); with `hard_gt` (int32
    [B,H*s,W*s]) and `hard_threshold` a fourth element: hard_label(prob, hard_gt, hard_threshold) f32 [B,H*s,W*s,C]
    from the same launch (no gradient flows through the Hardlabel op: hard_label_op_gpu.cu.cc:55-63)."""
    z = _dev(z, "z", torch.float32)
    bias = _dev(bias, "bias", torch.float32)
    B, H, W, C = z.shape
    Ho, Wo = H * stride, W * stride
    dev = z.device
    score = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev) if want_score else None
    prob = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev) if want_prob else None
    label = torch.empty((B, Ho, Wo), dtype=torch.int32, device=dev)
    if hard_gt is not None:
        if hard_threshold is None:
            raise ValueError("hard_gt needs hard_threshold (the Hardlabel op's `threshold` attribute)")
        # the Hardlabel op on the probabilities of the same launch (vgg16_convs.py:148-149): same bits as
        # hard_label(prob, gt, threshold), 432 MB of stores per 16 frames that fit under the head's own arithmetic
        gt = _dev(hard_gt, "hard_gt", torch.int32)
        if tuple(gt.shape) != (B, Ho, Wo):
            raise ValueError("hard_gt must be int32 [B, H*stride, W*stride]")
        hard = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
        check("pcnn_upscore_softmax_argmax_hard_fwd",
              lib().pcnn_upscore_softmax_argmax_hard_fwd(_ptr(z), _ptr(bias), B, H, W, C, int(kernel), int(stride),
                                                         1 if relu else 0, _ptr(score), _ptr(prob), _ptr(label),
                                                         _ptr(gt), float(hard_threshold), _ptr(hard), _stream(z)))
        return score, prob, label, hard
    check("pcnn_upscore_softmax_argmax_fwd",
          lib().pcnn_upscore_softmax_argmax_fwd(_ptr(z), _ptr(bias), B, H, W, C, int(kernel), int(stride),
                                                1 if relu else 0, _ptr(score), _ptr(prob), _ptr(label), _stream(z)))
    return score, prob, label


# ------------------------------------------------------------------------------------------------
class _AverageDistanceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, weight, point, symmetry, margin, num_rows=None):
        R, CH = prediction.shape
        C, P = point.shape[0], point.shape[1]
        dev = prediction.device
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        bottom_diff = torch.empty((R, CH), dtype=torch.float32, device=dev)
        nbytes = c_size_t(0)
        check("pcnn_average_distance_workspace_bytes",
              lib().pcnn_average_distance_workspace_bytes(R, C, P, ctypes.byref(nbytes)))
        ws = _ws(dev, "adl").get(nbytes.value, dev)
        check("pcnn_average_distance_fwd",
              lib().pcnn_average_distance_fwd(_ptr(prediction), _ptr(target), _ptr(weight), _ptr(point),
                                              _ptr(symmetry), R, C, P, float(margin), _ptr(num_rows), _ptr(loss),
                                              _ptr(bottom_diff), _ptr(ws), ws.numel(), _stream(prediction)))
        ctx.save_for_backward(bottom_diff)
        ctx.mark_non_differentiable(bottom_diff)
        return loss, bottom_diff

    @staticmethod
    def backward(ctx, grad_loss, _grad_diff):
        (bottom_diff,) = ctx.saved_tensors
        R, CH = bottom_diff.shape
        grad_loss = grad_loss.contiguous()
        out = torch.empty_like(bottom_diff)
        if R > 0:
            check("pcnn_average_distance_bwd",
                  lib().pcnn_average_distance_bwd(_ptr(grad_loss), _ptr(bottom_diff), R, CH, _ptr(out),
                                                  _stream(bottom_diff)))
        return out, None, None, None, None, None, None


def average_distance_loss(poses_pred, poses_target, poses_weight, points, symmetry, margin, name=None, num_rows=None):
    """Drop-in for `Network.average_distance_loss` (network.py:236-238).
    Returns (loss[1], bottom_diff[R,4C]). `num_rows` (device int32[1]): the op's true row count when
    the inputs are capacity-sized buffers of the sync-free Hough op (normalisation by that count)."""
    poses_pred = _dev(poses_pred, "prediction", torch.float32)
    poses_target = _dev(poses_target, "target", torch.float32)
    poses_weight = _dev(poses_weight, "weight", torch.float32)
    points = _dev(points, "point", torch.float32)
    symmetry = _dev(symmetry, "symmetry", torch.float32)
    for t, n in ((poses_pred, "prediction"), (poses_target, "target"), (poses_weight, "weight")):
        if t.dim() != 2:
            raise ValueError("%s must be 2-dimensional" % n)  # average_distance_loss_op.cc:278-285
    if points.dim() != 3:
        raise ValueError("point must be 3-dimensional")
    if symmetry.dim() != 1:
        raise ValueError("symmetry must be 1-dimensional")
    if poses_pred.shape[1] != POSE_CHANNELS * points.shape[0]:
        raise ValueError("prediction must be [R, 4*num_classes]")
    if num_rows is not None:
        num_rows = _dev(num_rows, "num_rows", torch.int32)
    return _AverageDistanceFn.apply(poses_pred, poses_target, poses_weight, points, symmetry, margin, num_rows)


# ------------------------------------------------------------------------------------------------
class _BackprojectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, label, depth, meta_data, label_3d, grid_size, kernel_size, threshold):
        B, H, W, Cd = data.shape
        Cl = label.shape[3]
        num_meta = meta_data.shape[-1]
        G = int(grid_size)
        dev = data.device
        top_data = torch.empty((B, G, G, G, Cd), dtype=torch.float32, device=dev)
        top_flag = torch.empty((B, G, G, G, Cd), dtype=torch.float32, device=dev)
        top_label = torch.empty((B, G, G, G, Cl), dtype=torch.float32, device=dev)
        # caller-owned scratch for the window-range table that lets 85 % of the voxels skip their depth scan
        need = c_size_t(0)
        check("pcnn_backproject_workspace_bytes", lib().pcnn_backproject_workspace_bytes(B, H, W, int(kernel_size), ctypes.byref(need)))
        ws = _ws(dev, "backproject").get(need.value, dev) if need.value else None
        check("pcnn_backproject_ws_fwd",
              lib().pcnn_backproject_ws_fwd(_ptr(data), _ptr(label), _ptr(depth), _ptr(meta_data), _ptr(label_3d),
                                            B, H, W, Cd, Cl, num_meta, G, int(kernel_size), float(threshold),
                                            _ptr(top_data), _ptr(top_label), _ptr(top_flag),
                                            _ptr(ws), c_size_t(need.value if ws is not None else 0), _stream(data)))
        ctx.save_for_backward(depth, meta_data)
        ctx.cfg = (B, H, W, Cd, num_meta, G)
        ctx.mark_non_differentiable(top_label, top_flag)
        return top_data, top_label, top_flag

    @staticmethod
    def backward(ctx, grad_data, _gl, _gf):
        depth, meta_data = ctx.saved_tensors
        B, H, W, Cd, num_meta, G = ctx.cfg
        grad_data = grad_data.contiguous()
        bottom = torch.empty((B, H, W, Cd), dtype=torch.float32, device=grad_data.device)
        check("pcnn_backproject_bwd",
              lib().pcnn_backproject_bwd(_ptr(grad_data), _ptr(depth), _ptr(meta_data), B, H, W, Cd, num_meta, G,
                                         _ptr(bottom), _stream(grad_data)))
        return bottom, None, None, None, None, None, None, None


def backproject(data, label, depth, meta_data, label_3d, grid_size, kernel_size, threshold, name=None):
    """Drop-in for `Network.backproject` (network.py:224-226).
    Returns (top_data[B,G,G,G,Cd], top_label[B,G,G,G,Cl], top_flag[B,G,G,G,Cd])."""
    data = _dev(data, "data", torch.float32)
    label = _dev(label, "label", torch.float32)
    depth = _dev(depth, "depth", torch.float32)
    meta_data = _dev(meta_data, "meta_data", torch.float32)
    label_3d = _dev(label_3d, "label_3d", torch.float32)
    if data.dim() != 4:
        raise ValueError("data must be 4-dimensional")       # backprojecting_op.cc:332-333
    if label.dim() != 4:
        raise ValueError("label must be 4-dimensional")      # :335-336
    if depth.dim() != 4:
        raise ValueError("depth must be 4-dimensional")      # :338-339
    if meta_data.dim() != 4:
        raise ValueError("meta data must be 4-dimensional")  # :341-342
    if label_3d.dim() != 5:
        raise ValueError("label 3D must be 5-dimensional")   # :344-345
    return _BackprojectFn.apply(data, label, depth, meta_data, label_3d, grid_size, kernel_size, threshold)
