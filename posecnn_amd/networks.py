"""`lib/networks` on MI355X: the layer DSL (`Network`) and the PoseCNN graph (`vgg16_convs`).

This mirrors the reference's op surface — same layer names, argument order and defaults as
lib/networks/network.py:159-222,303-310,321-340,361-445,474-506 and the same `setup()` chain as
lib/networks/vgg16_convs.py:79-212 — but executes eagerly on PyTorch-ROCm tensors and streams. On the
default inference path every dense contraction runs in the library's own gfx950 kernels behind
libposecnn_hip.so (posecnn_amd.ops): the 3x3 trunk as Winograd F(4x4,3x3) on the fp32 matrix cores
(csrc/winograd.hip, wino_mfma.hip, conv_first.hip), the 1x1 heads and fc6-8 on the row kernels of
csrc/fc_mfma.hip, next to the five custom layers. MIOpen / hipBLASLt are only reached
by the trainable training graph and by the non-default switches (`strict_numerics`, `fused_heads=False`,
`winograd_min_channels = 0`). All activations are NHWC (`[B,H,W,C]` contiguous), exactly what the custom
kernels index; library convolutions see them as channels-last NCHW views, so there is no layout copy
anywhere on the path.

TF1 -> PyTorch semantics (SURVEY.md §8a "Semantics ..."):
  conv       tf.nn.conv2d 'SAME' stride 1 + bias + ReLU unless relu=False      (network.py:159-188)
  max_pool   2x2 stride 2 'SAME' (all sizes stay even after pad_im(.,16))        (:303-310)
  deconv     tf.nn.conv2d_transpose, fixed diagonal bilinear filter, no bias    (:141-157,207-222)
  fc         NHWC-flattened input, weights [in,out], relu_layer / xw_plus_b      (:392-422)
  softmax_high_dimension / argmax_2d  -> fused gfx950 kernel                     (:474-488,432-434)
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import ops, pipeline

DEFAULT_PADDING = "SAME"


def layer(op):
    """Same decorator contract as network.py:40-59: feeds the current inputs to `op`, records the
    result under kwargs['name'] and makes it the next input."""

    def layer_decorated(self, *args, **kwargs):
        name = kwargs.setdefault("name", self.get_unique_name(op.__name__))
        if len(self.inputs) == 0:
            raise RuntimeError("No input variables found for layer %s." % name)
        elif len(self.inputs) == 1:
            layer_input = self.inputs[0]
        else:
            layer_input = list(self.inputs)
        if op.__name__ == "conv" and isinstance(layer_input, _RawConv) and layer_input.first is not None:
            pass  # a Winograd conv can take a pending first conv as it is (conv() decides)
        elif op.__name__ != "max_pool":
            # only max_pool fuses a deferred bias + ReLU; everybody else sees the activated tensor
            if isinstance(layer_input, list):
                layer_input = [self._activate(i) if isinstance(i, _RawConv) else i for i in layer_input]
            elif isinstance(layer_input, _RawConv):
                layer_input = self._activate(layer_input)
        layer_output = op(self, layer_input, *args, **kwargs)
        self.layers[name] = layer_output
        self.feed(layer_output)
        return self

    layer_decorated.__name__ = op.__name__
    return layer_decorated


def make_deconv_filter_1d(k):
    """network.py:141-150: f = ceil(k/2), c = (2f - 1 - f%2) / (2f), w[x] = 1 - |x/f - c|."""
    f = math.ceil(k / 2.0)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    return np.array([1 - abs(x / f - c) for x in range(k)], dtype=np.float64)


def is_bilinear_deconv_filter(w):
    """True for a conv_transpose filter in torch layout [c, c, k, k] (TF `[k, k, c_out, c_in]` after
    Network.load's permute) that equals make_deconv_filter (network.py:141-157):
    f32(outer(bilinear, bilinear)) on the channel diagonal, exact zeros elsewhere."""
    w = torch.as_tensor(w)
    if w.dim() != 4 or w.shape[0] != w.shape[1] or w.shape[2] != w.shape[3]:
        return False
    c, k = w.shape[0], w.shape[2]
    if k % 2 or k < 2:      # PoseCNN's deconvs are k = 4 / 16; 3x3 and 1x1 convs are never candidates
        return False
    f = torch.tensor(make_deconv_filter_1d(k), dtype=torch.float64)
    want = torch.outer(f, f).to(torch.float32)
    t = w.detach().to("cpu", torch.float32)
    diag = t[torch.arange(c), torch.arange(c)]                     # [c, k, k]
    if not torch.equal(diag, want.expand(c, k, k)):
        return False
    return float(t.abs().sum(dtype=torch.float64)) == float(diag.abs().sum(dtype=torch.float64))   # nothing off the diagonal


def _nchw(x):  # NHWC contiguous -> channels-last NCHW view (no copy)
    return x.permute(0, 3, 1, 2)


def _nhwc(x):  # channels-last NCHW -> NHWC contiguous view (no copy when channels_last)
    return x.permute(0, 2, 3, 1).contiguous()


class _Lazy(object):
    """A layer output that is only evaluated when somebody fetches or feeds it (TF evaluates only
    fetched tensors; eager code needs to be told)."""

    def __init__(self, fn):
        self.fn = fn


def raw_frame_blob(t):
    """A raw frame tensor -> the network's f32 input blob with the arithmetic of lib/fcn/test.py:56-74 (fcn._get_image_blob):
    uint8 BGR [B,H,W,3] -> float32(double(x) - PIXEL_MEANS); uint16 depth [B,H,W] -> clip(d / 2000, 0, 1) * 255 tiled to 3
    channels, minus the means. Only the configurations that do not fuse conv1_1 into conv1_2's input transform need it —
    the fused kernel forms these values itself (ops.conv3x3_c3_winograd43_raw)."""
    from .config import PIXEL_MEANS
    means = torch.from_numpy(np.asarray(PIXEL_MEANS, dtype=np.float64).reshape(1, 1, 1, 3)).to(t.device)
    if t.dtype == torch.uint8:
        return (t.to(torch.float64) - means).to(torch.float32).contiguous()
    if t.dtype == torch.uint16:
        d = (t.view(torch.int16).to(torch.int32) & 0xFFFF).to(torch.float64)
        if d.dim() == 4:
            d = d.reshape(d.shape[:3])
        # numpy's float32 / 2000.0 is a correctly rounded f32 division; the framework's f32 division on the GPU is not
        # (reciprocal multiply). A double division rounded to f32 is: d / 2000 with an integer d < 65536 is never within
        # 2^-36 (relative) of an f32 rounding boundary, so the second rounding cannot flip it.
        d = torch.clamp((d / 2000.0).to(torch.float32), 0, 1) * 255
        return (d.unsqueeze(-1).to(torch.float64) - means).to(torch.float32).contiguous()
    return t


def _is_raw(t):
    return isinstance(t, torch.Tensor) and t.dtype in (torch.uint8, torch.uint16)


class _RawConv(object):
    """A convolution output before bias + ReLU, handed to a consumer that fuses them (max_pool).
    Fetching the layer by name materialises the activated tensor (in place) like any other."""

    def __init__(self, y, bias, relu, wino=None, first=None, gemm=None, lazy12=None):
        # y: raw NHWC conv output, or None when the conv is still pending:
        # wino  = (M [n*n,T,C], B, H, W): in the Winograd domain, waiting for its output transform
        # first = (x [B,H,W,3], w [3,3,3,C]): a 3-channel first conv not evaluated yet — a following
        #         Winograd conv computes it fused with its own input transform
        # gemm  = (V [36,T,Cin], Ut [36,Cout,Cin], B, H, W): F(4x4,3x3) input transform done, the fused
        #         GEMM + output transform kernel still to run (with or without the max-pool)
        # lazy12 = (pending conv1_1 (_RawConv with `first`), name, w): conv1_2 on top of a pending conv1_1, nothing launched
        #         yet — a following 2x2 max_pool runs conv1_1 -> conv1_2 -> pool1 as ONE kernel; anything else that asks for the
        #         tensor gets the unfused pair
        self.y, self.bias, self.relu, self.out, self.wino, self.first, self.gemm = y, bias, relu, None, wino, first, gemm
        self.lazy12 = lazy12
        self.dual = False
        self.pooled = None   # the 2x2 max-pool of `out`, when the producing kernel wrote both

    @property
    def shape(self):
        if self.wino is not None:
            return (self.wino[1], self.wino[2], self.wino[3], self.wino[0].shape[2])
        if self.first is not None:
            return tuple(self.first[0].shape[:3]) + (self.first[1].shape[3],)
        if self.gemm is not None:
            return (self.gemm[2], self.gemm[3], self.gemm[4], self.gemm[1].shape[-2])
        if self.lazy12 is not None:
            return tuple(self.lazy12[0].first[0].shape[:3]) + (self.lazy12[2].shape[0],)
        if self.y is None and self.out is not None:
            return tuple(self.out.shape)
        return tuple(self.y.shape)


class Network(object):
    """Eager re-statement of the reference's graph-building `Network` (network.py:61-137).

    Variables live in `self.vars` under '<layer>/weights' and '<layer>/biases' (the names
    tf.variable_scope + make_var give them, network.py:169-185) and are created on first use.
    """

    def __init__(self, device="cuda", seed=3, init="he", trainable=True):
        self.inputs = []
        self.layers = {}
        self.vars = {}
        self.device = torch.device(device)
        self.trainable = trainable
        self.init = init
        self._gen = torch.Generator(device="cpu").manual_seed(seed)  # cfg.RNG_SEED = 3 (config.py)
        self.keep_prob_queue = 1.0
        self.conv_timing = None       # a list -> (name, executed flops, direct-conv flops, start, end) per conv
        # 3x3 / stride 1 / SAME convs with at least this many input channels are evaluated in the
        # Winograd domain (0 = never). tile 4 = F(4x4,3x3): 4x fewer multiplies, all f32, per-layer
        # error <= 1e-5 of the output range and end-to-end indistinguishable from the direct path at
        # the pipeline's tolerances (tests/test_gpu_pipeline.py); it beats the direct library
        # convolution on every VGG layer from conv1_2 on (tools/bench_layers.py). tile 2 =
        # F(2x2,3x3): 2.25x fewer multiplies, same error as a direct f32 convolution, wins from 128
        # input channels.
        self.winograd_min_channels = 64
        self.winograd_tile = 4
        self._wino_u = {}
        self._fc_cache = {}    # transposed / zero-padded fc weights (not in the Winograd filter cache)
        # F(4x4,3x3) layers with Cin, Cout multiples of 64: the 36 contractions + output transform run in
        # the library's own fp32-MFMA kernel (csrc/wino_mfma.hip); False = library batched GEMM + transform kernel
        self.winograd_mfma = True
        self.fuse_first_conv_into_winograd = True
        self.fused_first_conv = True  # 3-channel 3x3 convs go to the fused conv + bias + ReLU kernel
        self.fused_conv12 = True      # conv1_1 -> conv1_2 -> pool1 as one LDS-resident kernel (round 4): the grouped RGB-D trunk, and
                                      # (lazily, when the max_pool arrives) every single tower
        self.defer_act = frozenset()  # conv layers whose bias + ReLU is left to the following max_pool
        self.dual_pool = frozenset()  # ... and those whose un-pooled output other layers read too (Winograd only)
        self.rows_count = None        # device int32[1]: true row count of capacity-sized ROI rows fed to `fc` (or None)
        self.fc_skinny = True         # <= 32 capacity rows: fc6-8 on the weight-streaming kernel (csrc/fc_skinny.hip)
        self.small_heads = True       # 1/8-resolution head algebra (deconv + add + 1x1) in one launch (csrc/heads_small.hip)
        self.small_heads_max_pixels = 2 * 60 * 80

    # ---- plumbing ------------------------------------------------------------------------------
    def setup(self):
        raise NotImplementedError("Must be subclassed.")

    def feed(self, *args):
        assert len(args) != 0
        self.inputs = []
        for l in args:
            if isinstance(l, str):
                l = self.get_output(l)
            self.inputs.append(l)
        return self

    def get_output(self, name):
        try:
            out = self.layers[name]
        except KeyError:
            raise KeyError("Unknown layer name fed: %s" % name)
        if isinstance(out, _Lazy):
            out = self.layers[name] = out.fn()
        if isinstance(out, _RawConv):
            out = self.layers[name] = self._activate(out)
        return out

    def _activate(self, raw):
        if raw.out is None:
            if raw.wino is not None:
                m, B, H, W = raw.wino
                raw.out = ops.winograd_output(m, raw.bias, B, H, W, raw.relu, pool=False, tile=self.winograd_tile)
            elif raw.first is not None:
                raw.out = ops.conv3x3_c3(raw_frame_blob(raw.first[0]), raw.first[1], raw.bias, raw.relu)
            elif raw.gemm is not None:
                v, ut, B, H, W = raw.gemm
                raw.out = ops.winograd43_conv(v, ut, raw.bias, B, H, W, raw.relu, pool=0)
            elif raw.lazy12 is not None:   # somebody wants conv1_2 un-pooled: the unfused pair
                pf, name12, w2 = raw.lazy12
                B, H, W = pf.shape[:3]
                raw.out = ops.winograd43_conv(self._first_conv_v(pf), self._winograd_filter(name12, w2, transposed=True), raw.bias,
                                              B, H, W, raw.relu, pool=0)
            else:
                raw.out = self._bias_act(raw.y, raw.bias, raw.relu)
        return raw.out

    def _first_conv_v(self, pf):
        """V of conv1_2 with the pending conv1_1 `pf` evaluated inside its input transform (blobs or raw frames)."""
        x0 = pf.first[0]
        if _is_raw(x0):
            return ops.conv3x3_c3_winograd43_raw(x0 if x0.dtype == torch.uint8 else None, x0 if x0.dtype == torch.uint16 else None,
                                                 pf.first[1], pf.bias, pf.relu)
        return ops.conv3x3_c3_winograd43(x0, pf.first[1], pf.bias, pf.relu)

    def _winograd_filter(self, name, w, transposed=False):
        """U = G g G^T of a conv filter ([n*n, Cin, Cout]; transposed: [n*n, Cout, Cin] for the fused
        GEMM + output kernel), cached until the variable changes."""
        key = (w.data_ptr(), w._version, self.winograd_tile, transposed)
        hit = self._wino_u.get((name, transposed))
        if hit is None or hit[0] != key:
            u = ops.winograd_filter(w, self.winograd_tile)
            hit = (key, u.transpose(1, 2).contiguous() if transposed else u)
            self._wino_u[(name, transposed)] = hit
        return hit[1]

    def _winograd43_tail(self, name, v, w, b, relu, B_, H_, W_, c_i, c_o, timing):
        """Everything after the F(4x4,3x3) input transform: the library's fp32-MFMA kernel (contractions
        + output transform, optionally the following 2x2 max-pool) for Cin, Cout multiples of 64;
        framework batched GEMM + output transform kernel otherwise."""
        if self.winograd_mfma and c_i % 64 == 0 and c_o % 64 == 0:
            ut = self._winograd_filter(name, w, transposed=True)
            if timing is not None:
                timing[1].record()   # the MFMA kernel is timed by the library's own events
                self.conv_timing.append((name, timing[2], timing[2], timing[0], timing[1]))
            even = H_ % 2 == 0 and W_ % 2 == 0
            if name in self.defer_act and even:
                return _RawConv(None, b, relu, gemm=(v, ut, B_, H_, W_))      # the following max_pool decides
            if name in self.dual_pool and even:
                out = _RawConv(None, b, relu)
                out.out, out.pooled = ops.winograd43_conv(v, ut, b, B_, H_, W_, relu, pool=2)
                return out
            return ops.winograd43_conv(v, ut, b, B_, H_, W_, relu, pool=0)
        m = torch.bmm(v, self._winograd_filter(name, w))
        executed = 2.0 * m.numel() * c_i
        if timing is not None:
            timing[1].record()   # transform + GEMMs; the output transform is timed by the library
        if name in self.defer_act:
            out = _RawConv(None, b, relu, wino=(m, B_, H_, W_))
        elif name in self.dual_pool and H_ % 2 == 0 and W_ % 2 == 0:
            out = _RawConv(None, b, relu, wino=(m, B_, H_, W_))
            out.dual = True   # the following max_pool produces both tensors in one pass
        else:
            out = ops.winograd_output(m, b, B_, H_, W_, relu, pool=False, tile=4)
        if timing is not None:
            e0, e1, extra = timing
            self.conv_timing.append((name, executed + extra, 2.0 * B_ * H_ * W_ * c_o * c_i * 9 + extra, e0, e1))
        return out

    def get_unique_name(self, prefix):
        ident = sum(t.startswith(prefix) for t in self.layers) + 1
        return "%s_%d" % (prefix, ident)

    def make_var(self, name, shape, initializer, trainable=False):
        """network.py:108-110. A variable asks for gradients when both the network and the layer are
        trainable (the reference passes `trainable` to tf.get_variable the same way)."""
        if name not in self.vars:
            self.vars[name] = initializer(shape).to(self.device)
        v = self.vars[name]
        if self.trainable and trainable and not v.requires_grad and v.is_leaf:
            v.requires_grad_(True)
        if tuple(v.shape) != tuple(shape):
            raise ValueError("variable %s has shape %s, layer wants %s" % (name, tuple(v.shape), tuple(shape)))
        return v

    def _weight_init(self, fan_in):
        if self.init == "tf":  # tf.truncated_normal_initializer(0.0, stddev=0.001), network.py:170
            def f(shape):
                w = torch.empty(shape)
                torch.nn.init.trunc_normal_(w, 0.0, 0.001, -0.002, 0.002, generator=self._gen)
                return w
        else:  # He init: keeps random-weight activations in a sane f32 range through 13 layers
            def f(shape):
                return torch.randn(shape, generator=self._gen) * math.sqrt(2.0 / fan_in)
        return f

    def load(self, data_dict, ignore_missing=False):
        """network.py:71-107: `{layer: {'weights': [kh,kw,cin,cout] | [in,out], 'biases': [cout]}}`
        (the vgg16.npy / converted-checkpoint layout). A layer's weights also initialise its
        depth-tower twin '<layer>_p' — but only when the dict has no '<layer>_p' entry of its own
        (vgg16.npy initialisation of an RGBD network); a checkpoint that carries both towers loads
        each from its own entry whatever the key order. Constant bilinear `deconv` filters
        (make_deconv_filter, network.py:141-157, trainable=False) are recognised and dropped, so
        loaded and freshly built networks run the same interpolation kernel; a deconv filter that is
        NOT the fixed bilinear one is kept and switches the graph to the literal op order."""
        for op_name, params in data_dict.items():
            for suffix in ("", "_p"):
                if suffix and (op_name + suffix) in data_dict:
                    continue
                for pname, data in params.items():
                    key = "%s%s/%s" % (op_name, suffix, pname)
                    t = torch.as_tensor(np.asarray(data), dtype=torch.float32)
                    if pname == "weights" and t.dim() == 4:
                        t = t.permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last)
                    if key in self.vars and tuple(self.vars[key].shape) != tuple(t.shape):
                        if not ignore_missing:
                            raise ValueError("shape mismatch for %s" % key)
                        continue
                    if pname == "weights" and t.dim() == 4 and is_bilinear_deconv_filter(t):
                        self.vars.pop(key, None)   # the constant filter: the interpolation kernels ARE this filter
                        continue
                    if suffix == "" or key in self.vars:
                        self.vars[key] = t.to(self.device)
                        if pname == "weights" and t.dim() == 4 and op_name.startswith("upscore") and hasattr(self, "fused_heads"):
                            # a trained / non-bilinear deconv filter: conv1x1 and deconv no longer commute
                            self.fused_heads = False

    def load_file(self, path, ignore_missing=False):
        """Checkpoint ingestion. Accepts
          - `vgg16.npy`-style pickled dicts {layer: {'weights', 'biases'}} (network.py:71-107), and
          - TF checkpoint prefixes (`<prefix>.index` + `<prefix>.data-*`; `posecnn_amd/tf_checkpoint.py`),
          - `.npz` archives keyed by TF variable name ('conv1_1/weights', 'fc6/biases', ...) in TF
            layouts ([kh,kw,cin,cout] / [in,out]) — what three lines of TF dump from a checkpoint:
            `np.savez(out, **{v.name[:-2]: sess.run(v) for v in tf.global_variables()})`
            (INTEGRATION.md). Optimizer slots ('.../Momentum') are skipped."""
        if os.path.exists(str(path) + ".index"):
            # a TF checkpoint prefix (saver.restore, lib/fcn/test.py:1809-1811): read the tensor bundle
            from . import tf_checkpoint
            data = tf_checkpoint.to_layer_dict(tf_checkpoint.read_checkpoint(str(path)))
        elif str(path).endswith(".npz"):
            arch = np.load(path)
            data = {}
            for key in arch.files:
                if key.count("/") != 1 or key.rsplit("/", 1)[1] not in ("weights", "biases"):
                    continue
                layer_name, pname = key.rsplit("/", 1)
                data.setdefault(layer_name, {})[pname] = arch[key]
        else:
            data = np.load(path, allow_pickle=True, encoding="latin1").item()
        self.load(data, ignore_missing)
        return sorted(data)

    def save_npz(self, path):
        """Writes every variable under its TF name in TF layout (the inverse of load_file)."""
        out = {}
        for key, v in self.vars.items():
            t = v.detach().cpu()
            if key.endswith("/weights") and t.dim() == 4:
                t = t.permute(2, 3, 1, 0)  # [cout,cin,kh,kw] -> [kh,kw,cin,cout]
            out[key] = np.ascontiguousarray(t.numpy())
        np.savez(path, **out)

    # ---- dense layers (MIOpen / hipBLASLt) -------------------------------------------------------
    @layer
    def conv(self, input, k_h, k_w, c_o, s_h, s_w, name, reuse=None, relu=True, padding=DEFAULT_PADDING,
             group=1, trainable=True, biased=True, c_i=-1):
        assert padding in ("SAME", "VALID")
        if isinstance(input, tuple):
            input = input[0]
        raw_in = _is_raw(input)       # a frame as the sensor delivers it (uint8 BGR / uint16 depth): the blob has 3 channels
        if c_i == -1:
            c_i = 3 if raw_in else input.shape[-1]
        assert c_i % group == 0 and c_o % group == 0
        pending_first = input if isinstance(input, _RawConv) else None
        w = self.make_var(name + "/weights", (c_o, c_i // group, k_h, k_w),
                          lambda s: self._weight_init(c_i // group * k_h * k_w)(s).contiguous(memory_format=torch.channels_last),
                          trainable)
        b = self.make_var(name + "/biases", (c_o,), lambda s: torch.zeros(s), trainable) if biased else None
        assert s_h == 1 and s_w == 1 or padding == "VALID" or (k_h == 1 and k_w == 1), "strided SAME conv not on this path"
        pad = (k_h // 2, k_w // 2) if padding == "SAME" else 0
        if (self.fused_first_conv and b is not None and (k_h, k_w, c_i, group) == (3, 3, 3, 1) and padding == "SAME"
                and c_o % 64 == 0 and not (torch.is_grad_enabled() and (w.requires_grad or input.requires_grad))):
            # conv1_1: K = 27 is no GEMM; one HBM-bound kernel does conv + bias + ReLU — or, when a
            # Winograd conv follows, that layer's input transform kernel does it on the fly
            if self.fuse_first_conv_into_winograd and self.winograd_tile == 4 and self.winograd_min_channels and input.is_cuda:
                return _RawConv(None, b, relu, first=(input.contiguous(), w.permute(2, 3, 1, 0).contiguous()))
            return self._conv_first(raw_frame_blob(input), w, b, relu)
        if raw_in:
            input = raw_frame_blob(input)
        wino_ok = (b is not None and (k_h, k_w, s_h, s_w, group) == (3, 3, 1, 1, 1) and padding == "SAME")
        if pending_first is not None:
            if (wino_ok and self.winograd_tile == 4 and self.winograd_min_channels
                    and c_i >= self.winograd_min_channels and c_i % 64 == 0 and c_o % 4 == 0
                    and not (torch.is_grad_enabled() and w.requires_grad)):
                # conv1_1 -> conv1_2: the first conv is evaluated inside this layer's input transform
                B_, H_, W_ = pending_first.shape[:3]
                timing = None
                if self.conv_timing is not None:
                    timing = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), 2.0 * B_ * H_ * W_ * c_i * 27)
                    timing[0].record()
                if (self.fused_conv12 and self.winograd_mfma and name in self.defer_act and (c_i, c_o) == (64, 64)
                        and pending_first.first[1].shape[3] == 64 and H_ % 16 == 0 and W_ % 16 == 0):
                    # a 2x2 max_pool follows: conv1_1 -> conv1_2 -> pool1 as one LDS-resident kernel (csrc/conv_first.hip), launched
                    # by that max_pool; the library's own events time it
                    return _RawConv(None, b, relu, lazy12=(pending_first, name, w))
                v = self._first_conv_v(pending_first)
                return self._winograd43_tail(name, v, w, b, relu, B_, H_, W_, c_i, c_o, timing)
            input = self._activate(pending_first)
        if (wino_ok and input.is_cuda
                and self.winograd_min_channels and c_i >= self.winograd_min_channels and c_i % 4 == 0 and c_o % 4 == 0
                and (self.winograd_tile == 4 or (input.shape[1] % 2 == 0 and input.shape[2] % 2 == 0))
                and not (torch.is_grad_enabled() and (w.requires_grad or input.requires_grad))):
            B_, H_, W_, _ = input.shape
            tile = self.winograd_tile
            timing = None
            if self.conv_timing is not None:
                timing = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), 0.0)
                timing[0].record()
            v = ops.winograd_input(input, tile)
            if tile == 4:
                return self._winograd43_tail(name, v, w, b, relu, B_, H_, W_, c_i, c_o, timing)
            m = torch.bmm(v, self._winograd_filter(name, w))
            if timing is not None:
                timing[1].record()
                self.conv_timing.append((name, 2.0 * m.numel() * c_i, 2.0 * B_ * H_ * W_ * c_o * c_i * 9, timing[0], timing[1]))
            if name in self.defer_act:
                return _RawConv(None, b, relu, wino=(m, B_, H_, W_))
            return ops.winograd_output(m, b, B_, H_, W_, relu, pool=False, tile=tile)
        if self.conv_timing is not None and input.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = _nhwc(F.conv2d(_nchw(input), w, None, stride=(s_h, s_w), padding=pad, groups=group))
            e1.record()
            flops = 2.0 * y.numel() * (c_i // group) * k_h * k_w
            self.conv_timing.append((name, flops, flops, e0, e1))
        else:
            y = _nhwc(F.conv2d(_nchw(input), w, None, stride=(s_h, s_w), padding=pad, groups=group))
        if b is not None and name in self.defer_act and not (torch.is_grad_enabled() and y.requires_grad):
            return _RawConv(y, b, relu)
        if b is None and not relu:
            return y
        if b is None:
            return F.relu_(y)
        return self._bias_act(y, b, relu)  # bias_add + relu (network.py:181-187) in one pass, in place

    @layer
    def max_pool(self, input, k_h, k_w, s_h, s_w, name, padding=DEFAULT_PADDING):
        assert padding in ("SAME", "VALID")
        if isinstance(input, _RawConv):
            if input.pooled is not None and (k_h, k_w, s_h, s_w) == (2, 2, 2, 2):
                return input.pooled   # written by the convolution's own kernel (pool = 2)
            if (input.out is None and (k_h, k_w, s_h, s_w) == (2, 2, 2, 2)
                    and input.shape[1] % 2 == 0 and input.shape[2] % 2 == 0):
                if input.first is not None:
                    return F.max_pool2d(_nchw(self._activate(input)), 2, 2).permute(0, 2, 3, 1).contiguous()
                if input.gemm is not None:
                    v, ut, B_, H_, W_ = input.gemm
                    return ops.winograd43_conv(v, ut, input.bias, B_, H_, W_, input.relu, pool=1)
                if input.lazy12 is not None:
                    pf, name12, w2 = input.lazy12
                    ut = self._winograd_filter(name12, w2, transposed=True)
                    frag = self._wino_u.get((name12, "frag"))
                    if frag is None or frag[0] is not ut:   # the filter bank in the order the kernel's lanes consume it
                        frag = (ut, ops.conv12_fragment_major(ut))
                        self._wino_u[(name12, "frag")] = frag
                    x0, w1 = pf.first
                    if _is_raw(x0):
                        return ops.conv1_1_conv1_2_fused_raw(x0 if x0.dtype == torch.uint8 else None, x0 if x0.dtype == torch.uint16 else None,
                                                             w1, pf.bias, frag[1], input.bias, pf.relu, input.relu, ut2_layout=1)
                    return ops.conv1_1_conv1_2_fused(x0, w1, pf.bias, frag[1], input.bias, pf.relu, input.relu, groups=1, ut2_layout=1)
                if input.wino is not None and input.dual and self.winograd_tile == 4:
                    m, B_, H_, W_ = input.wino
                    input.out, pooled = ops.winograd43_output_both(m, input.bias, B_, H_, W_, input.relu)
                    return pooled
                if input.wino is not None:  # a Winograd output tile is exactly one pooling window
                    m, B_, H_, W_ = input.wino
                    return ops.winograd_output(m, input.bias, B_, H_, W_, input.relu, pool=True, tile=self.winograd_tile)
                return self._bias_relu_pool2(input.y, input.bias, input.relu)
            input = self._activate(input)
        H, W = input.shape[1], input.shape[2]
        if padding == "SAME" and (H % s_h or W % s_w):
            # TF 'SAME' pads at the bottom/right with -inf; not reached after pad_im(., 16)
            input = F.pad(input, (0, 0, 0, (-W) % s_w, 0, (-H) % s_h), value=float("-inf"))
        return _nhwc(F.max_pool2d(_nchw(input), (k_h, k_w), (s_h, s_w)))

    @layer
    def deconv(self, input, k_h, k_w, c_o, s_h, s_w, name, reuse=None, padding=DEFAULT_PADDING, trainable=True):
        """conv2d_transpose with the fixed bilinear filter of make_deconv_filter (network.py:141-157):
        weights[:, :, i, i] = outer(bilinear), zero elsewhere, so the dense [k,k,c_o,c_i]
        contraction equals a per-channel (depthwise) transposed convolution — adding the exact
        zeros of the off-diagonal taps changes no bit. 'SAME' with output = input * stride is
        PyTorch padding (k - s) / 2."""
        assert padding == "SAME"
        c_i = input.shape[-1]
        assert c_i == c_o and k_h == k_w and s_h == s_w, "PoseCNN deconvs are square, channel preserving"
        key = name + "/weights"
        if key in self.vars and self.vars[key].dim() == 4 and self.vars[key].shape[1] == c_i and c_i > 1:
            w = self.vars[key]  # a loaded dense [c_in, c_out, k, k] filter (checkpoint): honour it
            y = F.conv_transpose2d(_nchw(input), w, None, stride=(s_h, s_w), padding=((k_h - s_h) // 2, (k_w - s_w) // 2))
            return _nhwc(y)
        # the fixed filter: hand-written gfx950 interpolation kernel (MIOpen's depthwise
        # conv_transpose took 45 ms per call here — 83 % of the first profiled pipeline)
        return self._deconv_bilinear(input, k_h, s_h)

    # hooks the CPU checker overrides (tests/cpu_reference.py)
    def _bias_act(self, y, bias, relu):
        if torch.is_grad_enabled() and (y.requires_grad or bias.requires_grad):
            y = y + bias   # training: leave bias_add / relu to autograd
            return F.relu(y) if relu else y
        return ops.bias_act_(y, bias, relu)

    def _bias_relu_pool2(self, y, bias, relu):
        return ops.bias_relu_pool2(y, bias, relu)

    def _conv_first(self, x, w, bias, relu):
        # torch filter [c_o, c_i, kh, kw] -> the TF variable layout [kh, kw, c_i, c_o] the kernel reads
        return ops.conv3x3_c3(x.contiguous(), w.permute(2, 3, 1, 0).contiguous(), bias, relu)

    def _deconv_bilinear(self, x, k, s, add1=None, add2=None, bias=None, relu=False):
        return ops.deconv_bilinear(x, k, s, add1=add1, add2=add2, bias=bias, relu=relu)

    def _upscore_softmax_argmax(self, z, bias, k, s, relu=True, want_score=False, want_prob=True, hard_gt=None):
        return ops.upscore_softmax_argmax(z, bias, k, s, relu=relu, want_score=want_score, want_prob=want_prob,
                                          hard_gt=hard_gt, hard_threshold=self.threshold_label if hard_gt is not None else None)

    def _fused_hard_gt(self, z, k, s):
        """gt_label_2d when the `gt_label_weight` layer (hard_label on prob_normalized, vgg16_convs.py:148-149) can ride in
        the label head's launch: the layer is wanted (a with_losses graph, or fcn.im_segment_batch(with_losses=True) via
        `fuse_hard_label`), everything is on the GPU and nothing records a graph for autograd (the op's gradient is zero,
        but prob_normalized's own producer is then a framework op)."""
        if not (self.with_losses or self.fuse_hard_label) or torch.is_grad_enabled():
            return None
        g = self.layers.get('gt_label_2d')
        if not (isinstance(g, torch.Tensor) and g.is_cuda and z.is_cuda and g.dtype == torch.int32):
            return None
        B, H, W, _ = z.shape
        return g if tuple(g.shape) == (B, H * s, W * s) else None

    @layer
    def fc(self, input, num_out, name, num_in=-1, height=-1, width=-1, channel=-1, reuse=None, relu=True, trainable=True):
        if isinstance(input, tuple):
            input = input[0]
        if height > 0 and width > 0 and channel > 0:
            input = input.reshape(input.shape[0], height, width, channel)
        if input.dim() == 4:
            dim = input.shape[1] * input.shape[2] * input.shape[3]
            feed_in = input.reshape(-1, dim)  # NHWC flatten order, network.py:399-408
        else:
            dim = int(input.shape[-1]) if num_in == -1 else int(num_in)
            feed_in = input
        w = self.make_var(name + "/weights", (dim, num_out), self._weight_init(dim), trainable)
        b = self.make_var(name + "/biases", (num_out,), lambda s: torch.zeros(s), trainable)
        rows = getattr(self, "rows_count", None)
        route = self._fc_route(feed_in.is_cuda, feed_in.shape[0] if feed_in.dim() == 2 else None, dim, num_out,
                               self._fc_wants_grad(w, layer_trainable=trainable, feed_requires_grad=feed_in.requires_grad), rows is not None)
        if route == "skinny":
            # a handful of rows (the single-frame loop): the layer is a weight stream — csrc/fc_skinny.hip
            return ops.fc_skinny(feed_in.contiguous(), self._fc_wt(name, w), b, "relu" if relu else "none", num_rows=rows)
        if route == "rows":
            # capacity-sized rows behind the sync-free Hough layer: only the first *rows_count rows exist
            return ops.fc_rows(feed_in.contiguous(), self._fc_wt(name, w), b, relu, num_rows=rows)
        y = torch.addmm(b, feed_in, w)
        return F.relu(y) if relu else y

    def _fc_wants_grad(self, w, layer_trainable=True, feed_requires_grad=False):
        """Will a product with this weight be recorded for autograd? `w` may not exist yet (None) or may not carry its flag
        yet: `make_var` turns it on at the layer's first call when both the network and the layer are trainable."""
        will = (w is not None and w.requires_grad) or (self.trainable and layer_trainable and (w is None or w.is_leaf))
        return torch.is_grad_enabled() and (will or feed_requires_grad)

    def _fc_route(self, is_cuda, n_rows, dim, num_out, wants_grad, has_count):
        """Which implementation `fc` takes — ONE predicate for `fc` itself and for `fc_masks_dead_rows` (the two used
        to restate it separately): "skinny" (csrc/fc_skinny.hip: a device-counted buffer of at most SKINNY_MAX_ROWS rows),
        "rows" (csrc/fc_mfma.hip: any device-counted buffer whose shapes the kernel takes) or "addmm" (the framework: every
        trainable graph under autograd, CPU tensors, odd shapes). The two library routes never read a row at or past the count."""
        if not is_cuda or wants_grad or not has_count:
            return "addmm"
        if self.fc_skinny and n_rows is not None and 1 <= n_rows <= ops.SKINNY_MAX_ROWS and dim % 16 == 0:
            return "skinny"
        if dim % 64 == 0 and dim >= 128 and num_out % 64 == 0:
            return "rows"
        return "addmm"

    def fc_masks_dead_rows(self, name, dim, num_out, rows=None):
        """Will `fc(name)` on a capacity-sized, device-counted row buffer of `rows` rows run on the library's row kernels — which
        never read the rows at or past the count — or on the framework's addmm, which reads all of them? The caller
        (fcn.im_segment_batch) leaves dead rows of the pooled features unwritten only in the first case. Decided by `fc`'s own
        predicate (`_fc_route`), with the gradient flag the variable WILL have once `fc` has created / flagged it."""
        w = self.vars.get(name + "/weights")
        dev_ok = torch.device(self.device).type == "cuda"
        return self._fc_route(dev_ok, rows, dim, num_out, self._fc_wants_grad(w), True) != "addmm"

    def _fc_wt(self, name, w):
        """The TF weight variable [in, out] transposed to [out, in] (K contiguous rows for the kernels), cached."""
        key = (w.data_ptr(), w._version)
        hit = self._fc_cache.get(("fc", name))
        if hit is None or hit[0] != key:
            hit = (key, w.detach().t().contiguous())
            self._fc_cache[("fc", name)] = hit
        return hit[1]

    def _fc_skinny_ok(self, feed_in, dim, w):
        return (self.fc_skinny and getattr(self, "rows_count", None) is not None and feed_in.is_cuda and feed_in.dim() == 2
                and 1 <= feed_in.shape[0] <= ops.SKINNY_MAX_ROWS and dim % 16 == 0
                and not (torch.is_grad_enabled() and (w.requires_grad or feed_in.requires_grad)))

    def fc_tanh(self, num_out, name, tanh_name, num_in=-1):
        """`.fc(num_out, relu=False, name=name).tanh(name=tanh_name)` (fc8 -> poses_tanh, vgg16_convs.py:192-193);
        on the few-row path both come out of one launch."""
        x = self.inputs[0]
        if isinstance(x, tuple):
            x = x[0]
        dim = int(x.shape[-1]) if num_in == -1 else int(num_in)
        w = self.make_var(name + "/weights", (dim, num_out), self._weight_init(dim), True)
        b = self.make_var(name + "/biases", (num_out,), lambda s: torch.zeros(s), True)
        if x.dim() == 2 and self._fc_skinny_ok(x, dim, w):
            y, t = ops.fc_skinny(x.contiguous(), self._fc_wt(name, w), b, "tanh", num_rows=self.rows_count)
            self.layers[name], self.layers[tanh_name] = y, t
            return self.feed(t)
        if (x.dim() == 2 and getattr(self, "rows_count", None) is not None and x.is_cuda and dim % 64 == 0 and dim >= 128
                and num_out % 4 == 0 and not (torch.is_grad_enabled() and (w.requires_grad or x.requires_grad))):
            # more rows than the skinny kernel takes (a batch): the row kernel on a zero-padded filter, tanh in its epilogue
            key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
            hit = self._fc_cache.get(("fc_pad", name))
            if hit is None or hit[0] != key:
                npad = (num_out + 63) // 64 * 64
                wp = torch.zeros((npad, dim), dtype=torch.float32, device=w.device)
                wp[:num_out] = w.detach().t()
                bp = torch.zeros((npad,), dtype=torch.float32, device=w.device)
                bp[:num_out] = b.detach()
                hit = (key, wp, bp)
                self._fc_cache[("fc_pad", name)] = hit
            y, t = ops.fc_rows_cols(x.contiguous(), hit[1], hit[2], num_out, "tanh", num_rows=self.rows_count)
            self.layers[name], self.layers[tanh_name] = y, t
            return self.feed(t)
        return self.fc(num_out, relu=False, name=name, num_in=num_in).tanh(name=tanh_name)

    # ---- element-wise -----------------------------------------------------------------------------
    @layer
    def relu(self, input, name):
        return F.relu(input)

    @layer
    def concat(self, inputs, axis, name):
        inputs = [i[0] if isinstance(i, tuple) else i for i in inputs]
        return torch.cat(inputs, dim=axis)

    @layer
    def add(self, inputs, name):
        inputs = [i[0] if isinstance(i, tuple) else i for i in inputs]
        out = inputs[0] + inputs[1]  # tf.add_n
        for t in inputs[2:]:
            out = out + t
        return out

    @layer
    def multiply(self, input, name):
        a = input[0][0] if isinstance(input[0], tuple) else input[0]
        b = input[1][0] if isinstance(input[1], tuple) else input[1]
        return a * b

    @layer
    def l2_normalize(self, input, dim, name):
        # tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))
        return input * torch.rsqrt(torch.clamp((input * input).sum(dim=dim, keepdim=True), min=1e-12))

    @layer
    def dropout(self, input, keep_prob, name):
        if isinstance(input, tuple):
            input = input[0]
        if keep_prob is None or float(keep_prob) >= 1.0:
            return input  # keep_prob is fed as 1.0 at test time (lib/fcn/test.py:173-184)
        return F.dropout(input, 1.0 - float(keep_prob), training=True)

    @layer
    def tanh(self, input, name):
        if isinstance(input, tuple):
            input = input[0]
        return torch.tanh(input)

    @layer
    def softmax_high_dimension(self, input, num_classes, name):
        if isinstance(input, tuple):
            input = input[0]
        prob, label = ops.softmax_argmax(input, want_prob=True)
        self._argmax_cache = (prob, label)  # argmax_2d of this tensor comes out of the same pass
        return prob

    @layer
    def log_softmax_high_dimension(self, input, num_classes, name):
        if isinstance(input, tuple):
            input = input[0]
        return torch.log_softmax(input, dim=-1)

    @layer
    def argmax_2d(self, input, name):
        cache = getattr(self, "_argmax_cache", None)
        if cache is not None and cache[0] is input:
            return cache[1]
        return torch.argmax(input, dim=3).to(torch.int32)   # tf.argmax: first maximum

    # ---- custom layers (gfx950 kernels) ------------------------------------------------------------
    @layer
    def hough_voting_gpu(self, input, is_train, threshold, per_threshold, skip_pixels, name):
        return ops.hough_voting_gpu(input[0], input[1], input[2], input[3], input[4], is_train, threshold,
                                    per_threshold, skip_pixels, name=name)

    @layer
    def hough_voting_gpu_lowres(self, input, kernel, stride, is_train, threshold, per_threshold, skip_pixels, name):
        """Hough layer fed by the 1/stride-resolution vertex field (+ bias) instead of `vertex_pred`;
        same outputs as hough_voting_gpu on deconv(z) + bias (include/posecnn_hip.h)."""
        out = ops.hough_voting_gpu_lowres_padded(input[0], input[1], input[2], kernel, stride, input[3], input[4],
                                                 input[5], is_train, threshold, per_threshold, skip_pixels)
        r = int(out[5][0].item())
        return tuple(o[:r] for o in out[:5])

    @layer
    def roi_pool(self, input, pooled_height, pooled_width, spatial_scale, pool_channel, name):
        if isinstance(input[0], tuple):
            input[0] = input[0][0]
        return ops.roi_pool(input[0], input[1], pooled_height, pooled_width, spatial_scale, pool_channel, name=name)

    @layer
    def hard_label(self, input, threshold, name):
        return ops.hard_label(input[0], input[1], threshold, name=name)

    @layer
    def average_distance_loss(self, input, margin, name):
        return ops.average_distance_loss(input[0], input[1], input[2], input[3], input[4], margin, name=name)

    @layer
    def backproject(self, input, grid_size, kernel_size, threshold, name):
        return ops.backproject(input[0], input[1], input[2], input[3], input[4], grid_size, kernel_size, threshold, name=name)


class vgg16_convs(Network):
    """The PoseCNN network of tools/demo.py / tools/test_net.py (lib/networks/vgg16_convs.py).

    Constructor arguments follow vgg16_convs.py:5. `run(feed)` plays the role of
    `sess.run(net.enqueue_op, feed_dict)` + `sess.run([...])`: it binds the placeholders
    (data, [data_p], gt_label_2d, keep_prob, vertex_targets/weights, poses, extents, meta_data,
    points, symmetry) and evaluates the graph; outputs are read with `get_output(name)`.
    """

    def __init__(self, input_format, num_classes, num_units, scales, threshold_label, vote_threshold,
                 vertex_reg_2d=False, vertex_reg_3d=False, pose_reg=False, adaptation=False, trainable=True,
                 is_train=True, device="cuda", seed=3, init="he", with_losses=None, fused_heads=True,
                 want_prob=True, fused_pool=True, strict_numerics=False):
        Network.__init__(self, device=device, seed=seed, init=init, trainable=trainable)
        # strict_numerics: every 3x3 layer as a direct f32 convolution (library kernels) instead of Winograd
        # F(4x4,3x3). Both are exact in real arithmetic and f32 throughout; they differ in summation order
        # and Winograd's transform constants amplify rounding ~3x (DESIGN.md §4 has the measured end-to-end
        # distributions against a float64 trunk). bench.py runs the default (Winograd) and says so.
        self.strict_numerics = bool(strict_numerics)
        if self.strict_numerics:
            self.winograd_min_channels = 0
        # validation aid (tests/parity_study.py): None, or a torch dtype — the trunk is then evaluated as nine
        # shifted GEMMs per layer in that dtype (float64: the real-arithmetic reference; float32: a direct
        # convolution whose summation order is known), activations kept in that dtype between layers
        self.reference_trunk = None
        # conv -> max_pool pairs whose un-pooled activation nobody else consumes: one kernel does
        # bias + ReLU + 2x2 max from the raw convolution output (conv4_3 feeds score_conv4 and
        # roi_pool, so pool4 stays a plain max_pool)
        if fused_pool:
            self.defer_act = frozenset(n + sfx for n in ("conv1_2", "conv2_2", "conv3_3") for sfx in ("", "_p"))
            # conv4_3 feeds score_conv4 and roi_pool un-pooled AND pool4: the MFMA kernel writes both
            self.dual_pool = frozenset(("conv4_3", "conv4_3_p"))
        # fused_heads=False evaluates the heads in the reference's literal op order
        # (deconv -> 1x1 conv -> softmax -> argmax); True (default) uses the algebraically
        # identical low-resolution form + fused gfx950 epilogue (see setup()).
        # (the fused epilogues have no backward: a trainable training graph keeps the literal order)
        self.fused_heads = fused_heads and not (is_train and trainable)
        self.want_prob = want_prob  # prob_normalized is a fetched output in lib/fcn/test.py:193-195
        self.input_format = input_format
        self.num_classes = num_classes
        self.num_units = num_units
        self.scale = 1.0
        self.threshold_label = threshold_label
        self.vertex_reg_2d = vertex_reg_2d
        self.vertex_reg_3d = vertex_reg_3d
        self.vertex_reg = vertex_reg_2d or vertex_reg_3d
        self.pose_reg = pose_reg
        self.adaptation = adaptation
        self.is_train = 1 if is_train else 0
        self.skip_pixels = 10          # vgg16_convs.py:22,27
        self.vote_threshold = vote_threshold
        self.vote_percentage = 0.02    # vgg16_convs.py:24,29
        # TF evaluates only fetched tensors: the two training-loss layers (hard_label,
        # average_distance_loss) run at train time; at test time only when asked for.
        self.with_losses = bool(is_train) if with_losses is None else bool(with_losses)
        self.planted = None
        self.grouped_towers = True    # RGB-D inference: both towers as one grouped launch sequence
        self.merge_head_convs = True  # COLOR: the label and vertex 1x1 convolutions of a source as one product (fc_rows_split)
        self.fuse_hard_label = False  # fcn.im_segment_batch(with_losses=True) on a graph built without the loss layers: gt_label_weight from the label head's launch
        self.mfma_heads = True        # big batches: add_score / add_score_vertex + the 1/8-resolution `score` / `vertex_pred` products in one launch per head
        self.head_gemm = True         # 1x1 head convs (incl. the RGB-D concat) on the library's own fp32-MFMA row kernel
        self._head_wt = {}

    def run(self, feed, planted=None):
        self.layers = dict(feed)
        self.inputs = []
        self.keep_prob_queue = feed.get("keep_prob", 1.0)
        self.planted = planted
        self._argmax_cache = None
        self.setup()
        return self

    def _conv1x1_lowres(self, x, name, c_o, c_i):
        """The 1x1 conv `name` (same variables as Network.conv would create) applied WITHOUT bias
        or ReLU; returns (z, bias) for the fused deconv epilogue."""
        w = self.make_var(name + "/weights", (c_o, c_i, 1, 1),
                          lambda s: self._weight_init(c_i)(s).contiguous(memory_format=torch.channels_last))
        b = self.make_var(name + "/biases", (c_o,), lambda s: torch.zeros(s))
        return _nhwc(F.conv2d(_nchw(x), w, None)), b

    def _head_conv1x1(self, sources, c_o, name, relu=True):
        """`[concat(3) ->] conv(1, 1, c_o, 1, 1, name)` over the channel concatenation of `sources`
        (vgg16_convs.py:104-113 for RGB-D, :128-133 for COLOR) — same variables as Network.conv creates.
        On the GPU inference path a 1x1 convolution is a row product, and one over a concatenation is
        the SUM of one product per source: pcnn_fc_rows_fwd runs them back to back (`addend`), so the
        [B,h,w,1024] concatenation is neither written nor re-read, and bias + ReLU ride in the last
        product's epilogue."""
        xs = [self.get_output(n) for n in sources]
        cis = [int(x.shape[-1]) for x in xs]
        c_i = sum(cis)
        fast = (self.head_gemm and all(isinstance(x, torch.Tensor) and x.is_cuda for x in xs) and c_o % 64 == 0
                and all(c % 64 == 0 and c >= 128 for c in cis) and not (torch.is_grad_enabled() and self.trainable))
        if not fast:
            if len(sources) > 1:
                self.feed(*sources).concat(3, name='concat_' + sources[0].split('_')[0])
            else:
                self.feed(sources[0])
            return self.conv(1, 1, c_o, 1, 1, name=name, relu=relu, c_i=c_i)
        w = self.make_var(name + "/weights", (c_o, c_i, 1, 1),
                          lambda s: self._weight_init(c_i)(s).contiguous(memory_format=torch.channels_last), self.trainable)
        b = self.make_var(name + "/biases", (c_o,), lambda s: torch.zeros(s), self.trainable)
        key = (w.data_ptr(), w._version)
        hit = self._head_wt.get(name)
        if hit is None or hit[0] != key:
            w2 = w.detach().reshape(c_o, c_i)
            parts, off = [], 0
            for c in cis:
                parts.append(w2[:, off:off + c].contiguous())
                off += c
            hit = (key, parts, torch.zeros_like(b))
            self._head_wt[name] = hit
        parts, zero_b = hit[1], hit[2]
        B_, h, w_ = xs[0].shape[:3]
        partial = None
        for j in range(len(xs) - 1, 0, -1):     # the other towers first, as addends
            partial = ops.fc_rows(xs[j].reshape(-1, cis[j]), parts[j], zero_b, relu=False, addend=partial)
        out = ops.fc_rows(xs[0].reshape(-1, cis[0]), parts[0], b.detach(), relu=relu, addend=partial).view(B_, h, w_, c_o)
        self.layers[name] = out
        return self.feed(out)

    def _plant(self, key, name):
        """Benchmark aid (DESIGN.md §synthetic workload): add a low-resolution synthetic scene to
        the 1/8-resolution head features so that random-weight networks hand the Hough layer
        object-like label/vertex maps. No-op unless `run(..., planted=...)` supplies tensors."""
        if self.planted is not None and key in self.planted:
            t = self.layers[name] + self.planted[key]
            self.layers[name] = t
            self.feed(t)
        return self

    # the 3x3 layers of a VGG16 tower (vgg16_convs.py:36-52): (name, c_in, c_out, pool that follows)
    TRUNK = (("conv1_1", 3, 64, None), ("conv1_2", 64, 64, "pool1"), ("conv2_1", 64, 128, None), ("conv2_2", 128, 128, "pool2"),
             ("conv3_1", 128, 256, None), ("conv3_2", 256, 256, None), ("conv3_3", 256, 256, "pool3"),
             ("conv4_1", 256, 512, None), ("conv4_2", 512, 512, None), ("conv4_3", 512, 512, "pool4"),
             ("conv5_1", 512, 512, None), ("conv5_2", 512, 512, None), ("conv5_3", 512, 512, None))

    def _trunk_vars(self, suffix):
        """Creates (on first use) the variables of one tower in the order Network.conv would."""
        t = self.trainable
        out = []
        for name, ci, co, _ in self.TRUNK:
            w = self.make_var(name + suffix + "/weights", (co, ci, 3, 3),
                              lambda s_, ci=ci: self._weight_init(ci * 9)(s_).contiguous(memory_format=torch.channels_last), t)
            b = self.make_var(name + suffix + "/biases", (co,), lambda s_: torch.zeros(s_), t)
            out.append((w, b))
        return out

    def _can_group_towers(self):
        d = self.layers.get('data')
        dp = self.layers.get('data_p')
        same = isinstance(dp, torch.Tensor) and (dp.shape == d.shape if not _is_raw(d) else
                                                 (dp.dtype == torch.uint16 and tuple(dp.shape[:3]) == tuple(d.shape[:3]))) if isinstance(d, torch.Tensor) else False
        return (self.grouped_towers and self.input_format == 'RGBD' and isinstance(d, torch.Tensor) and d.is_cuda
                and self.winograd_mfma and self.winograd_tile == 4 and self.winograd_min_channels == 64
                and self.fused_first_conv and self.fuse_first_conv_into_winograd
                and not (torch.is_grad_enabled() and self.trainable)
                and d.shape[1] % 16 == 0 and d.shape[2] % 16 == 0 and same)

    def _trunk_grouped(self):
        """Both VGG16 towers of an RGB-D network (vgg16_convs.py:36-52 and :53-67) as ONE launch
        sequence: the colour and the depth blob are stacked along the batch axis and every trunk
        kernel runs once with `groups = 2` filter sets (image b uses set b // B). Identical arithmetic
        to running the towers one after the other — each image still meets only its tower's weights —
        with half the launches and twice the workgroups per launch (conv5_x alone has only 160
        workgroups per tower for the 256 CUs). Registers conv4_3 / conv5_3 (+ '_p') and pool4."""
        d, dp = self.get_output('data'), self.get_output('data_p')
        raw = _is_raw(d)                        # uint8 colour + uint16 depth frames: the first kernel forms the blobs itself
        if raw:
            x = None
            B, H, W = d.shape[:3]
            B2 = 2 * B
        else:
            x = pipeline.stacked_view(d, dp)        # free when the uploader placed the blobs back to back
            if x is None:
                x = torch.cat([d, dp], dim=0)
            B2, H, W, _ = x.shape
            B = B2 // 2
        va, vb = self._trunk_vars(""), self._trunk_vars("_p")
        key = tuple((w.data_ptr(), w._version, b.data_ptr(), b._version) for w, b in va + vb)
        hit = self._wino_u.get("grouped")
        if hit is None or hit[0] != key:
            packed = []
            for (name, ci, co, _), (w0, b0), (w1, b1) in zip(self.TRUNK, va, vb):
                bias = torch.stack([b0, b1]).contiguous()
                if ci == 3:   # the TF variable layout [ky, kx, ci, co] the first-conv kernel reads
                    wt = torch.stack([w0.permute(2, 3, 1, 0), w1.permute(2, 3, 1, 0)]).contiguous()
                else:
                    wt = torch.stack([ops.winograd_filter(w0, 4).transpose(1, 2), ops.winograd_filter(w1, 4).transpose(1, 2)]).contiguous()
                packed.append((wt, bias))
            hit = (key, packed)
            self._wino_u["grouped"] = hit
        packed = hit[1]
        y, h, w_ = None, H, W
        for li, (name, ci, co, pool) in enumerate(self.TRUNK):
            wt, bias = packed[li]
            if ci == 3:
                continue   # conv1_1 is evaluated inside conv1_2's input transform
            if li == 1 and self.fused_conv12 and pool == "pool1" and co == 64 and H % 16 == 0 and W % 16 == 0:
                # conv1_1 -> conv1_2 -> pool1 in one kernel: V (2.25 x 78.6 MB per frame) never touches HBM (csrc/conv_first.hip)
                frag = self._wino_u.get("conv12_frag")
                if frag is None or frag[0] is not wt:
                    # the filter bank fragment-major (one contiguous KB per B-operand load)
                    frag = (wt, ops.conv12_fragment_major(wt), 1)
                    self._wino_u["conv12_frag"] = frag
                y = (ops.conv1_1_conv1_2_fused_raw(d, dp, packed[0][0], packed[0][1], frag[1], bias, ut2_layout=frag[2]) if raw
                     else ops.conv1_1_conv1_2_fused(x, packed[0][0], packed[0][1], frag[1], bias, groups=2, ut2_layout=frag[2]))
                h, w_ = h // 2, w_ // 2
                continue
            if li == 1:
                v = (ops.conv3x3_c3_winograd43_raw(d, dp, packed[0][0], packed[0][1], True) if raw
                     else ops.conv3x3_c3_winograd43(x, packed[0][0], packed[0][1], True, groups=2))
            else:
                v = ops.winograd_input(y, 4)
            # conv4_3 is read un-pooled by score_conv4 and roi_pool AND pooled by pool4: the kernel writes
            # both (mode 2) whatever `dual_pool` / `fused_pool` say; the other conv -> pool pairs keep only
            # the pooled tensor (mode 1)
            mode = 0 if pool is None else (2 if name == "conv4_3" else 1)
            out = ops.winograd43_conv(v, wt, bias, B2, h, w_, True, pool=mode, groups=2)
            if mode == 2:
                full, y = out
            else:
                full = y = out
            if name in ("conv4_3", "conv5_3"):
                self.layers[name], self.layers[name + "_p"] = full[:B], full[B:]
            if pool is not None:
                h, w_ = h // 2, w_ // 2
                if pool == "pool4":
                    self.layers["pool4"], self.layers["pool4_p"] = y[:B], y[B:]
        return self

    def hbm_table(self, B, H, W):
        """Algorithmic HBM bytes per step of the library's streaming trunk kernels (for bench.py)."""
        towers = 2 if self.input_format == 'RGBD' else 1
        act = lambda div, ch: 4.0 * B * (H // div) * (W // div) * ch
        x_in = (act(2, 64) + act(2, 128) + act(4, 128) + 2 * act(4, 256) + act(8, 256) + 2 * act(8, 512) + 3 * act(16, 512))
        fused = self.fused_conv12 and self.winograd_mfma and H % 16 == 0 and W % 16 == 0   # (the first two layers as one kernel)
        t = {"wino43_input_kernel": towers * 3.25 * x_in}                             # reads X, writes V = 2.25 X
        if not fused:
            t["conv3x3_c3_wino43_kernel"] = towers * (act(1, 3) + 2.25 * act(1, 64))      # reads the frame, writes V of conv1_2
        return t

    def mfma_table(self, B, H, W):
        """Executed flops per step of the library's fp32-MFMA kernels (for bench.py)."""
        towers = 2 if self.input_format == 'RGBD' else 1
        tiles = lambda div: B * ((H // div + 3) // 4) * ((W // div + 3) // 4)
        div = {"1": 1, "2": 2, "3": 4, "4": 8, "5": 16}
        fused = self.fused_conv12 and self.winograd_mfma and H % 16 == 0 and W % 16 == 0
        fl = sum(2.0 * 36 * tiles(div[n[4]]) * ci * co for n, ci, co, _ in self.TRUNK if ci != 3 and not (fused and n == "conv1_2"))
        t = {"wino43_mfma_kernel": towers * fl}
        if fused:   # conv1_1 + conv1_2 + pool1 in one kernel: its matrix work is conv1_2's Winograd-domain contraction
            t["conv12_wino43_fused_kernel"] = towers * 2.0 * 36 * tiles(1) * 64 * 64
        return t

    def _trunk_reference(self, dtype):
        """The VGG16 tower(s) (vgg16_convs.py:36-67) as plain tensor algebra in `dtype`, image by image:
        conv3x3 'SAME' = sum over the 9 taps of [H*W, Cin] x [Cin, Cout] products of the zero-padded input,
        + bias, ReLU, 2x2 max-pool — no Winograd, no library convolution, activations never leave `dtype`
        until conv4_3 / pool4 / conv5_3 are handed to the heads as f32. Slow by design (validation only)."""
        towers = [("", "data")] + ([("_p", "data_p")] if self.input_format == "RGBD" else [])
        for sfx, src in towers:
            x = raw_frame_blob(self.get_output(src))
            vars_ = [(w.detach().to(dtype), b.detach().to(dtype)) for w, b in self._trunk_vars(sfx)]
            keep = {"conv4_3": [], "pool4": [], "conv5_3": []}
            for n in range(x.shape[0]):
                y = x[n:n + 1].to(dtype)
                for (name, ci, co, pool), (w, b) in zip(self.TRUNK, vars_):
                    _, h, w_, _ = y.shape
                    yp = F.pad(y, (0, 0, 1, 1, 1, 1))
                    acc = None
                    for ky in range(3):
                        for kx in range(3):
                            t_ = yp[:, ky:ky + h, kx:kx + w_, :].reshape(-1, ci) @ w[:, :, ky, kx].t()
                            acc = t_ if acc is None else acc + t_
                    y = torch.relu(acc + b).view(1, h, w_, co)
                    if name in keep:
                        keep[name].append(y.to(torch.float32))
                    if pool is not None:
                        y = y.view(1, h // 2, 2, w_ // 2, 2, co).amax(dim=(2, 4))
                        if pool in keep:
                            keep[pool].append(y.to(torch.float32))
            for k, v in keep.items():
                self.layers[k + sfx] = torch.cat(v, dim=0).contiguous()
        return self

    def setup(self):
        t = self.trainable
        if self.reference_trunk is not None:
            self._trunk_reference(self.reference_trunk)
            return self._setup_heads()
        if self._can_group_towers():
            self._trunk_grouped()
            return self._setup_heads()
        (self.feed('data')
             .conv(3, 3, 64, 1, 1, name='conv1_1', c_i=3, trainable=t)
             .conv(3, 3, 64, 1, 1, name='conv1_2', c_i=64, trainable=t)
             .max_pool(2, 2, 2, 2, name='pool1')
             .conv(3, 3, 128, 1, 1, name='conv2_1', c_i=64, trainable=t)
             .conv(3, 3, 128, 1, 1, name='conv2_2', c_i=128, trainable=t)
             .max_pool(2, 2, 2, 2, name='pool2')
             .conv(3, 3, 256, 1, 1, name='conv3_1', c_i=128, trainable=t)
             .conv(3, 3, 256, 1, 1, name='conv3_2', c_i=256, trainable=t)
             .conv(3, 3, 256, 1, 1, name='conv3_3', c_i=256, trainable=t)
             .max_pool(2, 2, 2, 2, name='pool3')
             .conv(3, 3, 512, 1, 1, name='conv4_1', c_i=256, trainable=t)
             .conv(3, 3, 512, 1, 1, name='conv4_2', c_i=512, trainable=t)
             .conv(3, 3, 512, 1, 1, name='conv4_3', c_i=512, trainable=t)
             .max_pool(2, 2, 2, 2, name='pool4')
             .conv(3, 3, 512, 1, 1, name='conv5_1', c_i=512, trainable=t)
             .conv(3, 3, 512, 1, 1, name='conv5_2', c_i=512, trainable=t)
             .conv(3, 3, 512, 1, 1, name='conv5_3', c_i=512, trainable=t))

        if self.input_format == 'RGBD':
            (self.feed('data_p')
                 .conv(3, 3, 64, 1, 1, name='conv1_1_p', c_i=3, trainable=t)
                 .conv(3, 3, 64, 1, 1, name='conv1_2_p', c_i=64, trainable=t)
                 .max_pool(2, 2, 2, 2, name='pool1_p')
                 .conv(3, 3, 128, 1, 1, name='conv2_1_p', c_i=64, trainable=t)
                 .conv(3, 3, 128, 1, 1, name='conv2_2_p', c_i=128, trainable=t)
                 .max_pool(2, 2, 2, 2, name='pool2_p')
                 .conv(3, 3, 256, 1, 1, name='conv3_1_p', c_i=128, trainable=t)
                 .conv(3, 3, 256, 1, 1, name='conv3_2_p', c_i=256, trainable=t)
                 .conv(3, 3, 256, 1, 1, name='conv3_3_p', c_i=256, trainable=t)
                 .max_pool(2, 2, 2, 2, name='pool3_p')
                 .conv(3, 3, 512, 1, 1, name='conv4_1_p', c_i=256, trainable=t)
                 .conv(3, 3, 512, 1, 1, name='conv4_2_p', c_i=512, trainable=t)
                 .conv(3, 3, 512, 1, 1, name='conv4_3_p', c_i=512, trainable=t)
                 .max_pool(2, 2, 2, 2, name='pool4_p')
                 .conv(3, 3, 512, 1, 1, name='conv5_1_p', c_i=512, trainable=t)
                 .conv(3, 3, 512, 1, 1, name='conv5_2_p', c_i=512, trainable=t)
                 .conv(3, 3, 512, 1, 1, name='conv5_3_p', c_i=512, trainable=t))
        return self._setup_heads()

    def _mfma_head_ok(self, c_i, c_o):
        return bool(self.mfma_heads) and c_i % 16 == 0 and c_o <= 96 and 4 * 64 * (c_i + 4) <= 60 * 1024

    def _small_head(self, s5, s4, up_name, add_name, drop_name, plant_key, conv_name, c_o, c_i):
        """`deconv(4,4,·,2,2)(s5) -> add(s4, ·) [-> plant] -> dropout(1.0)` and the bias-free 1x1 product `conv_name` of
        the fused-heads form, as ONE launch (ops.head_lowres). Registers the same layer names; the up-sampled
        tensor itself is only built if somebody fetches it. Returns (z, bias) like _conv1x1_lowres, or None when
        the launch does not apply (CPU checker, training graph, dropout active, loaded deconv filter)."""
        a, b5 = self.layers.get(s4), self.layers.get(s5)
        if not (self.small_heads and self.fused_heads and isinstance(a, torch.Tensor) and isinstance(b5, torch.Tensor)
                and a.is_cuda and a.dim() == 4 and a.shape[1] % 2 == 0 and a.shape[2] % 2 == 0
                # one launch instead of five (csrc/heads_small.hip). The 1x1 product runs on the matrix cores
                # (head_lowres_mfma_kernel, round 5: any batch) when the head fits it — units a multiple of 16, at most 96
                # outputs —, else, up to a frame or two (`small_heads_max_pixels`), on the vector ALUs out of LDS
                # (head_lowres_kernel, round 3: 32 pixels x U inputs and the U x Cout filter in 60 KB; at 16 frames it lost to
                # the library's 1x1 convolution, 237 vs 165 us). Heads that fit neither take deconv + add + 1x1.
                and c_i % 4 == 0
                and (self._mfma_head_ok(c_i, c_o)
                     or (a.shape[0] * a.shape[1] * a.shape[2] <= self.small_heads_max_pixels and 4 * (32 * c_i + c_i * c_o) <= 60 * 1024))
                and (self.keep_prob_queue is None or float(self.keep_prob_queue) >= 1.0)
                and (up_name + "/weights") not in self.vars
                and not (torch.is_grad_enabled() and self.trainable)):
            return None
        w = self.make_var(conv_name + "/weights", (c_o, c_i, 1, 1),
                          lambda s: self._weight_init(c_i)(s).contiguous(memory_format=torch.channels_last))
        b = self.make_var(conv_name + "/biases", (c_o,), lambda s: torch.zeros(s))
        key = (w.data_ptr(), w._version)
        hit = self._head_wt.get(("lowres", conv_name))
        if hit is None or hit[0] != key:
            hit = (key, w.detach().reshape(c_o, c_i).t().contiguous())     # [units, out]: the TF variable [1,1,in,out] as it is
            self._head_wt[("lowres", conv_name)] = hit
        planted = self.planted.get(plant_key) if self.planted is not None else None
        if not self._mfma_head_ok(c_i, c_o):
            add, z = ops.head_lowres(a, b5, hit[1], planted=planted, kernel=4, stride=2)
        else:
            hitm = self._head_wt.get(("lowres_mfma", conv_name))
            if hitm is None or hitm[0] != key:
                hitm = (key, ops.head_lowres_mfma_filter(hit[1]))
                self._head_wt[("lowres_mfma", conv_name)] = hitm
            add, z = ops.head_lowres_mfma(a, b5, hitm[1], c_o, planted=planted, kernel=4, stride=2)
        self.layers[up_name] = _Lazy(lambda: self._deconv_bilinear(b5, 4, 2))
        self.layers[add_name] = add
        self.layers[drop_name] = add
        self.feed(add)
        return z, b

    def _merged_head_convs(self):
        """COLOR networks (round 5, the single-frame loop's launch count): `score_conv5` [ReLU] and `score_conv5_vertex` [none] both
        read conv5_3, `score_conv4` / `score_conv4_vertex` both read conv4_3 (vgg16_convs.py:128-133,151-157) — one product per
        source instead of two (ops.fc_rows_split: the same kernel, the same bits per column). Only when every variable already
        exists (loaded / calibrated / a previous run): a lazily initialised network creates them in the reference's order first."""
        names = ('score_conv5', 'score_conv4', 'score_conv5_vertex', 'score_conv4_vertex')
        if not (self.input_format != 'RGBD' and self.vertex_reg and self.head_gemm and self.merge_head_convs
                and all((n + sfx) in self.vars for n in names for sfx in ('/weights', '/biases'))
                and not (torch.is_grad_enabled() and self.trainable)):
            return False
        srcs = {'conv5_3': ('score_conv5', 'score_conv5_vertex'), 'conv4_3': ('score_conv4', 'score_conv4_vertex')}
        xs = {k: self.get_output(k) for k in srcs}
        if not all(isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.shape[-1] % 64 == 0 and x.shape[-1] >= 128 for x in xs.values()):
            return False
        for src, (na, nb) in srcs.items():
            wa, wb = self.vars[na + '/weights'], self.vars[nb + '/weights']
            ba, bb = self.vars[na + '/biases'], self.vars[nb + '/biases']
            c_i = int(xs[src].shape[-1])
            if tuple(wa.shape[1:]) != (c_i, 1, 1) or tuple(wb.shape[1:]) != (c_i, 1, 1) or wa.shape[0] % 64 or wb.shape[0] % 64:
                return False
            key = tuple((t.data_ptr(), t._version) for t in (wa, wb, ba, bb))
            hit = self._head_wt.get(('merged', src))
            if hit is None or hit[0] != key:
                hit = (key, torch.cat([wa.detach().reshape(wa.shape[0], c_i), wb.detach().reshape(wb.shape[0], c_i)]).contiguous(),
                       torch.cat([ba.detach(), bb.detach()]).contiguous())
                self._head_wt[('merged', src)] = hit
            B_, h, w_ = xs[src].shape[:3]
            ya, yb = ops.fc_rows_split(xs[src].reshape(-1, c_i), hit[1], hit[2], wa.shape[0], relu_a=True, relu_b=False)
            self.layers[na] = ya.view(B_, h, w_, wa.shape[0])
            self.layers[nb] = yb.view(B_, h, w_, wb.shape[0])
        return True

    def _setup_heads(self):
        towers = ('', '_p') if self.input_format == 'RGBD' else ('',)
        merged = self._merged_head_convs()
        if not merged:
            self._head_conv1x1(['conv5_3' + t for t in towers], self.num_units, 'score_conv5')
            self._head_conv1x1(['conv4_3' + t for t in towers], self.num_units, 'score_conv4')
        small = self._small_head('score_conv5', 'score_conv4', 'upscore_conv5', 'add_score', 'dropout', 'add_score',
                                 'score', self.num_classes, self.num_units)
        if small is None:
            (self.feed('score_conv5')
                 .deconv(4, 4, self.num_units, 2, 2, name='upscore_conv5', trainable=False))

            (self.feed('score_conv4', 'upscore_conv5')
                 .add(name='add_score')
                 ._plant('add_score', 'add_score')
                 .dropout(self.keep_prob_queue, name='dropout'))

        if self.fused_heads:
            # deconv and the 1x1 `score` conv are both linear and act on different axes, so
            # conv1x1(deconv(x)) + b == deconv(conv1x1(x)) + b exactly in real arithmetic (borders
            # included): run the 64->C contraction at 1/8 resolution (64x fewer MACs) and let one
            # gfx950 kernel do deconv + bias + ReLU + softmax + argmax without ever writing the
            # [B,480,640,64] `upscore` or the full-resolution `score` to HBM.
            z, b = small if small is not None else self._conv1x1_lowres(self.get_output('dropout'), 'score', self.num_classes, self.num_units)
            k, s = int(16 * self.scale), int(8 * self.scale)
            hard_gt = self._fused_hard_gt(z, k, s)
            out = self._upscore_softmax_argmax(z, b, k, s, relu=True, want_score=self.with_losses,
                                               want_prob=self.want_prob, hard_gt=hard_gt)
            score, prob, label = out[:3]
            if hard_gt is not None:
                self.layers['gt_label_weight'] = out[3]
            if score is not None:
                self.layers['score'] = score
            self.layers['prob_normalized'] = prob
            self.layers['label_2d'] = label
            if self.with_losses:
                (self.feed('score')
                     .log_softmax_high_dimension(self.num_classes, name='prob'))
        else:
            (self.feed('dropout')
                 .deconv(int(16 * self.scale), int(16 * self.scale), self.num_units, int(8 * self.scale), int(8 * self.scale), name='upscore', trainable=False))

            (self.feed('upscore')
                 .conv(1, 1, self.num_classes, 1, 1, name='score', c_i=self.num_units))
            if self.with_losses:
                (self.feed('score')
                     .log_softmax_high_dimension(self.num_classes, name='prob'))

            (self.feed('score')
                 .softmax_high_dimension(self.num_classes, name='prob_normalized')
                 .argmax_2d(name='label_2d'))

        if self.with_losses and 'gt_label_weight' not in self.layers:   # (else: it left the label head's launch)
            (self.feed('prob_normalized', 'gt_label_2d')
                 .hard_label(threshold=self.threshold_label, name='gt_label_weight'))

        if self.vertex_reg:
            if not merged:
                self._head_conv1x1(['conv5_3'], 128, 'score_conv5_vertex', relu=False)
                self._head_conv1x1(['conv4_3'], 128, 'score_conv4_vertex', relu=False)
            small_v = self._small_head('score_conv5_vertex', 'score_conv4_vertex', 'upscore_conv5_vertex', 'add_score_vertex',
                                       'dropout_vertex', 'add_score_vertex', 'vertex_pred', 3 * self.num_classes, 128)
            if small_v is None:
                (self.feed('score_conv5_vertex')
                     .deconv(4, 4, 128, 2, 2, name='upscore_conv5_vertex', trainable=False))

                (self.feed('score_conv4_vertex', 'upscore_conv5_vertex')
                     .add(name='add_score_vertex')
                     ._plant('add_score_vertex', 'add_score_vertex')
                     .dropout(self.keep_prob_queue, name='dropout_vertex'))
            if self.fused_heads:
                # same commutation as the label head: 128->3C at 1/8 resolution, then one
                # interpolation pass writes vertex_pred (+ bias); `upscore_vertex` is never built
                # and the Hough layer interpolates just the pixels it samples, so `vertex_pred`
                # itself (81 MB/frame) is only materialised if somebody fetches it.
                zv, bv = small_v if small_v is not None else self._conv1x1_lowres(self.get_output('dropout_vertex'), 'vertex_pred', 3 * self.num_classes, 128)
                kv, sv = int(16 * self.scale), int(8 * self.scale)
                self.layers['vertex_pred_lowres'] = zv
                self.layers['vertex_pred_bias'] = bv
                self.layers['vertex_pred'] = _Lazy(lambda: self._deconv_bilinear(zv, kv, sv, bias=bv))
            else:
                (self.feed('dropout_vertex')
                     .deconv(int(16 * self.scale), int(16 * self.scale), 128, int(8 * self.scale), int(8 * self.scale), name='upscore_vertex', trainable=False)
                     .conv(1, 1, 3 * self.num_classes, 1, 1, name='vertex_pred', relu=False, c_i=128))

            if self.vertex_reg_2d:
                if self.fused_heads:
                    (self.feed('label_2d', 'vertex_pred_lowres', 'vertex_pred_bias', 'extents', 'meta_data', 'poses')
                         .hough_voting_gpu_lowres(kv, sv, self.is_train, self.vote_threshold, self.vote_percentage,
                                                  self.skip_pixels, name='hough'))
                else:
                    (self.feed('label_2d', 'vertex_pred', 'extents', 'meta_data', 'poses')
                         .hough_voting_gpu(self.is_train, self.vote_threshold, self.vote_percentage, self.skip_pixels, name='hough'))

                self.layers['rois'] = self.get_output('hough')[0]
                self.layers['poses_init'] = self.get_output('hough')[1]
                self.layers['poses_target'] = self.get_output('hough')[2]
                self.layers['poses_weight'] = self.get_output('hough')[3]

                if self.pose_reg:
                    # roi pooling without masking
                    (self.feed('conv5_3', 'rois')
                         .roi_pool(7, 7, 1.0 / 16.0, 0, name='pool5'))

                    (self.feed('conv4_3', 'rois')
                         .roi_pool(7, 7, 1.0 / 8.0, 0, name='pool4'))

                    (self.feed('pool5', 'pool4')
                         .add(name='pool_score')
                         .fc(4096, height=7, width=7, channel=512, name='fc6')
                         .dropout(self.keep_prob_queue, name='drop6')
                         .fc(4096, num_in=4096, name='fc7')
                         .dropout(self.keep_prob_queue, name='drop7')
                         .fc(4 * self.num_classes, relu=False, name='fc8')
                         .tanh(name='poses_tanh'))

                    (self.feed('poses_tanh', 'poses_weight')
                         .multiply(name='poses_mul')
                         .l2_normalize(dim=1, name='poses_pred'))

                    if self.with_losses:
                        (self.feed('poses_pred', 'poses_target', 'poses_weight', 'points', 'symmetry')
                             .average_distance_loss(margin=0.01, name='loss_pose'))

                    if self.adaptation:
                        self.layers['label_domain'] = self.get_output('hough')[4]
