"""CPU tests of the `lib/networks` surface: TF1 layer semantics of the dense layers (checked against
direct numpy loops written from the TF definitions), the fixed bilinear deconv filter, and the
`vgg16_convs` graph wiring (run end to end on the CPU restatement with the oracle)."""
import numpy as np
import pytest
import torch

from cpu_reference import run_cpu_pipeline, vgg16_convs_cpu
from posecnn_amd import config, synth
from posecnn_amd.networks import Network, make_deconv_filter_1d

F = np.float32


class Tiny(Network):
    def setup(self):
        pass


def test_make_deconv_filter_values():
    # network.py:141-150: k=4 -> taps .25 .75 .75 .25 ; k=16 -> c = 0.9375
    assert np.allclose(make_deconv_filter_1d(4), [0.25, 0.75, 0.75, 0.25])
    f16 = make_deconv_filter_1d(16)
    assert np.allclose(f16[:3], [1 - 0.9375, 1 - abs(1 / 8 - 0.9375), 1 - abs(2 / 8 - 0.9375)])
    assert np.isclose(f16.sum(), 8.0)  # stride-8 bilinear kernel: taps sum to the stride


def tf_conv_same(x, w, b):
    """tf.nn.conv2d NHWC, stride 1, 'SAME', filter [kh,kw,cin,cout] + bias (direct loops)."""
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (ph, kh - 1 - ph), (pw, kw - 1 - pw), (0, 0)))
    out = np.zeros((B, H, W, Cout), np.float64)
    for i in range(kh):
        for j in range(kw):
            out += np.einsum("bhwc,co->bhwo", xp[:, i:i + H, j:j + W, :].astype(np.float64), w[i, j].astype(np.float64))
    return out + b


def tf_conv2d_transpose_same(x, w, s):
    """tf.nn.conv2d_transpose NHWC 'SAME', output = input * s, filter [k,k,cout,cin]: scatter form.
    out[b, s*h + i - p, s*w + j - p, o] += x[b,h,w,c] * f[i,j,o,c] with p = (k - s) / 2."""
    B, H, W, Cin = x.shape
    k = w.shape[0]
    Cout = w.shape[2]
    p = (k - s) // 2
    out = np.zeros((B, H * s, W * s, Cout), np.float64)
    for h in range(H):
        for ww in range(W):
            for i in range(k):
                for j in range(k):
                    oy, ox = s * h + i - p, s * ww + j - p
                    if 0 <= oy < H * s and 0 <= ox < W * s:
                        out[:, oy, ox, :] += x[:, h, ww, :].astype(np.float64) @ w[i, j].T.astype(np.float64)
    return out


def test_conv_matches_tf_definition():
    rng = np.random.default_rng(0)
    net = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)
    x = rng.standard_normal((2, 6, 7, 5)).astype(F)
    w = rng.standard_normal((3, 3, 5, 4)).astype(F)
    b = rng.standard_normal(4).astype(F)
    net.load({"c": {"weights": w, "biases": b}})
    net.layers = {"data": torch.from_numpy(x)}
    y = net.feed("data").conv(3, 3, 4, 1, 1, name="c", c_i=5).get_output("c").numpy()
    assert np.allclose(y, np.maximum(tf_conv_same(x, w, b), 0), atol=1e-5)
    y2 = net.feed("data").conv(3, 3, 4, 1, 1, name="c", c_i=5, relu=False).layers["c"].numpy()
    assert np.allclose(y2, tf_conv_same(x, w, b), atol=1e-5)


@pytest.mark.parametrize("k,s", [(4, 2), (16, 8)])
def test_deconv_matches_dense_conv2d_transpose(k, s):
    rng = np.random.default_rng(1)
    C = 3
    net = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)  # the deconv layer -> oracle restatement
    x = rng.standard_normal((1, 4, 5, C)).astype(F)
    net.layers = {"x": torch.from_numpy(x)}
    y = net.feed("x").deconv(k, k, C, s, s, name="up", trainable=False).get_output("up").numpy()
    f1 = make_deconv_filter_1d(k)
    dense = np.zeros((k, k, C, C))
    for i in range(C):
        dense[:, :, i, i] = np.outer(f1, f1)  # weights[:, :, i, i] = bilinear (network.py:151-153)
    want = tf_conv2d_transpose_same(x, dense, s)
    assert y.shape == (1, 4 * s, 5 * s, C)
    assert np.allclose(y, want, atol=1e-5)


def test_max_pool_and_fc_flatten_order():
    rng = np.random.default_rng(2)
    net = Tiny(device="cpu", trainable=False)
    x = rng.standard_normal((2, 4, 6, 3)).astype(F)
    net.layers = {"x": torch.from_numpy(x)}
    y = net.feed("x").max_pool(2, 2, 2, 2, name="p").get_output("p").numpy()
    assert np.array_equal(y, x.reshape(2, 2, 2, 3, 2, 3).max(axis=(2, 4)))
    w = rng.standard_normal((4 * 6 * 3, 5)).astype(F)
    b = rng.standard_normal(5).astype(F)
    net.load({"fc": {"weights": w, "biases": b}})
    y = net.feed("x").fc(5, height=4, width=6, channel=3, name="fc", relu=False).get_output("fc").numpy()
    assert np.allclose(y, x.reshape(2, -1) @ w + b, atol=1e-4)  # NHWC flatten (network.py:399-408)


def test_vgg16_convs_graph_runs_on_cpu_reference():
    H, W = 96, 128
    K = config.DEMO_INTRINSICS.copy(); K[:2] *= W / 640.0
    net = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True,
                          trainable=False, is_train=False, init="he", with_losses=True)
    synth.init_calibrated(net)
    rng = np.random.default_rng(3)
    data = (rng.integers(0, 256, (1, H, W, 3)).astype(F) - config.PIXEL_MEANS).astype(F)
    # objects at this resolution are < 500 px: the Hough layer returns its dummy row
    planted, scenes = synth.make_planted_batch(0, 1, H=H, W=W, K=K, n_obj=2)
    pts = synth.make_model_points(22, 64)
    out = run_cpu_pipeline(net, data, K, config.LOV_EXTENTS, pts, config.LOV_SYMMETRY, planted=planted)
    assert out["label_2d"].shape == (1, H, W) and out["vertex_pred"].shape == (1, H, W, 66)
    assert net.get_output("conv4_3").shape == (1, H // 8, W // 8, 512)
    assert net.get_output("gt_label_weight").shape == (1, H, W, 22)
    assert out["rois"].shape[1] == 7 and out["poses_tanh"].shape[1] == 88
    # planted labels come through the bilinear deconv + identity head
    lab_lr = scenes[0]["label_lowres"]
    centre = out["label_2d"][0, 4::8, 4::8]
    assert (centre == lab_lr).mean() > 0.95
    # RGBD variant: two towers concatenated for the label head only (vgg16_convs.py:99-126)
    net2 = vgg16_convs_cpu("RGBD", 22, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True,
                           trainable=False, is_train=False, init="he")
    feed = {"data": torch.from_numpy(data[:, :32, :32]), "data_p": torch.from_numpy(data[:, :32, :32]),
            "gt_label_2d": torch.ones((1, 32, 32), dtype=torch.int32), "keep_prob": 1.0, "poses": torch.zeros((1, 13)),
            "extents": torch.from_numpy(config.LOV_EXTENTS), "meta_data": torch.from_numpy(config.make_meta_data(K)).reshape(1, 1, 1, 48),
            "points": torch.from_numpy(pts), "symmetry": torch.from_numpy(config.LOV_SYMMETRY)}
    with torch.no_grad():
        net2.run(feed)
    assert net2.vars["score_conv5/weights"].shape == (64, 1024, 1, 1)
    assert net2.vars["score_conv5_vertex/weights"].shape == (128, 512, 1, 1)
    assert "conv5_3_p" in net2.layers and net2.get_output("concat_conv5").shape[-1] == 1024


def test_checkpoint_round_trip_npz_and_npy(tmp_path):
    """save_npz writes TF-named / TF-layout variables; load_file reads them back (and the vgg16.npy
    dict format, network.py:71-107) into a second network that then computes the same outputs."""
    rng = np.random.default_rng(7)
    a = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)
    x = torch.from_numpy(rng.standard_normal((1, 6, 8, 3)).astype(F))
    a.layers = {"x": x}
    a.feed("x").conv(3, 3, 4, 1, 1, name="c1", c_i=3).fc(5, height=6, width=8, channel=4, name="f1", relu=False)
    path = str(tmp_path / "w.npz")
    a.save_npz(path)
    arch = np.load(path)
    assert arch["c1/weights"].shape == (3, 3, 3, 4) and arch["f1/weights"].shape == (6 * 8 * 4, 5)
    b = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)
    assert b.load_file(path) == ["c1", "f1"]
    b.layers = {"x": x}
    b.feed("x").conv(3, 3, 4, 1, 1, name="c1", c_i=3).fc(5, height=6, width=8, channel=4, name="f1", relu=False)
    assert torch.equal(a.get_output("f1"), b.get_output("f1"))
    npy = str(tmp_path / "vgg.npy")
    np.save(npy, {"c1": {"weights": arch["c1/weights"], "biases": arch["c1/biases"]}}, allow_pickle=True)
    c = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)
    assert c.load_file(npy) == ["c1"]
    c.layers = {"x": x}
    assert torch.equal(c.feed("x").conv(3, 3, 4, 1, 1, name="c1", c_i=3).get_output("c1"), a.get_output("c1"))


def test_deferred_bias_relu_is_fused_only_into_max_pool():
    """A conv listed in `defer_act` hands its raw output to the next layer: max_pool fuses bias + ReLU,
    every other consumer (and a fetch by name) sees the activated tensor."""
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((1, 6, 8, 5)).astype(F))
    w = rng.standard_normal((3, 3, 5, 4)).astype(F); b = rng.standard_normal(4).astype(F)
    outs = {}
    for defer in (False, True):
        net = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)
        net.defer_act = frozenset(["c"]) if defer else frozenset()
        net.load({"c": {"weights": w, "biases": b}})
        net.layers = {"x": x}
        with torch.no_grad():
            net.feed("x").conv(3, 3, 4, 1, 1, name="c", c_i=5).max_pool(2, 2, 2, 2, name="p")
            pooled = net.get_output("p").clone()
            net.feed("x").conv(3, 3, 4, 1, 1, name="c", c_i=5).relu(name="r")
            outs[defer] = (pooled, net.get_output("r").clone(), net.get_output("c").clone())
    for a, b_ in zip(outs[False], outs[True]):
        assert torch.equal(a, b_)
    assert float(outs[True][2].min()) >= 0  # fetched by name: activated


def test_load_keeps_depth_tower_weights_and_drops_constant_deconv_filters():
    """(a) a checkpoint that carries '<layer>_p' entries loads each tower from its own
    entry whatever the dict order; vgg16.npy-style dicts (no '_p' keys) still initialise both towers;
    (b) the constant bilinear deconv filters of a checkpoint do not divert `deconv` to the dense path."""
    from posecnn_amd.networks import is_bilinear_deconv_filter
    rng = np.random.default_rng(4)
    wa = rng.standard_normal((3, 3, 3, 64)).astype(F); wb = rng.standard_normal((3, 3, 3, 64)).astype(F)
    ba = rng.standard_normal(64).astype(F); bb = rng.standard_normal(64).astype(F)

    def fresh():
        net = vgg16_convs_cpu("RGBD", 22, 64, (1.0,), 1.0, -1.0, trainable=False, is_train=False)
        for n in ("conv1_1", "conv1_1_p"):   # the variables exist once the graph has been built
            net.vars[n + "/weights"] = torch.zeros((64, 3, 3, 3)); net.vars[n + "/biases"] = torch.zeros(64)
        return net
    for order in (("conv1_1", "conv1_1_p"), ("conv1_1_p", "conv1_1")):
        d = {k: ({"weights": wa, "biases": ba} if k == "conv1_1" else {"weights": wb, "biases": bb}) for k in order}
        net = fresh(); net.load(d)
        assert torch.equal(net.vars["conv1_1/biases"], torch.from_numpy(ba)), order
        assert torch.equal(net.vars["conv1_1_p/biases"], torch.from_numpy(bb)), order
        assert torch.equal(net.vars["conv1_1_p/weights"], torch.from_numpy(wb).permute(3, 2, 0, 1)), order
    net = fresh(); net.load({"conv1_1": {"weights": wa, "biases": ba}})     # vgg16.npy: one entry feeds both towers
    assert torch.equal(net.vars["conv1_1_p/biases"], torch.from_numpy(ba))

    f16 = make_deconv_filter_1d(16)
    tf_filter = np.zeros((16, 16, 64, 64), np.float64)                       # [k, k, c_out, c_in], network.py:151-157
    for i in range(64):
        tf_filter[:, :, i, i] = np.outer(f16, f16)
    tf_filter = tf_filter.astype(F)
    assert is_bilinear_deconv_filter(torch.from_numpy(tf_filter).permute(3, 2, 0, 1))
    assert not is_bilinear_deconv_filter(torch.from_numpy(wa).permute(3, 2, 0, 1))
    net = fresh(); assert net.fused_heads
    net.load({"upscore": {"weights": tf_filter}})
    assert "upscore/weights" not in net.vars and net.fused_heads           # same kernels as an unloaded network
    trained = tf_filter.copy(); trained[3, 3, 0, 1] = 0.01                   # somebody fine-tuned the filter
    net.load({"upscore": {"weights": trained}})
    assert "upscore/weights" in net.vars and not net.fused_heads           # literal op order, dense conv_transpose
