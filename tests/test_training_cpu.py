"""CPU checks of the training-graph restatements in oracle/ (deconv gradient, smooth-L1 vertex
loss) against independent numpy / torch formulations, and of the host-side loss helpers."""
import numpy as np
import torch

import oracle
from posecnn_amd import train
from posecnn_amd.networks import make_deconv_filter_1d

F = np.float32


def test_oracle_deconv_bwd_is_the_transpose_of_the_forward():
    rng = np.random.default_rng(0)
    for (B, H, W, C, k, s) in [(1, 5, 6, 3, 4, 2), (2, 3, 4, 2, 16, 8), (1, 4, 4, 1, 4, 4)]:
        x = rng.standard_normal((B, H, W, C)).astype(F)
        g = rng.standard_normal((B, H * s, W * s, C)).astype(F)
        y = oracle.deconv_bilinear(x, k, s)
        gx = oracle.deconv_bilinear_bwd(g, k, s)
        # <deconv(x), g> == <x, deconv^T(g)>
        assert abs(float((y.astype(np.float64) * g).sum()) - float((x.astype(np.float64) * gx).sum())) < 1e-3
        # and against torch's conv_transpose2d autograd with the dense-diagonal filter
        f = make_deconv_filter_1d(k)
        w = torch.zeros((C, 1, k, k), dtype=torch.float64)
        w[:, 0] = torch.from_numpy(np.outer(f, f).astype(F).astype(np.float64))
        xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
        yt = torch.nn.functional.conv_transpose2d(xt, w, stride=s, padding=(k - s) // 2, groups=C)
        yt.backward(torch.from_numpy(g.astype(np.float64)).permute(0, 3, 1, 2))
        assert np.allclose(gx, xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-5)


def test_oracle_smooth_l1_matches_the_tf_formula():
    rng = np.random.default_rng(1)
    for n, sigma in [(1000, 1.0), (300001, 1.0), (4097, 3.0), (0, 1.0)]:
        p = (rng.standard_normal(n) * 2).astype(F); t = (rng.standard_normal(n) * 2).astype(F)
        w = (rng.random(n) < 0.3).astype(F)
        out, grad = oracle.smooth_l1_vertex(p, t, w, sigma)
        pt = torch.from_numpy(p.astype(np.float64)).requires_grad_(True)
        tt, wt = torch.from_numpy(t.astype(np.float64)), torch.from_numpy(w.astype(np.float64))
        s2 = sigma ** 2
        diff = wt * (pt - tt); ad = diff.abs(); sign = (ad < 1.0 / s2).double()
        in_loss = diff ** 2 * (s2 / 2.0) * sign + (ad - 0.5 / s2) * (1 - sign)
        loss = in_loss.sum() / (wt.sum() + 1e-10)
        loss.backward()
        lv = float(loss.detach())
        assert abs(out[0] - lv) <= 1e-5 * max(1.0, abs(lv))
        assert abs(out[2] - float(wt.sum())) <= 1e-3
        if n:
            assert np.allclose(grad, pt.grad.numpy(), atol=1e-6, rtol=1e-4)


def test_host_loss_helpers():
    rng = np.random.default_rng(2)
    scores = torch.log_softmax(torch.from_numpy(rng.standard_normal((1, 4, 5, 6)).astype(F)), dim=-1)
    labels = torch.zeros((1, 4, 5, 6)); labels[..., 2] = 1
    ce = train.loss_cross_entropy_single_frame(scores, labels)
    assert abs(float(ce) + float(scores[..., 2].mean())) < 1e-6
    q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((3, 8)).astype(F)), dim=1)
    assert float(train.loss_quaternion(q, q, torch.ones((3, 8)))) < 1e-6

    class Net:  # regularisation covers weights and biases of every layer (network.py:171,184,417)
        device = "cpu"
        vars = {"a/weights": torch.full((2, 2), 2.0), "a/biases": torch.full((2,), 1.0), "other": torch.ones(3)}
    assert abs(float(train.regularization_loss(Net, 0.1)) - 0.05 * (16 + 2)) < 1e-6
    s = train.SolverWrapper(Net)
    s.iter = 60000
    assert abs(s.learning_rate() - 0.001 * 0.1 ** 2) < 1e-12   # staircase decay (train.py:531-536)
