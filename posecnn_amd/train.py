"""Training-graph pieces of the hot path: the losses `lib/fcn/train.py` builds on
top of `vgg16_convs` and a minimal solver loop, so the network can be fine-tuned on MI355X.

    loss = loss_cls + VERTEX_W * smooth_l1(vertex) + POSE_W * loss_pose + loss_regu     (train.py:488-519)

The custom layers bring their own gradients (ROI pooling, average-distance loss, backprojection;
zeros for Hough voting and hard labels, as the reference registers them); the fixed bilinear
deconvs and the vertex loss have hand-written gfx950 backward kernels; convolutions / fc layers go
through the framework's autograd. A trainable `vgg16_convs(is_train=True)` evaluates the heads in
the reference's literal op order (the fused inference epilogues have no backward).
"""
import torch

from . import ops


class TrainConfig(object):
    """cfg.TRAIN defaults of lib/fcn/config.py:54-125."""
    LEARNING_RATE = 0.001
    MOMENTUM = 0.9
    GAMMA = 0.1
    STEPSIZE = 30000
    WEIGHT_REG = 0.0001
    VERTEX_W = 5.0
    POSE_W = 1.0
    SNAPSHOT_ITERS = 10000


def loss_cross_entropy_single_frame(scores, labels):
    """train.py:455-466. scores = log-softmax [B,H,W,C] (`prob`), labels = `gt_label_weight`."""
    cross_entropy = -(labels * scores).sum(dim=3)
    return cross_entropy.sum() / (labels.sum() + 1e-10)


def smooth_l1_loss_vertex(vertex_pred, vertex_targets, vertex_weights, sigma=1.0):
    """train.py:564-573 (one streaming forward and one backward kernel)."""
    return ops.smooth_l1_loss_vertex(vertex_pred, vertex_targets, vertex_weights, sigma)


def loss_quaternion(pose_pred, pose_targets, pose_weights):
    """train.py:469-477 (the alternative pose loss the reference keeps around)."""
    distances = 1 - (pose_pred * pose_targets).sum(dim=1) ** 2
    weights = pose_weights.mean(dim=1)
    return (weights * distances).sum() / (weights.sum() + 1e-10)


def regularization_loss(net, scale):
    """tf.contrib.layers.l2_regularizer(scale) on every conv / fc variable (network.py:171,417):
    scale * sum(v^2) / 2, summed like tf.add_n(get_regularization_losses()) (train.py:483)."""
    total = None
    for name in sorted(net.vars):
        v = net.vars[name]
        if not name.endswith(("/weights", "/biases")):
            continue
        term = (v * v).sum() * (0.5 * scale)
        total = term if total is None else total + term
    return total if total is not None else torch.zeros((), device=net.device)


def build_losses(net, cfg=TrainConfig):
    """The SINGLE_FRAME / VERTEX_REG_2D / POSE_REG branch of train_net (train.py:488-519), evaluated
    on the layers of a finished `net.run(feed)`. Returns a dict of scalar tensors."""
    out = {}
    out["loss_cls"] = loss_cross_entropy_single_frame(net.get_output("prob"), net.get_output("gt_label_weight"))
    out["loss_regu"] = regularization_loss(net, cfg.WEIGHT_REG)
    loss = out["loss_cls"] + out["loss_regu"]
    if net.vertex_reg:
        out["loss_vertex"] = cfg.VERTEX_W * smooth_l1_loss_vertex(
            net.get_output("vertex_pred"), net.get_output("vertex_targets"), net.get_output("vertex_weights"))
        loss = loss + out["loss_vertex"]
        if net.pose_reg and net.vertex_reg_2d:
            out["loss_pose"] = cfg.POSE_W * net.get_output("loss_pose")[0].reshape(())
            loss = loss + out["loss_pose"]
    out["loss"] = loss
    return out


class SolverWrapper(object):
    """train.py:25-56,206-261 reduced to what runs on one GPU: momentum SGD with the staircase
    exponential decay of train.py:531-536, snapshot / restore of the variables."""

    def __init__(self, net, cfg=TrainConfig):
        self.net, self.cfg = net, cfg
        self.optimizer = None
        self.iter = 0

    def learning_rate(self):
        return self.cfg.LEARNING_RATE * self.cfg.GAMMA ** (self.iter // self.cfg.STEPSIZE)

    def _params(self):
        return [v for _, v in sorted(self.net.vars.items()) if v.requires_grad]

    def train_step(self, feed):
        """One sess.run([loss, ..., train_op]) of train_model_vertex_pose (train.py:241)."""
        with torch.enable_grad():
            self.net.run(feed)
            losses = build_losses(self.net, self.cfg)
            if self.optimizer is None:  # variables are created by the first run
                # tf.train.MomentumOptimizer: accum = m * accum + grad; var -= lr * accum
                self.optimizer = torch.optim.SGD(self._params(), lr=self.learning_rate(), momentum=self.cfg.MOMENTUM)
            for g in self.optimizer.param_groups:
                g["lr"] = self.learning_rate()
            self.optimizer.zero_grad(set_to_none=True)
            losses["loss"].backward()
            self.optimizer.step()
        self.iter += 1
        return {k: float(v.detach()) for k, v in losses.items()}

    def snapshot(self, path):
        state = {"vars": {k: v.detach().cpu() for k, v in self.net.vars.items()}, "iter": self.iter,
                 "optimizer": None if self.optimizer is None else self.optimizer.state_dict()}
        torch.save(state, path)

    def restore(self, path):
        state = torch.load(path, map_location="cpu")
        for k, v in state["vars"].items():
            t = v.to(self.net.device)
            if k in self.net.vars and self.net.vars[k].requires_grad:
                with torch.no_grad():
                    self.net.vars[k].copy_(t)
            else:
                self.net.vars[k] = t
        self.iter = int(state.get("iter", 0))
        if state.get("optimizer") is not None and self.optimizer is not None:
            self.optimizer.load_state_dict(state["optimizer"])
