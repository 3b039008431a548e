"""Scratch: why do translations differ by mm when no voter changes? (round 3 investigation)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_study as ps
from posecnn_amd import config, synth, fcn
from posecnn_amd.networks import vgg16_convs
dev = torch.device("cuda:0")
H, W, C, B = 480, 640, 22, 4
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
net = vgg16_convs("RGBD", C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False, is_train=False, seed=3, init="he", with_losses=False, device=dev)
synth.init_planted_heads(net)
K = config.DEMO_INTRINSICS.copy(); ext = config.LOV_EXTENTS
pts = T(synth.make_model_points(C, 256))
rng = np.random.default_rng(2024)
def run(path, data, data_p, planted):
    net.reference_trunk = {"float64": torch.float64, "taps_f32": torch.float32}.get(path)
    net.winograd_min_channels = 0 if path == "library" else 64
    with torch.no_grad():
        det = fcn.im_segment_batch(net, data, K, ext, pts, config.LOV_SYMMETRY, data_p=data_p, planted=planted)
        n = int(det.count.item())
        out = {"label": det.label_2d.clone(), "rows": det.rows[:n].cpu().numpy(), "vertex": net.get_output("vertex_pred").clone(),
               "zv": net.get_output("vertex_pred_lowres").clone(), "pool": net.get_output("pool_score")[:n].clone(),
               "fc6": net.get_output("fc6")[:n].clone(), "fc8": net.get_output("fc8")[:n].clone(), "c4": net.get_output("conv4_3").clone(), "c5": net.get_output("conv5_3").clone()}
    net.reference_trunk = None; net.winograd_min_channels = 64
    return out
for b0 in range(0, 16, B):
    data, data_p = ps._rgbd_inputs(rng, B, H, W)
    planted_np, _ = synth.make_planted_batch(1000 + b0, B, H=H, W=W, C=C, K=K, n_obj=5)
    planted = {k: T(v) for k, v in planted_np.items()}
    data, data_p = T(data), T(data_p)
    ref = run("float64", data, data_p, planted)
    for p in ("taps_f32", "winograd"):
        got = run(p, data, data_p, planted)
        d = np.abs(got["rows"] - ref["rows"])
        i = int(np.argmax(d[:, 11:14].max(1)))
        j = int(np.argmax(d[:, 7:11].max(1)))
        print(p, "batch", b0, "zv maxdiff", float((got["zv"] - ref["zv"]).abs().max()), "zv absmax", float(ref["zv"].abs().max()),
              "c4 maxdiff", float((got["c4"]-ref["c4"]).abs().max()), "c4 absmax", float(ref["c4"].abs().max()))
        print("  worst trans row", i, "diff", d[i, 11:14], "ref", ref["rows"][i], "\n   got", got["rows"][i])
        n, c = int(ref["rows"][i, 0]), int(ref["rows"][i, 1]); r = ref["rows"][i]
        cx, cy = int(round((r[2]+r[4])/2)), int(round((r[3]+r[5])/2))
        for nm, o in (("ref", ref), ("got", got)):
            idx, inl, dd = ps.voters(ref["label"][n], o["vertex"][n], c, cx, cy, ext, K)
            print("   ", nm, "voters", int(inl.sum()), "mean d (f64)", float(dd[inl].double().mean()), "kernel tz", o["rows"][i, 13], "score", o["rows"][i, 6])
        idx, inl_r, dr = ps.voters(ref["label"][n], ref["vertex"][n], c, cx, cy, ext, K)
        idx, inl_g, dg = ps.voters(ref["label"][n], got["vertex"][n], c, cx, cy, ext, K)
        print("    max |d_got - d_ref| over voters", float((dg - dr)[inl_r].abs().max()), " over all sampled", float((dg-dr).abs().max()))
        print("  worst quat row", j, "diff", d[j, 7:11], "pool maxdiff", float((got["pool"][j]-ref["pool"][j]).abs().max()), "pool absmax", float(ref["pool"][j].abs().max()),
              "fc6 maxdiff", float((got["fc6"][j]-ref["fc6"][j]).abs().max()), "fc6 absmax", float(ref["fc6"][j].abs().max()),
              "fc8 maxdiff", float((got["fc8"][j]-ref["fc8"][j]).abs().max()), "fc8 absmax", float(ref["fc8"][j].abs().max()))
