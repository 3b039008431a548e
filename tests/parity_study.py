#!/usr/bin/env python
"""End-to-end numerics study of the trunk (VERDICT r2 "Next" #1): how far do detections move when the
same 480x640 RGB-D frames go through differently computed VGG16 trunks?

  float64    the real-arithmetic reference: nine shifted GEMMs per layer in float64 (vgg16_convs._trunk_reference)
  taps_f32   the same algebra in float32 — a direct convolution whose summation order is known
  library    F.conv2d (MIOpen picks the algorithm; `strict_numerics=True` / winograd_min_channels = 0)
  winograd   the product default: F(4x4,3x3) on the fp32 matrix cores (csrc/wino_mfma.hip)

Everything behind the trunk (heads, softmax/argmax, Hough voting, RoI pooling, fc6-8) is the same f32
kernel sequence in every run, so differences are the trunk's rounding and what the pipeline makes of it.
Per path, against float64: label flips, detections that moved to another Hough cell, max |d box|,
max |d quaternion|, the distribution of |d translation| (absolute, and relative to |t|), and the Hough
voters of each detection that changed side of the hard inlier test (hough_voting_gpu_op.cu.cc:269-294) —
counted by re-evaluating the canonical predicate (SURVEY.md §8a HOUGH) in torch on each run's own vertex field.

Scale matters when reading the absolute numbers. Round 3 ran this on He-initialised weights fed raw pixel
values (|x| ~ 100): conv4_3 reached ~800, the 1/8-resolution vertex field ~45 and fc8 ~1500 where a trained
PoseCNN has O(1-10), and noise classes produced junk detections at exp(5) "metres" — north_star's absolute
1e-4 was unassertable there (profiles/r03_parity_study.json). Round 4 runs it on the CALIBRATED synthetic
network (synth.init_calibrated: every layer's output std ~ 1, |fc8| <~ 3, depths 0.5-2 m), the same weights
bench.py measures, and tests/test_gpu_round3.py asserts the tolerance literally. Relative figures are kept.

TEST INFRASTRUCTURE (imported by tests/test_gpu_round3.py; `python tests/parity_study.py --frames 64
--out profiles/r04_parity_study.json` writes the table DESIGN.md §4 quotes). Needs a GPU.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from posecnn_amd import config, synth  # noqa: E402

F = np.float32
PATHS = ("float64", "taps_f32", "library", "winograd")


def _rgbd_inputs(rng, B, H, W):
    im = rng.integers(0, 256, (B, H, W, 3)).astype(F)
    depth = rng.integers(0, 3000, (B, H, W, 1)).astype(F)
    data = (im - config.PIXEL_MEANS).astype(F)
    data_p = (np.tile(np.clip(depth / 2000.0, 0, 1) * 255, (1, 1, 1, 3)) - config.PIXEL_MEANS).astype(F)   # test.py:70-74
    return data, data_p


def _project_box_thr(ext_c, d, fx, fy, px, py):
    """PROJECT_BOX of SURVEY.md §8a (hough_voting_gpu_op.cu.cc:84-120) for a vector of depths d (torch f32)."""
    import torch
    hx, hy, hz = (float(np.float32(float(e) * 0.5)) for e in ext_c)
    xs, ys = [], []
    for sx in (-1.0, 1.0):
        for sy in (-1.0, 1.0):
            for sz in (-1.0, 1.0):
                Z = sz * hz + d
                xs.append(fx * ((sx * hx) / Z) + px)
                ys.append(fy * ((sy * hy) / Z) + py)
    xs, ys = torch.stack(xs), torch.stack(ys)
    w = xs.max(0).values - xs.min(0).values + 1.0
    h = ys.max(0).values - ys.min(0).values + 1.0
    return torch.maximum(w, h) * 0.6


def voters(label_n, vertex_n, c, cx, cy, ext, K, skip=10):
    """(pixel indices sampled for class c, inlier mask at cell (cx, cy), depths) — the vote predicate of
    compute_hough_kernel restated op by op in torch f32 (no FMA; exp is torch's, which only matters within
    1 ulp of the window edge)."""
    import torch
    W = label_n.shape[1]
    idx = torch.nonzero(label_n.reshape(-1) == c).reshape(-1)[::skip]
    x, y = idx % W, idx // W
    f = vertex_n[y, x]
    u, v, d = f[:, 3 * c], f[:, 3 * c + 1], torch.exp(f[:, 3 * c + 2])
    dx, dy = (cx - x).to(torch.float32), (cy - y).to(torch.float32)
    cosv = (u * dx + v * dy) / (torch.sqrt(u * u + v * v) * torch.sqrt(dx * dx + dy * dy))
    thr = _project_box_thr(ext[c], d, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
    inl = (cosv > 0.9) & (dx.abs() < thr) & (dy.abs() < thr)
    return idx, inl, d


def run_study(device, n_frames=32, batch=4, H=480, W=640, C=22, seed=2024, paths=PATHS, n_obj=5, log=None):
    import torch
    from posecnn_amd import fcn
    from posecnn_amd.networks import vgg16_convs
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    net = vgg16_convs("RGBD", C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=False, seed=3, init="he", with_losses=False, device=device)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0
    ext = config.LOV_EXTENTS[:C]
    pts = T(synth.make_model_points(C, 256))
    rng = np.random.default_rng(seed)
    acc = {p: {"label_flips": 0, "pixels": 0, "detections": 0, "missing_or_extra": 0, "cell_moved": 0, "box": [], "quat": [],
               "trans": [], "rel_trans": [], "planted": [], "depth": [], "votes": [], "changed": [], "bound_excess": [],
               "fc8_rel": [], "fc8_scale": [], "conv5_3_rel_err": 0.0, "max_prob_diff": 0.0, "field_rel_err": 0.0, "field_abs_err": 0.0,
               "field_absmax": 0.0}
           for p in paths if p != "float64"}

    def run(path, data, data_p, planted):
        net.reference_trunk = {"float64": torch.float64, "taps_f32": torch.float32}.get(path)
        net.winograd_min_channels = 0 if path == "library" else 64
        with torch.no_grad():
            det = fcn.im_segment_batch(net, data, K, ext, pts, config.LOV_SYMMETRY[:C], data_p=data_p, planted=planted)
            n = int(det.count.item())
            out = {"label": det.label_2d.clone(), "rows": det.rows[:n].cpu().numpy(), "prob": net.get_output("prob_normalized").clone(),
                   "conv5_3": net.get_output("conv5_3").clone(), "vertex": net.get_output("vertex_pred").clone(),
                   "zv": net.get_output("vertex_pred_lowres").clone(), "fc8": net.get_output("fc8")[:n].cpu().numpy()}
        net.reference_trunk = None
        net.winograd_min_channels = 64
        return out

    for b0 in range(0, n_frames, batch):
        B = min(batch, n_frames - b0)
        data, data_p = _rgbd_inputs(rng, B, H, W)
        planted_np, scenes = synth.make_planted_batch(1000 + b0, B, H=H, W=W, C=C, K=K, n_obj=n_obj)
        planted = {k: T(v) for k, v in planted_np.items()}
        data, data_p = T(data), T(data_p)
        ref = run("float64", data, data_p, planted)
        rkey = {(int(r[0]), int(r[1])): (r, f8) for r, f8 in zip(ref["rows"], ref["fc8"])}
        real = {(n_, o[0]) for n_, sc in enumerate(scenes) for o in sc["objects"]}
        for p in acc:
            got = run(p, data, data_p, planted)
            a = acc[p]
            a["label_flips"] += int((got["label"] != ref["label"]).sum())
            a["pixels"] += ref["label"].numel()
            a["max_prob_diff"] = max(a["max_prob_diff"], float((got["prob"] - ref["prob"]).abs().max()))
            a["conv5_3_rel_err"] = max(a["conv5_3_rel_err"], float((got["conv5_3"] - ref["conv5_3"]).abs().max() / ref["conv5_3"].abs().max()))
            a["field_abs_err"] = max(a["field_abs_err"], float((got["zv"] - ref["zv"]).abs().max()))
            a["field_absmax"] = max(a["field_absmax"], float(ref["zv"].abs().max()))
            a["field_rel_err"] = max(a["field_rel_err"], float((got["zv"] - ref["zv"]).abs().max() / ref["zv"].abs().max()))
            gkey = {(int(r[0]), int(r[1])): (r, f8) for r, f8 in zip(got["rows"], got["fc8"])}
            a["missing_or_extra"] += len(set(gkey) ^ set(rkey))
            for key in sorted(set(gkey) & set(rkey)):
                (g, gf8), (r, rf8) = gkey[key], rkey[key]
                n, c = key
                a["detections"] += 1
                # the winning cell: the box is centre -/+ extent, the centre an integer cell
                gc = (int(round((g[2] + g[4]) / 2)), int(round((g[3] + g[5]) / 2)))
                rc = (int(round((r[2] + r[4]) / 2)), int(round((r[3] + r[5]) / 2)))
                if gc != rc:
                    a["cell_moved"] += 1
                    continue   # a different (equally voted) cell: counted, not part of the distributions
                a["box"].append(float(np.abs(g[2:6] - r[2:6]).max()))
                a["quat"].append(float(np.abs(g[7:11] - r[7:11]).max()))
                a["trans"].append(float(np.abs(g[11:14] - r[11:14]).max()))
                a["rel_trans"].append(float(np.abs(g[11:14] - r[11:14]).max() / max(np.abs(r[11:14]).max(), 1e-6)))
                a["planted"].append(key in real)
                a["depth"].append(float(r[13]))
                a["fc8_scale"].append(float(np.abs(rf8).max()))
                a["fc8_rel"].append(float(np.abs(gf8 - rf8).max() / max(np.abs(rf8).max(), 1e-6)))
                a.setdefault("fc8_abs", []).append(float(np.abs(gf8 - rf8).max()))
                # voters of that cell under each run's own field, on the reference's label map
                _, ig, dg = voters(ref["label"][n], got["vertex"][n], c, rc[0], rc[1], ext, K)
                _, ir, dr = voters(ref["label"][n], ref["vertex"][n], c, rc[0], rc[1], ext, K)
                changed = int((ig != ir).sum())
                nv = int(ir.sum())
                a["votes"].append(nv)
                a["changed"].append(changed)
                # |d mean depth| <= changed * spread / voters (+ rounding of the sequential f32 sum)
                both = ig | ir
                spread = float((dr[both].max() - dr[both].min())) if bool(both.any()) else 0.0
                # |d mean depth| <= changed * spread / voters; on top of that the field's own rounding: d = exp(z) moves by
                # d * |dz|, |dz| <= the field error of this run (measured above) — relative to |t| both ways
                tmax = max(float(np.abs(r[11:14]).max()), 1e-6)
                bound = (changed * spread / max(min(nv, int(ig.sum())), 1)) / max(abs(float(r[13])), 1e-6) + 1.5 * a["field_abs_err"] + 1e-6
                a["bound_excess"].append(float(np.abs(g[11:14] - r[11:14]).max()) / tmax - bound)
        if log:
            log("frames %d..%d done" % (b0, b0 + B - 1))

    out = {"frames": n_frames, "height": H, "width": W, "classes": C, "objects_per_frame": n_obj, "reference": "float64 trunk (nine shifted GEMMs per layer)", "paths": {}}
    for p, a in acc.items():
        tr, ch = np.asarray(a["trans"]), np.asarray(a["changed"])
        rt, pl, dp = np.asarray(a["rel_trans"]), np.asarray(a["planted"], dtype=bool), np.asarray(a["depth"])
        vt = np.asarray(a["votes"])
        sup = vt >= 100
        q = lambda v, x: float(np.quantile(v, x)) if len(v) else None
        out["paths"][p] = {
            "label_flips": a["label_flips"], "pixels": a["pixels"], "max_prob_diff": a["max_prob_diff"],
            "conv5_3_rel_err": a["conv5_3_rel_err"], "detections_compared": a["detections"],
            "detections_missing_or_extra": a["missing_or_extra"], "winning_cell_moved": a["cell_moved"],
            "max_box_diff_px": max(a["box"]) if a["box"] else None, "max_quat_diff": max(a["quat"]) if a["quat"] else None,
            "trans_diff_median": q(tr, 0.5), "trans_diff_p90": q(tr, 0.9), "trans_diff_p99": q(tr, 0.99),
            "trans_diff_max": float(tr.max()) if len(tr) else None,
            "trans_rel_diff_median": q(rt, 0.5), "trans_rel_diff_p99": q(rt, 0.99), "trans_rel_diff_max": float(rt.max()) if len(rt) else None,
            # "supported": >= 100 voters agree on the cell — the planted objects as the Hough layer sees them. The rest are
            # classes with > 500 label pixels whose votes do not agree (planted objects reduced to slivers by occlusion, noise
            # classes): 0-2 votes, "depths" of exp(5).
            "supported_detections": int(sup.sum()), "supported_trans_diff_max": float(tr[sup].max()) if sup.any() else None,
            "supported_trans_rel_diff_max": float(rt[sup].max()) if sup.any() else None,
            "supported_depth_max_m": float(dp[sup].max()) if sup.any() else None,
            "depth_min_m": float(dp.min()) if len(dp) else None, "depth_max_m": float(dp.max()) if len(dp) else None,
            "unsupported_detections": int((~sup).sum()), "unsupported_depth_max_m": float(dp[~sup].max()) if (~sup).any() else None,
            "unsupported_max_votes": int(vt[~sup].max()) if (~sup).any() else None,
            "unsupported_trans_diff_max": float(tr[~sup].max()) if (~sup).any() else None,
            "vertex_field_abs_err": a["field_abs_err"], "vertex_field_absmax": a["field_absmax"], "vertex_field_rel_err": a["field_rel_err"],
            "fc8_rel_err_max": max(a["fc8_rel"]) if a["fc8_rel"] else None, "fc8_abs_err_max": max(a.get("fc8_abs", [0.0])),
            "fc8_absmax_median": q(np.asarray(a["fc8_scale"]), 0.5),
            "voters_total": int(np.sum(a["votes"])), "voters_changed_total": int(ch.sum()) if len(ch) else 0,
            "detections_with_changed_voters": int((ch > 0).sum()) if len(ch) else 0,
            "trans_diff_max_when_no_voter_changed": float(tr[ch == 0].max()) if len(tr) and (ch == 0).any() else None,
            "trans_diff_max_when_voters_changed": float(tr[ch > 0].max()) if len(tr) and (ch > 0).any() else None,
            "max_excess_over_voter_bound": max(a["bound_excess"]) if a["bound_excess"] else None,
        }
    return out


AMPLITUDES = (30.0, 10.0, 3.0, 1.0, 0.3)


def run_margin_sweep(device, amplitudes=AMPLITUDES, n_frames=8, batch=4, H=480, W=640, C=22, seed=2026, n_obj=5, log=None):
    """The margin behind "label maps bit-exact" (VERDICT r5 "Next" #2; reference argmax: lib/networks/network.py:432-434 on the
    softmax of :474-488). Every parity scene plants a logit of 30 over the O(1) output of the randomly initialised score
    heads, so only object-boundary pixels are ever near a tie. Here the planted logit is lowered — 30, 10, 3, 1, 0.3 — on
    `n_frames` full-size RGB-D frames; for each f32 trunk (taps_f32 / library / winograd) against the float64 trunk:

      label_flips        pixels whose argmax differs from the float64 run's
      gap_*              the float64 run's top-1 minus top-2 log-probability per pixel (= the score gap the argmax decides on):
                         its minimum, and how many pixels sit under 1e-3 / 1e-4 / 1e-5 / 1e-6
      flip_gap_max       the LARGEST such gap among the flipped pixels: labels can only differ where the two best classes are
                         closer than the trunks' rounding difference, so this is the margin the claim needs
      score_err_max      max |log p_f32 - log p_f64| over all classes of all pixels: the rounding difference itself

    TF1/cuDNN's own summation order is unknowable offline, so "bit-exact against the reference" can only be
    a statement of this kind: exact wherever the decision margin exceeds the f32 rounding of the trunk — measured here."""
    import torch
    from posecnn_amd import fcn
    from posecnn_amd.networks import vgg16_convs
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    net = vgg16_convs("RGBD", C, 64, (1.0,), 1.0, -1.0, vertex_reg_2d=True, pose_reg=True, trainable=False,
                      is_train=False, seed=3, init="he", with_losses=False, device=device)
    synth.init_calibrated(net)
    K = config.DEMO_INTRINSICS.copy()
    K[:2] *= W / 640.0
    ext = config.LOV_EXTENTS[:C]
    pts = T(synth.make_model_points(C, 256))
    f32_paths = tuple(p for p in PATHS if p != "float64")

    def run(path, data, data_p, planted):
        net.reference_trunk = {"float64": torch.float64, "taps_f32": torch.float32}.get(path)
        net.winograd_min_channels = 0 if path == "library" else 64
        try:
            with torch.no_grad():
                det = fcn.im_segment_batch(net, data, K, ext, pts, config.LOV_SYMMETRY[:C], data_p=data_p, planted=planted)
                return det.label_2d.clone(), net.get_output("prob_normalized").clone()
        finally:
            net.reference_trunk = None
            net.winograd_min_channels = 64

    out = {"frames": n_frames, "height": H, "width": W, "classes": C, "objects_per_frame": n_obj,
           "reference": "float64 trunk (nine shifted GEMMs per layer); heads, softmax and argmax are the same f32 kernels in every run",
           "what": "planted logit amplitude sweep: label flips of each f32 trunk against the float64 trunk, and the float64 run's "
                   "top-1 / top-2 log-probability gap at the flipped pixels", "amplitudes": {}}
    for amp in amplitudes:
        rng = np.random.default_rng(seed)     # the same frames at every amplitude
        acc = {p: {"label_flips": 0, "flip_gap_max": 0.0, "score_err_max": 0.0, "object_pixels_flipped": 0} for p in f32_paths}
        gaps = {"pixels": 0, "gap_min": float("inf"), "under_1e-3": 0, "under_1e-4": 0, "under_1e-5": 0, "under_1e-6": 0,
                "object_pixels": 0, "planted_label_recovered": 0}
        for b0 in range(0, n_frames, batch):
            B = min(batch, n_frames - b0)
            data, data_p = _rgbd_inputs(rng, B, H, W)
            planted_np, scenes = synth.make_planted_batch(3000 + b0, B, H=H, W=W, C=C, K=K, n_obj=n_obj)
            planted_np = dict(planted_np, add_score=(planted_np["add_score"] * F(amp / 30.0)).astype(F))   # the label logit only
            planted = {k: T(v) for k, v in planted_np.items()}
            data, data_p = T(data), T(data_p)
            lab64, prob64 = run("float64", data, data_p, planted)
            lp64 = torch.log(prob64.double().clamp_min(1e-300))
            top2 = torch.topk(lp64, 2, dim=-1).values
            gap = (top2[..., 0] - top2[..., 1])
            gaps["pixels"] += gap.numel()
            gaps["gap_min"] = min(gaps["gap_min"], float(gap.min()))
            for name, thr in (("under_1e-3", 1e-3), ("under_1e-4", 1e-4), ("under_1e-5", 1e-5), ("under_1e-6", 1e-6)):
                gaps[name] += int((gap < thr).sum())
            low = torch.from_numpy(np.stack([s["label_lowres"] for s in scenes])).to(device)
            full = low.repeat_interleave(8, dim=1).repeat_interleave(8, dim=2)       # the planted scene, nearest-neighbour
            gaps["object_pixels"] += int((full > 0).sum())
            gaps["planted_label_recovered"] += int(((lab64 == full) & (full > 0)).sum())
            for p in f32_paths:
                lab, prob = run(p, data, data_p, planted)
                flip = lab != lab64
                a = acc[p]
                a["label_flips"] += int(flip.sum())
                a["object_pixels_flipped"] += int((flip & (full > 0)).sum())
                if bool(flip.any()):
                    a["flip_gap_max"] = max(a["flip_gap_max"], float(gap[flip].max()))
                a["score_err_max"] = max(a["score_err_max"], float((torch.log(prob.double().clamp_min(1e-300)) - lp64).abs().max()))
            if log:
                log("amplitude %g: frames %d..%d done" % (amp, b0, b0 + B - 1))
        out["amplitudes"]["%g" % amp] = {"float64_gaps": gaps, "paths": acc}
    agree = [a for a in amplitudes if all(v["label_flips"] == 0 for v in out["amplitudes"]["%g" % a]["paths"].values())]
    flipped = [a for a in amplitudes if any(v["label_flips"] for v in out["amplitudes"]["%g" % a]["paths"].values())]
    out["smallest_amplitude_with_zero_flips_on_all_f32_trunks"] = min(agree) if agree else None
    out["largest_amplitude_with_any_flip"] = max(flipped) if flipped else None
    out["flip_gap_max_overall"] = max((v["flip_gap_max"] for a in out["amplitudes"].values() for v in a["paths"].values()), default=0.0)
    return out


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--margin-sweep", action="store_true",
                    help="the planted-logit amplitude sweep instead of the detection study (python tests/parity_study.py --margin-sweep "
                         "--frames 8 --out profiles/r06_margin_study.json)")
    a = ap.parse_args()
    if a.margin_sweep:
        res = run_margin_sweep(torch.device("cuda:0"), n_frames=a.frames, batch=a.batch, log=lambda m: print(m, file=sys.stderr, flush=True))
        txt = json.dumps(res, indent=1)
        print(txt)
        if a.out:
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            open(a.out, "w").write(txt + "\n")
        return
    res = run_study(torch.device("cuda:0"), a.frames, a.batch, log=lambda m: print(m, file=sys.stderr, flush=True))
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
