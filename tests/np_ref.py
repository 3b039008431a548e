"""Independent numpy restatement of the canonical op specifications, written from the
reference kernels without looking at oracle/pcnn_oracle.c's structure: a second opinion that must
agree with the C oracle bit for bit on small cases. float32 arithmetic throughout (np.float32
scalars / arrays), one rounding per operation.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np

F = np.float32


def exp_f32(x):
    """canonical expf: IEEE-double range reduction + degree-13 Horner, one rounding to float."""
    x = np.asarray(x, dtype=np.float32)
    xd = np.clip(x.astype(np.float64), -150.0, 130.0)
    kd = np.floor(xd * 1.4426950408889634074 + 0.5)
    r = (xd - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10
    coef = [1.6059043836821614599e-10, 2.0876756987868098979e-09, 2.5052108385441718775e-08,
            2.7557319223985890653e-07, 2.7557319223985892511e-06, 2.4801587301587301566e-05,
            1.9841269841269841253e-04, 1.3888888888888889419e-03, 8.3333333333333332177e-03,
            4.1666666666666664354e-02, 1.6666666666666665741e-01, 0.5, 1.0, 1.0]
    p = np.full_like(xd, coef[0])
    for c in coef[1:]:
        p = p * r + c
    with np.errstate(over="ignore", under="ignore"):
        out = (p * np.ldexp(1.0, kd.astype(np.int64))).astype(np.float32)
    return np.where(np.isnan(x), x, out)


def exp_softmax_f32(x):
    """canonical exp of the softmax layers: the all-f32 sequence of exp_softmax_f32 (pcnn_device.h) /
    oracle_exp_softmax — every numpy operation below is one IEEE float32 operation."""
    x = np.asarray(x, dtype=np.float32)
    xc = np.minimum(np.maximum(x, F(-104.0)), F(88.0)).astype(F)
    with np.errstate(invalid="ignore", under="ignore", over="ignore"):
        kf = np.rint(xc * F(1.44269502)).astype(F)
        r = (xc - kf * F(0.693145752)).astype(F)
        r = (r - kf * F(1.42860677e-06)).astype(F)
        p = np.full_like(r, F(1.98412698e-04))
        for c in (1.38888889e-03, 8.33333377e-03, 4.16666679e-02, 1.66666672e-01, 0.5, 1.0, 1.0):
            p = (p * r + F(c)).astype(F)
        k = np.where(np.isnan(kf), 0, kf).astype(np.int64)
        low = k < -126
        s1 = np.ldexp(F(1.0), np.where(low, k + 64, k).astype(np.int32)).astype(F)
        out = (p * s1).astype(F)
        out = np.where(low, (out * F(5.42101086e-20)).astype(F), out)
    return np.where(np.isnan(x), x, out).astype(F)


def project_box(cls, extents, meta, d):
    """hough_voting_gpu_op.cu.cc:84-120, factor 0.6; d may be an array."""
    d = np.asarray(d, dtype=np.float32)
    h = (extents[cls].astype(np.float64) * 0.5).astype(np.float32)
    fx, px, fy, py = F(meta[0]), F(meta[2]), F(meta[4]), F(meta[5])
    xs, ys = [], []
    with np.errstate(all="ignore"):
        for sz in (+1, -1):
            Z = (F(sz) * h[2] + d).astype(np.float32)
            for sy in (+1, -1):
                for sx in (+1, -1):
                    xs.append((fx * (F(sx) * h[0] / Z).astype(np.float32)).astype(np.float32) + px)
                    ys.append((fy * (F(sy) * h[1] / Z).astype(np.float32)).astype(np.float32) + py)
        xs = np.stack(xs).astype(np.float32)
        ys = np.stack(ys).astype(np.float32)
        # fmin/fmax ignore NaN, starting from +-1e8
        minx = np.fmin(F(1e8), np.fmin.reduce(xs, axis=0))
        maxx = np.fmax(F(-1e8), np.fmax.reduce(xs, axis=0))
        miny = np.fmin(F(1e8), np.fmin.reduce(ys, axis=0))
        maxy = np.fmax(F(-1e8), np.fmax.reduce(ys, axis=0))
        w = (maxx - minx).astype(np.float32) + F(1)
        hh = (maxy - miny).astype(np.float32) + F(1)
        return (np.fmax(w, hh) * F(0.6)).astype(np.float32)


def _angle_pass(cxg, cyg, x, y, u, v, inlier):
    with np.errstate(all="ignore"):
        dx = (cxg - x).astype(np.float32)
        dy = (cyg - y).astype(np.float32)
        n1 = np.sqrt(F(u * u) + F(v * v), dtype=np.float32)
        n2 = np.sqrt((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32), dtype=np.float32)
        dot = (F(u) * dx).astype(np.float32) + (F(v) * dy).astype(np.float32)
        q = dot / (n1 * n2).astype(np.float32)
        return q > F(inlier), np.abs(dx), np.abs(dy)


def hough_space(labelmap, vertmap, extents, meta, cls, skip, inlier=0.9):
    """votes + hough_data of every cell for one class (compute_hough_kernel, .cu.cc:253-333)."""
    H, W = labelmap.shape
    C = vertmap.shape[2] // 3
    idx = np.flatnonzero(labelmap.ravel() == cls)[::skip]
    ys, xs = idx // W, idx % W
    u = vertmap[ys, xs, 3 * cls].astype(np.float32)
    v = vertmap[ys, xs, 3 * cls + 1].astype(np.float32)
    d = exp_f32(vertmap[ys, xs, 3 * cls + 2])
    thr = project_box(cls, extents, meta, d)
    cyg, cxg = np.mgrid[0:H, 0:W]
    votes = np.zeros((H, W), np.float32)
    sumd = np.zeros((H, W), np.float32)
    passes = []
    for i in range(len(idx)):
        ok, ax, ay = _angle_pass(cxg, cyg, xs[i], ys[i], u[i], v[i], inlier)
        passes.append((ok, ax, ay))
        m = ok & (ax < thr[i]) & (ay < thr[i])
        votes = (votes + m.astype(np.float32)).astype(np.float32)
        sumd = np.where(m, (sumd + d[i]).astype(np.float32), sumd)
    hd = np.zeros((H, W, 3), np.float32)
    has = votes > 0
    with np.errstate(all="ignore"):
        dist = np.where(has, sumd / np.where(has, votes, F(1)), F(0)).astype(np.float32)
    thr2 = project_box(cls, extents, meta, dist)
    bw = np.full((H, W), -1, np.float32)
    bh = np.full((H, W), -1, np.float32)
    for ok, ax, ay in passes:
        inside = ok & (ax < thr2) & (ay < thr2)
        bw = np.where(inside & (ax > bw), ax, bw)
        bh = np.where(inside & (ay > bh), ay, bh)
    hd[..., 0] = np.where(has, dist, 0)
    hd[..., 1] = np.where(has, F(2) * bh, 0)
    hd[..., 2] = np.where(has, F(2) * bw, 0)
    return votes, hd, len(idx)


def hough_voting(label, vertex, extents, meta, is_train=0, vote_thr=-1.0, per_thr=0.02, skip=10,
                 inlier=0.9, label_thr=500, MAX_ROI=128):
    """Inference-mode rows (is_train must be 0 here): (top_box, top_pose), canonical order."""
    assert not is_train
    B, H, W = label.shape
    C = vertex.shape[3] // 3
    cap = MAX_ROI // B
    boxes, poses = [], []
    for n in range(B):
        md = meta.reshape(B, -1)[n]
        fx, px, fy, py = F(md[0]), F(md[2]), F(md[4]), F(md[5])
        slots = [c for c in range(1, C) if int((label[n] == c).sum()) > label_thr]
        maxima = []
        for c in slots:
            hs, hd, _ = hough_space(label[n], vertex[n], extents, md, c, skip, inlier)
            if vote_thr > 0:
                for cy in range(H):
                    for cx in range(W):
                        v = hs[cy, cx]
                        if not (v > F(vote_thr)) or not (hd[cy, cx, 1] > 0 and hd[cy, cx, 2] > 0):
                            continue
                        win = hs[max(cy - 3, 0):cy + 4, max(cx - 3, 0):cx + 4]
                        if (win > v).any():
                            continue
                        if F(v) / F(hd[cy, cx, 1] * hd[cy, cx, 2]) < F(per_thr):
                            continue
                        maxima.append((c, cx, cy, v, hd[cy, cx]))
            else:
                best = int(np.argmax(hs.ravel()))  # first maximum
                maxima.append((c, best % W, best // W, hs.ravel()[best], hd.reshape(-1, 3)[best]))
        for c, cx, cy, v, hd3 in maxima[:cap]:
            k = 0.5 + float(F(0.05))
            dist, bh2, bw2 = hd3
            boxes.append([F(n), F(c), F(cx - float(bw2) * k), F(cy - float(bh2) * k),
                          F(cx + float(bw2) * k), F(cy + float(bh2) * k), F(v)])
            rx = F(F(cx) - px) / fx
            ry = F(F(cy) - py) / fy
            poses.append([F(1), F(0), F(0), F(0), F(rx * F(dist)), F(ry * F(dist)), F(dist)])
    if not boxes:
        return np.zeros((1, 7), np.float32), np.zeros((1, 7), np.float32)
    return np.array(boxes, np.float32), np.array(poses, np.float32)


def c_round(x):
    """C roundf: half away from zero."""
    x = np.float32(x)
    return int(np.sign(x) * np.floor(np.abs(x) + F(0.5)))


def roi_pool(data, rois, PH, PW, scale, pool_channel):
    """roi_pooling_op_gpu.cu.cc:20-101"""
    B, H, W, C = data.shape
    R = rois.shape[0]
    Cout = 1 if pool_channel else C
    top = np.zeros((R, PH, PW, Cout), np.float32)
    arg = np.full((R, PH, PW, Cout), -1, np.int32)
    for n in range(R):
        b, cls = int(rois[n, 0]), int(rois[n, 1])
        sw, sh = c_round(F(rois[n, 2]) * F(scale)), c_round(F(rois[n, 3]) * F(scale))
        ew, eh = c_round(F(rois[n, 4]) * F(scale)), c_round(F(rois[n, 5]) * F(scale))
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = F(rh) / F(PH), F(rw) / F(PW)
        if b < 0 or b >= B:
            continue
        for ph in range(PH):
            for pw in range(PW):
                hs = min(max(int(np.floor(F(ph) * bh)) + sh, 0), H)
                he = min(max(int(np.ceil(F(ph + 1) * bh)) + sh, 0), H)
                ws = min(max(int(np.floor(F(pw) * bw)) + sw, 0), W)
                we = min(max(int(np.ceil(F(pw + 1) * bw)) + sw, 0), W)
                if he <= hs or we <= ws:
                    continue
                chans = [cls] if pool_channel else range(C)
                for oc, c in enumerate(chans):
                    if c < 0 or c >= C:
                        continue
                    patch = data[b, hs:he, ws:we, c]
                    k = int(np.argmax(patch.ravel()))  # first max, h-major
                    val = patch.ravel()[k]
                    if val > -np.finfo(np.float32).max:
                        hh, ww = hs + k // (we - ws), ws + k % (we - ws)
                        top[n, ph, pw, oc] = val
                        arg[n, ph, pw, oc] = (hh * W + ww) * C + c
                    else:
                        top[n, ph, pw, oc] = -np.finfo(np.float32).max
    return top, arg


def hard_label(prob, gt, thr):
    """hard_label_op_gpu.cu.cc:17-29"""
    C = prob.shape[-1]
    p = prob.reshape(-1, C)
    g = gt.reshape(-1)
    out = np.zeros_like(p)
    ok = (g >= 0) & (g < C)
    gi = np.where(ok, g, 0)
    sel = ok & ((g > 0) | (p[np.arange(len(g)), gi] < F(thr)))
    out[np.flatnonzero(sel), gi[sel]] = 1
    return out.reshape(prob.shape)


def _rot(q):
    s, u, v, w = (F(t) for t in q)
    return np.array([
        [s * s + u * u - v * v - w * w, F(2) * (u * v - s * w), F(2) * (u * w + s * v)],
        [F(2) * (u * v + s * w), s * s - u * u + v * v - w * w, F(2) * (v * w - s * u)],
        [F(2) * (u * w - s * v), F(2) * (v * w + s * u), s * s - u * u - v * v + w * w]], np.float32)


def _mv(R, pts):
    """rows of pts times R^T with the reference's left-to-right sums (no matmul reassociation)."""
    out = np.empty_like(pts)
    for j in range(3):
        out[:, j] = ((R[j, 0] * pts[:, 0]).astype(np.float32) + (R[j, 1] * pts[:, 1]).astype(np.float32)
                     ).astype(np.float32) + (R[j, 2] * pts[:, 2]).astype(np.float32)
    return out.astype(np.float32)


def average_distance(pred, target, weight, point, symmetry, margin):
    """average_distance_loss_op_gpu.cu.cc:35-252,323-335"""
    R = pred.shape[0]
    C, P = point.shape[0], point.shape[1]
    diff_out = np.zeros((R, 4 * C), np.float32)
    total = F(0)
    for n in range(R):
        w4 = weight[n].reshape(C, 4)[:, 0]
        cs = np.flatnonzero(w4 > 0)
        if len(cs) == 0:
            continue
        c = int(cs[0])
        Rg, Ru = _rot(target[n, 4 * c:4 * c + 4]), _rot(pred[n, 4 * c:4 * c + 4])
        s, u, v, w = (F(t) for t in pred[n, 4 * c:4 * c + 4])
        D = [np.array([[2 * s, -2 * w, 2 * v], [2 * w, 2 * s, -2 * u], [-2 * v, 2 * u, 2 * s]], np.float32),
             np.array([[2 * u, 2 * v, 2 * w], [2 * v, -2 * u, -2 * s], [2 * w, 2 * s, -2 * u]], np.float32),
             np.array([[-2 * v, 2 * u, 2 * s], [2 * u, 2 * v, 2 * w], [-2 * s, 2 * w, -2 * v]], np.float32),
             np.array([[-2 * w, -2 * s, 2 * u], [2 * s, -2 * w, 2 * v], [2 * u, 2 * v, 2 * w]], np.float32)]
        pts = point[c].astype(np.float32)
        X1 = _mv(Ru, pts)
        X2all = _mv(Rg, pts)
        lb = F(0)
        g = [F(0)] * 4
        den = F(R * P)
        for p in range(P):
            if symmetry[c] > 0:
                e = (X1[p] - X2all).astype(np.float32)
                dd = ((e[:, 0] * e[:, 0]).astype(np.float32) + (e[:, 1] * e[:, 1]).astype(np.float32)
                      ).astype(np.float32) + (e[:, 2] * e[:, 2]).astype(np.float32)
                q = int(np.argmin(dd))  # first minimum
            else:
                q = p
            e = (X1[p] - X2all[q]).astype(np.float32)
            dist = F(F(e[0] * e[0]) + F(e[1] * e[1])) + F(e[2] * e[2])
            if dist < F(margin):
                continue
            lb = F(lb + F(np.float64(F(dist - F(margin))) / (2.0 * R * P)))
            for k4 in range(4):
                acc = F(0)
                for j in range(3):
                    for k in range(3):
                        acc = F(acc + F(F(F(e[j] * pts[p, k]) * D[k4][j, k]) / den))
                g[k4] = F(g[k4] + acc)
        # NOTE: per-point partial sums `acc` start from 0 per point in the reference too
        # (diffs[index_diff + k] is a per-(roi, point) slot), then are summed over p ascending.
        for k4 in range(4):
            diff_out[n, 4 * c + k4] = g[k4]
        total = F(total + lb)
    return np.array([total], np.float32), diff_out


def backproject(data, label, depth, meta, label_3d, G, ksize, thr):
    """backprojecting_op_gpu.cu.cc:17-126"""
    B, H, W, Cd = data.shape
    Cl = label.shape[3]
    top_data = np.zeros((B, G, G, G, Cd), np.float32)
    top_flag = np.zeros((B, G, G, G, Cd), np.float32)
    top_label = np.zeros((B, G, G, G, Cl), np.float32)
    for n in range(B):
        m = meta.reshape(B, -1)[n].astype(np.float32)
        for d in range(G):
            for h in range(G):
                for w in range(G):
                    X = F(F(d) * m[42]) + m[45]
                    Y = F(F(h) * m[43]) + m[46]
                    Z = F(F(w) * m[44]) + m[47]
                    X1 = F(F(F(m[18] * X) + F(m[19] * Y)) + F(m[20] * Z)) + m[21]
                    Y1 = F(F(F(m[22] * X) + F(m[23] * Y)) + F(m[24] * Z)) + m[25]
                    Z1 = F(F(F(m[26] * X) + F(m[27] * Y)) + F(m[28] * Z)) + m[29]
                    x1 = F(F(m[0] * X1) + F(m[1] * Y1)) + F(m[2] * Z1)
                    x2 = F(F(m[3] * X1) + F(m[4] * Y1)) + F(m[5] * Z1)
                    x3 = F(F(m[6] * X1) + F(m[7] * Y1)) + F(m[8] * Z1)
                    with np.errstate(all="ignore"):
                        a, b = F(x1) / F(x3), F(x2) / F(x3)
                    if not (np.isfinite(a) and np.isfinite(b)):
                        px = 0 if np.isnan(a) else (2**31 - 1 if a > 0 else -2**31)
                        py = 0 if np.isnan(b) else (2**31 - 1 if b > 0 else -2**31)
                    else:
                        px, py = c_round(a), c_round(b)
                    acc = np.zeros(Cd, np.float32)
                    accl = np.zeros(Cl, np.float32)
                    cnt = 0
                    for x in range(max(px - ksize, 0), min(px + ksize, W - 1) + 1):
                        for y in range(max(py - ksize, 0), min(py + ksize, H - 1) + 1):
                            if np.abs(F(depth[n, y, x] - Z1)) < F(thr):
                                cnt += 1
                                acc = (acc + data[n, y, x]).astype(np.float32)
                                accl = (accl + label[n, y, x]).astype(np.float32)
                    if cnt == 0:
                        top_label[n, d, h, w] = label_3d[n, d, h, w]
                    else:
                        top_data[n, d, h, w] = acc / F(cnt)
                        top_flag[n, d, h, w] = 1
                        top_label[n, d, h, w] = accl / F(cnt)
    return top_data, top_label, top_flag


def softmax_argmax(score):
    """network.py:474-488, 432-434 with the canonical softmax exp"""
    s = score.astype(np.float32)
    m = np.fmax.reduce(s, axis=-1, keepdims=True)
    e = exp_softmax_f32((s - m).astype(np.float32))
    tot = np.zeros(s.shape[:-1] + (1,), np.float32)
    for c in range(s.shape[-1]):
        tot = (tot + e[..., c:c + 1]).astype(np.float32)
    p = (e / tot).astype(np.float32)
    return p, np.argmax(p, axis=-1).astype(np.int32)
