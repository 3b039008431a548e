"""Test scenes for the ICP slice (TEST INFRASTRUCTURE): an analytic box renderer standing in for the reference's OpenGL
vertex / normal render (lib/synthesize/synthesize.cpp:1972-1991) and small SE(3) helpers."""
import numpy as np


def rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def pose(R, t):
    T = np.zeros((3, 4))
    T[:, :3], T[:, 3] = R, t
    return T


def compose(A, B):
    """A * B for 3x4 rigid transforms"""
    return pose(A[:, :3] @ B[:, :3], A[:, :3] @ B[:, 3] + A[:, 3])


def pose_error(A, B):
    """(rotation angle in degrees, translation distance) between two 3x4 poses"""
    Rd = A[:, :3] @ B[:, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(Rd) - 1) / 2, -1, 1)))
    return ang, float(np.linalg.norm(A[:, 3] - B[:, 3]))


def render_box(T, half, K, H, W):
    """Ray-cast an axis-aligned box (half extents `half`, object frame) placed at pose T (camera <- object).
    Returns (vertex_map f32 [H,W,3] in camera coordinates, 0 on the background; normal_map f32 [H,W,3]; mask)."""
    R, t = T[:, :3], T[:, 3]
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, dtype=np.float64)], axis=-1)   # camera rays (z = 1)
    o = -R.T @ t                      # camera centre in the object frame
    dd = d @ R                        # ray directions in the object frame (R^T d)
    half = np.asarray(half, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-half - o) / dd
        t2 = (half - o) / dd
    tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
    tmin, tmax = tn.max(-1), tf.min(-1)
    hit = (tmax >= tmin) & (tmin > 0)
    axis = tn.argmax(-1)
    n_obj = np.zeros((H, W, 3))
    sgn = -np.sign(np.take_along_axis(dd, axis[..., None], -1))[..., 0]
    np.put_along_axis(n_obj, axis[..., None], sgn[..., None], -1)
    vert = d * tmin[..., None]        # camera frame: z = tmin because the rays have z = 1
    norm = n_obj @ R.T
    vert[~hit] = 0
    norm[~hit] = 0
    return vert.astype(np.float32), norm.astype(np.float32), hit


def scene(T_true, T_init, half, K, H, W, factor=10000.0, obj_id=3, noise=0.0, rng=None):
    """live depth (uint16, quantised by `factor`) + label from the box at T_true; predicted maps from the box at T_init"""
    v_true, _, hit = render_box(T_true, half, K, H, W)
    z = v_true[..., 2].astype(np.float64)
    if noise and rng is not None:
        z = z + rng.standard_normal(z.shape) * noise * hit
    depth = np.clip(np.round(z * factor), 0, 65535).astype(np.uint16)
    label = np.where(hit, obj_id, 0).astype(np.int32)
    pv, pn, _ = render_box(T_init, half, K, H, W)
    return depth, label, pv, pn


# ---- triangle meshes for the renderer tests ---------------------------------------------------------------------------
def box_mesh(half):
    """12 triangles, 24 vertices (one quad per face so that the per-vertex normals are the face normals)"""
    hx, hy, hz = half
    verts, norms, faces = [], [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            a, b = (axis + 1) % 3, (axis + 2) % 3
            base = len(verts)
            for sa, sb in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                p = [0.0, 0.0, 0.0]
                p[axis] = sgn * half[axis]
                p[a] = sa * half[a]
                p[b] = sb * half[b]
                verts.append(p)
                n = [0.0, 0.0, 0.0]
                n[axis] = sgn
                norms.append(n)
            faces += [[base, base + 1, base + 2], [base, base + 2, base + 3]]
    return np.asarray(verts, np.float32), np.asarray(norms, np.float32), np.asarray(faces, np.int32)


def icosphere(radius=1.0, subdivisions=2, scale=(1.0, 1.0, 1.0)):
    """Subdivided icosahedron (20 * 4^s faces) scaled per axis into an ellipsoid; returns (vertices, unit-sphere normals, faces).
    (For the ellipsoid the true normal is n ~ p / scale^2; callers that need it compute it from the vertices.)"""
    t = (1.0 + 5 ** 0.5) / 2
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    v = np.asarray(v)
    sc = np.asarray(scale, np.float64) * radius
    n = v / sc
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return (v * sc).astype(np.float32), n.astype(np.float32), np.asarray(f, np.int32)


def depth_scene_from_mesh(render_fn, T_true, K, H, W, factor=10000.0, obj_id=3):
    """depth (uint16) + label of a mesh at T_true through `render_fn(pose[1,3,4]) -> vertices [1,H,W,4]`"""
    v = np.asarray(render_fn(T_true[None]))[0]
    hit = np.isfinite(v[..., 2])
    z = np.where(hit, v[..., 2], 0.0).astype(np.float64)
    depth = np.clip(np.round(z * factor), 0, 65535).astype(np.uint16)
    label = np.where(hit, obj_id, 0).astype(np.int32)
    return depth, label


HYPOTHESIS_DZ = (0.0, -0.02, -0.01, 0.01, 0.02, 0.03, 0.04, 0.05)    # synthesize.cpp:2252-2270


def quat2mat(q):
    w, x, y, z = [float(v) for v in q]
    s = 2.0 / (w * w + x * x + y * y + z * z)
    return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])


def solve_icp_reference(label, depth, K, factor, obj, T_co, mesh, max_error=0.01, iterations=8, depth_range=(0.25, 6.0), radius=0.01,
                        q_t=None, polish_evaluations=50):
    """Synthesizer::solveICP for ONE object on the CPU checker (oracle_*):
    render -> masked backprojection -> translation estimate -> Nelder-Mead polish -> 8 depth hypotheses x (render + ICP) -> SegICP score.
    T_co: the network's pose (3x4, camera <- object); q_t: the same as (quaternion, translation) when the caller wants the
    rx, ry of synthesize.cpp:2209-2214 taken from it. Returns a dict (T_new, T_icp, hits, choose, pairs, agree)."""
    import oracle
    v, n, f = mesh
    H, W = label.shape
    T_co = np.array(T_co, dtype=np.float64)
    maps = oracle.render_mesh(v, n, f, T_co[None], K, H, W, depth_range, model_index=obj - 1)
    live = oracle.icp_backproject(depth, label, obj, K, factor)
    sums, mask = oracle.icp_center(label, live, maps["canonical"][0], maps["vertices"][0], maps["normals"][0], obj, max_error)
    c = int(sums[3])
    t_in = T_co[:, 3].copy() if q_t is None else np.asarray(q_t[4:7], dtype=np.float64)
    Tz = T_co[2, 3]
    if c > 0:
        Tz = float(np.float32(sums[2]) / np.float32(c))
        rx = t_in[0] / t_in[2] if t_in[2] else 0.0
        ry = t_in[1] / t_in[2] if t_in[2] else 0.0
        T_co[:, 3] = (rx * Tz, ry * Tz, Tz)
        if polish_evaluations:
            pv = oracle.render_mesh(v, n, f, T_co[None], K, H, W, depth_range, want=("vertices",))["vertices"][0]
            x, _, _ = oracle.icp_polish(label, live, pv, obj, depth_range, polish_evaluations)
            T_co = compose(pose(quat2mat(x[:4]), x[4:7]), T_co)
            Tz = T_co[2, 3]
    T_new = T_co.copy()
    hyps = np.repeat(T_co[None], len(HYPOTHESIS_DZ), 0)
    hyps[:, 2, 3] = Tz + np.asarray(HYPOTHESIS_DZ)
    pm = oracle.render_mesh(v, n, f, hyps, K, H, W, depth_range, want=("vertices", "normals"))
    upd, _ = oracle.icp_refine(np.repeat(live[None], len(hyps), 0), pm["vertices"], pm["normals"], K, depth_range, max_error, iterations)
    hyps = np.stack([compose(U, T) for U, T in zip(upd, hyps)])
    pairs = int(sums[4])
    choose, hits = 0, None
    if pairs > 0:
        hits = oracle.icp_score(live, maps["canonical"][0], mask, hyps, radius)
        choose = int(np.argmax(hits))
    return {"T_new": T_new, "T_icp": hyps[choose], "hyps": hyps, "hits": hits, "choose": choose, "pairs": pairs, "agree": c}


def random_sheet(seed, n_points=400, tilt=(0.4, -0.25), z0=0.8, half=0.12):
    """A random Delaunay triangulation of a square sheet (slivers and all) in the plane z = z0 + tilt . (x, y): returns
    (vertices f32 [n,3], faces int32 [m,3], hull test `inside(x, y, margin)`)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-half, half, (n_points, 2))
    pts[:4] = [[-half, -half], [half, -half], [half, half], [-half, half]]         # the hull is the square itself
    tri = Delaunay(pts)
    z = z0 + tilt[0] * pts[:, 0] + tilt[1] * pts[:, 1]
    v = np.concatenate([pts, z[:, None]], axis=1).astype(np.float32)
    f = tri.simplices.astype(np.int32)
    f[::2] = f[::2, ::-1]                                                           # mixed windings
    return v, f
