"""Where does adl_terms_kernel's launch go? (round 5) Rows with pose targets on a 3024-row capacity buffer, P = 2620 points:
    python tools/probe_adl.py
prints the launch time for: no symmetric row / only symmetric rows / the bench's mix, with the symmetric rows contiguous or spread."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posecnn_amd import config, ops, synth, _lib
dev = torch.device("cuda:0")
C, P, CAP = 22, config.NUM_MODEL_POINTS, 3024
pts = torch.from_numpy(synth.make_model_points(C, P, extents=config.LOV_EXTENTS)).to(dev)
rng = np.random.default_rng(0)


def case(n_live, sym_rows, spread):
    """n_live rows with targets; `sym_rows` of them of a symmetric class (16), contiguous at the front or spread evenly"""
    w = np.zeros((CAP, 4 * C), np.float32); t = np.zeros_like(w); p = np.zeros_like(w)
    sym = set(np.linspace(0, n_live - 1, sym_rows).astype(int).tolist()) if spread else set(range(sym_rows))
    for r in range(n_live):
        c = 16 if r in sym else 3
        q = rng.standard_normal(4); q /= np.linalg.norm(q); q2 = rng.standard_normal(4); q2 /= np.linalg.norm(q2)
        w[r, 4 * c:4 * c + 4] = 1; t[r, 4 * c:4 * c + 4] = q; p[r, 4 * c:4 * c + 4] = q2
    sy = torch.from_numpy(config.LOV_SYMMETRY.astype(np.float32)).to(dev)
    args = [torch.from_numpy(a).to(dev) for a in (p, t, w)]
    cnt = torch.tensor([n_live], dtype=torch.int32, device=dev)
    f = lambda: ops.average_distance_loss(args[0], args[1], args[2], pts, sy, 0.01, num_rows=cnt)
    for _ in range(3): f()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    rep = _lib.profile_report(); _lib.profile_enable(False)
    return {k: round(v["avg_us"], 1) for k, v in rep.items()}


only = os.environ.get("ADL_PROBE_ONLY")   # one case (for a rocprofv3 --pmc pass)
if only == "63 live, all symmetric":
    print(json.dumps({only: case(63, 63, False)})); sys.exit(0)
out = {"684 live, 0 symmetric": case(684, 0, False), "684 live, 63 symmetric contiguous": case(684, 63, False),
       "684 live, 63 symmetric spread": case(684, 63, True), "63 live, all symmetric": case(63, 63, False),
       "9 live, all symmetric": case(9, 9, False), "1 live, symmetric": case(1, 1, False), "1 live, not symmetric": case(1, 0, False),
       "63 live, 0 symmetric": case(63, 0, False)}
print(json.dumps(out, indent=1))
