#!/usr/bin/env python
"""Generates tests/golden/demo_frames.npz and tests/golden/lov_models.npz from the reference's own DATA
fixtures ("data, not expected outputs"), so that GPU parity tests can run on realistic
geometry on a box without /root/reference:

  demo_frames.npz   depth  uint16 [5,480,640]  data/demo_images/00000{1..5}-depth.png (factor 10000, tools/demo.py:101)
                    label  uint8  [5,480,640]  a deterministic segmentation of each depth image into up to six YCB
                                               classes (valid-depth quantile bands, biggest connected blob per band)
  lov_models.npz    points f32 [22,2620,3]     data/LOV/models/*/points.xyz cut to the shortest model (lov.py:141-158)
                    extents f32 [22,3]         data/LOV/extents.txt (lov.py:161-170)

Run here (the reference tree is mounted in this container only):  python tests/golden/make_demo_fixtures.py
"""
import os
import sys

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from posecnn_amd import config, datasets  # noqa: E402

REF = "/root/reference"
CLASSES = (1, 5, 11, 16, 21, 14)   # incl. the two classes lov.py:38 marks symmetric (16, 21)


def segment(depth):
    """uint16 depth -> uint8 label: six quantile bands of the valid depths; in each band the largest connected
    component (after a 5x5 opening) becomes one object of CLASSES[band]."""
    valid = (depth > 0) & (depth < 20000)
    label = np.zeros(depth.shape, np.uint8)
    if valid.sum() < 1000:
        return label
    qs = np.quantile(depth[valid], np.linspace(0, 1, len(CLASSES) + 1))
    for b, cls in enumerate(CLASSES):
        m = valid & (depth >= qs[b]) & (depth <= qs[b + 1])
        m = ndimage.binary_opening(m, structure=np.ones((5, 5), bool))
        comp, n = ndimage.label(m)
        if n == 0:
            continue
        sizes = ndimage.sum(m, comp, index=np.arange(1, n + 1))
        label[comp == (1 + int(np.argmax(sizes)))] = cls
    return label


def main():
    depths, labels = [], []
    for i in range(1, 6):
        d = datasets.read_depth(os.path.join(REF, "data", "demo_images", "%06d-depth.png" % i))
        depths.append(d)
        labels.append(segment(d))
    out = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out, "demo_frames.npz"), depth=np.stack(depths), label=np.stack(labels))
    _, pts_all = datasets.load_object_points(os.path.join(REF, "data", "LOV", "models"), config.LOV_CLASSES)
    ext = datasets.load_object_extents(os.path.join(REF, "data", "LOV", "extents.txt"), 22)
    np.savez_compressed(os.path.join(out, "lov_models.npz"), points=pts_all.astype(np.float32), extents=ext)
    for f in ("demo_frames.npz", "lov_models.npz"):
        print(f, os.path.getsize(os.path.join(out, f)))
    for k, l in enumerate(labels):
        print("frame", k + 1, {int(c): int((l == c).sum()) for c in np.unique(l) if c})


if __name__ == "__main__":
    main()
