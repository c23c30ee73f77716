"""Generates tests/golden/quadtree_nodes.npz from the REFERENCE's FixedQuadTree (run only in the build container):

    cd /root/repo/tests/golden && python make_golden_quadtree.py

src/UCF_VIT/dataloaders/quadtree.py is loaded from its file with an EMPTY stand-in module for `cv2` (absent here): the code path that
is recorded — FixedQuadTree._build_tree / Rect.contains, pure numpy integer logic — never touches cv2 (only set_area / serialize
do, and those are not called).  The fixtures are data only: edge maps (uint8, 0 / 255 like cv.Canny output) and the node lists
[x1, x2, y1, y2] + values the reference builds from them.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
spec = importlib.util.spec_from_file_location("ref_quadtree", "/root/reference/src/UCF_VIT/dataloaders/quadtree.py")
Q = importlib.util.module_from_spec(spec)
spec.loader.exec_module(Q)


def edge_map(kind, H, W, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "sparse":
        m = rng.random((H, W)) < 0.03
    elif kind == "dense":
        m = rng.random((H, W)) < 0.5
    elif kind == "blobs":                                   # edges concentrated in a few regions, like an object outline
        m = np.zeros((H, W), dtype=bool)
        for _ in range(5):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(4, max(5, H // 5))
            yy, xx = np.ogrid[:H, :W]
            d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
            m |= np.abs(d - r) < 1.0
    elif kind == "empty":
        m = np.zeros((H, W), dtype=bool)
    else:                                                   # one corner only: the tree digs down to 2-pixel nodes and stops early
        m = np.zeros((H, W), dtype=bool)
        m[: H // 8, : W // 8] = rng.random((H // 8, W // 8)) < 0.7
    return m.astype(np.uint8) * 255


rec = {}
cases = [("sparse", 64, 64, 16), ("dense", 64, 64, 49), ("blobs", 224, 224, 196), ("empty", 32, 32, 16), ("corner", 64, 64, 196),
         ("blobs", 256, 256, 1024), ("sparse", 224, 224, 49), ("dense", 128, 128, 400)]
for i, (kind, H, W, L) in enumerate(cases):
    edges = edge_map(kind, H, W, 100 + i)
    qt = Q.FixedQuadTree(domain=edges, fixed_length=L)
    nodes = np.array([list(r.get_coord()) for r, _ in qt.nodes], dtype=np.int32)
    vals = np.array([v for _, v in qt.nodes], dtype=np.int64)
    rec[f"edges{i}"] = edges
    rec[f"nodes{i}"] = nodes
    rec[f"values{i}"] = vals
    rec[f"L{i}"] = np.int32(L)
    print(i, kind, (H, W), "fixed_length", L, "-> nodes", len(nodes), "max value", int(vals.max()) if len(vals) else 0)
rec["n_cases"] = np.int32(len(cases))
np.savez_compressed(os.path.join(HERE, "quadtree_nodes.npz"), **rec)
print("done")
