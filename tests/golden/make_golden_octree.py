"""Generates tests/golden/octree_nodes.npz from the REFERENCE's FixedOctTree (run only in the build container):

    cd /root/repo/tests/golden && python make_golden_octree.py

src/UCF_VIT/dataloaders/octree.py is loaded from its file with an EMPTY stand-in module for `cv2` (absent here; octree.py imports it
but never calls it); scipy (RegularGridInterpolator, used by serialize) is the real package.  Recorded: edge volumes (uint8), the node
lists [x1, x2, y1, y2, z1, z2] + values of _build_tree, and for small cases the serialized patches, sizes and centres.  Data only.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
spec = importlib.util.spec_from_file_location("ref_octree", "/root/reference/src/UCF_VIT/dataloaders/octree.py")
O = importlib.util.module_from_spec(spec)
spec.loader.exec_module(O)

rec = {}
# (kind, side, fixed_length, norm_factor, serialize patch size or 0)
cases = [("sparse", 32, 64, 255, 4), ("dense", 32, 120, 255, 0), ("corner", 32, 64, 255, 0), ("empty", 16, 15, 255, 2), ("multi", 64, 512, 85, 0),
         ("sparse", 64, 225, 255, 0)]
for i, (kind, n, L, norm, p) in enumerate(cases):
    rng = np.random.Generator(np.random.PCG64(200 + i))
    if kind == "sparse":
        dom = (rng.random((n, n, n)) < 0.01).astype(np.uint8) * 255
    elif kind == "dense":
        dom = (rng.random((n, n, n)) < 0.4).astype(np.uint8) * 255
    elif kind == "corner":
        dom = np.zeros((n, n, n), dtype=np.uint8)
        dom[: n // 4, : n // 4, : n // 4] = (rng.random((n // 4,) * 3) < 0.6) * 255
    elif kind == "empty":
        dom = np.zeros((n, n, n), dtype=np.uint8)
    else:                       # what Patchify_3D builds for 3 channels: counter (0..3) * norm_factor with norm_factor = int(255 / 3)
        dom = (rng.integers(0, 4, (n, n, n)) * (rng.random((n, n, n)) < 0.05) * norm).astype(np.uint8)
    t = O.FixedOctTree(domain=dom, fixed_length=L, norm_factor=norm)
    nodes = np.array([list(c.get_coord()) for c, _ in t.nodes], dtype=np.int32)
    vals = np.array([v for _, v in t.nodes], dtype=np.int64)
    rec[f"domain{i}"], rec[f"nodes{i}"], rec[f"values{i}"] = dom, nodes, vals
    rec[f"L{i}"], rec[f"norm{i}"], rec[f"p{i}"] = np.int32(L), np.int32(norm), np.int32(p)
    print(i, kind, n, "fixed_length", L, "norm", norm, "-> nodes", len(nodes), "max value", int(vals.max()))
    if p:
        img = rng.random((n, n, n, 1)).astype(np.float32)
        seq_patch, seq_size, seq_pos = t.serialize(img, size=(p, p, p, 1))
        rec[f"img{i}"] = img
        rec[f"seq{i}"] = np.asarray(seq_patch, dtype=np.float32)            # [L, p, p, p, 1] (Patchify_3D: np.asarray(seq_img, float32))
        rec[f"size{i}"] = np.asarray(seq_size, dtype=np.float32)
        rec[f"pos{i}"] = np.asarray(seq_pos, dtype=np.float32)
        print("   serialized", rec[f"seq{i}"].shape)
rec["n_cases"] = np.int32(len(cases))
np.savez_compressed(os.path.join(HERE, "octree_nodes.npz"), **rec)
print("done")
