"""GPU-side fixed-length quadtree patcher: tree logic bit-exact against node lists built by the reference's
own FixedQuadTree (tests/golden/quadtree_nodes.npz), CPU oracle pinned by the same fixtures; the bicubic patch resampling is
checked against the oracle (torch bicubic = the cv2.INTER_CUBIC formula; cv2 is absent: that half is parity unpinned)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

from oracle import quadtree_ref as QR


def _cases():
    g = load_golden("quadtree_nodes.npz")
    for i in range(int(g["n_cases"])):
        yield i, g[f"edges{i}"].numpy(), int(g[f"L{i}"]), g[f"nodes{i}"].numpy(), g[f"values{i}"].numpy()


def test_oracle_tree_matches_reference_node_lists():
    for i, edges, L, nodes, vals in _cases():
        n, v = QR.build_tree(edges, L)
        assert np.array_equal(np.array(n, dtype=np.int32), nodes), i
        assert np.array_equal(np.array(v, dtype=np.int64), vals), i


@pytest.mark.gpu
def test_hip_tree_is_bit_exact_vs_reference_node_lists():
    from UCF_VIT._hip import ops
    for i, edges, L, nodes, vals in _cases():
        e = torch.from_numpy(edges).unsqueeze(0).cuda()
        out_nodes, out_vals, count, seq_ps = ops.quadtree_build(e, L)
        n = int(count[0])
        assert n == len(nodes), (i, n, len(nodes))
        assert np.array_equal(out_nodes[0, :n].cpu().numpy(), nodes), i
        assert np.array_equal(out_vals[0, :n].cpu().numpy().astype(np.int64), vals), i
        # padding as FixedQuadTree.serialize pads: size 0, position (-1, -1)
        sp = seq_ps[0].cpu().numpy()
        assert np.array_equal(sp[:n, 0], (nodes[:, 1] - nodes[:, 0]).astype(np.float32))
        assert np.array_equal(sp[:n, 1], ((nodes[:, 1] + nodes[:, 0]) / 2).astype(np.float32))
        assert np.array_equal(sp[:n, 2], ((nodes[:, 3] + nodes[:, 2]) / 2).astype(np.float32))
        assert (sp[n:, 0] == 0).all() and (sp[n:, 1:] == -1).all()


@pytest.mark.gpu
def test_hip_tree_batched_random_maps_vs_oracle():
    """a batch of different maps in one launch (one workgroup per image), odd densities, ties everywhere"""
    from UCF_VIT._hip import ops
    rng = np.random.Generator(np.random.PCG64(7))
    B, H, W, L = 9, 128, 128, 196
    maps = np.stack([(rng.random((H, W)) < d).astype(np.uint8) * 255 for d in (0.0, 0.001, 0.01, 0.05, 0.2, 0.5, 0.9, 1.0, 0.03)])
    out_nodes, out_vals, count, _ = ops.quadtree_build(torch.from_numpy(maps).cuda(), L)
    for b in range(B):
        n, v = QR.build_tree(maps[b], L)
        assert int(count[b]) == len(n), b
        assert np.array_equal(out_nodes[b, :len(n)].cpu().numpy(), np.array(n, dtype=np.int32)), b
        assert np.array_equal(out_vals[b, :len(n)].cpu().numpy(), np.array(v, dtype=np.int32)), b


@pytest.mark.gpu
@pytest.mark.parametrize("C", [3, 1])
def test_hip_serialize_matches_oracle(C):
    """patch gather + bicubic resample to p x p + the reference's [S, p, p, C] -> [C, S, p*p] reshape; PARITY UNPINNED against cv2
    (absent): the oracle resamples with torch's bicubic kernel, the same formula as cv2.INTER_CUBIC on float input"""
    from UCF_VIT._hip import ops
    rng = np.random.Generator(np.random.PCG64(11))
    H = W = 128
    L, p = 49, 8
    B = 3
    maps = np.stack([(rng.random((H, W)) < d).astype(np.uint8) * 255 for d in (0.02, 0.3, 0.0)])
    imgs = rng.random((B, H, W, C)).astype(np.float32) * 255
    e = torch.from_numpy(maps).cuda()
    nodes, vals, count, seq_ps = ops.quadtree_build(e, L)
    seq = ops.quadtree_serialize(torch.from_numpy(imgs).cuda(), nodes, count, p)        # [B, C, L, p*p]
    assert tuple(seq.shape) == (B, C, L, p * p)
    for b in range(B):
        n, _ = QR.build_tree(maps[b], L)
        want, size, pos = QR.serialize(imgs[b], n, L, p)
        got = seq[b].cpu().numpy()
        got = got if C > 1 else got[0]
        assert np.abs(got - want).max() <= 1e-3 * 255, b            # fp32 bicubic: same taps, different summation order
        assert np.array_equal(seq_ps[b, :, 0].cpu().numpy(), size) and np.array_equal(seq_ps[b, :, 1:].cpu().numpy(), pos)


@pytest.mark.gpu
def test_patchify_module_feeds_the_adaptive_model():
    """UCF_VIT.dataloaders.transform.Patchify (batched, on the device) -> VIT(adaptive_patching=True): shapes and layout line up"""
    from UCF_VIT.dataloaders.transform import Patchify
    from UCF_VIT.simple.arch import VIT
    rng = np.random.Generator(np.random.PCG64(3))
    B, H, L, p = 4, 64, 16, 8
    edges = torch.from_numpy((rng.random((B, H, H)) < 0.05).astype(np.uint8) * 255).cuda()
    img = torch.from_numpy(rng.random((B, H, H, 3)).astype(np.float32)).cuda()
    seq, size, pos, nodes, count = Patchify(L, p, 3)(img, edges)
    assert tuple(seq.shape) == (B, 3, L, p * p) and tuple(size.shape) == (B, L) and tuple(pos.shape) == (B, L, 2)
    assert int(count.min()) == L                      # dense enough maps: every image reaches the fixed length
    # leaves tile the image exactly once
    area = ((nodes[..., 1] - nodes[..., 0]) * (nodes[..., 3] - nodes[..., 2])).sum(dim=1)
    assert bool((area == H * H).all())
    m = VIT(img_size=[H, H], patch_size=p, in_chans=3, num_classes=5, embed_dim=64, depth=1, num_heads=2, adaptive_patching=True, fixed_length=L,
            use_adaptive_pos_emb=True).cuda()
    out = m(seq, ["red", "green", "blue"], torch.cat([size.unsqueeze(-1), pos], dim=-1))
    assert tuple(out.shape) == (B, 5) and bool(torch.isfinite(out).all())


# ------------------------------------------------------------------------------------------------ octree (3-D volumes)
def _ocases():
    g = load_golden("octree_nodes.npz")
    for i in range(int(g["n_cases"])):
        yield i, g, g[f"domain{i}"].numpy(), int(g[f"L{i}"]), int(g[f"norm{i}"]), int(g[f"p{i}"]), g[f"nodes{i}"].numpy(), g[f"values{i}"].numpy()


def test_oracle_octree_matches_reference():
    """node lists bit-exact; the trilinear leaf resampling against the reference's own scipy RegularGridInterpolator output"""
    for i, g, dom, L, norm, p, nodes, vals in _ocases():
        n, v = QR.build_octree(dom, L, norm)
        assert np.array_equal(np.array(n, dtype=np.int32), nodes), i
        assert np.array_equal(np.array(v, dtype=np.int64), vals), i
        if p:
            seq, size, pos = QR.serialize3(g[f"img{i}"].numpy(), n, L, p)
            assert np.abs(seq - g[f"seq{i}"].numpy()).max() < 1e-5, i
            assert np.array_equal(size, g[f"size{i}"].numpy()) and np.array_equal(pos, g[f"pos{i}"].numpy()), i


@pytest.mark.gpu
def test_hip_octree_is_bit_exact_and_serializes_like_the_reference():
    from UCF_VIT._hip import ops
    for i, g, dom, L, norm, p, nodes, vals in _ocases():
        d = torch.from_numpy(dom).unsqueeze(0).cuda()
        out_nodes, out_vals, count, seq_ps = ops.octree_build(d, L, norm)
        n = int(count[0])
        assert n == len(nodes), (i, n, len(nodes))
        assert np.array_equal(out_nodes[0, :n].cpu().numpy(), nodes), i
        assert np.array_equal(out_vals[0, :n].cpu().numpy().astype(np.int64), vals), i
        if p:
            img = g[f"img{i}"].cuda().unsqueeze(0)
            seq = ops.octree_serialize(img, out_nodes, count, p, flat=False)                 # [1, L, p, p, p, C]
            assert np.abs(seq[0].cpu().numpy() - g[f"seq{i}"].numpy()).max() < 1e-5, i
            assert np.array_equal(seq_ps[0, :, 0].cpu().numpy(), g[f"size{i}"].numpy())
            assert np.array_equal(seq_ps[0, :, 1:].cpu().numpy(), g[f"pos{i}"].numpy())


@pytest.mark.gpu
def test_hip_octree_batched_vs_oracle():
    from UCF_VIT._hip import ops
    rng = np.random.Generator(np.random.PCG64(17))
    B, N, L, p = 4, 32, 120, 4
    doms = np.stack([(rng.random((N, N, N)) < d).astype(np.uint8) * 255 for d in (0.0, 0.002, 0.05, 0.6)])
    imgs = rng.random((B, N, N, N, 2)).astype(np.float32)
    out_nodes, out_vals, count, seq_ps = ops.octree_build(torch.from_numpy(doms).cuda(), L, 255)
    seq = ops.octree_serialize(torch.from_numpy(imgs).cuda(), out_nodes, count, p, flat=False)
    for b in range(B):
        n, v = QR.build_octree(doms[b], L, 255)
        assert int(count[b]) == len(n), b
        assert np.array_equal(out_nodes[b, :len(n)].cpu().numpy(), np.array(n, dtype=np.int32)), b
        want, size, pos = QR.serialize3(imgs[b], n, L, p)
        assert np.abs(seq[b].cpu().numpy() - want).max() < 1e-5, b
        assert np.array_equal(seq_ps[b, :, 0].cpu().numpy(), size) and np.array_equal(seq_ps[b, :, 1:].cpu().numpy(), pos)


@pytest.mark.gpu
def test_patchify_3d_module_shapes_and_tiling():
    from UCF_VIT.dataloaders.transform import Patchify_3D
    rng = np.random.Generator(np.random.PCG64(5))
    B, N, L, p = 2, 32, 64, 4
    dom = torch.from_numpy((rng.random((B, N, N, N)) < 0.02).astype(np.uint8) * 255).cuda()
    vol = torch.from_numpy(rng.random((B, N, N, N, 1)).astype(np.float32)).cuda()
    seq, size, pos, nodes, count = Patchify_3D(L, p, 1)(vol, dom)
    assert tuple(seq.shape) == (B, 1, L, p ** 3) and tuple(size.shape) == (B, L) and tuple(pos.shape) == (B, L, 3)
    ext = nodes[..., 1::2] - nodes[..., 0::2]
    assert bool((ext.prod(dim=-1).sum(dim=1) == N ** 3).all())          # the leaves tile the volume exactly once
